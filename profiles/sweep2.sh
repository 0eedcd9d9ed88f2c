#!/bin/bash
CFG=${1:-cfg2}
for PRE in 0 1 2; do for NT in 512 1024 256; do
  r=$(CNF_MFMA_NT=$NT CNF_MFMA_PRE=$PRE timeout 120 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g samples*steps/s  kernel %.3f ms  frac %.3f  path %s' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel_path']))" 2>&1 | tail -1)
  echo "PRE=$PRE NT=$NT : $r"
done; done

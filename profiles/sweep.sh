#!/bin/bash
# A/B sweep of launch variants of the fused MFMA kernel (env knobs read at cnf_create).
CFG=${1:-cfg2}
for NT in 512 768 1024 256; do for PRIO in 0 1 2; do for Q in 0 1; do
  r=$(CNF_MFMA_NT=$NT CNF_MFMA_PRIO=$PRIO CNF_MFMA_QUEUE=$Q timeout 120 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g samples*steps/s  kernel %.3f ms  frac %.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))")
  echo "NT=$NT PRIO=$PRIO QUEUE=$Q : $r"
done; done; done

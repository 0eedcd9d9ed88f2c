#!/bin/bash
# Profile recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect.sh <round-tag> <config> [extra bench args]
# 1. kernel trace + stats of the default bench command
# 2. PMC passes in their own runs (no tracing domains besides --kernel-trace), one small
#    counter group per pass: SQ issue/wait, MFMA busy, LDS, clock, HBM read, HBM write.
set -u
TAG=${1:-r1}; CFG=${2:-cfg2}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cnf::" not in k: continue
        agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"  {c:32s} launches={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}\n")
print(open(out + "/pmc_summary.txt").read())
PY
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-200'
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|gpu)" | head -0

#!/bin/bash
# Profile recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect.sh <round-tag> <config> [extra bench args]
# 1. kernel trace + stats of the bench command.  bench.py pre-rolls >= 2 s of back-to-back launches before its
#    timed steps (several hundred launches of the solve kernel), so the CSV average IS the sustained-clock
#    duration and roofline.frac = flop_per_launch / (CSV average) reproduces the printed line.
#    A second stats file restricted to the launches of the timed region + the last quarter of the pre-roll
#    (warm-ups excluded) is derived from the kernel trace: <config>_kernel_stats_steady.json
# 2. PMC passes in their own runs (no tracing domains besides --kernel-trace), one small
#    counter group per pass: SQ issue/wait, MFMA busy, LDS, clock, HBM read, HBM write; the clock each
#    launch ran at = GRBM_GUI_ACTIVE / 8 XCDs / duration is written next to the times.
set -u
TAG=${1:-r2}; CFG=${2:-cfg2}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 40 --warmup 3 --no-cpu-baseline --secondary none $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
# PMC passes: a short pre-roll is enough (counters are per launch), 24 launches each
# (--secondaries none: the default cfg2 line would otherwise run every secondary workload under every counter pass - serialised
# launches - and the passes ran into their time limit: rounds 5's cfg2 summary had lost its SQ / WRITE passes that way)
PBENCH="python3 $ROOT/bench.py --config $CFG --steps 20 --warmup 2 --preroll-seconds 0.5 --no-cpu-baseline --secondary none --secondaries none --bf16x6-secondary off $*"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- $PBENCH > $OUT/pmc$i.log 2>&1
done
python3 $ROOT/profiles/summarise.py $OUT $CFG

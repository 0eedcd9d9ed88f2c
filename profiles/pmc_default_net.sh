#!/bin/bash
# PMC passes of the solve kernel that serves ICNF(nvariables = NV) (profiles/default_net_profile_target.py): run through gpurun
# from the repo root:  bash profiles/pmc_default_net.sh <tag> <nv>
# PMC_TARGET=<script under profiles/> and PMC_FILTER=<substring of the kernel names to keep> select another target
# (the gradient: PMC_TARGET=default_net_grad_profile_target.py PMC_FILTER=kernel)
set -u
TAG=${1:-r4}; NV=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${TAG}_nv${NV}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NV
TARGET=${PMC_TARGET:-default_net_profile_target.py}
export PMC_FILTER=${PMC_FILTER:-solve_kernel}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ROOT/profiles/$TARGET > $OUT/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ["PMC_FILTER"] not in k: continue
        tot[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[1:] if len(v) > 1 else v     # the first launch carries the first-use overheads
        print(f"   {c:28s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY

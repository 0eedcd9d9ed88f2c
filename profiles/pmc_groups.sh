#!/bin/bash
# PMC passes with caller-chosen counter groups:  bash profiles/pmc_groups.sh <tag> <target.py> <kernel-name filter> "<group 1>" "<group 2>" ...
# (one rocprofv3 run per group, --kernel-trace only; per-kernel averages over the launches after the first)
set -u
TAG=$1; TARGET=$2; export PMC_FILTER=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcg_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
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
        v = v[1:] if len(v) > 1 else v
        print(f"   {c:36s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
rm -rf $OUT

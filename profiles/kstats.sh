#!/bin/bash
# rocprofv3 kernel-trace + stats of a python target; the per-kernel summary CSV lands in gpurun_out/<tag>_kernel_stats.csv
#   bash profiles/kstats.sh <tag> <target.py> [VAR=value ...]
TAG=$1; TARGET=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 $TARGET > gpurun_out/${TAG}_prof.log 2>&1
f=$(find gpurun_out/prof_$TAG -name "*_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
head -14 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200

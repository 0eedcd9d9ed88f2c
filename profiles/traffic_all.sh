#!/bin/bash
# HBM traffic per STEP of every workload on bench.py's default line: bash profiles/traffic_all.sh <tag>   (GPU box, repo root)
# Two PMC passes per workload (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md section HBM), summed over every kernel
# dispatch of profiles/step_target.py and divided by its REPS; FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request), both
# are in KiB.  Writes gpurun_out/<tag>/traffic.json; profiles/make_pmc_traffic.py folds it into profiles/pmc_traffic.json.
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export REPS=3
for WL in cfg2 cfg2p cfg3 cfg4 cfg5 cfg2:grad cfg3:grad cfg4:grad nv20 nv20:grad; do
  export WL
  D=$OUT/traffic_$(echo $WL | tr ':' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/f -- python3 $ROOT/profiles/step_target.py > $D.f.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/w -- python3 $ROOT/profiles/step_target.py > $D.w.log 2>&1
done
python3 - <<PY
import csv, glob, json, os
out = {}
for d in sorted(glob.glob("$OUT/traffic_*")):
    if not os.path.isdir(d): continue
    wl = os.path.basename(d)[len("traffic_"):].replace("_grad", ":grad")
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}; per = {}
    for f in glob.glob(d + "/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cnf::" not in r["Kernel_Name"]: continue
            c = r["Counter_Name"]
            if c in tot:
                v = float(r["Counter_Value"]); tot[c] += v
                k = r["Kernel_Name"].split("(")[0][:60]
                per.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                per[k][c] += v
                if c == "FETCH_SIZE": per[k]["launches"] += 1
    reps = $REPS
    out[wl] = {"bytes_per_step": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / reps,
               "read_bytes_per_step": 2 * tot["FETCH_SIZE"] * 1024 / reps, "write_bytes_per_step": tot["WRITE_SIZE"] * 1024 / reps,
               "kernels": {k: {"bytes_per_step": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / reps, "launches_per_step": v["launches"] / reps}
                           for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:6]}}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["bytes_per_step"] / 1e6, 1) for k, v in out.items()}))
PY

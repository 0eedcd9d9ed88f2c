#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a profiles/traffic_all.sh run: python profiles/make_pmc_traffic.py profiles/r5/<tag>_traffic.json
HBM bytes per STEP of each workload of bench.py's default line (every kernel of the step), which bench.py copies into
roofline.traffic / secondaries[*].roofline.traffic together with the ratio to the algorithmic bytes."""
import glob, hashlib, json, os, sys
src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_digest(root=ROOT):
    """sha256 over the library's kernel sources (csrc/*.hip, *.h, sorted by name): bench.py compares it with the tree it runs from
    and marks `traffic_stale` when a kernel changed after the PMC passes were collected."""
    h = hashlib.sha256()
    d = os.path.join(root, "continuousnormalizingflows.jl_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]

t = json.load(open(src))
out = {"_note": "HBM bytes per step of the workload = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over every kernel dispatch of "
                "profiles/step_target.py, separate --pmc passes (profiles/traffic_all.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md: "
                "gfx950 counts 64 B per 128-B request).  bench.py copies `bytes` into roofline.traffic and names `source`."}
out["_csrc_sha16"] = csrc_digest()
for k, v in t.items():
    out[k] = {"bytes": int(v["bytes_per_step"]), "read": int(v["read_bytes_per_step"]), "write": int(v["write_bytes_per_step"]),
              "source": os.path.relpath(os.path.abspath(src), ROOT),
              "top_kernels": {kk: int(vv["bytes_per_step"]) for kk, vv in list(v["kernels"].items())[:3]}}
old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
if "cfg1" in old and "cfg1" not in out:
    out["cfg1"] = old["cfg1"]
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v["bytes"] for k, v in out.items() if not k.startswith("_")}))

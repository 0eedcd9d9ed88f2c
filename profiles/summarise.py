#!/usr/bin/env python3
"""Summarise one profiles/collect.sh output directory: PMC means per kernel (pmc_summary.txt), the steady-state
launch statistics of the dominant kernel from the kernel trace (kernel_stats_steady.json: warm-ups excluded,
>= 20 launches), and the clock each PMC launch ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration)."""
import collections
import csv
import glob
import json
import sys

out, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)            # (pass dir, dispatch id) -> duration ns for the clock estimate
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cnf::" not in k:
            continue
        agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
clock = {}
for f in glob.glob(out + "/pmc*/**/*kernel_trace.csv", recursive=True):
    pdir = f.split("/pmc")[1].split("/")[0]
    cc = glob.glob(out + "/pmc" + pdir + "/**/*counter_collection.csv", recursive=True)
    if not cc:
        continue
    gui = collections.defaultdict(float)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            gui[r["Dispatch_Id"]] = float(r["Counter_Value"])
    if not gui:
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cnf::" not in k or r["Dispatch_Id"] not in gui:
            continue
        ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        clock.setdefault(k.split("(")[0], []).append(gui[r["Dispatch_Id"]] / 8.0 / ns * 1e3)   # MHz
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"  {c:32s} launches={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}\n")
        if k in clock:
            v = clock[k]
            fo.write(f"  {'clock_MHz (GUI_ACTIVE/8/duration)':32s} launches={len(v):3d} mean={sum(v)/len(v):.6g} "
                     f"min={min(v):.6g} max={max(v):.6g}  [under the PMC pass; counters serialise launches]\n")
print(open(out + "/pmc_summary.txt").read())

# steady-state statistics of the dominant kernel from the --stats run's kernel trace
tr = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
if tr:
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        rows[r["Kernel_Name"].split("(")[0]].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"])))
    tot = {k: sum(e - s for s, e in v) for k, v in rows.items()}
    dom = max(tot, key=tot.get)
    v = sorted(rows[dom])
    d = [(e - s) / 1e6 for s, e in v]
    n = len(d)
    q = max(1, n // 4)
    steady = d[-max(20, q):] if n >= 20 else d
    js = dict(kernel=dom, config=cfg, launches=n, avg_ms_all=sum(d) / n, min_ms=min(d), max_ms=max(d),
              first_quarter_avg_ms=sum(d[:q]) / q, last_quarter_avg_ms=sum(d[-q:]) / q,
              steady_launches=len(steady), steady_avg_ms=sum(steady) / len(steady),
              note="steady = the last max(20, n/4) launches: bench.py's timed steps and the tail of its pre-roll; "
                   "warm-up launches are at the head of the list and excluded")
    json.dump(js, open(out + "/kernel_stats_steady.json", "w"), indent=1)
    print(json.dumps(js, indent=1))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[:1]:
    print("".join(l[:200] + "\n" for l in open(f).read().splitlines()[:4]))

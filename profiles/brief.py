#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line read from stdin (optionally prefixed by a tag given as argv[1])."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    out = [tag, d["config"]["name"], f"{d['value']:.4g}", f"frac={r['frac']:.4f}", f"kernel_ms={r['kernel_ms']:.4f}", f"loss={d['loss']:.6f}"]
    if "secondary" in d:
        s = d["secondary"]
        out += ["|", s["config"]["name"], f"{s['value']:.4g}", f"frac={s['roofline']['frac']:.4f}"]
    print(" ".join(str(x) for x in out))

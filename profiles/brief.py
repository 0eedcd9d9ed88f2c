#!/usr/bin/env python3
"""One-line summary of bench.py JSON lines: `brief.py FILE...` (the file name is the tag), or from stdin with an optional tag
as argv[1].  An argument that names an existing file is always read as a file (never waits on stdin)."""
import json
import os
import sys


def lines():
    files = [a for a in sys.argv[1:] if os.path.isfile(a)]
    if files:
        for f in files:
            for line in open(f):
                yield os.path.basename(f), line
    else:
        t = sys.argv[1] if len(sys.argv) > 1 else ""
        for line in sys.stdin:
            yield t, line


for tag, line in lines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    out = [tag, d["config"]["name"], f"{d['value']:.4g}", f"frac={r['frac']:.4f}", f"kernel_ms={r['kernel_ms']:.4f}", f"loss={d['loss']:.6f}"]
    if "secondary" in d:
        s = d["secondary"]   # (round 6: the stdout digest carries {name, ms, frac, value, ...}; the full record has config / roofline)
        name = s["config"]["name"] if "config" in s else s.get("name", "?")
        frac = s["roofline"]["frac"] if "roofline" in s else s.get("frac", float("nan"))
        out += ["|", name, f"{s['value']:.4g}", f"frac={frac:.4f}"]
    print(" ".join(str(x) for x in out))

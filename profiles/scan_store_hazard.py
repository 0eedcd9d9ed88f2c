#!/usr/bin/env python3
"""Static check of every gfx950 code object of the library: no 96- / 128-bit vector-memory store may be followed DIRECTLY by a VALU
instruction that writes one of the store's data registers.

Why: such a store reads its data registers after it has issued ("VMEM store more than 8 bytes followed by a write of the VGPRs
holding the write data": one wait state, CDNA3 ISA guide, manually inserted wait states).  The compiler provides the wait state for
the stores it understands, but for buffer stores with the offset in a scalar register on gfx950 it did not: the 8-tile instances of
cnf_coop_grad.hip zeroed an accumulator in the register that held the fourth dword of the last operand store, and under memory
back-pressure the store wrote that zero (round 4: layer-1 cotangent off by 1e-3 at more workgroups than compute units; fixed by
CNF_STORE_DATA_HAZARD in csrc/cnf_coop_dev.h).  Usage: python profiles/scan_store_hazard.py [dir with .o files]; exit code 1 on a hit.
"""
import glob, os, re, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(.*)")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_code_object(path):
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True).stdout.split("\n")
    ins, func = [], None
    for l in out:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            func = m.group(1)
            continue
        t = l.split("//")[0].strip()
        if t:
            ins.append((func, t))
    hits = []
    for i, (fn, t) in enumerate(ins[:-1]):
        m = STORE.match(t)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        data = regs(ops[0]) if m.group(1).startswith("buffer") else (regs(ops[1]) if len(ops) > 1 else set())
        nxt = ins[i + 1][1]
        mm = re.match(r"^(v_\w+)\s+([^,]+)", nxt)
        if mm and not mm.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and regs(mm.group(2).strip()) & data:
            hits.append((fn, t, nxt))
    return hits


def scan_objects(obj_dir):
    """-> (number of code objects scanned, list of hits)"""
    hits, n = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(glob.glob(os.path.join(obj_dir, "*.o"))):
            # llvm-objdump --offloading writes the bundled device code objects next to the input: work on a copy
            c = os.path.join(tmp, os.path.basename(o))
            os.symlink(os.path.abspath(o), c)
            subprocess.run([OBJDUMP, "--offloading", c], capture_output=True, cwd=tmp)
            for co in glob.glob(c + ".*gfx950"):
                n += 1
                hits += [(os.path.basename(o),) + h for h in scan_code_object(co)]
    return n, hits


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "continuousnormalizingflows.jl_amd", "csrc")
    n, hits = scan_objects(d)
    for h in hits[:20]:
        print("HAZARD", h)
    print(f"{n} code objects scanned, {len(hits)} stores followed directly by a write of their data registers")
    sys.exit(1 if hits else 0)

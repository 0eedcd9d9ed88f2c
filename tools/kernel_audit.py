#!/usr/bin/env python3
"""Per-kernel register / spill / scratch audit of a built object: `python tools/kernel_audit.py csrc/cnf_coop_grad3.o [filter]`.
Reads the AMDGPU metadata notes of the embedded gfx950 code object (llvm-readelf --notes) and counts a few instruction classes in
its disassembly (v_readlane = scalar-spill reloads, v_accvgpr moves, scratch accesses, MFMAs)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract(obj, tmp):
    # llvm-objdump --offloading writes the bundled device code object next to (a copy of) the input
    import glob
    import shutil
    c = shutil.copy(obj, tmp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", c], capture_output=True, cwd=tmp)
    return glob.glob(os.path.join(tmp, "*gfx950*"))[0]


def main():
    obj = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        dev = extract(obj, tmp)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", dev], text=True)
        kernels = {}
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            d = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in
                 ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
            d["agpr_count"] = int(blk.split()[0])
            kernels[name] = d
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", dev], text=True)
        cur = None
        counts = {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                counts[cur] = dict(mfma=0, readlane=0, writelane=0, accmov=0, scratch=0, vmov=0, branch=0, total=0)
                continue
            if cur is None or not line.startswith("\t"):
                continue
            toks = line.split()
            ins = toks[0] if toks else ""
            c = counts[cur]
            c["total"] += 1
            if ins.startswith("v_mfma"): c["mfma"] += 1
            elif ins.startswith("v_readlane"): c["readlane"] += 1
            elif ins.startswith("v_writelane"): c["writelane"] += 1
            elif ins.startswith("v_accvgpr"): c["accmov"] += 1
            elif ins.startswith("scratch_"): c["scratch"] += 1
            elif ins.startswith("v_mov_b32") or ins.startswith("v_mov_b64"): c["vmov"] += 1
            elif ins.startswith("s_cbranch") or ins.startswith("s_branch"): c["branch"] += 1
        for name, d in kernels.items():
            dem = subprocess.check_output(["c++filt", name], text=True).strip()
            if flt and flt not in dem:
                continue
            c = counts.get(name, {})
            print(f"{dem[:110]}\n    vgpr {d['vgpr_count']} agpr {d['agpr_count']} sgpr {d['sgpr_count']} vspill {d['vgpr_spill_count']} sspill {d['sgpr_spill_count']} "
                  f"scratch {d['private_segment_fixed_size']} B | insts {c.get('total')} mfma {c.get('mfma')} readlane {c.get('readlane')} writelane {c.get('writelane')} "
                  f"accmov {c.get('accmov')} scratch-ops {c.get('scratch')} v_mov {c.get('vmov')} branches {c.get('branch')}")


if __name__ == "__main__":
    main()

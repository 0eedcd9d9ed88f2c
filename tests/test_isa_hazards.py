"""Static ISA check of the built code objects (no GPU): see profiles/scan_store_hazard.py."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def test_no_wide_store_is_followed_directly_by_a_write_of_its_data_registers():
    """A 96- / 128-bit vector-memory store reads its data registers after it has issued; the instruction behind it must not write
    them (one wait state).  The compiler did not provide it for buffer stores with a scalar offset on gfx950, and the cooperative
    reverse sweep's 8-tile instances lost entries of an operand array to it (round 4).  Every code object of the library is
    disassembled and scanned; CNF_STORE_DATA_HAZARD (csrc/cnf_coop_dev.h) is what keeps the count at zero."""
    import scan_store_hazard as S
    obj_dir = os.path.join(ROOT, "continuousnormalizingflows.jl_amd", "csrc")
    if not glob.glob(os.path.join(obj_dir, "*.o")) or not os.path.exists(S.OBJDUMP):
        pytest.skip("object files of the library not in the tree (built by __graft_entry__.build())")
    n, hits = S.scan_objects(obj_dir)
    assert n >= 20, n
    assert not hits, hits[:5]

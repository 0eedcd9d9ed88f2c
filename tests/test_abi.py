"""CPU: the C-ABI library loads, exports every symbol include/cnf.h declares, validates
configurations, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cnf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cnf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("cnf_create", "cnf_destroy", "cnf_set_params", "cnf_aug_f", "cnf_integrate_fixed",
              "cnf_inference_fixed", "cnf_loss_sums", "cnf_last_error", "cnf_version",
              "cnf_kernel_path"):
        assert s in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert set(pkg._lib.EXPORTS) == set(declared_symbols())
    assert lib.cnf_version() == 1
    info = lib.cnf_build_info().decode()          # which compiler made the code objects (recorded by the Makefile)
    assert "clang" in info.lower() and "gfx950" in info, info


def test_config_struct_matches_header(pkg):
    # 5 + 9 + 8 + 8 int32 fields
    assert C.sizeof(pkg._lib.CnfConfig) == 4 * (5 + 9 + 8 + 8)


def _cfg(pkg, **kw):
    c = pkg._lib.CnfConfig()
    c.nvars, c.naug, c.ncond, c.autonomous, c.n_layers = 2, 0, 0, 0, 2
    c.widths[0], c.widths[1], c.widths[2] = 3, 16, 2
    c.acts[0], c.acts[1] = 1, 0
    c.mode, c.nprobes = 0, 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("bad", [dict(nvars=0), dict(n_layers=0), dict(n_layers=9), dict(mode=7),
                                 dict(nprobes=0), dict(naug=1), dict(kernel_path=5)])
def test_create_rejects_inconsistent_config(pkg, bad):
    lib = pkg._lib.load()
    h = C.c_void_p()
    rc = lib.cnf_create(C.byref(h), C.byref(_cfg(pkg, **bad)))
    assert rc == pkg._lib.ERR_INVALID
    assert lib.cnf_last_error().decode().startswith("cnf_create")
    assert not h.value


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_gpu_means_loud_failure_not_fallback(pkg):
    lib = pkg._lib.load()
    h = C.c_void_p()
    rc = lib.cnf_create(C.byref(h), C.byref(_cfg(pkg)))
    assert rc == pkg._lib.ERR_NO_DEVICE
    assert "no CPU fallback" in lib.cnf_last_error().decode()
    with pytest.raises(pkg._lib.CnfError):
        pkg._lib.check(rc)


def test_null_arguments_are_errors_not_crashes(pkg):
    lib = pkg._lib.load()
    assert lib.cnf_create(None, None) == pkg._lib.ERR_INVALID
    assert lib.cnf_destroy(None) == 0
    assert lib.cnf_aug_f(None, None, None, 0.0, None, None, 4, None) == pkg._lib.ERR_INVALID
    assert lib.cnf_inference_fixed(None, 0, 1, 0.0, 1.0, None, None, None, 4, None, None, None,
                                   None) == pkg._lib.ERR_INVALID


def test_comm_entry_points_fail_cleanly_without_a_device(pkg):
    """The RCCL entry points validate their arguments and report through status codes (no GPU here: cnf_comm_init must
    fail with NO_DEVICE or COMM, never crash); the unique id is 128 bytes as include/cnf.h says."""
    lib = pkg._lib.load()
    hdr = open(os.path.join(ROOT, "include", "cnf.h")).read()
    assert "#define CNF_COMM_ID_BYTES 128" in hdr and pkg._lib.COMM_ID_BYTES == 128
    assert lib.cnf_comm_unique_id(None) == pkg._lib.ERR_INVALID
    assert lib.cnf_comm_destroy(None) == 0
    assert lib.cnf_allreduce_loss(None, None, 0, None, None) == pkg._lib.ERR_INVALID
    assert lib.cnf_allreduce_sum(None, None, 0, 0, None) == pkg._lib.ERR_INVALID
    c = C.c_void_p()
    buf = (C.c_char * 128)()
    assert lib.cnf_comm_init(C.byref(c), 2, 2, buf, 0) == pkg._lib.ERR_INVALID      # rank out of range
    if not torch.cuda.is_available():
        assert lib.cnf_comm_init(C.byref(c), 0, 1, buf, 0) in (pkg._lib.ERR_NO_DEVICE, pkg._lib.ERR_COMM)
        assert not c.value


def test_cpp_host_example_compiles_against_the_header(tmp_path):
    """examples/abi_demo.cpp (a torch-free C++/HIP host on the C ABI) must keep compiling against include/cnf.h."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-c",
                    os.path.join(ROOT, "examples", "abi_demo.cpp"), "-o", str(tmp_path / "abi_demo.o")], check=True, timeout=300)


def test_the_switchboard_is_the_only_reader_of_the_environment(pkg, monkeypatch):
    """include/cnf.h: cnf_tuning - every A/B and test switch in one documented struct, read from the CNF_* variables in ONE place
    (csrc/cnf_tuning.hip: once, at the board's first use, and on cnf_set_tuning(NULL) - creating a handle never touches the board) and changed through cnf_set_tuning."""
    import glob
    import os
    csrc = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "csrc")
    readers = [os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))
               if "getenv" in open(f).read()]
    assert sorted(readers) == ["cnf_tuning.hip"], readers
    # every field of the header's struct is bound, documented with its variable, and round-trips (no GPU needed)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(csrc)), "include", "cnf.h")).read()
    body = re.search(r"typedef struct cnf_tuning \{(.*?)\} cnf_tuning;", hdr, re.S).group(1)
    fields = re.findall(r"int32_t (\w+);\s*/\* (CNF_[A-Z0-9_]+), default (-?\d+):", body)
    assert [f for f, _, _ in fields] == list(pkg._lib.TUNING_FIELDS) and len(fields) >= 30
    integration = open(os.path.join(os.path.dirname(os.path.dirname(csrc)), "INTEGRATION.md")).read()
    for f, env, _ in fields:
        assert env == "CNF_" + f.upper(), (f, env)
        assert env in integration, f"{env} is not described in INTEGRATION.md"
        monkeypatch.delenv(env, raising=False)
    base = pkg.reload_tuning()
    assert base == {f: int(d) for f, _, d in fields}
    old = pkg.set_tuning(tile_split=0, lg_nw=8)
    assert old == {"tile_split": 1, "lg_nw": 4} and pkg.get_tuning()["tile_split"] == 0 and pkg.get_tuning()["lg_nw"] == 8
    monkeypatch.setenv("CNF_COOPD", "2")
    assert pkg.reload_tuning()["coopd"] == 2 and pkg.get_tuning()["tile_split"] == 1      # reload = defaults + environment
    monkeypatch.delenv("CNF_COOPD")
    assert pkg.reload_tuning() == base

"""ctypes binding of libcnf_hip.so — the C ABI declared in include/cnf.h.

The product path has no CPU fallback: if the shared library is missing or cannot be loaded
this module raises, and every compute entry point of the library itself returns an error when
no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# CNF_HIP_LIB overrides the library path (same variable as the Julia binding, julia/hip_ext/libcnf.jl): A/B builds
LIB_PATH = os.environ.get("CNF_HIP_LIB") or os.path.join(_PKG, "libcnf_hip.so")
CSRC = os.path.join(_PKG, "csrc")
MAX_LAYERS = 8

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_NO_PARAMS, ERR_HIP, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5
ACT_IDENTITY, ACT_TANH, ACT_SOFTPLUS = 0, 1, 2
MODE_HUTCH_VJP, MODE_HUTCH_JVP, MODE_EXACT = 0, 1, 2
ALG_RK4, ALG_TSIT5 = 0, 1
ALG_VCABM = 2   # CNF_ALG_VCABM: taken by cnf_loss_adaptive only - the multistep solve has its own entry points (cnf_vcabm_*, cnf_solve_vcabm)
VCABM_MAX_ORDER = 12
PATH_AUTO, PATH_SIMT, PATH_MFMA, PATH_LAYERED = 0, 1, 2, 3
STEP_FSAL, STEP_RETRY = 1, 2
ARITH_F32, ARITH_BF16X6 = 0, 1
ERR_COMM = -6
COMM_ID_BYTES = 128
DTYPE_F32, DTYPE_F64 = 0, 1

EXPORTS = ("cnf_version", "cnf_build_info", "cnf_get_tuning", "cnf_set_tuning", "cnf_last_error", "cnf_create", "cnf_destroy", "cnf_set_params",
           "cnf_kernel_path", "cnf_solve_controller", "cnf_repack_on_device", "cnf_aug_f", "cnf_integrate_fixed", "cnf_inference_fixed",
           "cnf_loss_sums", "cnf_loss_mean", "cnf_loss_grad_fixed", "cnf_loss_grad_grid", "cnf_grad_path", "cnf_step_embedded", "cnf_assemble_u0",
           "cnf_epilogue", "cnf_vcabm_begin", "cnf_vcabm_attempt", "cnf_vcabm_accept", "cnf_vcabm_state", "cnf_solve_vcabm", "cnf_solve_tsit5", "cnf_loss_adaptive", "cnf_loss_grad_adaptive",
           "cnf_integrate_fixed_dt", "cnf_inference_fixed_dt",
           "cnf_comm_unique_id", "cnf_comm_init", "cnf_comm_init_all", "cnf_comm_destroy", "cnf_comm_rank", "cnf_comm_size",
           "cnf_comm_group_start", "cnf_comm_group_end", "cnf_allreduce_loss", "cnf_allreduce_sum",
           "cnf_kernel_family", "cnf_kernel_family_for", "cnf_kernel_name", "cnf_grad_path_for", "cnf_grad_form_for")
FAMILY_SIMT, FAMILY_PER_WAVE, FAMILY_COOP, FAMILY_COOPX, FAMILY_TILE_SPLIT, FAMILY_LAYERED, FAMILY_COOPD = 0, 1, 2, 3, 4, 5, 6
FAMILY_NAMES = ("simt", "per_wave", "coop", "coopx", "tile_split", "layered", "coopd")


class CnfConfig(C.Structure):
    _fields_ = [("nvars", C.c_int32), ("naug", C.c_int32), ("ncond", C.c_int32),
                ("autonomous", C.c_int32), ("n_layers", C.c_int32),
                ("widths", C.c_int32 * (MAX_LAYERS + 1)), ("acts", C.c_int32 * MAX_LAYERS),
                ("mode", C.c_int32), ("nprobes", C.c_int32),
                ("reg_z", C.c_int32), ("reg_j", C.c_int32), ("reg_aug", C.c_int32),
                ("device_id", C.c_int32), ("kernel_path", C.c_int32), ("arith", C.c_int32)]


TUNING_FIELDS = ("tile_split", "solve2", "solve2_pair", "coopd", "coopd_grad", "coop_grad", "coop_grad_mid", "coop_grad3", "coop_grad3_gib", "grad_layered", "jvp_grad_twin", "probe_grad_twin", "adaptive_ckpt", "layered_loss_by_solve", "device_controller", "dc_per_cu", "mfma_coop", "mfma_coopx", "mfma_nt", "mfma_pre", "mfma_prio", "mfma_queue", "cg_one_per_cu", "layered_min_b", "layered_kc", "layered_no_kckpt", "layered_act_gib", "lg_gemm", "lg_spw", "lg_nw", "lg_gemm2_wide", "lg_wgrad_per_cu", "lg_wgrad_t1", "lg_wgrad_t2")


class CnfTuning(C.Structure):
    """cnf_tuning (include/cnf.h): the library's switchboard, one int32 per switch."""
    _fields_ = [(f, C.c_int32) for f in TUNING_FIELDS]


class SolveStats(C.Structure):
    """cnf_solve_stats (include/cnf.h)."""
    _fields_ = [("naccept", C.c_int32), ("nreject", C.c_int32), ("nf", C.c_int32), ("max_order", C.c_int32)]


class CnfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcnf_hip error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into continuousnormalizingflows.jl_amd/libcnf_hip.so."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4", "all"], stdout=out)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    # torch ships its own libamdhip64.so.7; import it first so both sides share one HIP runtime
    # (device pointers and streams are then interchangeable).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH)
    vp, fp, szp = C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)
    lib.cnf_version.restype = C.c_int
    lib.cnf_last_error.restype = C.c_char_p
    lib.cnf_create.argtypes = [C.POINTER(vp), C.POINTER(CnfConfig)]
    lib.cnf_destroy.argtypes = [vp]
    lib.cnf_set_params.argtypes = [vp, fp, C.c_size_t, szp, szp, C.c_int, vp]
    lib.cnf_kernel_path.argtypes = [vp]
    lib.cnf_solve_controller.argtypes = [vp]
    lib.cnf_repack_on_device.argtypes = [vp]
    lib.cnf_grad_path.argtypes = [vp]
    lib.cnf_grad_path_for.argtypes = [vp, C.c_int64, C.c_int, C.c_int]
    lib.cnf_grad_form_for.argtypes = [vp, C.c_int64, C.c_int, C.c_int, C.c_int]
    lib.cnf_kernel_family.argtypes = [vp]
    lib.cnf_kernel_family_for.argtypes = [vp, C.c_int64, C.c_int]
    lib.cnf_kernel_name.argtypes = [vp]
    lib.cnf_kernel_name.restype = C.c_char_p
    lib.cnf_get_tuning.argtypes = [C.POINTER(CnfTuning)]
    lib.cnf_set_tuning.argtypes = [C.POINTER(CnfTuning)]
    lib.cnf_build_info.argtypes = []
    lib.cnf_build_info.restype = C.c_char_p
    lib.cnf_loss_grad_grid.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), fp, fp, fp, C.c_int64,
                                       C.POINTER(C.c_float), fp, fp, fp, vp]
    lib.cnf_step_embedded.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp, C.c_int64, C.c_float,
                                      C.c_float, fp, fp, vp]
    lib.cnf_assemble_u0.argtypes = [vp, fp, C.c_int64, fp, vp]
    lib.cnf_epilogue.argtypes = [vp, fp, C.c_int64, fp, fp, vp]
    lib.cnf_vcabm_begin.argtypes = [vp, C.c_float, fp, fp, fp, C.c_int64, vp]
    lib.cnf_vcabm_attempt.argtypes = [vp, C.c_int, C.c_float, fp, fp, C.c_int64, C.c_float, C.c_float, fp, vp]
    lib.cnf_vcabm_accept.argtypes = [vp, fp, fp, C.c_int64, C.c_float, C.c_float, fp, vp]
    lib.cnf_vcabm_state.argtypes = [vp, C.c_int64, fp, C.POINTER(C.c_double), vp]
    lib.cnf_solve_tsit5.argtypes = [vp, C.c_float, C.c_float, fp, fp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                    fp, C.POINTER(SolveStats), C.POINTER(C.c_float), C.c_int32, vp]
    lib.cnf_loss_grad_adaptive.argtypes = [vp, C.c_float, C.c_float, fp, fp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                           C.POINTER(C.c_float), fp, fp, fp, C.POINTER(SolveStats), C.POINTER(C.c_float), C.c_int32, vp]
    lib.cnf_solve_vcabm.argtypes = [vp, C.c_float, C.c_float, fp, fp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                    fp, C.POINTER(SolveStats), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32, vp]
    lib.cnf_loss_adaptive.argtypes = [vp, C.c_int, C.c_float, C.c_float, fp, fp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                      C.POINTER(C.c_double), fp, fp, fp, fp, C.POINTER(SolveStats), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                      C.c_int32, vp]
    lib.cnf_aug_f.argtypes = [vp, fp, fp, C.c_float, fp, fp, C.c_int64, vp]
    lib.cnf_integrate_fixed.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp,
                                        C.c_int64, fp, vp]
    lib.cnf_inference_fixed.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp,
                                        C.c_int64, fp, fp, fp, vp]
    lib.cnf_integrate_fixed_dt.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, fp, fp, fp, C.c_int64, fp, vp]
    lib.cnf_inference_fixed_dt.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, fp, fp, fp, C.c_int64, fp, fp, fp, vp]
    lib.cnf_loss_sums.argtypes = [vp, fp, fp, C.c_int64, fp, vp]
    lib.cnf_loss_mean.argtypes = [vp, fp, fp, C.c_int64, C.POINTER(C.c_double), fp, fp, vp]
    lib.cnf_loss_grad_fixed.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp, C.c_int64,
                                        C.POINTER(C.c_float), fp, fp, fp, vp]
    lib.cnf_comm_unique_id.argtypes = [vp]
    lib.cnf_comm_init.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int]
    lib.cnf_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    lib.cnf_comm_destroy.argtypes = [vp]
    lib.cnf_comm_rank.argtypes = [vp]
    lib.cnf_comm_size.argtypes = [vp]
    lib.cnf_allreduce_loss.argtypes = [vp, fp, C.c_int64, fp, vp]
    lib.cnf_allreduce_sum.argtypes = [vp, fp, C.c_size_t, C.c_int, vp]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the ABI is incomplete
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise CnfError(rc, load().cnf_last_error().decode())


def stream_ptr(device) -> C.c_void_p:
    """hipStream_t of torch's current stream on `device`, as the void* the ABI takes."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a tensor; None -> NULL."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def clamp_maxiters(sol_kwargs) -> int:
    """sol_kwargs.maxiters as the C int the ABI takes (the reference's usage example passes typemax(Int))."""
    return min(int(sol_kwargs.get("maxiters", 100000)), 2 ** 31 - 1)


def get_tuning() -> dict:
    """The switchboard's current values (cnf_get_tuning)."""
    t = CnfTuning()
    rc = load().cnf_get_tuning(C.byref(t))
    if rc:
        raise CnfError(rc, "cnf_get_tuning")
    return {f: int(getattr(t, f)) for f in TUNING_FIELDS}


def set_tuning(**kw) -> dict:
    """Change switches of the library's process-wide switchboard (cnf_set_tuning); returns the previous values of the ones
    changed.  Note that cnf_create re-reads the CNF_* environment variables: set switches AFTER the handles they concern exist."""
    t = CnfTuning()
    lib = load()
    rc = lib.cnf_get_tuning(C.byref(t))
    if rc:
        raise CnfError(rc, "cnf_get_tuning")
    old = {}
    for k, v in kw.items():
        if k not in TUNING_FIELDS:
            raise KeyError(f"cnf_tuning has no field {k!r}")
        old[k] = int(getattr(t, k))
        setattr(t, k, int(v))
    rc = lib.cnf_set_tuning(C.byref(t))
    if rc:
        raise CnfError(rc, "cnf_set_tuning")
    return old


def reload_tuning() -> dict:
    """Re-read the switchboard from its defaults and the CNF_* environment variables (cnf_set_tuning(NULL); creating a handle does not),
    so that a variable changed after a handle was created takes effect for it too."""
    rc = load().cnf_set_tuning(None)
    if rc:
        raise CnfError(rc, "cnf_set_tuning")
    return get_tuning()

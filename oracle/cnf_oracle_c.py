"""ctypes binding of oracle/libcnf_oracle.so (CPU fp32 restatement).  TEST INFRASTRUCTURE ONLY.

Parity unpinned by the reference (see cnf_oracle.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this.  Matrices are shaped like their Julia
counterparts, (rows, B); they are converted to the column-major memory layout internally.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcnf_oracle.so")
MAX_LAYERS = 8


class Cfg(C.Structure):
    _fields_ = [("nvars", C.c_int32), ("naug", C.c_int32), ("ncond", C.c_int32),
                ("autonomous", C.c_int32), ("n_layers", C.c_int32),
                ("widths", C.c_int32 * (MAX_LAYERS + 1)), ("acts", C.c_int32 * MAX_LAYERS),
                ("mode", C.c_int32), ("nprobes", C.c_int32),
                ("reg_z", C.c_int32), ("reg_j", C.c_int32), ("reg_aug", C.c_int32)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cnf_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "clean", "all"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        fp, szp = C.POINTER(C.c_float), C.POINTER(C.c_size_t)
        _lib.cnf_oracle_aug_f.argtypes = [C.POINTER(Cfg), fp, szp, szp, fp, C.c_float, fp, fp,
                                          C.c_int64, fp, C.c_int]
        _lib.cnf_oracle_integrate_fixed.argtypes = [C.POINTER(Cfg), fp, szp, szp, C.c_int, C.c_int,
                                                    C.c_float, C.c_float, fp, fp, fp, C.c_int64,
                                                    fp, C.c_int]
        _lib.cnf_oracle_inference_fixed.argtypes = [C.POINTER(Cfg), fp, szp, szp, C.c_int, C.c_int,
                                                    C.c_float, C.c_float, fp, fp, fp, C.c_int64,
                                                    fp, fp, fp, C.c_int]
    return _lib


def set_fast_tanh(on: bool) -> None:
    lib().cnf_oracle_set_fast_tanh(int(bool(on)))


def isa() -> str:
    """The vector ISA the block products run on (selected at load time from cpuid)."""
    return {2: "avx512", 1: "avx2"}.get(int(lib().cnf_oracle_isa()), "scalar")


def max_threads() -> int:
    return int(lib().cnf_oracle_max_threads())


def _cfg(spec) -> Cfg:
    c = Cfg()
    c.nvars, c.naug, c.ncond, c.autonomous = spec.nvars, spec.naug, spec.ncond, int(spec.autonomous)
    c.n_layers = len(spec.acts)
    for i, w in enumerate(spec.widths):
        c.widths[i] = w
    for i, a in enumerate(spec.acts):
        c.acts[i] = a
    c.mode, c.nprobes = spec.mode, spec.nprobes
    c.reg_z, c.reg_j, c.reg_aug = int(spec.reg_z), int(spec.reg_j), int(spec.reg_aug)
    return c


def _cm(a: Optional[np.ndarray]):
    """(rows,B) array -> float32 memory in Julia column-major order, + ctypes pointer."""
    if a is None:
        return None, None
    m = np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)
    return m, m.ctypes.data_as(C.POINTER(C.c_float))


def _offs(spec):
    w, b, _ = spec.param_offsets()
    return (C.c_size_t * len(w))(*w), (C.c_size_t * len(b))(*b)


def aug_f(spec, p, u, t, eps, ys, nthreads=1):
    S, B = u.shape
    pm = np.ascontiguousarray(p, dtype=np.float32)
    um, up = _cm(u); em, ep = _cm(eps); ym, yp = _cm(ys)
    du = np.empty((B, S), dtype=np.float32)
    w, b = _offs(spec)
    rc = lib().cnf_oracle_aug_f(C.byref(_cfg(spec)), pm.ctypes.data_as(C.POINTER(C.c_float)), w, b,
                                up, float(t), ep, yp, B, du.ctypes.data_as(C.POINTER(C.c_float)),
                                nthreads)
    if rc:
        raise RuntimeError(f"cnf_oracle_aug_f rc={rc}")
    return du.T


def integrate_fixed(spec, p, u0, t0, t1, nsteps, alg, eps, ys, nthreads=1):
    S, B = u0.shape
    pm = np.ascontiguousarray(p, dtype=np.float32)
    um, up = _cm(u0); em, ep = _cm(eps); ym, yp = _cm(ys)
    u1 = np.empty((B, S), dtype=np.float32)
    w, b = _offs(spec)
    rc = lib().cnf_oracle_integrate_fixed(C.byref(_cfg(spec)), pm.ctypes.data_as(C.POINTER(C.c_float)),
                                          w, b, alg, nsteps, t0, t1, up, ep, yp, B,
                                          u1.ctypes.data_as(C.POINTER(C.c_float)), nthreads)
    if rc:
        raise RuntimeError(f"cnf_oracle_integrate_fixed rc={rc}")
    return u1.T


def inference_fixed(spec, p, xs, t0, t1, nsteps, alg, eps, ys=None, nthreads=1):
    nv, B = xs.shape
    pm = np.ascontiguousarray(p, dtype=np.float32)
    xm, xp = _cm(xs); em, ep = _cm(eps); ym, yp = _cm(ys)
    logp = np.empty(B, dtype=np.float32)
    regs = np.empty((3, B), dtype=np.float32)
    uf = np.empty((B, spec.S), dtype=np.float32)
    w, b = _offs(spec)
    fp = C.POINTER(C.c_float)
    rc = lib().cnf_oracle_inference_fixed(C.byref(_cfg(spec)), pm.ctypes.data_as(fp), w, b, alg,
                                          nsteps, t0, t1, xp, ep, yp, B, logp.ctypes.data_as(fp),
                                          regs.ctypes.data_as(fp), uf.ctypes.data_as(fp), nthreads)
    if rc:
        raise RuntimeError(f"cnf_oracle_inference_fixed rc={rc}")
    return logp, (regs[0], regs[1], regs[2]), uf.T

"""The host side of the adaptive solvers (`SciMLBase.solve` in `base_sol`, src/core/base_icnf.jl:134-140): OrdinaryDiffEq's
step-size (and, for VCABM, order) policies over the device passes of the C ABI.  A single-process solve runs the same
policies inside the library (`cnf_solve_vcabm`, `cnf_solve_tsit5`: one call per solve); the loops here are what a sharded
solve runs, all-reducing the error sums between the passes so that every rank takes the steps of the unsharded solve
(`icnf.adaptive_policy = "python"` forces them for one process - the two are tested to agree bit for bit)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from ._lib import ptr as _ptr, stream_ptr as _stream_ptr


# get_current_alg_order(VCABM(), cache) when the initial step is chosen: the cache starts at order 1
VCABM_INITDT_ORDER = 1.0


def _solve_errors(call):
    """Run a whole-solve library call; its status codes become the exceptions of the host loops."""
    try:
        _lib.check(call())
    except _lib.CnfError as err:
        if "maxiters reached" in str(err):
            raise RuntimeError("adaptive solve: maxiters reached") from None
        if "non-finite" in str(err):
            raise FloatingPointError("adaptive solve: " + str(err).split(": ", 2)[-1]) from None
        raise


def _adaptive_integrate(icnf, h, u0: torch.Tensor, t0: float, t1: float,
                        e: Optional[torch.Tensor], y: Optional[torch.Tensor], group=None, _tsit5: bool = False, _sp=None) -> torch.Tensor:
    """Adaptive Tsit5 from t0 to t1 (either direction) on the (B, S) state u0: what `SciMLBase.solve(prob, Tsit5();
    reltol, abstol)` does in `base_sol` (src/core/base_icnf.jl:134-140), restated from OrdinaryDiffEq's documented
    algorithm — Hairer's initial step, embedded 4th-order error estimate scaled by `abstol + reltol max(|u_prev|, |u|)`
    under the RMS norm over the WHOLE state, PI controller (beta1 = 7/50, beta2 = 2/25, gamma = 9/10, qmin = 1/5,
    qmax = 10, qoldinit = 1e-4), first-same-as-last.  Every attempt is one `cnf_step_embedded` call; the host only
    handles the controller's scalars.  The error norm couples all columns, so with torch.distributed initialised
    the squared sum and the element count are all-reduced and every rank takes the same steps.  The step sequence
    of the Julia implementation cannot be checked here (no Julia); parity is against the fp64 oracle's restatement
    of the same algorithm and against fine fixed-step solves."""
    from .sharding import allsum as _allsum, is_sharded
    if icnf._solver() == _lib.ALG_VCABM and not _tsit5:
        return _vcabm_integrate(icnf, h, u0, t0, t1, e, y, group=group, _sp=_sp)
    kw = icnf.sol_kwargs
    reltol, abstol = float(kw.get("reltol", 1e-4)), float(kw.get("abstol", 1e-4))
    maxiters = _lib.clamp_maxiters(kw)
    dev = icnf.device
    B, S = u0.shape
    lib, st = h.lib, (_sp if _sp is not None else _stream_ptr(dev))
    sharded = is_sharded(group)     # group=False: a rank-local solve, no collectives
    tdir = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
    stats = {"naccept": 0, "nreject": 0, "nf": 0, "dts": [], "alg_used": "Tsit5"}
    icnf.last_solve_stats = stats
    # an EMPTY shard of a sharded solve must still take part in every all-reduce (it contributes zeros and follows
    # the other ranks' steps); only an unsharded empty batch returns at once
    if span == 0.0 or (B == 0 and not sharded):
        return u0.clone()
    if not sharded and getattr(icnf, "adaptive_policy", "library") == "library":
        # single process: the same controller restated inside the library (cnf_solve_tsit5), one call per solve
        cap = 4096
        if getattr(icnf, "_ts_records", None) is None:      # the record array of the call, made once
            icnf._ts_records = (_lib.SolveStats(), (C.c_float * cap)())
        ss, dts = icnf._ts_records
        u0 = u0.contiguous()
        out = torch.empty_like(u0)
        _solve_errors(lambda: lib.cnf_solve_tsit5(h.ptr, t0, t1, _ptr(u0), _ptr(e), _ptr(y), B, abstol, reltol,
                                                  float(kw["dt"]) if "dt" in kw else 0.0, maxiters, _ptr(out), C.byref(ss), dts, cap, st))
        stats.update(naccept=ss.naccept, nreject=ss.nreject, nf=ss.nf, dts=[float(v) for v in dts[:min(ss.naccept, cap)]],
                     controller="device" if lib.cnf_solve_controller(h.ptr) == 1 else "host")
        return out

    def allsum(vals):
        return _allsum(vals, dev, group if sharded else False)

    def f(u, t):
        du = torch.empty_like(u)
        if B:
            _lib.check(lib.cnf_aug_f(h.ptr, _ptr(du), _ptr(u), t, _ptr(e), _ptr(y), B, st))
        stats["nf"] += 1
        return du

    if "dt" in kw:
        dt = min(abs(float(kw["dt"])), span)
    else:   # ode_determine_initdt (Hairer, Noersett, Wanner I, II.4) with the RMS norm over all S*B entries
        sk = abstol + u0.abs() * reltol
        f0 = f(u0, t0)
        s0, s1, n = allsum([float(((u0 / sk).double() ** 2).sum()), float(((f0 / sk).double() ** 2).sum()), float(B * S)])
        d0, d1 = math.sqrt(s0 / n), math.sqrt(s1 / n)
        dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        dt0 = min(dt0, span)
        if not (math.isfinite(dt0) and dt0 > 0.0):
            raise FloatingPointError("adaptive solve: non-finite state or dynamics at t0 (no initial step)")
        f1 = f(u0 + tdir * dt0 * f0, t0 + tdir * dt0)
        (s2,) = allsum([float((((f1 - f0) / sk).double() ** 2).sum())])
        d2 = math.sqrt(s2 / n) / dt0
        dmax = max(d1, d2)
        dt1 = max(1e-6, dt0 * 1e-3) if dmax <= 1e-15 else 10.0 ** (-(2.0 + math.log10(dmax)) / 5.0)
        dt = min(100.0 * dt0, dt1, span)
        if not (math.isfinite(dt) and dt > 0.0):
            raise FloatingPointError("adaptive solve: non-finite state or dynamics at t0 (no initial step)")
    beta1, beta2, gamma, qmin, qmax, qold = 7.0 / 50.0, 2.0 / 25.0, 0.9, 0.2, 10.0, 1e-4
    u = u0.contiguous().clone()
    un = torch.empty_like(u)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    t, flags = t0, 0
    ntot = allsum([float(B * S)])[0] if sharded else float(B * S)
    for _ in range(maxiters):
        if abs(t1 - t) <= 1e-7 * max(1.0, span):
            break
        last = dt >= abs(t1 - t) * (1.0 - 1e-6)
        step = abs(t1 - t) if last else dt          # tstop: never step over t1
        if B:
            _lib.check(lib.cnf_step_embedded(h.ptr, _lib.ALG_TSIT5, flags, t, tdir * step, _ptr(u), _ptr(e), _ptr(y), B,
                                             abstol, reltol, _ptr(un), _ptr(err), st))
        stats["nf"] += 6 if flags else 7
        (ssq,) = allsum([float(err.item())])
        eest = math.sqrt(ssq / ntot)
        if not math.isfinite(eest):
            raise FloatingPointError("adaptive solve: non-finite error estimate (unstable dynamics)")
        q11 = eest ** beta1 if eest > 0.0 else 0.0
        q = 1.0 / qmax if eest == 0.0 else max(1.0 / qmax, min(1.0 / qmin, (q11 / qold ** beta2) / gamma))
        if eest <= 1.0:   # accept
            t = t1 if last else t + tdir * step
            u, un = un, u
            stats["naccept"] += 1
            stats["dts"].append(tdir * step)
            qold = max(eest, 1e-4)
            dt = step / q
            flags = _lib.STEP_FSAL
        else:             # reject: same (t, u), smaller step
            stats["nreject"] += 1
            dt = step / min(1.0 / qmin, q11 / gamma)
            flags = _lib.STEP_RETRY
    else:
        raise RuntimeError("adaptive solve: maxiters reached")
    return u


def _vcabm_integrate(icnf, h, u0: torch.Tensor, t0: float, t1: float,
                     e: Optional[torch.Tensor], y: Optional[torch.Tensor], group=None, _sp=None) -> torch.Tensor:
    """`SciMLBase.solve(prob, VCABM(); reltol, abstol)` of `base_sol` (src/core/base_icnf.jl:134-140) - the reference's
    default solver - from t0 to t1 (either direction) on the (B, S) state u0.  The device keeps the multistep state and
    does the PECE passes (`cnf_vcabm_*`, csrc/cnf_vcabm.hip); this loop is the host side of the solver: order ramp
    1 -> 3 over the first steps, then Shampine-Gordon order selection from the error estimates of orders k-2 .. k+1, the
    integral step-size controller dt / clamp(EEst^(1/(k+1)) / gamma, 1/qmax, 1/qmin) (gamma = 9/10, qmin = 1/5,
    qmax = 10; the same factor after a rejection) and Hairer's initial step.  Restated from the published algorithm
    (see oracle/cnf_oracle64.py::integrate_vcabm, the fp64 oracle this is tested against); the Julia package's exact
    step / order sequence cannot be checked here.  A single-process solve runs the same policy inside the library
    (`cnf_solve_vcabm`, one call per solve; `icnf.adaptive_policy = "python"` keeps this loop - the two are tested to take
    identical steps).  The error norms run over the whole S x B state, so a sharded
    solve all-reduces the squared sums and every rank takes the same steps."""
    from .sharding import allsum as _allsum, is_sharded
    kw = icnf.sol_kwargs
    reltol, abstol = float(kw.get("reltol", 1e-4)), float(kw.get("abstol", 1e-4))
    maxiters = _lib.clamp_maxiters(kw)
    dev = icnf.device
    B, S = u0.shape
    lib, st = h.lib, (_sp if _sp is not None else _stream_ptr(dev))
    sharded = is_sharded(group)     # group=False: a rank-local solve, no collectives
    tdir = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
    stats = {"naccept": 0, "nreject": 0, "nf": 0, "dts": [], "orders": [], "alg_used": "VCABM"}
    icnf.last_solve_stats = stats
    # an empty shard of a sharded solve still joins every all-reduce (zeros) and follows the other ranks' steps
    if span == 0.0 or (B == 0 and not sharded):
        return u0.clone()

    def allsum(vals):
        return _allsum(vals, dev, group if sharded else False)

    u0 = u0.contiguous()
    if not sharded and getattr(icnf, "adaptive_policy", "library") == "library":
        # single process: the same policy restated inside the library (cnf_solve_vcabm), one call per solve
        cap = 4096
        if getattr(icnf, "_vc_records", None) is None:      # the record arrays of the call, made once (two 16 KB ctypes arrays cost 10 us)
            icnf._vc_records = (_lib.SolveStats(), (C.c_float * cap)(), (C.c_int32 * cap)())
        ss, dts, orders = icnf._vc_records
        out = torch.empty_like(u0)
        _solve_errors(lambda: lib.cnf_solve_vcabm(h.ptr, t0, t1, _ptr(u0), _ptr(e), _ptr(y), B, abstol, reltol,
                                                  float(kw["dt"]) if "dt" in kw else 0.0, maxiters, _ptr(out), C.byref(ss), dts, orders, cap, st))
        m = min(ss.naccept, cap)
        stats.update(naccept=ss.naccept, nreject=ss.nreject, nf=ss.nf, dts=[float(v) for v in dts[:m]], orders=[int(v) for v in orders[:m]],
                     controller="device" if lib.cnf_solve_controller(h.ptr) == 1 else "host")
        return out
    if B:
        _lib.check(lib.cnf_vcabm_begin(h.ptr, t0, _ptr(u0), _ptr(e), _ptr(y), B, st))
    stats["nf"] += 1
    ntot = allsum([float(B * S)])[0] if sharded else float(B * S)
    if "dt" in kw:
        dt = min(abs(C.c_float(float(kw["dt"])).value), span)      # Float32, as the library entry takes it
    else:   # ode_determine_initdt (Hairer, Noersett, Wanner I, II.4); its last step divides by get_current_alg_order(alg, cache), which for the variable-order Adams family is the CURRENT order of the cache = 1 at the start (the +1 / 'order 7' of round 1 had no source; recalled, not read - no Julia here)
        du = torch.empty_like(u0)

        def f(u, t):
            if B:
                _lib.check(lib.cnf_aug_f(h.ptr, _ptr(du), _ptr(u), t, _ptr(e), _ptr(y), B, st))
            stats["nf"] += 1
            return du.clone()

        sk = abstol + u0.abs() * reltol
        f0 = f(u0, t0)
        s0, s1 = allsum([float(((u0 / sk).double() ** 2).sum()), float(((f0 / sk).double() ** 2).sum())])
        d0, d1 = math.sqrt(s0 / ntot), math.sqrt(s1 / ntot)
        dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        dt0 = min(dt0, span)
        if not (math.isfinite(dt0) and dt0 > 0.0):
            raise FloatingPointError("adaptive solve: non-finite state or dynamics at t0 (no initial step)")
        f1 = f(u0 + tdir * dt0 * f0, t0 + tdir * dt0)
        (s2,) = allsum([float((((f1 - f0) / sk).double() ** 2).sum())])
        d2 = math.sqrt(s2 / ntot) / dt0
        dmax = max(d1, d2)
        dt1 = max(1e-6, dt0 * 1e-3) if dmax <= 1e-15 else 10.0 ** (-(2.0 + math.log10(dmax)) / VCABM_INITDT_ORDER)
        dt = min(100.0 * dt0, dt1, span)
        if not (math.isfinite(dt) and dt > 0.0):
            raise FloatingPointError("adaptive solve: non-finite state or dynamics at t0 (no initial step)")
    gamma, qmin, qmax = 0.9, 0.2, 10.0
    err3 = torch.zeros(3, dtype=torch.float64, device=dev)
    errp = torch.zeros(1, dtype=torch.float64, device=dev)
    t, k, step = t0, 1, 1
    for _ in range(maxiters):
        if abs(t1 - t) <= 1e-7 * max(1.0, span):
            break
        last = dt >= abs(t1 - t) * (1.0 - 1e-6)
        hstep = abs(t1 - t) if last else dt          # tstop: never step over t1
        if B:
            _lib.check(lib.cnf_vcabm_attempt(h.ptr, k, tdir * hstep, _ptr(e), _ptr(y), B, abstol, reltol, _ptr(err3), st))
        stats["nf"] += 1
        s_k, s_km1, s_km2 = allsum(err3.tolist())
        eest = math.sqrt(s_k / ntot)
        if not math.isfinite(eest):
            raise FloatingPointError("adaptive solve: non-finite error estimate (unstable dynamics)")
        if eest > 1.0:    # reject: same state, smaller step, same order
            stats["nreject"] += 1
            dt = hstep / max(1.0 / qmax, min(1.0 / qmin, eest ** (1.0 / (k + 1)) / gamma))
            continue
        select = step > 4 and k >= 3
        lower = select and max(math.sqrt(s_km2 / ntot), math.sqrt(s_km1 / ntot)) <= eest
        want_up = select and not lower and k < _lib.VCABM_MAX_ORDER
        if B:
            _lib.check(lib.cnf_vcabm_accept(h.ptr, _ptr(e), _ptr(y), B, abstol, reltol, _ptr(errp) if want_up else None, st))
        stats["nf"] += 1
        knew = k
        if not select:
            knew = min(k + 1, 3)
        elif lower:
            knew = k - 1
        elif want_up:
            (s_kp1,) = allsum(errp.tolist())
            if math.sqrt(s_kp1 / ntot) < eest:
                knew = k + 1
                eest = 1.0     # keeps the step size (up to the safety factor) across the order change
        q = 1.0 / qmax if eest == 0.0 else max(1.0 / qmax, min(1.0 / qmin, eest ** (1.0 / (knew + 1)) / gamma))
        t = t1 if last else t + tdir * hstep
        stats["naccept"] += 1
        stats["dts"].append(tdir * hstep)
        stats["orders"].append(k)
        k, step = knew, step + 1
        dt = hstep / q
    else:
        raise RuntimeError("adaptive solve: maxiters reached")
    out = torch.empty_like(u0)
    if B:
        _lib.check(lib.cnf_vcabm_state(h.ptr, B, _ptr(out), None, st))
    return out

"""fp64 autograd oracle for the batched augmented-ODE hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED BY THE REFERENCE: impICNF/ContinuousNormalizingFlows.jl v0.31.0 is pure
Julia, there is no Julia in the build container or on the GPU box, and the reference's own
tests hold no golden vectors for this path (test/ci_tests/smoke_tests.jl:69-156 only check
`!isnothing`; regression_tests.jl:28 is `@test true`).  This file is therefore an
*independent second implementation* written from the reference's source text, not the
reference itself.  It is deliberately built on different machinery than the product
(torch.autograd VJPs, torch.func JVPs/Jacobians, float64) so that an error in the
hand-written backward pass of the C restatement / HIP kernels cannot hide in it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates (all citations relative to /root/reference):
  * MLP input rows [z; t; ys]            src/layers/cond_layer.jl:7-31, src/core/icnf.jl:147-153,
                                         src/core/base_icnf.jl:49-60
  * Dense chain  h = act(W h + b)        src/core/icnf.jl:67-71 (Lux.Chain of Lux.Dense)
  * Hutchinson VJP dynamics              src/core/icnf.jl:517-559, src/core/utils.jl:150-159
  * Hutchinson JVP dynamics              src/core/icnf.jl:561-603, src/core/utils.jl:161-170
  * exact-trace dynamics (TestMode)      src/core/icnf.jl:297-339, src/core/utils.jl:79-88
  * regularisers  |zdot|_2, |eps^T J|_2  src/core/icnf.jl:184-205, 229-251
  * state assembly u0=[x;0], logp        src/core/base_icnf.jl:247-296, 158-172
  * |z_aug|_2                            src/core/base_icnf.jl:106-132
  * loss                                 src/core/icnf.jl:628-649
  * fixed-step RK4 / Tsit5               user-supplied sol_kwargs in the reference
                                         (src/core/base_icnf.jl:138); tableaux from SURVEY.md §8 A4.

Array convention: every matrix argument is shaped like its Julia counterpart, (rows, B)
with one column per sample.  Flat parameter vector `p` uses the Lux/ComponentArrays layout:
layer_1.weight (out x in, column-major), layer_1.bias, layer_2.weight, ...
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

ACT_IDENTITY, ACT_TANH, ACT_SOFTPLUS = 0, 1, 2
MODE_HUTCH_VJP, MODE_HUTCH_JVP, MODE_EXACT = 0, 1, 2
ALG_RK4, ALG_TSIT5 = 0, 1

# Tsitouras 2011 5(4) tableau.
TSIT5_C = (0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0)
TSIT5_A = (
    (),
    (0.161,),
    (-0.008480655492356989, 0.335480655492357),
    (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
    (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
    (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
     -0.028269050394068383),
)
TSIT5_B = (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742,
           -3.290069515436081, 2.324710524099774)

RK4_C = (0.0, 0.5, 0.5, 1.0)
RK4_A = ((), (0.5,), (0.0, 0.5), (0.0, 0.0, 1.0))
RK4_B = (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0)


def tableau(alg: int):
    if alg == ALG_RK4:
        return RK4_C, RK4_A, RK4_B
    if alg == ALG_TSIT5:
        return TSIT5_C, TSIT5_A, TSIT5_B
    raise ValueError(f"unknown alg {alg}")


@dataclass
class Spec:
    """Configuration carrier; mirrors the fields of ICNF that reach the hot path
    (src/core/icnf.jl:16-141)."""
    nvars: int
    naug: int = 0
    ncond: int = 0
    autonomous: bool = False
    widths: Sequence[int] = ()      # [n_in, h1, ..., D]
    acts: Sequence[int] = ()        # one per Dense layer
    mode: int = MODE_HUTCH_VJP
    nprobes: int = 1
    reg_z: bool = False             # NORM_Z  and TrainMode{true}
    reg_j: bool = False             # NORM_J  and TrainMode{true}
    reg_aug: bool = False           # NORM_Z_AUG and AUGMENTED and TrainMode{true}

    @property
    def D(self) -> int:
        return self.nvars + self.naug

    @property
    def n_in(self) -> int:
        return self.D + (0 if self.autonomous else 1) + self.ncond

    @property
    def S(self) -> int:
        return self.D + 3

    def check(self):
        assert len(self.widths) == len(self.acts) + 1
        assert self.widths[0] == self.n_in, (self.widths, self.n_in)
        assert self.widths[-1] == self.D

    def param_offsets(self) -> Tuple[List[int], List[int], int]:
        w_off, b_off, o = [], [], 0
        for l in range(len(self.acts)):
            fin, fout = self.widths[l], self.widths[l + 1]
            w_off.append(o)
            o += fin * fout
            b_off.append(o)
            o += fout
        return w_off, b_off, o


def unpack_params(spec: Spec, p: np.ndarray, dtype=torch.float64):
    """Flat Lux blob -> [(W (out,in), b (out))]; weight stored column-major (out x in)."""
    w_off, b_off, n = spec.param_offsets()
    assert p.shape == (n,), (p.shape, n)
    out = []
    for l in range(len(spec.acts)):
        fin, fout = spec.widths[l], spec.widths[l + 1]
        W = np.asarray(p[w_off[l]:w_off[l] + fin * fout]).reshape(fin, fout).T  # (out,in)
        b = np.asarray(p[b_off[l]:b_off[l] + fout])
        out.append((torch.tensor(np.ascontiguousarray(W), dtype=dtype),
                    torch.tensor(b, dtype=dtype)))
    return out


def _act(a: torch.Tensor, kind: int) -> torch.Tensor:
    if kind == ACT_IDENTITY:
        return a
    if kind == ACT_TANH:
        return torch.tanh(a)
    if kind == ACT_SOFTPLUS:
        return torch.nn.functional.softplus(a)
    raise ValueError(kind)


def _net(spec: Spec, layers, z: torch.Tensor, t: float, ys: Optional[torch.Tensor]):
    """z: (D,B) -> zdot (D,B).  Input rows [z; t; ys] (cond_layer.jl:7-31)."""
    rows = [z]
    if not spec.autonomous:
        rows.append(torch.full((1, z.shape[1]), float(t), dtype=z.dtype))
    if spec.ncond:
        rows.append(ys)
    h = torch.cat(rows, dim=0)
    for (W, b), kind in zip(layers, spec.acts):
        h = _act(W @ h + b[:, None], kind)
    return h


def aug_f(spec: Spec, p: np.ndarray, u: np.ndarray, t: float, eps: Optional[np.ndarray],
          ys: Optional[np.ndarray]) -> np.ndarray:
    """One dynamics call: u (S,B) -> du (S,B) = [zdot; ldot; Edot; ndot]
    (src/core/icnf.jl:517-536 Hutchinson, :297-316 exact).  eps is (K*D, B): probe k
    occupies rows k*D:(k+1)*D.  K>1 is the build-defined extension of SURVEY.md §8(a0):
    ldot = -(1/K) sum_k <eps_k^T J, eps_k>,  ndot = (1/K) sum_k |eps_k^T J|_2."""
    spec.check()
    layers = unpack_params(spec, np.asarray(p, dtype=np.float64))
    D, B = spec.D, u.shape[1]
    z = torch.tensor(np.asarray(u[:D], dtype=np.float64), requires_grad=True)
    yt = None if ys is None else torch.tensor(np.asarray(ys, dtype=np.float64))
    f = lambda zz: _net(spec, layers, zz, t, yt)
    zdot = f(z)
    ldot = torch.zeros(B, dtype=torch.float64)
    ndot = torch.zeros(B, dtype=torch.float64)
    if spec.mode == MODE_EXACT:
        # full per-sample Jacobian, then trace (utils.jl:79-88, icnf.jl:312)
        tr = torch.zeros(B, dtype=torch.float64)
        for i in range(D):
            seed = torch.zeros_like(zdot)
            seed[i] = 1.0
            (gi,) = torch.autograd.grad(zdot, z, seed, retain_graph=True)
            tr = tr + gi[i]
        ldot = -tr
    else:
        K = spec.nprobes
        e = torch.tensor(np.asarray(eps, dtype=np.float64))
        assert e.shape == (K * D, B)
        for k in range(K):
            ek = e[k * D:(k + 1) * D]
            if spec.mode == MODE_HUTCH_VJP:
                (g,) = torch.autograd.grad(zdot, z, ek, retain_graph=True)   # eps^T J
            else:
                _, g = torch.func.jvp(f, (z.detach(),), (ek,))             # J eps
            ldot = ldot - (g * ek).sum(0) / K
            if spec.reg_j:
                ndot = ndot + torch.linalg.vector_norm(g, dim=0) / K
    Edot = torch.linalg.vector_norm(zdot, dim=0) if spec.reg_z and spec.mode != MODE_EXACT \
        else torch.zeros(B, dtype=torch.float64)
    if spec.mode == MODE_EXACT:
        ndot = torch.zeros(B, dtype=torch.float64)
    du = torch.cat([zdot.detach(), ldot[None], Edot.detach()[None], ndot[None]], dim=0)
    return du.detach().numpy()


def integrate_fixed(spec: Spec, p, u0: np.ndarray, t0: float, t1: float, nsteps: int, alg: int,
                    eps, ys) -> np.ndarray:
    """base_sol with a fixed-step explicit RK method (src/core/base_icnf.jl:134-140 with
    sol_kwargs=(alg, adaptive=false, dt)).  Stage time t_n + c_i dt, t_n = t0 + n dt."""
    c, a, b = tableau(alg)
    dt = (t1 - t0) / nsteps
    u = np.asarray(u0, dtype=np.float64).copy()
    for n in range(nsteps):
        tn = t0 + n * dt
        ks = []
        for i in range(len(c)):
            ui = u.copy()
            for j, aij in enumerate(a[i]):
                if aij != 0.0:
                    ui += dt * aij * ks[j]
            ks.append(aug_f(spec, p, ui, tn + c[i] * dt, eps, ys))
        for bi, ki in zip(b, ks):
            u += dt * bi * ki
    return u


def fixed_dt_grid(t0: float, t1: float, dt: float):
    """The step times OrdinaryDiffEq takes with `adaptive = false, dt` on tspan (t0, t1) (base_sol,
    src/core/base_icnf.jl:134-140; what STEER's drawn t1 meets, base_icnf.jl:23-43): steps of |dt| towards t1 and a
    SHORTER LAST STEP that lands on t1 (t1 is a tstop) — unless what remains is within 100 eps(Float32) of the larger
    end point, which its floating-point fix-up treats as already being t1 (then the steps are the equal division).
    The reference's tspan and dt are Float32 (`tspan::NTuple{2, T}`, src/core/icnf.jl:22; T = Float32), so the plan is made
    from their Float32 values; a Float32 dt a hair above span / n (0.1f0 on (0, 1)) makes the n-th step overshoot t1 by less
    than that tolerance, and the same fix-up snaps it onto t1: n equal steps, not n - 1 and a tail.
    Returns the list [t0, ..., t1]."""
    t0, t1, dt = (float(np.float32(v)) for v in (t0, t1, dt))
    span, adt = abs(t1 - t0), abs(dt)
    assert adt > 0.0
    tdir = 1.0 if t1 >= t0 else -1.0
    n = int(math.floor(span / adt + 1e-9))
    tol = 100.0 * float(np.finfo(np.float32).eps) * max(abs(t0), abs(t1))
    if span - n * adt > tol and adt - (span - n * adt) <= tol:
        n += 1
    rem = span - n * adt
    if rem <= tol:
        if n == 0:
            return [t0]
        return [t0 + (t1 - t0) * i / n for i in range(n)] + [t1]
    return [t0 + tdir * adt * i for i in range(n + 1)] + [t1]


def integrate_grid(spec: Spec, p, u0: np.ndarray, tgrid, alg: int, eps, ys) -> np.ndarray:
    """Fixed steps on a given grid of times (either direction): step n runs from tgrid[n] to tgrid[n+1]."""
    u = np.asarray(u0, dtype=np.float64).copy()
    for ta, tb in zip(tgrid[:-1], tgrid[1:]):
        u = integrate_fixed(spec, p, u, ta, tb, 1, alg, eps, ys)
    return u


def integrate_fixed_dt(spec: Spec, p, u0: np.ndarray, t0: float, t1: float, dt: float, alg: int, eps, ys) -> np.ndarray:
    """base_sol with sol_kwargs = (alg, adaptive = false, dt): steps of dt and a shorter last step (fixed_dt_grid)."""
    return integrate_grid(spec, p, u0, fixed_dt_grid(t0, t1, dt), alg, eps, ys)


TSIT5_BTILDE = (-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                0.5823571654525552, -0.45808210592918697, 1.0 / 66.0)


def integrate_adaptive_tsit5(spec: Spec, p, u0: np.ndarray, t0: float, t1: float, reltol: float, abstol: float,
                             eps=None, ys=None, dt0: Optional[float] = None, maxiters: int = 100000):
    """base_sol with Tsit5(), adaptive (src/core/base_icnf.jl:134-140), restated in float64 from OrdinaryDiffEq's
    documented algorithm: Hairer's initial step, embedded 4th-order estimate dt sum btilde_i k_i scaled by
    abstol + reltol max(|u_prev|, |u|), RMS norm over the whole S x B state, PI controller (beta1 = 7/50, beta2 = 2/25,
    gamma = 9/10, qmin = 1/5, qmax = 10, qoldinit = 1e-4), first-same-as-last.  Returns (u1, stats).
    The Julia package is not available here, so its exact step sequence is unverified: this is the oracle the HIP
    path's adaptive mode is compared with, itself checked against fine fixed-step solves."""
    c, a, b = tableau(ALG_TSIT5)
    bt = TSIT5_BTILDE
    f = lambda u, t: aug_f(spec, p, u, t, eps, ys).astype(np.float64)
    u = np.asarray(u0, dtype=np.float64).copy()
    n = u.size
    tdir = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
    rms = lambda x: math.sqrt(float((x * x).sum()) / n)
    k1 = f(u, t0)
    nf = 1
    if dt0 is None:
        sk = abstol + np.abs(u) * reltol
        d0, d1 = rms(u / sk), rms(k1 / sk)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        h0 = min(h0, span)
        f1 = f(u + tdir * h0 * k1, t0 + tdir * h0)
        nf += 1
        d2 = rms((f1 - k1) / sk) / h0
        dm = max(d1, d2)
        h1 = max(1e-6, h0 * 1e-3) if dm <= 1e-15 else 10.0 ** (-(2.0 + math.log10(dm)) / 5.0)
        dt = min(100.0 * h0, h1, span)
    else:
        dt = min(abs(dt0), span)
    beta1, beta2, gamma, qmin, qmax, qold = 7.0 / 50.0, 2.0 / 25.0, 0.9, 0.2, 10.0, 1e-4
    t = t0
    stats = {"naccept": 0, "nreject": 0, "dts": []}
    for _ in range(maxiters):
        if abs(t1 - t) <= 1e-12 * max(1.0, span):
            break
        last = dt >= abs(t1 - t) * (1.0 - 1e-12)
        h = tdir * (abs(t1 - t) if last else dt)
        ks = [k1]
        for i in range(1, 6):
            ui = u.copy()
            for j, aij in enumerate(a[i]):
                if aij != 0.0:
                    ui = ui + h * aij * ks[j]
            ks.append(f(ui, t + c[i] * h))
        un = u.copy()
        for bi, ki in zip(b, ks):
            un = un + h * bi * ki
        k7 = f(un, t + h)
        nf += 6
        ut = h * sum(bti * ki for bti, ki in zip(bt, ks + [k7]))
        eest = rms(ut / (abstol + np.maximum(np.abs(u), np.abs(un)) * reltol))
        q11 = eest ** beta1 if eest > 0 else 0.0
        q = 1.0 / qmax if eest == 0 else max(1.0 / qmax, min(1.0 / qmin, (q11 / qold ** beta2) / gamma))
        if eest <= 1.0:
            t = t1 if last else t + h
            u, k1 = un, k7
            qold = max(eest, 1e-4)
            stats["naccept"] += 1
            stats["dts"].append(h)
            dt = abs(h) / q
        else:
            stats["nreject"] += 1
            dt = abs(h) / min(1.0 / qmin, q11 / gamma)
    else:
        raise RuntimeError("maxiters")
    stats["nf"] = nf
    return u, stats


def adams_moulton_gammas(n: int):
    """gamma*_0 .. gamma*_{n-1} of the Adams-Moulton family: sum_{m=0}^{j} gamma*_m / (j - m + 1) = [j == 0]
    (Hairer, Noersett, Wanner I, III.1 eq. 1.9'): 1, -1/2, -1/12, -1/24, -19/720, -3/160, ..."""
    from fractions import Fraction
    gs = [Fraction(1)]
    for j in range(1, n):
        gs.append(-sum(gs[m] / (j - m + 1) for m in range(j)))
    return [float(x) for x in gs]


class VcabmStepper:
    """The multistep state and the PECE passes of the variable-coefficient Adams method (Hairer, Noersett, Wanner I, III.5,
    eqs. 5.7-5.10; Shampine & Gordon's error estimates), one attempt at a time so a test can drive the HIP entry points
    and this with the same (order, dt) script.  Orders / indices as in the text: order k = number of predictor terms."""

    def __init__(self, f, u0, t0, abstol, reltol, max_order: int = 12):
        self.f, self.u, self.t = f, np.asarray(u0, dtype=np.float64).copy(), t0
        self.abstol, self.reltol, self.max_order = abstol, reltol, max_order
        self.fn = f(self.u, t0)
        self.hist, self.phistar_prev, self.pending = [], [], None
        self.gstar = adams_moulton_gammas(max_order + 3)
        self.n = self.u.size

    def _rms_sq(self, x):
        return float((x * x).sum())

    def attempt(self, k: int, h: float):
        """-> (u_new, [sum of squared scaled errors of orders k, k-1, k-2])"""
        assert 1 <= k <= min(self.max_order, len(self.hist) + 1)
        assert min(k, len(self.hist)) <= len(self.phistar_prev), "the order can rise by at most one per accepted step"
        dts = [h] + self.hist
        m = min(k + 1, len(self.hist) + 1)          # differences the history supports
        # beta_j(n) = prod_{i<j} (t_{n+1} - t_{n-i}) / (t_n - t_{n-1-i});  Phi_j(n) = Phi_{j-1}(n) - Phi*_{j-1}(n-1)
        beta = [1.0]
        for j in range(1, m):
            beta.append(beta[-1] * sum(dts[:j]) / sum(dts[1:j + 1]))
        phi = [self.fn]
        for j in range(1, m):
            phi.append(phi[-1] - self.phistar_prev[j - 1])
        phistar = [bj * pj for bj, pj in zip(beta, phi)]
        # g_j(n) = c_{j,1}: c_{0,q} = 1/q, c_{1,q} = 1/(q(q+1)), c_{j,q} = c_{j-1,q} - c_{j-1,q+1} h / (t_{n+1} - t_{n-j+1})
        ng = k + 1
        c = [1.0 / q for q in range(1, ng + 2)]
        g = [c[0]]
        for j in range(1, ng):
            if j == 1:
                c = [1.0 / (q * (q + 1)) for q in range(1, ng + 1)]
            else:
                xi = sum(dts[:j])
                c = [c[q] - c[q + 1] * h / xi for q in range(len(c) - 1)]
            g.append(c[0])
        up = self.u + h * sum(g[j] * phistar[j] for j in range(k))                # P
        dp = self.f(up, self.t + h)                                               # E
        pn = self._diffs(dp, k, phistar)
        un = up + h * g[k] * pn[k]                                                # C
        sc = self.abstol + np.maximum(np.abs(self.u), np.abs(un)) * self.reltol
        errs = [self._rms_sq(h * (g[k] - g[k - 1]) * pn[k] / sc),
                self._rms_sq(h * (g[k - 1] - g[k - 2]) * pn[k - 1] / sc) if k >= 2 else 0.0,
                self._rms_sq(h * (g[k - 2] - g[k - 3]) * pn[k - 2] / sc) if k >= 3 else 0.0]
        self.pending = (k, h, phistar, un, sc)
        self.last_g, self.last_beta = g, beta
        return un, errs

    @staticmethod
    def _diffs(d, upto, phistar):   # Phi_j(n+1), j = 0 .. upto, from a derivative at t_{n+1}
        out = [d]
        for j in range(1, upto + 1):
            out.append(out[-1] - phistar[j - 1])
        return out

    def accept(self, want_up: bool = False):
        """commit the pending attempt (final E); -> sum of squared scaled errors of order k+1, or None"""
        k, h, phistar, un, sc = self.pending
        fnew = self.f(un, self.t + h)                                             # E
        up = None
        if want_up:
            assert k < self.max_order and len(self.hist) >= k
            up = self._rms_sq(h * self.gstar[k + 1] * self._diffs(fnew, k + 1, phistar)[k + 1] / sc)
        self.u, self.fn, self.t = un, fnew, self.t + h
        self.hist = [h] + self.hist[:self.max_order]
        self.phistar_prev, self.pending = phistar, None
        return up


def integrate_vcabm(spec: Spec, p, u0: np.ndarray, t0: float, t1: float, reltol: float, abstol: float,
                    eps=None, ys=None, dt0: Optional[float] = None, maxiters: int = 100000, max_order: int = 12):
    """base_sol with the reference's DEFAULT algorithm VCABM() (src/core/icnf.jl:84-89, base_icnf.jl:134-140): the
    variable-step, variable-order Adams predictor-corrector in variable-coefficient (divided-difference) form, restated in
    float64 from the published algorithm - Hairer, Noersett, Wanner I, III.5 (beta_j, Phi_j, Phi*_j, g_j recurrences)
    with Shampine & Gordon's PECE step, error estimate and order selection - in the arrangement
    OrdinaryDiffEqAdamsBashforthMoulton uses as far as it can be recalled without the package: order k starts at 1 and
    rises by one per accepted step up to 3 during the first 4 steps; predictor with k terms, corrector adds the
    (k+1)-th, error estimate dt (g_k - g_{k-1}) Phi_k(n+1) scaled by abstol + reltol max(|u_prev|, |u|) under the
    RMS norm over the WHOLE state; after that the estimates for orders k-1, k-2 (from the predictor differences) and
    k+1 (dt gamma*_{k+1} Phi_{k+1}(n+1) from the re-evaluated derivative) lower the order when max(err_{k-2}, err_{k-1})
    <= err_k, or raise it when err_{k+1} < err_k (the step-size error is then set to 1); integral step-size controller
    dt / clamp(EEst^(1/(k+1)) / gamma, 1/qmax, 1/qmin), gamma = 9/10, qmin = 1/5, qmax = 10, same factor on a rejection;
    Hairer's initial step with the exponent 1 / get_current_alg_order = 1 / (current order of the cache) = 1.  PARITY UNPINNED: the Julia package is absent, so its exact
    step and order sequence is unverified; this is the oracle the HIP path's VCABM mode is compared with, itself
    checked against fine fixed-step solves and on a linear field.  Returns (u1, stats)."""
    f = lambda u, t: aug_f(spec, p, u, t, eps, ys).astype(np.float64)
    n = np.asarray(u0).size
    tdir = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
    rms = lambda x: math.sqrt(float((x * x).sum()) / n)
    s = VcabmStepper(f, u0, t0, abstol, reltol, max_order)
    nf = 1
    if dt0 is None:
        u, fn = s.u, s.fn
        sk = abstol + np.abs(u) * reltol
        d0, d1 = rms(u / sk), rms(fn / sk)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        h0 = min(h0, span)
        f1 = f(u + tdir * h0 * fn, t0 + tdir * h0)
        nf += 1
        d2 = rms((f1 - fn) / sk) / h0
        dm = max(d1, d2)
        h1 = max(1e-6, h0 * 1e-3) if dm <= 1e-15 else 10.0 ** (-(2.0 + math.log10(dm)) / 1.0)
        dt = min(100.0 * h0, h1, span)
    else:
        dt = min(abs(dt0), span)
    gamma, qmin, qmax = 0.9, 0.2, 10.0
    t, k, step = t0, 1, 1
    stats = {"naccept": 0, "nreject": 0, "dts": [], "orders": []}
    for _ in range(maxiters):
        if abs(t1 - t) <= 1e-12 * max(1.0, span):
            break
        last = dt >= abs(t1 - t) * (1.0 - 1e-12)
        h = tdir * (abs(t1 - t) if last else dt)
        _, errs = s.attempt(k, h)
        nf += 1
        eest = math.sqrt(errs[0] / n)
        if not eest <= 1.0:
            stats["nreject"] += 1
            dt = abs(h) / max(1.0 / qmax, min(1.0 / qmin, eest ** (1.0 / (k + 1)) / gamma))
            continue
        select = step > 4 and k >= 3
        lower = select and max(math.sqrt(errs[2] / n), math.sqrt(errs[1] / n)) <= eest
        want_up = select and not lower and k < max_order
        up = s.accept(want_up)
        nf += 1
        knew = k
        if not select:
            knew = min(k + 1, 3)
        elif lower:
            knew = k - 1
        elif want_up and math.sqrt(up / n) < eest:
            knew = k + 1
            eest = 1.0
        q = 1.0 / qmax if eest == 0 else max(1.0 / qmax, min(1.0 / qmin, eest ** (1.0 / (knew + 1)) / gamma))
        t = t1 if last else t + h
        s.t = t
        stats["naccept"] += 1
        stats["dts"].append(h)
        stats["orders"].append(k)
        k, step = knew, step + 1
        dt = abs(h) / q
    else:
        raise RuntimeError("maxiters")
    stats["nf"] = nf
    return s.u, stats


def std_normal_logpdf(z: np.ndarray) -> np.ndarray:
    """logpdf of MvNormal(Zeros(d), Eye(d)) per column (src/core/icnf.jl:76-79)."""
    d = z.shape[0]
    return -0.5 * d * math.log(2.0 * math.pi) - 0.5 * (z * z).sum(0)


def inference_fixed(spec: Spec, p, xs: np.ndarray, t0, t1, nsteps, alg, eps, ys=None, dt=None):
    """inference_prob + inference_sol for MatrixMode (src/core/base_icnf.jl:247-296,158-172).
    Returns logp (B,), (Edot, ndot, Adot) each (B,), u_final (S,B).  With `dt` given, nsteps is ignored and the solve
    takes OrdinaryDiffEq's fixed-dt steps (integrate_fixed_dt)."""
    B = xs.shape[1]
    u0 = np.concatenate([np.asarray(xs, dtype=np.float64),
                         np.zeros((spec.naug + 3, B))], axis=0)
    if dt is not None:
        u1 = integrate_fixed_dt(spec, p, u0, t0, t1, dt, alg, eps, ys)
    else:
        u1 = integrate_fixed(spec, p, u0, t0, t1, nsteps, alg, eps, ys)
    D = spec.D
    z, dlogp = u1[:D], u1[D]
    logp = std_normal_logpdf(z) - dlogp
    if spec.reg_aug and spec.naug > 0:
        Adot = np.sqrt((z[spec.nvars:] ** 2).sum(0))
    else:
        Adot = np.zeros(B)
    return logp, (u1[D + 1].copy(), u1[D + 2].copy(), Adot), u1


def loss(spec: Spec, p, xs, t0, t1, nsteps, alg, eps, ys=None, lambdas=(0.0, 0.0, 0.0)):
    """mean(-logp + l1 E + l2 n + l3 A)  (src/core/icnf.jl:628-649)."""
    logp, (E, n, A), _ = inference_fixed(spec, p, xs, t0, t1, nsteps, alg, eps, ys)
    return float(np.mean(-logp + lambdas[0] * E + lambdas[1] * n + lambdas[2] * A))


# ----------------------------------------------------------------------------------------
# deterministic synthetic inputs
# ----------------------------------------------------------------------------------------
def make_spec(nvars, hidden: Sequence[int], act=ACT_TANH, naug=0, ncond=0, autonomous=False,
              **kw) -> Spec:
    D = nvars + naug
    n_in = D + (0 if autonomous else 1) + ncond
    widths = [n_in, *hidden, D]
    acts = [act] * len(hidden) + [ACT_IDENTITY]
    s = Spec(nvars=nvars, naug=naug, ncond=ncond, autonomous=autonomous, widths=widths,
             acts=acts, **kw)
    s.check()
    return s


def glorot_params(spec: Spec, rng: np.random.Generator, bias_scale: float = 0.0) -> np.ndarray:
    """W ~ U(-sqrt(6/(in+out)), +), b = bias_scale*N(0,1); flat Lux layout, float32."""
    parts = []
    for l in range(len(spec.acts)):
        fin, fout = spec.widths[l], spec.widths[l + 1]
        lim = math.sqrt(6.0 / (fin + fout))
        W = rng.uniform(-lim, lim, size=(fout, fin))
        parts.append(W.T.reshape(-1))                 # column-major (out x in)
        parts.append(bias_scale * rng.standard_normal(fout))
    return np.concatenate(parts).astype(np.float32)


def synth_inputs(spec: Spec, B: int, seed: int, bias_scale: float = 0.0):
    rng = np.random.default_rng(seed)
    p = glorot_params(spec, rng, bias_scale)
    xs = rng.standard_normal((spec.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((spec.nprobes * spec.D, B)).astype(np.float32)
    ys = rng.standard_normal((spec.ncond, B)).astype(np.float32) if spec.ncond else None
    return p, xs, eps, ys


# ----------------------------------------------------------------------------------------
# Groundwork for SURVEY.md §8(f) rank 2 — parameter gradient of the loss (training).
# The reference differentiates `loss` through SciMLBase.solve with QuadratureAdjoint
# (src/core/icnf.jl:90-99; src/exts/mlj_ext/core_icnf.jl:42-51).  With a fixed-step solver the
# exact gradient of the *discrete* loss is obtained by reverse-mode through the RK steps
# (discretise-then-optimise); this fp64 autograd version is the oracle a future HIP backward
# kernel will be checked against.  Hutchinson VJP and JVP modes.
# ----------------------------------------------------------------------------------------
def loss_and_grad(spec: Spec, p, xs, t0, t1, nsteps, alg, eps, ys=None, lambdas=(0.0, 0.0, 0.0), wrt_x=False, tgrid=None):
    """Returns (loss, dloss/dp) with p in the flat Lux layout, all in float64; with wrt_x also dloss/dxs.
    tgrid (nsteps + 1 times) replaces the uniform grid: fixed steps frozen from an adaptive solve."""
    spec.check()
    D, K = spec.D, spec.nprobes
    pt = torch.tensor(np.asarray(p, dtype=np.float64), requires_grad=True)
    w_off, b_off, _ = spec.param_offsets()
    layers = []
    for l in range(len(spec.acts)):
        fin, fout = spec.widths[l], spec.widths[l + 1]
        W = pt[w_off[l]:w_off[l] + fin * fout].reshape(fin, fout).t()
        layers.append((W, pt[b_off[l]:b_off[l] + fout]))
    x = torch.tensor(np.asarray(xs, dtype=np.float64), requires_grad=bool(wrt_x))
    B = x.shape[1]
    e = None if spec.mode == MODE_EXACT else torch.tensor(np.asarray(eps, dtype=np.float64))
    yt = None if ys is None else torch.tensor(np.asarray(ys, dtype=np.float64))

    def f_aug(u, t):
        z = u[:D]
        if not z.requires_grad:
            z = z.clone().requires_grad_(True)
        zdot = _net(spec, layers, z, t, yt)
        ldot = torch.zeros(B, dtype=torch.float64)
        ndot = torch.zeros(B, dtype=torch.float64)
        if spec.mode == MODE_EXACT:   # TestMode: ldot = -tr J, no regularisers (src/core/icnf.jl:297-339)
            for i in range(D):
                seed = torch.zeros_like(zdot)
                seed[i] = 1.0
                (gi,) = torch.autograd.grad(zdot, z, seed, create_graph=True)
                ldot = ldot - gi[i]
            return torch.cat([zdot, ldot[None], torch.zeros(2, B, dtype=torch.float64)], dim=0)
        for k in range(K):
            ek = e[k * D:(k + 1) * D]
            if spec.mode == MODE_HUTCH_VJP:
                (g,) = torch.autograd.grad(zdot, z, ek, create_graph=True)          # eps^T J
            else:
                # J eps by the double-backward identity: v -> J^T v is linear, its derivative applied to eps is J eps
                v = torch.zeros_like(zdot, requires_grad=True)
                (jt,) = torch.autograd.grad(zdot, z, v, create_graph=True)
                (g,) = torch.autograd.grad(jt, v, ek, create_graph=True)
            ldot = ldot - (g * ek).sum(0) / K
            if spec.reg_j:
                ndot = ndot + torch.sqrt((g * g).sum(0)) / K
        Edot = torch.sqrt((zdot * zdot).sum(0)) if spec.reg_z else torch.zeros(B, dtype=torch.float64)
        return torch.cat([zdot, ldot[None], Edot[None], ndot[None]], dim=0)

    c, a, b = tableau(alg)
    dt = (t1 - t0) / nsteps
    u = torch.cat([x, torch.zeros(spec.naug + 3, B, dtype=torch.float64)], dim=0)
    for n in range(nsteps):
        tn = t0 + n * dt
        if tgrid is not None:
            tn, dt = float(tgrid[n]), float(tgrid[n + 1]) - float(tgrid[n])
        ks = []
        for i in range(len(c)):
            ui = u
            for j, aij in enumerate(a[i]):
                if aij != 0.0:
                    ui = ui + dt * aij * ks[j]
            ks.append(f_aug(ui, tn + c[i] * dt))
        for bi, ki in zip(b, ks):
            u = u + dt * bi * ki
    z, dlogp = u[:D], u[D]
    logp = -0.5 * D * math.log(2.0 * math.pi) - 0.5 * (z * z).sum(0) - dlogp
    A = torch.sqrt((z[spec.nvars:] ** 2).sum(0)) if (spec.reg_aug and spec.naug > 0) \
        else torch.zeros(B, dtype=torch.float64)
    L = (-logp + lambdas[0] * u[D + 1] + lambdas[1] * u[D + 2] + lambdas[2] * A).mean()
    if wrt_x:
        gp, gx = torch.autograd.grad(L, (pt, x))
        return float(L.detach()), gp.detach().numpy(), gx.detach().numpy()
    (gp,) = torch.autograd.grad(L, pt)
    return float(L.detach()), gp.detach().numpy()

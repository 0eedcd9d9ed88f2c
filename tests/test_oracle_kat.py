"""CPU: analytic known-answer tests that pin the oracles without the reference
(SURVEY.md §8(c)): closed-form linear flow, convergence orders of the integrators,
VJP/JVP agreement, exact trace vs one-hot probes vs finite differences, zero weights,
K-probe mean, column independence."""
import math

import numpy as np
import pytest
from scipy.linalg import expm


def linear_spec(o64, D, mode):
    s = o64.Spec(nvars=D, autonomous=True, widths=[D, D], acts=[o64.ACT_IDENTITY], mode=mode)
    s.check()
    return s


def linear_problem(D, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((D, D)) * 0.4
    b = rng.standard_normal(D) * 0.2
    p = np.concatenate([A.T.reshape(-1), b])
    xs = rng.standard_normal((D, 6))
    M = np.zeros((D + 1, D + 1)); M[:D, :D] = A; M[:D, D] = b
    E = expm(M)
    z1 = E[:D, :D] @ xs + E[:D, D:]
    logp = -0.5 * D * math.log(2 * math.pi) - 0.5 * (z1 ** 2).sum(0) + np.trace(A)
    return A, p, xs, z1, logp


@pytest.mark.parametrize("which", ["fp64", "c"])
def test_linear_field_closed_form(which, oracles):
    o64, oc = oracles
    D = 3
    spec = linear_spec(o64, D, o64.MODE_EXACT)
    A, p, xs, z1, logp_ref = linear_problem(D, 0)
    if which == "fp64":
        logp, _, u1 = o64.inference_fixed(spec, p, xs, 0.0, 1.0, 40, o64.ALG_TSIT5, None)
        tol = 1e-9
    else:
        logp, _, u1 = oc.inference_fixed(spec, p.astype(np.float32), xs.astype(np.float32), 0.0, 1.0,
                                         40, o64.ALG_TSIT5, None)
        tol = 2e-5
    assert np.max(np.abs(u1[:D] - z1)) < tol
    assert np.max(np.abs(u1[D] + np.trace(A))) < tol       # dlogp = -tr(A) (t1 - t0)
    assert np.max(np.abs(logp - logp_ref)) < 10 * tol


@pytest.mark.parametrize("alg,order", [(0, 4), (1, 5)])
def test_integrator_convergence_order(alg, order, oracles):
    o64, _ = oracles
    D = 3
    spec = linear_spec(o64, D, o64.MODE_EXACT)
    A, p, xs, z1, _ = linear_problem(D, 1)
    errs = []
    for n in (2, 4, 8):
        u1 = o64.integrate_fixed(spec, p, np.concatenate([xs, np.zeros((3, xs.shape[1]))]), 0.0, 1.0,
                                 n, alg, None, None)
        errs.append(np.max(np.abs(u1[:D] - z1)))
    slopes = [math.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert all(abs(s - order) < 0.6 for s in slopes), (errs, slopes)


def test_vjp_and_jvp_give_the_same_trace_scalar(oracles):
    o64, oc = oracles
    kw = dict(nvars=4, hidden=[16, 16], ncond=2, reg_z=True, reg_j=True)
    sv = o64.make_spec(mode=o64.MODE_HUTCH_VJP, **kw)
    sj = o64.make_spec(mode=o64.MODE_HUTCH_JVP, **kw)
    p, xs, eps, ys = o64.synth_inputs(sv, 9, 5, bias_scale=0.2)
    u = np.concatenate([xs, np.zeros((3, 9), np.float32)])
    for f, tol in ((o64.aug_f, 1e-12), (oc.aug_f, 2e-6)):
        dv = f(sv, p, u, 0.2, eps, ys)
        dj = f(sj, p, u, 0.2, eps, ys)
        D = sv.D
        assert np.max(np.abs(dv[:D + 2] - dj[:D + 2])) < tol     # zdot, ldot, Edot agree
        assert np.max(np.abs(dv[D + 2] - dj[D + 2])) > 1e-3      # |eps^T J| != |J eps|


def test_exact_trace_equals_onehot_probes_and_finite_differences(oracles):
    o64, oc = oracles
    D = 5
    se = o64.make_spec(nvars=D, hidden=[24, 24], mode=o64.MODE_EXACT)
    sk = o64.make_spec(nvars=D, hidden=[24, 24], nprobes=D)
    p, xs, _, _ = o64.synth_inputs(se, 7, 6, bias_scale=0.2)
    u = np.concatenate([xs, np.zeros((3, 7), np.float32)]).astype(np.float64)
    onehot = np.tile(np.eye(D).reshape(D * D, 1), (1, 7))       # probe k = e_k
    ex = o64.aug_f(se, p, u, 0.4, None, None)
    hk = o64.aug_f(sk, p, u, 0.4, onehot, None)
    assert np.max(np.abs(ex[D] - D * hk[D])) < 1e-12              # mean over K=D probes -> x D
    h = 1e-6
    tr = np.zeros(7)
    for i in range(D):
        up, um = u.copy(), u.copy()
        up[i] += h; um[i] -= h
        tr += (o64.aug_f(se, p, up, 0.4, None, None)[i] - o64.aug_f(se, p, um, 0.4, None, None)[i]) / (2 * h)
    assert np.max(np.abs(ex[D] + tr)) < 1e-7
    exc = oc.aug_f(se, p, u.astype(np.float32), 0.4, None, None)
    assert np.max(np.abs(exc[D] - ex[D])) < 5e-6


def test_zero_weights_translate_by_bias(oracles):
    o64, oc = oracles
    spec = o64.make_spec(nvars=3, hidden=[8])
    _, _, n = spec.param_offsets()
    p = np.zeros(n, np.float32)
    _, b_off, _ = spec.param_offsets()
    bN = np.array([0.3, -0.2, 0.1], np.float32)
    p[b_off[-1]:b_off[-1] + 3] = bN
    rng = np.random.default_rng(2)
    xs = rng.standard_normal((3, 5)).astype(np.float32)
    eps = rng.standard_normal((3, 5)).astype(np.float32)
    ref = -1.5 * math.log(2 * math.pi) - 0.5 * ((xs + bN[:, None]) ** 2).sum(0)
    for f, tol in ((o64.inference_fixed, 1e-6), (oc.inference_fixed, 1e-5)):
        logp = f(spec, p, xs, 0.0, 1.0, 8, o64.ALG_RK4, eps)[0]
        assert np.max(np.abs(logp - ref)) < tol


def test_k_probe_result_is_mean_of_single_probe_runs(oracles):
    o64, oc = oracles
    K = 3
    sk = o64.make_spec(nvars=4, hidden=[16, 16], nprobes=K, reg_z=True, reg_j=True)
    s1 = o64.make_spec(nvars=4, hidden=[16, 16], nprobes=1, reg_z=True, reg_j=True)
    p, xs, eps, _ = o64.synth_inputs(sk, 6, 11, bias_scale=0.1)
    lk, (Ek, nk, _), _ = oc.inference_fixed(sk, p, xs, 0.0, 1.0, 6, o64.ALG_TSIT5, eps)
    runs = [oc.inference_fixed(s1, p, xs, 0.0, 1.0, 6, o64.ALG_TSIT5, eps[k * 4:(k + 1) * 4]) for k in range(K)]
    assert np.max(np.abs(lk - np.mean([r[0] for r in runs], 0))) < 1e-5
    assert np.max(np.abs(nk - np.mean([r[1][1] for r in runs], 0))) < 1e-5
    assert np.max(np.abs(Ek - runs[0][1][0])) < 1e-6


def test_columns_are_independent(oracles):
    """Fixed-step integration keeps samples independent: permuting the
    columns permutes the outputs bit-for-bit."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 70, 12)
    perm = np.random.default_rng(0).permutation(70)
    a = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 3, o64.ALG_TSIT5, eps)[0]
    b = oc.inference_fixed(spec, p, xs[:, perm], 0.0, 1.0, 3, o64.ALG_TSIT5, eps[:, perm])[0]
    assert np.array_equal(a[perm], b)


def test_torch_f32_gemm_restatement_matches_the_fp64_oracle(oracles):
    """oracle/cnf_oracle_torch32.py (bench.py's CPU-favourable timing cross-check): RNODE, two probes, conditioned."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnf_oracle_torch32 as t32
    o64, _ = oracles
    spec = o64.make_spec(nvars=5, naug=1, ncond=3, hidden=[32, 32], act=2, nprobes=2, reg_z=True, reg_j=True)
    p, xs, eps, ys = o64.synth_inputs(spec, 33, 9, bias_scale=0.2)
    ref = o64.inference_fixed(spec, p, xs, 0.0, 1.0, 8, o64.ALG_TSIT5, eps, ys)[0]
    got = t32.inference_fixed(spec, p, xs, 0.0, 1.0, 8, o64.ALG_TSIT5, eps, ys)
    assert np.max(np.abs(got - ref)) < 5e-5


def test_two_hidden_layer_trace_identity(oracles):
    """The identity behind the fused exact-trace shortcut for the reference's default architecture (two hidden layers):
    tr J = act'_2^T Q act'_1 with the constant Q = W_2 .* (W_1[:,0:D] W_3)^T  (csrc/cnf_mfma_kernel.h, mfma_pack)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=5, naug=2, ncond=3, hidden=[24, 40], act=2, mode=2)
    p, xs, _, ys = o64.synth_inputs(spec, 9, 4, bias_scale=0.3)
    u = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, 9))])
    u[:spec.D] += 0.1 * np.random.default_rng(0).standard_normal((spec.D, 9))
    t = 0.37
    du = o64.aug_f(spec, p, u, t, None, ys)
    layers = [(np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64)) for W, b in o64.unpack_params(spec, p)]
    (W1, b1), (W2, b2), (W3, b3) = layers
    D = spec.D
    x = np.vstack([u[:D], np.full((1, 9), t), ys.astype(np.float64)])
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))
    a1 = W1 @ x + b1[:, None]
    h1 = np.log1p(np.exp(a1))
    a2 = W2 @ h1 + b2[:, None]
    Q = W2 * (W1[:, :D] @ W3).T
    tr = np.einsum("aj,ab,bj->j", sig(a2), Q, sig(a1))
    assert np.max(np.abs(du[D] + tr)) < 1e-12


# ---- the restatement of the reference's default solver (VCABM) ----

def test_adams_moulton_gammas_known_values(oracles):
    o64, _ = oracles
    g = o64.adams_moulton_gammas(8)   # Hairer, Noersett, Wanner I, table III.1.2
    want = [1, -1 / 2, -1 / 12, -1 / 24, -19 / 720, -3 / 160, -863 / 60480, -275 / 24192]
    assert np.allclose(g, want, rtol=0, atol=1e-15)


def test_variable_coefficient_adams_is_exact_on_polynomials(oracles):
    """u' = (q + 1) t^q with arbitrary step sizes: an order-k attempt (k predictor terms + one corrector term = the
    interpolation polynomial of degree k through k + 1 derivative values) integrates degree q <= k exactly whatever the
    step ratios, its order-k error estimate - a k-th divided difference - vanishes for q < k, and with constant steps
    the g_j are the classical Adams-Bashforth gammas.  Pins the beta / Phi / g recurrences of the restatement."""
    o64, _ = oracles
    rng = np.random.default_rng(4)
    for q in range(0, 9):
        f = lambda u, t, q=q: np.full_like(u, (q + 1) * t ** q)
        s = o64.VcabmStepper(f, np.zeros((1, 3)), 0.3, 1.0, 0.0)
        for n in range(14):
            h = float(rng.uniform(0.02, 0.2))
            k = min(n + 1, 12)
            t_old, u_old = s.t, s.u.copy()
            un, errs = s.attempt(k, h)
            exact = (t_old + h) ** (q + 1) - t_old ** (q + 1)
            if q <= k:
                assert np.max(np.abs(un - u_old - exact)) < 1e-11, (q, n, k)
            else:
                assert np.max(np.abs(un - u_old - exact)) > 1e-13
            if q < k:
                assert errs[0] < 1e-20, (q, n, k, errs)
            up = s.accept(want_up=(k < 12 and len(s.hist) >= k))
            if up is not None and q < k + 1:
                assert up < 1e-20
    # constant steps: g_j = gamma_j = 1, 1/2, 5/12, 3/8, 251/720 (Adams-Bashforth)
    s = o64.VcabmStepper(lambda u, t: np.ones_like(u), np.zeros((1, 1)), 0.0, 1.0, 0.0)
    for n in range(6):
        s.attempt(min(n + 1, 5), 0.1)
        s.accept()
    un, _ = s.attempt(5, 0.1)
    assert abs(float(un.ravel()[0]) - 0.7) < 1e-14
    assert np.allclose(s.last_g, [1, 1 / 2, 5 / 12, 3 / 8, 251 / 720, 95 / 288], rtol=0, atol=1e-14)
    assert np.allclose(s.last_beta, 1.0, rtol=0, atol=1e-14)


def test_vcabm_restatement_on_the_linear_field(oracles):
    """Closed form e^{A} z0: the adaptive solve lands within a small multiple of the tolerance at every tolerance, uses
    fewer derivative evaluations than adaptive Tsit5 at tight tolerances, climbs to high order, and runs backwards."""
    o64, _ = oracles
    D = 3
    spec = linear_spec(o64, D, o64.MODE_EXACT)
    A, p, xs, z1, _ = linear_problem(D, 2)
    u0 = np.concatenate([xs, np.zeros((3, xs.shape[1]))])
    prev = None
    for tol in (1e-4, 1e-6, 1e-8, 1e-10):
        u1, st = o64.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol)
        err = np.max(np.abs(u1[:D] - z1))
        assert err < 50 * tol, (tol, err)
        assert np.max(np.abs(u1[D] + np.trace(A))) < 50 * tol
        assert st["nf"] == 2 + 2 * st["naccept"] + st["nreject"]                 # f0, Hairer's probe, PECE
        assert st["orders"][:4] == [1, 2, 3, 3]
        if prev is not None:
            assert err < prev
        prev = err
    _, ts = o64.integrate_adaptive_tsit5(spec, p, u0, 0.0, 1.0, 1e-10, 1e-10)
    assert st["nf"] < ts["nf"] and max(st["orders"]) >= 8
    back, sb = o64.integrate_vcabm(spec, p, u1, 1.0, 0.0, 1e-10, 1e-10)
    assert sb["dts"][0] < 0 and np.max(np.abs(back[:D] - xs)) < 1e-7


def test_vcabm_restatement_reproduces_its_fixture(oracles):
    """tests/golden/vcabm_default_softplus_aug.npz (tests/golden/make_golden.py::make_vcabm_fixture): the restatement of the
    default solver is pinned against accidental change - same accepted / rejected counts, order history, steps and state."""
    import os
    from conftest import GOLDEN
    o64, _ = oracles
    f = np.load(os.path.join(GOLDEN, "vcabm_default_softplus_aug.npz"))
    spec = o64.make_spec(nvars=2, naug=3, hidden=[24, 24], act=2, reg_z=True, reg_j=True)
    u0 = np.vstack([f["xs"].astype(np.float64), np.zeros((spec.naug + 3, f["xs"].shape[1]))])
    for tag in ("a", "b", "a0", "b0"):
        tol = float(f[f"tol_{tag}"])
        u1, st = o64.integrate_vcabm(spec, f["p"], u0, 0.0, 1.0, tol, tol, f["eps"], dt0=2.0 ** -7 if tag.endswith("0") else None)
        assert (st["naccept"], st["nreject"]) == (int(f[f"naccept_{tag}"]), int(f[f"nreject_{tag}"]))
        assert st["orders"] == f[f"orders_{tag}"].tolist()
        assert np.allclose(st["dts"], f[f"dts_{tag}"], rtol=1e-9, atol=0) and np.allclose(u1, f[f"u1_{tag}"], rtol=0, atol=1e-10)


# ---- derived, not recalled: what can be computed exactly is, with sympy / rational arithmetic ----

def test_adams_gammas_derived_exactly(oracles):
    """The Adams-Moulton gamma*_j = int_{-1}^{0} binom(s + j - 1, j) ds and the Adams-Bashforth gamma_j = int_0^1 binom(s + j - 1, j) ds
    (Hairer-Noersett-Wanner I, III.1) computed by exact polynomial integration, 14 terms - the stepper's order runs to 12 - against
    the recurrence the restatement uses, and against gamma_j = sum_{m <= j} gamma*_m."""
    import sympy as sp
    o64, _ = oracles
    s = sp.symbols("s")
    n = 14
    gstar, gab = [], []
    for j in range(n):
        binom = sp.Integer(1)
        for i in range(j):
            binom = binom * (s + i) / (i + 1)          # binom(s + j - 1, j) = s (s+1) ... (s+j-1) / j!
        gstar.append(sp.integrate(sp.expand(binom), (s, -1, 0)))
        gab.append(sp.integrate(sp.expand(binom), (s, 0, 1)))
    assert gstar[:5] == [1, sp.Rational(-1, 2), sp.Rational(-1, 12), sp.Rational(-1, 24), sp.Rational(-19, 720)]
    assert gab[:5] == [1, sp.Rational(1, 2), sp.Rational(5, 12), sp.Rational(3, 8), sp.Rational(251, 720)]
    got = o64.adams_moulton_gammas(n)
    assert np.max(np.abs(np.array(got, dtype=np.float64) - np.array([float(g) for g in gstar]))) < 1e-15
    for j in range(n):                                  # gamma_j = sum_{m=0}^{j} gamma*_m
        assert sum(gstar[: j + 1]) == gab[j]
    # with constant steps the variable-coefficient stepper's g_j must be these gamma_j (its predictor is Adams-Bashforth)
    f = lambda u, t: np.ones_like(u)
    st = o64.VcabmStepper(f, np.zeros((1, 1)), 0.1, 1.0, 0.0)
    if hasattr(st, "constant_step_g"):
        assert np.allclose(st.constant_step_g(8), [float(g) for g in gab[:8]], atol=1e-14)


def _rooted_tree_conditions(A, b, c, order):
    """Order conditions of an explicit Runge-Kutta pair up to `order` (<= 5): list of (name, sum, 1 / gamma(tree))."""
    A, b, c = np.asarray(A, dtype=np.float64), np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64)
    Ac, Ac2, Ac3, AAc = A @ c, A @ c ** 2, A @ c ** 3, A @ (A @ c)
    conds = [("b 1", b.sum(), 1.0)]
    if order >= 2: conds += [("b c", b @ c, 1 / 2)]
    if order >= 3: conds += [("b c^2", b @ c ** 2, 1 / 3), ("b A c", b @ Ac, 1 / 6)]
    if order >= 4:
        conds += [("b c^3", b @ c ** 3, 1 / 4), ("b c.Ac", b @ (c * Ac), 1 / 8), ("b A c^2", b @ Ac2, 1 / 12), ("b A A c", b @ AAc, 1 / 24)]
    if order >= 5:
        conds += [("b c^4", b @ c ** 4, 1 / 5), ("b c^2.Ac", b @ (c ** 2 * Ac), 1 / 10), ("b c.Ac^2", b @ (c * Ac2), 1 / 15),
                  ("b c.AAc", b @ (c * AAc), 1 / 30), ("b (Ac)^2", b @ (Ac * Ac), 1 / 20), ("b A c^3", b @ Ac3, 1 / 20),
                  ("b A(c.Ac)", b @ (A @ (c * Ac)), 1 / 40), ("b A A c^2", b @ (A @ Ac2), 1 / 60), ("b A A A c", b @ (A @ AAc), 1 / 120)]
    return conds


def test_tsit5_pair_satisfies_its_order_conditions(oracles):
    """Tsitouras' 5(4) pair as the oracle (and the kernels, csrc/cnf_common.h) carry it: the 7-stage FSAL tableau satisfies all 17
    rooted-tree conditions of order 5 with its weights b, the embedded weights b - btilde satisfy the 8 conditions of order 4 and
    VIOLATE order 5 (so dt sum btilde_i k_i is an O(dt^5) error estimate, not zero), sum btilde = 0 and the row sums equal c.
    The constants are decimal literals of the published tableau, so 'satisfies' means to a few ulp of 1 - this pins every digit
    of btilde, which the restatement recalled from memory."""
    o64, _ = oracles
    c6, a6, b6, bt = o64.TSIT5_C, o64.TSIT5_A, o64.TSIT5_B, o64.TSIT5_BTILDE
    assert len(bt) == 7
    A = np.zeros((7, 7))
    for i, row in enumerate(a6):
        A[i, : len(row)] = row
    A[6, :6] = b6                                         # first-same-as-last: stage 7 = f(u_new)
    c = np.array(list(c6) + [1.0])
    b = np.array(list(b6) + [0.0])
    assert np.max(np.abs(A.sum(1) - c)) < 5e-15
    for name, got, want in _rooted_tree_conditions(A, b, c, 5):
        assert abs(got - want) < 5e-15, (name, got, want)
    bhat = b - np.array(bt)
    assert abs(np.sum(bt)) < 5e-16
    for name, got, want in _rooted_tree_conditions(A, bhat, c, 4):
        assert abs(got - want) < 5e-15, ("embedded", name, got, want)
    viol = max(abs(got - want) for name, got, want in _rooted_tree_conditions(A, bhat, c, 5)[8:])
    assert viol > 1e-4, viol                              # genuinely 4th order
    # classical RK4 for completeness
    cr, ar, br = o64.RK4_C, o64.RK4_A, o64.RK4_B
    Ar = np.zeros((4, 4))
    for i, row in enumerate(ar):
        Ar[i, : len(row)] = row
    for name, got, want in _rooted_tree_conditions(Ar, br, cr, 4):
        assert abs(got - want) < 1e-15, (name, got, want)


def test_tanh_fast_departure_is_measured(oracles):
    """Lux runs NNlib.tanh_fast on CPU Float32 arrays where the kernels (and the parity oracle) evaluate tanh itself.  The rational
    approximation is restated in oracle/cnf_oracle.c (cnf_oracle_set_fast_tanh); here its departure is a NUMBER: max |tanh_fast - tanh|
    over a dense grid, and max |delta logp| over 4096 columns of BASELINE config 2 (D = 8, 3 x 64, RK4 x 40) - far inside the 1e-4
    log-density tolerance (DESIGN.md section 2 quotes both)."""
    o64, oc = oracles
    spec = o64.make_spec(8, [64, 64, 64])
    B = 4096
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240612)
    oc.set_fast_tanh(False)
    exact = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 40, o64.ALG_RK4, eps, None, nthreads=8)[0]
    oc.set_fast_tanh(True)
    try:
        fast = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 40, o64.ALG_RK4, eps, None, nthreads=8)[0]
    finally:
        oc.set_fast_tanh(False)
    dlogp = float(np.max(np.abs(fast - exact)))
    x = np.linspace(-9.0, 9.0, 2_000_001)
    x2 = (x.astype(np.float32) ** 2).astype(np.float64)
    n = 1.0 + x2 * (0.1346604 + x2 * (0.0035974074 + x2 * (2.2332108e-5 + x2 * 1.587199e-8)))
    d = 1.0 + x2 * (0.4679937 + x2 * (0.026262015 + x2 * (0.0003453992 + x2 * 8.7767893e-7)))
    approx = np.where(x2 < 66.0, x * n / d, np.sign(x))
    dtanh = float(np.max(np.abs(approx - np.tanh(x))))
    print(f"tanh_fast: max|tanh_fast - tanh| = {dtanh:.3e}; cfg2 (4096 columns, RK4 x 40): max|dlogp| = {dlogp:.3e}")
    assert dtanh < 5e-7
    assert 0.0 < dlogp < 2e-5

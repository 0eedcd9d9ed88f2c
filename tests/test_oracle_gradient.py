"""CPU: groundwork for the parameter-gradient row: the fp64 autograd
gradient of the discrete loss agrees with central finite differences of the oracle's own loss, and
the committed fixture pins it for the future HIP backward kernel."""
import os

import numpy as np

from conftest import GOLDEN


import pytest


@pytest.mark.parametrize("nprobes,mode", [(1, 0), (3, 0), (2, 1)])
def test_loss_gradient_matches_finite_differences(nprobes, mode, oracles):
    o64, _ = oracles
    spec = o64.make_spec(nvars=3, hidden=[8, 8], reg_z=True, reg_j=True, nprobes=nprobes, mode=mode)
    p, xs, eps, _ = o64.synth_inputs(spec, 5, 31, bias_scale=0.2)
    lam = (0.01, 0.02, 0.0)
    L, g = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, 4, o64.ALG_RK4, eps, None, lam)
    assert abs(L - o64.loss(spec, p.astype(np.float64), xs, 0.0, 1.0, 4, o64.ALG_RK4, eps, None, lam)) < 1e-12
    rng = np.random.default_rng(0)
    idx = rng.choice(p.size, 12, replace=False)
    h = 1e-6
    for i in idx:
        pp, pm = p.astype(np.float64).copy(), p.astype(np.float64).copy()
        pp[i] += h; pm[i] -= h
        fd = (o64.loss(spec, pp, xs, 0.0, 1.0, 4, o64.ALG_RK4, eps, None, lam)
              - o64.loss(spec, pm, xs, 0.0, 1.0, 4, o64.ALG_RK4, eps, None, lam)) / (2 * h)
        assert abs(fd - g[i]) < 1e-7 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_gradient_fixture_is_reproducible(oracles):
    o64, _ = oracles
    f = np.load(os.path.join(GOLDEN, "grad_cfg2_d8_3x64_tsit5.npz"))
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    L, g = o64.loss_and_grad(spec, f["p"], f["xs"], 0.0, 1.0, int(f["nsteps"]), o64.ALG_TSIT5, f["eps"])
    assert abs(L - float(f["loss"])) < 1e-12
    assert np.max(np.abs(g - f["grad"])) < 1e-12


def test_frozen_grid_gradient_against_the_exact_flow():
    """What cnf_loss_grad_adaptive differs from (include/cnf.h): under the reference's default sol_kwargs it trains with
    QuadratureAdjoint on the VCABM solution (src/core/icnf.jl:84-99); the library returns the exact gradient of the Tsit5 solve on the
    frozen accepted steps.  Both approximate the gradient of the exact flow's loss: central differences of the loss on a VCABM solve
    at 1e-10 along random directions.  Bounds for the default architecture at nvariables = 1 (nvariables = 8 and both weight scales:
    profiles/r6/r6f_adaptive_gradient_gap.json, same script)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "adaptive_gradient_gap.py")
    spec = importlib.util.spec_from_file_location("adaptive_gradient_gap", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.run(1, 1.0, ndir=3)
    assert fresh["tsit5_steps"] <= 6 and fresh["rel_gap_frozen_vs_tight"] < 1e-6, fresh["rel_gap_frozen_vs_tight"]
    stiff = mod.run(1, 3.0, ndir=3)
    assert stiff["tsit5_steps"] > 10
    # within the solver tolerance's own accuracy, and no worse than differentiating the adaptive VCABM solve itself
    assert stiff["rel_gap_frozen_vs_tight"] < 2e-4, stiff["rel_gap_frozen_vs_tight"]
    assert stiff["rel_loss_gap_frozen_vs_tight"] < 1e-4
    committed = __import__("json").load(open(os.path.join(os.path.dirname(path), "r6", "r6f_adaptive_gradient_gap.json")))
    worst = max(r["rel_gap_frozen_vs_tight"] for r in committed)
    assert worst < 5e-3 and {(r["nvariables"], r["weight_scale"]) for r in committed} == {(1, 1.0), (1, 3.0), (8, 1.0), (8, 3.0)}

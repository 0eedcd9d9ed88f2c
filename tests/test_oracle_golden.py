"""CPU: the fp32 C restatement (oracle/cnf_oracle.c) against the committed fp64 fixtures.

Tolerances are fp32-vs-fp64 rounding: a single dynamics call agrees to ~1e-5 relative; a
40-step solve accumulates 160-240 calls, so log-densities are held to 5e-5 absolute (the
north_star bound for the product is 1e-4)."""
import numpy as np
import pytest

from conftest import GOLDEN_NAMES, load_golden


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_c_oracle_aug_f_matches_golden(name, oracles):
    _, oc = oracles
    spec, meta, g = load_golden(name)
    du = oc.aug_f(spec, g["p"], g["u"], float(g["t"]), g["eps"], g["ys"], nthreads=2)
    scale = 1.0 + np.abs(g["du"])
    assert np.max(np.abs(du - g["du"]) / scale) < 2e-5


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_c_oracle_inference_matches_golden(name, oracles):
    _, oc = oracles
    spec, meta, g = load_golden(name)
    logp, (E, n, A), u1 = oc.inference_fixed(spec, g["p"], g["xs"], 0.0, 1.0, meta["nsteps"],
                                             meta["alg"], g["eps"], g["ys"], nthreads=2)
    assert np.max(np.abs(logp - g["logp"])) < 5e-5
    assert np.max(np.abs(E - g["E"])) < 5e-5
    assert np.max(np.abs(n - g["n"])) < 5e-5
    assert np.max(np.abs(A - g["A"])) < 5e-5
    assert np.max(np.abs(u1 - g["u1"])) < 5e-5


def test_c_oracle_thread_count_does_not_change_bits(oracles):
    o64, oc = oracles
    spec = o64.make_spec(8, [64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 200, 3)
    a = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 5, o64.ALG_TSIT5, eps, nthreads=1)[0]
    b = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 5, o64.ALG_TSIT5, eps, nthreads=4)[0]
    assert np.array_equal(a, b)


def test_c_oracle_ragged_batch_and_single_column(oracles):
    o64, oc = oracles
    spec = o64.make_spec(2, [32, 32])
    p, xs, eps, _ = o64.synth_inputs(spec, 130, 4)   # 130 = 2 blocks of 64 + 2
    full = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 4, o64.ALG_RK4, eps)[0]
    one = oc.inference_fixed(spec, p, xs[:, 129:], 0.0, 1.0, 4, o64.ALG_RK4, eps[:, 129:])[0]
    assert np.array_equal(full[129:], one)

"""Generate the committed golden fixtures from the independent fp64 autograd oracle
(oracle/cnf_oracle64.py).  Run from the repo root:  python tests/golden/make_golden.py

The reference itself cannot be executed (pure Julia, no toolchain here) and holds no golden
vectors for this path, so these fixtures pin the C restatement and the HIP kernels against an
independent second implementation — "parity unpinned by the reference" (DESIGN.md §oracle).
Each .npz holds the inputs (p, xs, eps, ys, u, t) in float32 and the fp64 outputs of one
dynamics call (du) and of a fixed-step solve (logp, E, n, A, u1).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cnf_oracle64 as o  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name -> (make_spec kwargs, B, alg, nsteps, seed)
CASES = {
    "cfg1_ffjord_d2_2x32_tsit5": (dict(nvars=2, hidden=[32, 32]), 32, o.ALG_TSIT5, 40, 20240613),
    "cfg2_ffjord_d8_3x64_rk4": (dict(nvars=8, hidden=[64, 64, 64]), 24, o.ALG_RK4, 40, 20240614),
    "cfg2p_ffjord_d8_3x64_tsit5": (dict(nvars=8, hidden=[64, 64, 64]), 24, o.ALG_TSIT5, 40, 20240614),
    "cfg3_rnode_d8_3x64_tsit5_k4": (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True,
                                         reg_j=True), 16, o.ALG_TSIT5, 40, 20240615),
    "cfg4_ffjord_d32_3x256_rk4": (dict(nvars=32, hidden=[256, 256, 256]), 4, o.ALG_RK4, 40, 20240616),
    "cfg5_cond_d8c8_3x128_exact_rk4": (dict(nvars=8, ncond=8, hidden=[128, 128, 128],
                                            mode=o.MODE_EXACT), 8, o.ALG_RK4, 40, 20240617),
    "default_softplus_aug_train": (dict(nvars=1, naug=2, hidden=[16, 16], act=o.ACT_SOFTPLUS,
                                        reg_z=True, reg_j=True, reg_aug=True), 16, o.ALG_TSIT5, 20, 7),
    "jvp_d3_2x16_autonomous": (dict(nvars=3, hidden=[16, 16], autonomous=True,
                                    mode=o.MODE_HUTCH_JVP, reg_z=True, reg_j=True), 8, o.ALG_RK4, 10, 8),
    "ragged_d5_widths_24_40_softplus_tanh": (None, 10, o.ALG_TSIT5, 10, 9),
}


def build_spec(name, kw):
    if kw is not None:
        return o.make_spec(**kw)
    # non-uniform widths and mixed activations: exercises the generic path
    s = o.Spec(nvars=4, naug=1, ncond=2, widths=[8, 24, 40, 5],
               acts=[o.ACT_SOFTPLUS, o.ACT_TANH, o.ACT_IDENTITY], nprobes=2, reg_z=True, reg_j=True,
               reg_aug=True)
    s.check()
    return s


def main():
    index = {}
    for name, (kw, B, alg, nsteps, seed) in CASES.items():
        spec = build_spec(name, kw)
        p, xs, eps, ys = o.synth_inputs(spec, B, seed, bias_scale=0.1)
        rng = np.random.default_rng(seed + 1)
        u = rng.standard_normal((spec.S, B)).astype(np.float32)
        t = np.float32(0.37)
        du = o.aug_f(spec, p, u, float(t), eps, ys)
        logp, (E, n, A), u1 = o.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys)
        arrays = dict(p=p, xs=xs, eps=eps, u=u, t=t, du=du, logp=logp, E=E, n=n, A=A, u1=u1)
        if ys is not None:
            arrays["ys"] = ys
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(nvars=spec.nvars, naug=spec.naug, ncond=spec.ncond,
                           autonomous=bool(spec.autonomous), widths=list(spec.widths),
                           acts=list(spec.acts), mode=spec.mode, nprobes=spec.nprobes,
                           reg_z=bool(spec.reg_z), reg_j=bool(spec.reg_j), reg_aug=bool(spec.reg_aug),
                           B=B, alg=alg, nsteps=nsteps, seed=seed)
        print(name, "logp[:3] =", logp[:3])
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()


def make_gradient_fixture():
    """Groundwork for the parameter-gradient row: dloss/dp of the discrete loss (cfg2 shape, B = 8,
    Tsit5 x 10) from oracle/cnf_oracle64.py::loss_and_grad."""
    spec = o.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o.synth_inputs(spec, 8, 20240618, bias_scale=0.1)
    L, g = o.loss_and_grad(spec, p, xs, 0.0, 1.0, 10, o.ALG_TSIT5, eps)
    np.savez_compressed(os.path.join(OUT, "grad_cfg2_d8_3x64_tsit5.npz"), p=p, xs=xs, eps=eps, nsteps=10,
                        loss=L, grad=g)


if __name__ == "__main__":
    make_gradient_fixture()


def make_vcabm_fixture():
    """The reference's default solver on a small flow (default-style softplus net, nvariables = 2, naugments = 3, RNODE terms on,
    B = 24, tolerances 1e-4 - the defaults - and 1e-6): final state, accepted / rejected counts, order history and steps of
    oracle/cnf_oracle64.py::integrate_vcabm."""
    spec = o.make_spec(nvars=2, naug=3, hidden=[24, 24], act=2, reg_z=True, reg_j=True)
    p, xs, eps, _ = o.synth_inputs(spec, 24, 20240704, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, 24))])
    out = dict(p=p, xs=xs, eps=eps)
    for tag, tol in (("a", 1e-4), ("b", 1e-6)):
        u1, st = o.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol, eps)
        out.update({f"tol_{tag}": tol, f"u1_{tag}": u1, f"naccept_{tag}": st["naccept"], f"nreject_{tag}": st["nreject"],
                    f"orders_{tag}": np.array(st["orders"]), f"dts_{tag}": np.array(st["dts"])})
    np.savez_compressed(os.path.join(OUT, "vcabm_default_softplus_aug.npz"), **out)


if __name__ == "__main__":
    make_vcabm_fixture()

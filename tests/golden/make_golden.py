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
    # tags a / b: Hairer's initial step (tiny: the exponent is 1 / current order = 1, so the first steps grow tenfold each and a
    # Float32 run may take a few more or fewer steps); a0 / b0: a common explicit initial step 2^-7, where the HIP path must
    # reproduce the decisions step for step
    for tag, tol, dt0 in (("a", 1e-4, None), ("b", 1e-6, None), ("a0", 1e-4, 2.0 ** -7), ("b0", 1e-6, 2.0 ** -7)):
        u1, st = o.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol, eps, dt0=dt0)
        out.update({f"tol_{tag}": tol, f"u1_{tag}": u1, f"naccept_{tag}": st["naccept"], f"nreject_{tag}": st["nreject"],
                    f"orders_{tag}": np.array(st["orders"]), f"dts_{tag}": np.array(st["dts"])})
    np.savez_compressed(os.path.join(OUT, "vcabm_default_softplus_aug.npz"), **out)


if __name__ == "__main__":
    make_vcabm_fixture()


def make_generate_fixtures():
    """`generate` (src/core/base_icnf.jl:351-404, 185-194): the reversed-tspan solve from a given base sample z0 - the final
    state u1 (whose first nvars rows are the samples) of oracle/cnf_oracle64.py::integrate_fixed(t1 -> t0), for the headline
    shape (Tsit5), a conditioned exact-trace flow (RK4) and an augmented softplus flow; plus the same three under
    OrdinaryDiffEq's fixed-dt stepping with a span that is not a multiple of dt (a shorter last step: what a STEER-drawn
    end time meets, base_icnf.jl:23-43), forwards."""
    cases = {
        "generate_cfg2p": (dict(nvars=8, hidden=[64, 64, 64]), 16, o.ALG_TSIT5, 40, 31),
        "generate_cfg5_cond_exact": (dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=o.MODE_EXACT), 8, o.ALG_RK4, 40, 32),
        "generate_aug_softplus": (dict(nvars=1, naug=2, hidden=[16, 16], act=o.ACT_SOFTPLUS, reg_z=True, reg_j=True,
                                       reg_aug=True), 12, o.ALG_TSIT5, 20, 33),
    }
    for name, (kw, B, alg, nsteps, seed) in cases.items():
        spec = o.make_spec(**kw)
        p, _, eps, ys = o.synth_inputs(spec, B, seed, bias_scale=0.1)
        rng = np.random.default_rng(seed + 7)
        z0 = rng.standard_normal((spec.D, B)).astype(np.float32)
        u0 = np.concatenate([z0.astype(np.float64), np.zeros((3, B))], axis=0)
        u1 = o.integrate_fixed(spec, p, u0, 1.0, 0.0, nsteps, alg, eps, ys)            # reversed tspan
        t1s = 1.0 + 0.1 * (2.0 * rng.random() - 1.0)                                  # a STEER-style end time
        xs = z0[:spec.nvars]
        logp_s, regs_s, u1_s = o.inference_fixed(spec, p, xs, 0.0, t1s, 0, alg, eps, ys, dt=1.0 / nsteps)
        arrays = dict(p=p, z0=z0, eps=eps, u1=u1, nsteps=nsteps, alg=alg, t1_steer=t1s, logp_steer=logp_s, u1_steer=u1_s,
                      E_steer=regs_s[0], n_steer=regs_s[1], A_steer=regs_s[2], spec=json.dumps(dict(kw)))
        if ys is not None:
            arrays["ys"] = ys
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        print(name, "x[:, 0] =", u1[:spec.nvars, 0][:3], "t1_steer =", t1s, "grid steps =", len(o.fixed_dt_grid(0.0, t1s, 1.0 / nsteps)) - 1)


if __name__ == "__main__":
    make_generate_fixtures()

"""Wide conditioned / several-probe / exact-trace flows: the extended cooperative kernel (csrc/cnf_coop_x.hip) against the
layer-wise path they took before (CNF_MFMA_COOPX=0), whole 40-step solves, one MI355X."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
CASES = [
    ("cfg5 flow at H=256: D=8, C=8, 3x256, exact trace, RK4x40, B=16384", dict(nvars=8, ncond=8, hidden=[256] * 3, mode=2), 16384, "rk4"),
    ("cfg5 flow at H=192: D=8, C=8, 3x192, exact trace, RK4x40, B=16384", dict(nvars=8, ncond=8, hidden=[192] * 3, mode=2), 16384, "rk4"),
    ("CondFFJORD D=8, C=8, 3x256, Hutchinson(1), Tsit5x40, B=32768", dict(nvars=8, ncond=8, hidden=[256] * 3), 32768, "tsit5"),
    ("RNODE D=32, 3x256, Hutchinson(4), RK4x40, B=32768", dict(nvars=32, hidden=[256] * 3, nprobes=4, reg_z=True, reg_j=True), 32768, "rk4"),
]
out = {}
for name, kw, B, alg in CASES:
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 5)
    t = lambda a: None if a is None else torch.tensor(np.ascontiguousarray(a.T), device=dev).t()
    X, E, Y, P = t(xs), t(eps), t(ys), torch.tensor(p, device=dev)
    r = {}
    for tag, env in (("coopx", "1"), ("layered", "0")):
        os.environ["CNF_MFMA_COOPX"] = env; pkg.reload_tuning()
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        reg = bool(spec.reg_z or spec.reg_j)
        icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nconditions=spec.ncond, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(),
                        steer_rate=0.0, lambda1=0.01 if spec.reg_z else 0.0, lambda2=0.01 if spec.reg_j else 0.0, lambda3=0.0, nprobes=spec.nprobes,
                        device=dev, sol_kwargs=dict(alg=pkg.Tsit5() if alg == "tsit5" else pkg.RK4(), adaptive=False, nsteps=40))
        mode = pkg.TestMode() if spec.mode == 2 else pkg.TrainMode(reg)
        args = (X,) + ((Y,) if Y is not None else ()) + (P, {})
        fn = lambda: pkg.inference(icnf, mode, *args, eps=E, _raw=True)
        lp = fn()[0]
        torch.cuda.synchronize()
        reps = 5 if tag == "coopx" else 2
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        r[tag + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
        r[tag + "_path"] = icnf.kernel_path(mode)
        r[tag + "_logp0"] = float(lp[0])
    r["speedup"] = round(r["layered_ms"] / r["coopx_ms"], 2)
    r["max_abs_dlogp_between_paths"] = abs(r["coopx_logp0"] - r["layered_logp0"])
    out[name] = r
del os.environ["CNF_MFMA_COOPX"]; pkg.reload_tuning()
print(json.dumps(out, indent=1))

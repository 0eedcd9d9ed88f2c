"""Latency of one solve at the reference's own batch sizes: fixed-step Tsit5 x 40 and the adaptive
default-tolerance Tsit5 solve, for the cfg1 net (D=2, 2x32) and the cfg2 net (D=8, 3x64), B = 16 .. 16384.
Prints one JSON object: {net: {B: {"fixed_ms": .., "adaptive_ms": .. (one launch), "adaptive_hostloop_ms": .., "vcabm_ms": .. (the reference's default solver, host policy loop)}}}."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
out = {"lib": os.environ.get("CNF_HIP_LIB", "default")}
nets = (("d2_2x32", dict(nvars=2, hidden=[32, 32])), ("d8_3x64", dict(nvars=8, hidden=[64] * 3)))
for name, kw in nets:
    out[name] = {}
    for B in (16, 256, 1024, 4096, 16384):
        spec = o64.make_spec(**kw)
        p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
        X = torch.tensor(xs.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
        E = torch.tensor(eps.T.copy(), device=dev).t()
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        m = pkg.TrainMode(False)
        r = {}
        for tag, sk in (("fixed", dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40)), ("adaptive", dict(alg=pkg.Tsit5(), adaptive=True, reltol=1e-4, abstol=1e-4)),
                        ("vcabm", dict(reltol=1e-4, abstol=1e-4))):
            icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), steer_rate=0.0,
                            lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev, sol_kwargs=sk)
            fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            r[tag + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
            if tag != "fixed":
                st = getattr(icnf, "last_solve_stats", None) or {}
                r[tag + "_steps"] = st.get("naccept"), st.get("nreject")
        if os.environ.get("CNF_DEVICE_CONTROLLER", "1") != "0":     # the host-loop twin of the adaptive Tsit5 solve
            os.environ["CNF_DEVICE_CONTROLLER"] = "0"; pkg.reload_tuning()
            icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), steer_rate=0.0,
                            lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=True, reltol=1e-4, abstol=1e-4))
            fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30): fn()
            torch.cuda.synchronize()
            r["adaptive_hostloop_ms"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
            del os.environ["CNF_DEVICE_CONTROLLER"]; pkg.reload_tuning()
        out[name][str(B)] = r
print(json.dumps(out))

import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
for name, kw, alg in (("d8_3x64_tsit5", dict(nvars=8, hidden=[64] * 3), "tsit5"), ("d8_3x64_rk4", dict(nvars=8, hidden=[64] * 3), "rk4")):
    out[name] = {}
    for B in (16, 256, 1024, 2048, 4096):
        spec = o64.make_spec(**kw)
        p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
        X = torch.tensor(xs.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev); E = torch.tensor(eps.T.copy(), device=dev).t()
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        m = pkg.TrainMode(False)
        icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0,
                        device=dev, sol_kwargs=dict(alg=pkg.Tsit5() if alg == "tsit5" else pkg.RK4(), adaptive=False, nsteps=40))
        r = {}
        for tag, env in (("wave", "0"), ("split", "2")):
            os.environ["CNF_TILE_SPLIT"] = env; pkg.reload_tuning()
            fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E, _raw=True)
            for _ in range(5): fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            t0 = time.perf_counter()
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            r[tag + "_wall_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
            r[tag + "_kernel_ms"] = round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)
        out[name][str(B)] = r
print(json.dumps(out))

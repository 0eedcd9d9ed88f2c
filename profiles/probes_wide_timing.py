"""Loss + gradient with K = 4 probes on wide nets, B = 32 768, 40 steps: probe by probe on the one-probe cooperative / dealt reverse sweep
(cnf_handle::grad_twin, round 5) against the configuration's own layer-wise gradient (CNF_PROBE_GRAD_TWIN=0), and the one-probe gradient
of the same shape for scale.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
ACTS = ["identity", "tanh", "softplus"]
out = {}
for name, kw, B in (("d32_3x256_rk4", dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 32768),
                    ("nv16", dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), 32768),
                    ("nv20", dict(nvars=20, naug=21, hidden=[168, 168], act=2, reg_z=True, reg_j=True, reg_aug=True), 32768)):
    res = {}
    for tag, K, env in (("k1", 1, "1"), ("k4_probe_by_probe", 4, "2"), ("k4_layerwise", 4, "0")):   # 2: the loop on three hidden layers too
        os.environ["CNF_PROBE_GRAD_TWIN"] = env; pkg.reload_tuning()
        spec = o64.make_spec(nprobes=K, **kw)
        p, xs, eps, _ = o64.synth_inputs(spec, B, 3, bias_scale=0.1)
        X, E, P = (torch.tensor(np.ascontiguousarray(a), device=dev) for a in (xs, eps, p))
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
        alg = pkg.RK4() if "rk4" in name else pkg.Tsit5()
        icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), nprobes=K,
                        steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.01 if spec.reg_aug else 0.0, device=dev,
                        sol_kwargs=dict(alg=alg, adaptive=False, nsteps=40))
        m = pkg.TrainMode(True)
        for _ in range(2): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3
        for _ in range(n): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
        torch.cuda.synchronize()
        res[tag] = dict(ms=round(1e3 * (time.perf_counter() - t0) / n, 2), grad_path=icnf.grad_path(m, B=B, alg=0 if "rk4" in name else 1), loss=float(v))
    res["speedup"] = round(res["k4_layerwise"]["ms"] / res["k4_probe_by_probe"]["ms"], 3)
    out[name] = res
os.environ.pop("CNF_PROBE_GRAD_TWIN", None)
print(json.dumps(out))

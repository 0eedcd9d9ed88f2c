"""Loss + gradient in Hutchinson JVP mode without the |J eps| regulariser: the VJP twin's fused reverse sweeps (cnf_handle::grad_twin,
round 5) against the mode's own layer-wise gradient (CNF_JVP_GRAD_TWIN=0); Tsit5 x 40, B = 32 768 (3 x 64: 65 536)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
ACTS = ["identity", "tanh", "softplus"]
CASES = {"d8_3x64": (dict(nvars=8, hidden=[64, 64, 64], mode=1), 65536),
         "d32_3x256": (dict(nvars=32, hidden=[256, 256, 256], mode=1), 32768),
         "default_nv20": (dict(nvars=20, naug=21, hidden=[168, 168], act=2, mode=1, reg_z=True, reg_aug=True), 32768)}
out = {}
for name, (kw, B) in CASES.items():
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 3, bias_scale=0.1)
    X, E, P = (torch.tensor(np.ascontiguousarray(a), device=dev) for a in (xs, eps, p))
    row = {}
    for tag, env in (("twin", "1"), ("layerwise", "0")):
        os.environ["CNF_JVP_GRAD_TWIN"] = env; pkg.reload_tuning()
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
        icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nn=pkg.Chain(*layers), compute_mode=pkg.HIPJacVecMatrixMode(),
                        steer_rate=0.0, lambda1=0.01 if spec.reg_z else 0.0, lambda2=0.0, lambda3=0.01 if spec.reg_aug else 0.0, device=dev,
                        sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        m = pkg.TrainMode(bool(spec.reg_z or spec.reg_aug))
        for _ in range(2): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
        torch.cuda.synchronize()
        row[tag] = dict(ms=1e3 * (time.perf_counter() - t0) / n, grad_path=icnf.grad_path(m, B=B, alg=1), loss=float(v))
    row["speedup"] = row["layerwise"]["ms"] / row["twin"]["ms"]
    out[name] = row
print(json.dumps(out))

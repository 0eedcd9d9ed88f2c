import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
ACTS = ["identity", "tanh", "softplus"]
out = {}
for name, kw, B in (("d32_3x256_k4_rk4", dict(nvars=32, hidden=[256, 256, 256], nprobes=4, reg_z=True, reg_j=True), 32768),
                    ("d32_3x256_k1_rk4", dict(nvars=32, hidden=[256, 256, 256], nprobes=1, reg_z=True, reg_j=True), 32768),
                    ("nv20_k4", dict(nvars=20, naug=21, hidden=[168, 168], act=2, nprobes=4, reg_z=True, reg_j=True, reg_aug=True), 32768),
                    ("nv20_k1", dict(nvars=20, naug=21, hidden=[168, 168], act=2, nprobes=1, reg_z=True, reg_j=True, reg_aug=True), 32768)):
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 3, bias_scale=0.1)
    X, E, P = (torch.tensor(np.ascontiguousarray(a), device=dev) for a in (xs, eps, p))
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
    alg = pkg.RK4() if "rk4" in name else pkg.Tsit5()
    icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), nprobes=spec.nprobes,
                    steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.01 if spec.reg_aug else 0.0, device=dev,
                    sol_kwargs=dict(alg=alg, adaptive=False, nsteps=40))
    m = pkg.TrainMode(True)
    for _ in range(2): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): v, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
    torch.cuda.synchronize()
    out[name] = dict(ms=1e3 * (time.perf_counter() - t0) / n, grad_path=icnf.grad_path(m, B=B, alg=0 if "rk4" in name else 1))
print(json.dumps(out))

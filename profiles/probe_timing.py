import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); o64, oc = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
for nv, K in ((8, 4), (7, 4), (7, 2), (7, 8), (7, 1)):
    spec = o64.make_spec(nvars=nv, hidden=[64, 64, 64], nprobes=K, reg_z=True, reg_j=True)
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(4)]
    icnf = pkg.ICNF(nvariables=nv, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=2), steer_rate=0.0,
                    lambda1=0.01, lambda2=0.01, lambda3=0.0, nprobes=K, device=dev,
                    sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    B = 65536
    p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
    X = torch.tensor(xs.T.copy(), device=dev).t(); E = torch.tensor(eps.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
    m = pkg.TrainMode(True)
    r = {}
    for name, fn in (("infer", lambda: pkg.loss(icnf, m, X, P, {}, eps=E)), ("grad", lambda: pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E))):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(4): fn()
        t1.record(); torch.cuda.synchronize()
        r[name + "_ms"] = t0.elapsed_time(t1) / 4
    out[f"nvars{nv}_K{K}"] = r
print(json.dumps(out, indent=1))

"""Timing of cnf_loss_grad_fixed (forward solve with checkpoints + reverse sweep) at the headline
shape: FFJORD D=8, 3x64 tanh, B=65536, 40 fixed steps."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
spec = o64.make_spec(nvars=8, hidden=[64, 64, 64]); B = 65536
p, xs, eps, _ = o64.synth_inputs(spec, B, 20240612)
out = {}
for name, alg in (("RK4", pkg.RK4()), ("Tsit5", pkg.Tsit5())):
    icnf = pkg.ICNF(nvariables=8, naugments=0, steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev,
                    nn=pkg.Chain(pkg.Dense(9, 64, "tanh"), pkg.Dense(64, 64, "tanh"), pkg.Dense(64, 64, "tanh"), pkg.Dense(64, 8)),
                    sol_kwargs=dict(alg=alg, adaptive=False, nsteps=40))
    X = torch.tensor(xs.T.copy(), device=dev).t(); E = torch.tensor(eps.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
    mode = pkg.TrainMode(False)
    for _ in range(2): pkg.loss_and_gradient(icnf, mode, X, P, {}, eps=E)
    torch.cuda.synchronize()
    n = 5
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): val, g = pkg.loss_and_gradient(icnf, mode, X, P, {}, eps=E)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / n
    t0.record()
    for _ in range(n): pkg.inference(icnf, mode, X, P, {}, eps=E)
    t1.record(); torch.cuda.synchronize()
    inf = t0.elapsed_time(t1) / n
    out[name] = dict(grad_ms=ms, inference_ms=inf, ratio=ms / inf, samples_steps_per_s=B * 40 / (ms * 1e-3),
                     loss=float(val), grad_norm=float(g.norm()))
print(json.dumps(out, indent=1))

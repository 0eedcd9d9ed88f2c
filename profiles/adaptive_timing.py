"""Adaptive solves (reltol = abstol = 1e-4, the reference's defaults; Tsit5 and the reference's default VCABM) against the
fixed 40-step solve at cfg2 scale."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
B = 65536
p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
X = torch.tensor(xs.T.copy(), device=dev).t(); E = torch.tensor(eps.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
out = {}
for name, kw in (("fixed_tsit5_40", dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40)),
                 ("adaptive_1e-4", dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4)),
                 ("adaptive_1e-6", dict(alg=pkg.Tsit5(), reltol=1e-6, abstol=1e-6)),
                 ("vcabm_1e-4", dict(alg=pkg.VCABM(), reltol=1e-4, abstol=1e-4)),
                 ("vcabm_1e-6", dict(alg=pkg.VCABM(), reltol=1e-6, abstol=1e-6))):
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(4)]
    icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0,
                    device=dev, sol_kwargs=kw)
    m = pkg.TrainMode(False)
    fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E)[0]
    ref = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): lp = fn()
    torch.cuda.synchronize()
    r = {"ms": 1e3 * (time.perf_counter() - t0) / 5}
    if icnf.adaptive:
        st = icnf.last_solve_stats
        r.update(naccept=st["naccept"], nreject=st["nreject"], nf=st["nf"])
        if "orders" in st: r["orders"] = st["orders"]
    g = lambda: pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)[1]
    g(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g()
    torch.cuda.synchronize()
    r["loss_and_gradient_ms"] = 1e3 * (time.perf_counter() - t0) / 5
    r["grad_path"] = icnf.grad_path(m)
    out[name] = r
    if name == "fixed_tsit5_40": base = lp
    else: r["max_abs_dlogp_vs_fixed40"] = float((lp - base).abs().max())
print(json.dumps(out))

import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for nv in [int(v) for v in os.environ.get("NVS", "16,20,24,32,40,47").split(",")]:
    icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    B = 32768
    X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev)
    E = torch.randn(B, icnf.D, device=dev).t()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode, name in ((pkg.TrainMode(True), "train"), (pkg.TestMode(), "test")):
        for _ in range(2): pkg.inference(icnf, mode, X, P, st, eps=E)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(3): pkg.inference(icnf, mode, X, P, st, eps=E)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 3
        W = icnf.nn.widths
        flop = 2 * 2 * (W[0]*W[1] + W[1]*W[2] + W[2]*W[3]) * B * 240
        out[f"nv{nv}_{name}"] = dict(widths=W, path=icnf.kernel_path(mode), ms=round(ms,2), tflops_fwd_vjp=round(flop/ms/1e9,1) if name=="train" else None)
    m = pkg.TrainMode(True)
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
    torch.cuda.synchronize(); t0.record()
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
    t1.record(); torch.cuda.synchronize()
    out[f"nv{nv}_grad"] = dict(ms=round(t0.elapsed_time(t1) / 2,1), path=icnf.grad_path(m))
for k,v in out.items(): print(k, v)

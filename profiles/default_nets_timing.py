import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, oc = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
for nv in [int(v) for v in os.environ.get("NVS", "1,2,4,8,12,15").split(",")]:
    icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    B = 65536
    X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev)
    E = torch.randn(B, icnf.D, device=dev).t()
    for mode, name in ((pkg.TrainMode(True), "train"), (pkg.TestMode(), "test")):
        for _ in range(2): pkg.inference(icnf, mode, X, P, st, eps=E)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5): pkg.inference(icnf, mode, X, P, st, eps=E)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        out[f"nv{nv}_{name}"] = dict(widths=icnf.nn.widths, path=icnf.kernel_path(mode), ms=ms, samples_steps_per_s=B * 40 / ms * 1e3)
    if True:
        m = pkg.TrainMode(True)
        for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
        torch.cuda.synchronize(); t0.record()
        for _ in range(3): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
        t1.record(); torch.cuda.synchronize()
        out[f"nv{nv}_grad"] = dict(ms=t0.elapsed_time(t1) / 3)
print(json.dumps(out))

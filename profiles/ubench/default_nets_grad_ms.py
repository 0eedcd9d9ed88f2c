"""loss + gradient ms of the default architecture at several nvariables (B = 32 768, Tsit5 x 40): for A/B runs of environment
switches that are read once per process (e.g. CNF_LG_WGRAD_PER_CU).  python profiles/ubench/default_nets_grad_ms.py [nv ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
nvs = [int(v) for v in sys.argv[1:]] or [12, 16, 20, 24, 28, 32, 40]
out = []
for nv in nvs:
    B = 32768
    ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
    X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, ic.D, device=dev).t()
    m = pkg.TrainMode(True)
    for _ in range(2): pkg.loss_and_gradient(ic, m, X, P, st, eps=E)
    torch.cuda.synchronize(); t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3): pkg.loss_and_gradient(ic, m, X, P, st, eps=E)
    t1.record(); torch.cuda.synchronize()
    out.append((nv, round(t0.elapsed_time(t1) / 3, 2)))
print(os.environ.get("CNF_LG_WGRAD_PER_CU", "default"), out, flush=True)

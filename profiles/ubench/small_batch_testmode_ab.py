import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
for nv in (32, 40, 45):
    for B in (512, 4096):
        res = {}
        for env in ("1", "0"):
            os.environ["CNF_COOPD"] = env
            ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
            ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
            X = torch.randn(B, nv, device=dev).t()
            m = pkg.TestMode()
            for _ in range(2): pkg.inference(ic, m, X, P, st)
            torch.cuda.synchronize(); t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5): pkg.inference(ic, m, X, P, st)
            t1.record(); torch.cuda.synchronize()
            res[env] = (round(t0.elapsed_time(t1) / 5, 3), ic.kernel_family(m, B=B))
        print("TestMode", nv, B, res, flush=True)

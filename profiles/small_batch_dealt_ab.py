"""A/B of the dealt kernels at small batches (CNF_COOPD: 1 = the dispatch's choice, 2 = dealt forced, 0 = dealt off):
TrainMode inference and loss + gradient in ms, Tsit5 x 40 fixed steps.  gpurun -- python profiles/small_batch_dealt_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return round(t0.elapsed_time(t1) / n, 3)


for nv in (20, 32, 40):
    for B in (256, 1024, 4096):
        res = {}
        for env in ("1", "2", "0"):
            os.environ["CNF_COOPD"] = env; pkg.reload_tuning()
            ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
            ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
            X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, ic.D, device=dev).t()
            m = pkg.TrainMode(True)
            res[env] = (timed(lambda: pkg.inference(ic, m, X, P, st, eps=E)), timed(lambda: pkg.loss_and_gradient(ic, m, X, P, st, eps=E)),
                        ic.kernel_family(m, B=B))
        print(nv, B, res, flush=True)

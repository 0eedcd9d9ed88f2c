import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
torch.manual_seed(0)
for nv in (7, 8, 10, 12, 15):
    for B in (1024, 65536):
        icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
        m = pkg.TrainMode(True)
        r = {}
        for name, env in (("slab", None), ("slab_own_forward", "ckpt0"), ("layered", "1")):   # slab: one forward solve for loss terms and checkpoints (round 5)
            os.environ.pop("CNF_GRAD_LAYERED", None); os.environ.pop("CNF_ADAPTIVE_CKPT", None)
            if env == "1": os.environ["CNF_GRAD_LAYERED"] = "1"
            if env == "ckpt0": os.environ["CNF_ADAPTIVE_CKPT"] = "0"
            pkg.reload_tuning()
            fn = lambda: pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): v, g = fn()
            torch.cuda.synchronize()
            r[name + "_ms"] = round(1e3 * (time.perf_counter() - t0) / 3, 2)
        os.environ.pop("CNF_GRAD_LAYERED", None); os.environ.pop("CNF_ADAPTIVE_CKPT", None); pkg.reload_tuning()
        out[f"nv{nv}_B{B}"] = r
print(json.dumps(out))

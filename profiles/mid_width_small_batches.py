"""Two-hidden-layer nets of 7 - 8 hidden tiles (the default architecture at nvariables = 12 ... 15): loss + gradient on the auxiliary
cooperative sweep (the default at every batch size since round 5) against the slab-accumulator kernel (CNF_COOP_GRAD_MID=0), Tsit5 x 40
and under the adaptive solver (tolerance 1e-4), B = 64 ... 65 536."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for nv in (12, 13, 14, 15):
    for B in (64, 1024, 4000, 8192, 65536):
        r = {}
        for name, env in (("slab", "0"), ("aux", "1")):
            os.environ["CNF_COOP_GRAD_MID"] = env; pkg.reload_tuning()
            for tag, kw in (("fixed", dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40)), ("adaptive", dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))):
                if tag == "adaptive" and B > 8192: continue
                icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=kw)
                ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
                X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
                m = pkg.TrainMode(True)
                fn = lambda: pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
                fn(); fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): v, g = fn()
                torch.cuda.synchronize()
                r[f"{name}_{tag}_ms"] = round(1e3 * (time.perf_counter() - t0) / 3, 2)
        out[f"nv{nv}_B{B}"] = r
os.environ.pop("CNF_COOP_GRAD_MID", None)
print(json.dumps(out))

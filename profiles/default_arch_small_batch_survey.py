"""Survey: the reference's default architecture ICNF(nvariables = nv) at B = 1024 and 8192, Tsit5 x 40: inference and loss + gradient per
call, with the kernel family and the gradient path each call takes - to spot dispatch anomalies (a jump between neighbours)."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for nv in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 15, 16, 18, 20, 24, 28, 30, 32, 40, 47):
    for B in (1024, 8192):
        icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
        m = pkg.TrainMode(True)
        r = {}
        for name, fn in (("inference_ms", lambda: pkg.inference(icnf, m, X, P, st, eps=E)), ("loss_grad_ms", lambda: pkg.loss_and_gradient(icnf, m, X, P, st, eps=E))):
            fn(); fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): fn()
            torch.cuda.synchronize()
            r[name] = round(1e3 * (time.perf_counter() - t0) / 3, 2)
        r["family"] = icnf.kernel_family(m, B=B); r["grad_path"] = icnf.grad_path(m, B=B, alg=1)
        out[f"nv{nv}_B{B}"] = r
print(json.dumps(out))

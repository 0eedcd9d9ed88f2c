# Forward solves at full size: the dealt kernels against the extended kernel (CNF_COOPD=0) on the same plan - TrainMode and TestMode
# of the default architecture, tanh flows of the same sizes - every column, B = 32 768 (128 workgroups x 4 .. 8 super-tiles each).
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
B = int(os.environ.get("DG_B", "32768"))
cases = [("default", nv, mode) for nv in (16, 22, 29, 32, 38, 43, 47) for mode in ("train", "test")] + [("tanh", h, "train") for h in (176, 232, 264, 340)]
worst = 0.0
for kind, n, mode in cases:
    torch.manual_seed(2)
    if kind == "default":
        mk = lambda: pkg.ICNF(nvariables=n, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=20))
        nv = n
    else:
        nv, D = 24, 24
        layers = [pkg.Dense(D + 1, n, "tanh"), pkg.Dense(n, n, "tanh"), pkg.Dense(n, D, "identity")]
        mk = lambda: pkg.ICNF(nvariables=nv, naugments=0, nn=pkg.Chain(*layers), device=dev, steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.0,
                              sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=20))
    ic = mk()
    ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
    X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, ic.D, device=dev).t()
    m = pkg.TrainMode(True) if mode == "train" else pkg.TestMode()
    out = {}
    for tag, env in (("dealt", "1"), ("extended", "0")):
        os.environ["CNF_COOPD"] = env; pkg.reload_tuning()
        ic2 = mk()
        r = pkg.inference(ic2, m, X, P, st, eps=E)
        out[tag] = (r[0].double().cpu(), ic2.kernel_family(m, B=B))
    d = float((out["dealt"][0] - out["extended"][0]).abs().max()); sc = float(out["extended"][0].abs().max())
    worst = max(worst, d / (1.0 + sc))
    print(f"{kind} {n} {mode}: families {out['dealt'][1]} / {out['extended'][1]}, max |dlogp| {d:.2e} at |logp| <= {sc:.1f}", flush=True)
print("worst relative", worst)

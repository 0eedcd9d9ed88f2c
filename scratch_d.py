import sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, oc = entry.load_oracle()
dev = torch.device("cuda:0")
def run(nv, B, alg, nsteps, check=True, reps=3):
    res = {}
    for flag in ("1", "0"):
        os.environ["CNF_COOPD"] = flag
        icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5() if alg else pkg.RK4(), adaptive=False, nsteps=nsteps))
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        g = torch.Generator(device="cpu").manual_seed(nv)
        X = torch.randn(B, nv, generator=g).to(dev).t(); P = ps.to(dev)
        E = torch.randn(B, icnf.D, generator=g).to(dev).t()
        mode = pkg.TrainMode(True)
        fam = icnf.kernel_family(mode, B=B)
        out = pkg.inference(icnf, mode, X, P, st, eps=E)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps): out = pkg.inference(icnf, mode, X, P, st, eps=E)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
        W = icnf.nn.widths
        flop = 2 * 2 * (W[0]*W[1] + W[1]*W[2] + W[2]*W[3]) * B * nsteps * (6 if alg else 4)
        res[flag] = (fam, ms, flop / ms / 1e9, out[0].cpu().numpy(), [r.cpu().numpy() for r in out[1]])
    a, b = res["1"], res["0"]
    d = np.abs(a[3] - b[3]).max(); dr = max(np.abs(x - y).max() for x, y in zip(a[4], b[4]))
    print(f"nv={nv} B={B} alg={alg} nsteps={nsteps}: {a[0]} {a[1]:.2f} ms {a[2]:.1f} TF | {b[0]} {b[1]:.2f} ms {b[2]:.1f} TF | max|dlogp| {d:.2e} max|dregs| {dr:.2e}", flush=True)
for nv, B in ((22, 100), (23, 77), (24, 130), (27, 64), (29, 50)):
    run(nv, B, 1, 3, reps=1)
    run(nv, B, 0, 2, reps=1)
for nv in (16, 20, 22, 24, 28, 29):
    run(nv, 32768, 1, 40)

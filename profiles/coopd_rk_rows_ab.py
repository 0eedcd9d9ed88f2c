"""A/B of the dealt forward kernel's Runge-Kutta rows: three rows in LDS + three parked (CNF_COOPD=1, the default where LDS has the
room) against all six in the plan's global ring (CNF_COOPD=3), reference's default architecture at nvariables = NV, B = 32 768,
40 Tsit5 steps, TrainMode: bit-identity of every output and ms per solve."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = []
for nv in [int(v) for v in os.environ.get("NV", "16,18,20,21").split(",")]:
    res = {}
    for tag, sw in (("lds_rows", "1"), ("global_ring", "3")):
        os.environ["CNF_COOPD"] = sw
        pkg.reload_tuning()
        icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        B = 32768
        g = torch.Generator().manual_seed(1)
        X = torch.randn(B, nv, generator=g).to(dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, generator=g).to(dev).t()
        for _ in range(3): r = pkg.inference(icnf, pkg.TrainMode(True), X, P, st, eps=E)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): r = pkg.inference(icnf, pkg.TrainMode(True), X, P, st, eps=E)
        e1.record(); torch.cuda.synchronize()
        res[tag] = (e0.elapsed_time(e1) / 10, [t.clone() for t in r[0]] if isinstance(r[0], (tuple, list)) else [r[0].clone()])
    same = all(torch.equal(a, b) for a, b in zip(res["lds_rows"][1], res["global_ring"][1]))
    out.append(dict(nv=nv, ms_lds_rows=res["lds_rows"][0], ms_global_ring=res["global_ring"][0], bit_identical=same))
    print(json.dumps(out[-1]), flush=True)

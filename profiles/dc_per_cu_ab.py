"""One-launch VCABM solves at two workgroups per CU (CNF_DC_PER_CU=2: capacity 32 768 samples instead of 16 384) against the default:
`loss` per call of the reference's default flow ICNF(nvariables = 1) at B = 1024 ... 65 536."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for per_cu in ("1", "2"):
    os.environ["CNF_DC_PER_CU"] = per_cu; pkg.reload_tuning()
    res = {}
    for B in (1024, 4096, 8192, 16384, 20480, 24576, 28672, 32768, 65536):
        r = torch.distributions.Beta(2.0, 4.0).sample((1, B)).float().to(dev)
        icnf = pkg.ICNF(nvariables=1, device=dev)
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        ps = ps.to(dev)
        mode = pkg.TrainMode(True)
        for _ in range(5): v = pkg.loss(icnf, mode, r, ps, st)
        best = 1e9
        for _rep in range(5):      # best of five runs of 30 calls (the host-loop path's wall time jitters with the host)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30
            for _ in range(n): v = pkg.loss(icnf, mode, r, ps, st)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / n)
        res[str(B)] = dict(ms=round(best, 4), controller=icnf.last_solve_stats["controller"],
                           steps=icnf.last_solve_stats["naccept"], loss=float(v))
    out["per_cu_" + per_cu] = res
print(json.dumps(out))

"""The reference's own PkgBenchmark suite (benchmark/benchmarks.jl: ICNF(; nvariables = 1), every default - default net,
default solver VCABM at 1e-4, default lambdas / steer rate - on 2^10 samples of Beta(2, 4)) on the HIP path:
  direct/train   loss(icnf, TrainMode{true}(), r, ps, st)          direct/test   loss(icnf, TestMode(), r, ps, st)
  AD-1-order/train, AD-1-order/test                                 the gradient of each with respect to ps
(the suite's `inplace` twins are the same calls here).  Also the same four at 2^16 samples."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for ndata in (2 ** 10, 2 ** 16):
    r = torch.distributions.Beta(2.0, 4.0).sample((1, ndata)).float().to(dev)
    icnf = pkg.ICNF(nvariables=1, device=dev)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    ps = ps.to(dev)
    tn, tt = pkg.TrainMode(True), pkg.TestMode()
    cases = {"direct/train": lambda: pkg.loss(icnf, tn, r, ps, st),
             "direct/test": lambda: pkg.loss(icnf, tt, r, ps, st),
             "AD-1-order/train": lambda: pkg.loss_and_gradient(icnf, tn, r, ps, st)[1],
             "AD-1-order/test": lambda: pkg.loss_and_gradient(icnf, tt, r, ps, st)[1]}
    res = {}
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n, reps = 20, []
        for _rep in range(3):       # three runs of 20 calls; `ms` is the best one (a run now and then catches a host hiccup: `ms_runs`)
            t0 = time.perf_counter()
            for _ in range(n):
                v = fn()
            torch.cuda.synchronize()
            reps.append(1e3 * (time.perf_counter() - t0) / n)
        res[name] = {"ms": min(reps), "ms_runs": [round(x, 4) for x in reps], "steps": icnf.last_solve_stats["naccept"],
                     "rejected": icnf.last_solve_stats["nreject"]}
    res["alg"] = type(icnf.sol_kwargs["alg"]).__name__
    res["grad_path"] = {"train": icnf.grad_path(tn), "test": icnf.grad_path(tt)}
    out[f"ndata_{ndata}"] = res
print(json.dumps(out))

"""`ICNF(; nvariables = n)` with EVERY default (net 4 n_in wide softplus, naugments = n + 1, VCABM at 1e-4, lambda = 0.01, STEER) for
n = 1 .. 16: loss and loss + gradient in TrainMode{true} and TestMode at 1024 and 65 536 samples, and which gradient
implementation served it (1 fused, 2 layer-wise)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
out = {}
for n in (1, 2, 3, 4, 6, 8, 12, 15, 16):
    row = {}
    for ndata in (1024, 65536):
        r = torch.distributions.Beta(2.0, 4.0).sample((n, ndata)).float().to(dev)
        icnf = pkg.ICNF(nvariables=n, device=dev)
        ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
        ps = ps.to(dev)
        tn, tt = pkg.TrainMode(True), pkg.TestMode()
        cases = {"loss_train": lambda: pkg.loss(icnf, tn, r, ps, st), "loss_test": lambda: pkg.loss(icnf, tt, r, ps, st),
                 "grad_train": lambda: pkg.loss_and_gradient(icnf, tn, r, ps, st)[1],
                 "grad_test": lambda: pkg.loss_and_gradient(icnf, tt, r, ps, st)[1]}
        res = {}
        for name, fn in cases.items():
            if name == "grad_test" and ndata == 65536 and icnf.grad_path(tt) == 2 and n > 8:
                continue                                   # D passes of the layer-wise path at 65 536 columns: minutes; skipped
            fn(); fn()
            torch.cuda.synchronize()
            k = 5
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            res[name] = round(1e3 * (time.perf_counter() - t0) / k, 3)
        res["steps"] = icnf.last_solve_stats["naccept"]
        row[f"B{ndata}"] = res
    row["D"] = icnf.D
    row["hidden"] = icnf.nn.widths[1]
    row["grad_path"] = {"train": icnf.grad_path(tn), "test": icnf.grad_path(tt)}
    out[f"n{n}"] = row
print(json.dumps(out))

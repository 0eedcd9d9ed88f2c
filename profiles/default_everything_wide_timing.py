"""ICNF(nvariables = nv) with EVERY default (architecture, lambdas, the adaptive default solver at 1e-4, steer off for timing):
loss and loss + gradient at B = 32 768 - the gradient differentiates the adaptive solve on its frozen grid."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
for nv in (16, 20, 32):
    icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    B = 32768
    X = (0.5 * torch.randn(B, nv, device=dev)).t().contiguous(); P = ps.to(dev)
    m = pkg.TrainMode(True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): pkg.loss(icnf, m, X, P, st)
    torch.cuda.synchronize(); t0.record()
    for _ in range(3): pkg.loss(icnf, m, X, P, st)
    t1.record(); torch.cuda.synchronize()
    fwd = t0.elapsed_time(t1) / 3
    nsteps = icnf.last_solve_stats.get("naccept")
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st)
    torch.cuda.synchronize(); t0.record()
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st)
    t1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(nv=nv, widths=icnf.nn.widths, alg=type(icnf.sol_kwargs["alg"]).__name__, loss_ms=round(fwd, 1), loss_steps=nsteps,
                          grad_ms=round(t0.elapsed_time(t1) / 2, 1), grad_path=icnf.grad_path(m), grad_steps=len(icnf.last_solve_stats.get("tgrid", [])) - 1)))

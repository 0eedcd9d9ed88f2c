"""Loss + gradient of the reference's default architecture at nvariables 12 / 15 (two softplus layers of 104 / 128), B = 65 536,
Tsit5 x 40.  These shapes keep their per-wave forward plan (the one-launch adaptive solvers hang off it); from 4096 columns on their
gradient runs on the cooperative reverse sweep through an auxiliary cooperative plan + image (cnf_handle::plan_cg):
86.4 / 105.9 ms on the slab-accumulator kernel (CNF_COOP_GRAD_MID=0) -> 77.3 / 79.8 ms."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
for nv in tuple(int(v) for v in os.environ.get("NVS", "12,15").split(",")):
    icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    B = 65536
    X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
    m = pkg.TrainMode(True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): pkg.inference(icnf, m, X, P, st, eps=E)
    torch.cuda.synchronize(); t0.record()
    for _ in range(3): pkg.inference(icnf, m, X, P, st, eps=E)
    t1.record(); torch.cuda.synchronize()
    fwd = t0.elapsed_time(t1) / 3
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
    torch.cuda.synchronize(); t0.record()
    for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
    t1.record(); torch.cuda.synchronize()
    print(nv, icnf.nn.widths, "fwd path", icnf.kernel_path(m), round(fwd, 1), "ms; grad path", icnf.grad_path(m), round(t0.elapsed_time(t1) / 2, 1), "ms")

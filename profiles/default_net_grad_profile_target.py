"""rocprofv3 target: loss + gradient of the reference's default architecture at nvariables = NV (default 20: cooperative gradient)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
nv = int(os.environ.get("NV", "20"))
icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
B = 32768
X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
for _ in range(4): pkg.loss_and_gradient(icnf, pkg.TrainMode(True), X, P, st, eps=E)
torch.cuda.synchronize()

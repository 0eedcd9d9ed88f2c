import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
icnf = pkg.ICNF(nvariables=8, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
B = 65536
X = torch.randn(B, 8, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
m = pkg.TrainMode(True)
for _ in range(2): pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)
torch.cuda.synchronize()

"""one loss + gradient and one inference of the default architecture on the layer-wise path (nvariables >= 48), for a kernel trace:
python profiles/ubench/layerwise_probe.py [nv] [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, ic.D, device=dev).t()
m = pkg.TrainMode(True)
for name, fn in (("inference", lambda: pkg.inference(ic, m, X, P, st, eps=E)), ("loss+gradient", lambda: pkg.loss_and_gradient(ic, m, X, P, st, eps=E))):
    fn(); torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize()
    print(nv, B, name, round((time.time() - t) * 1e3, 2), "ms", ic.kernel_family(m, B=B), flush=True)

# nvariables = 15 (8 hidden tiles: the plan's layout IS the configuration) at full size: the cooperative sweep, the dealt sweep forced
# onto the same plan, and the slab-accumulator kernel - which two agree?
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
nv = int(os.environ.get("NV", "15")); B = int(os.environ.get("DG_B", "32768")); nsteps = int(os.environ.get("NSTEPS", "40"))
torch.manual_seed(0)
icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=nsteps))
ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev); E = torch.randn(B, icnf.D, device=dev).t()
m = pkg.TrainMode(True)
res = {}
for tag, env in (("coop", dict(CNF_COOPD_GRAD="0", CNF_COOP_GRAD_MID="1")), ("dealt", dict(CNF_COOPD_GRAD="2", CNF_COOP_GRAD_MID="1")), ("slab", dict(CNF_COOPD_GRAD="0", CNF_COOP_GRAD_MID="0"))):
    os.environ.update(env)
    ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=nsteps))
    l, g = pkg.loss_and_gradient(ic, m, X, P, st, eps=E)[:2]
    res[tag] = (float(l), g.double().cpu())
for a in res:
    for b in res:
        if a < b: print(a, b, float((res[a][1] - res[b][1]).norm() / res[a][1].norm()), float((res[a][1] - res[b][1]).abs().max()))
print({k: v[0] for k, v in res.items()})
W = icnf.nn.widths
off = 0
for l in range(len(W) - 1):
    for nm, n in (("W", W[l] * W[l + 1]), ("b", W[l + 1])):
        a_, b_ = res["coop"][1][off:off + n], res["slab"][1][off:off + n]
        print(f"layer {l + 1} {nm}: rel {float((a_ - b_).norm() / b_.norm()):.3e} max {float((a_ - b_).abs().max()):.3e}")
        off += n

# Cooperative reverse sweep (cnf_coop_grad.hip) with two workgroups per CU against one per CU (CNF_CG_ONE_PER_CU=1) and against the
# layer-wise path (CNF_COOP_GRAD=0), shapes of 8 hidden tiles: does co-residency change the result?
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
B = int(os.environ.get("DG_B", "16384"))
only = os.environ.get("ONLY")
shapes = [("3x128 tanh D=8", 8, 0, [128, 128, 128], "tanh"), ("3x128 tanh D=24", 20, 4, [128, 128, 128], "tanh"), ("2x128 softplus D=31", 15, 16, [128, 128], "softplus"),
          ("3x192 tanh D=16", 16, 0, [192, 192, 192], "tanh"), ("3x256 tanh D=32", 32, 0, [256, 256, 256], "tanh"),
          ("3x320 tanh D=40", 40, 0, [320, 320, 320], "tanh"), ("2x384 softplus D=64", 32, 32, [384, 384], "softplus"), ("2x176 tanh D=20", 20, 0, [176, 176], "tanh")]
for name, nv, na, hid, act in shapes:
    if only and only not in name: continue
    D = nv + na
    w = [D + 1] + hid + [D]
    layers = [pkg.Dense(w[i], w[i + 1], act if i + 2 < len(w) else "identity") for i in range(len(w) - 1)]
    res = {}
    torch.manual_seed(1)
    X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, D, device=dev).t()
    P = None
    for tag, env in (("two", dict(CNF_CG_ONE_PER_CU="0", CNF_COOP_GRAD="1")), ("one", dict(CNF_CG_ONE_PER_CU="1", CNF_COOP_GRAD="1")), ("layered", dict(CNF_CG_ONE_PER_CU="0", CNF_COOP_GRAD="0"))):
        os.environ.update(env)
        ic = pkg.ICNF(nvariables=nv, naugments=na, nn=pkg.Chain(*layers), device=dev, steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.0,
                      sol_kwargs=dict(alg=pkg.RK4(), adaptive=False, nsteps=2))
        if P is None:
            ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
        m = pkg.TrainMode(True)
        l, g = pkg.loss_and_gradient(ic, m, X, P, st, eps=E)[:2]
        res[tag] = (g.double().cpu(), ic.grad_path(m, B=B, alg=0))
    r = res["layered"][0]
    off = 0
    for l in range(len(w) - 1):
        for nm, n in (("W", w[l] * w[l + 1]), ("b", w[l + 1])):
            a_, b_ = res["two"][0][off:off + n], r[off:off + n]
            print(f"   layer {l + 1} {nm}: rel {float((a_ - b_).norm() / b_.norm()):.2e}", end="")
            off += n
    print()
    print(name, "paths", {k: v[1] for k, v in res.items()}, "two-vs-layered %.2e  one-vs-layered %.2e" % (float((res["two"][0] - r).norm() / r.norm()), float((res["one"][0] - r).norm() / r.norm())), flush=True)

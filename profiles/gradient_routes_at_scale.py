# Every fused gradient route against the layer-wise path (CNF_GRAD_LAYERED=1) at a batch that gives the kernels more workgroups than
# compute units - the regime in which the store-data hazard of DESIGN 8.5 showed and which the small parity cases do not reach.
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
B = int(os.environ.get("DG_B", "16384"))
def default_net(nv): D = 2 * nv + 1; return (f"ICNF(nvariables={nv})", nv, nv + 1, [4 * (D + 1)] * 2, "softplus", 1)
shapes = [("3x64 tanh D=8 (cfg2)", 8, 0, [64, 64, 64], "tanh", 1), ("3x64 tanh D=8 K=4 (cfg3)", 8, 0, [64, 64, 64], "tanh", 4),
          default_net(2), default_net(4), default_net(7), default_net(8), default_net(10), default_net(12), default_net(14), default_net(16), default_net(22),
          default_net(24), default_net(29), default_net(30), default_net(32), default_net(38), default_net(43), default_net(47),
          ("3x128 tanh D=8", 8, 0, [128, 128, 128], "tanh", 1), ("3x256 tanh D=32 (cfg4)", 32, 0, [256, 256, 256], "tanh", 1), ("2x232 tanh D=40", 40, 0, [232, 232], "tanh", 1)]
only = os.environ.get("ONLY")
worst = 0.0
for name, nv, na, hid, act, K in shapes:
    if only and only not in name: continue
    D = nv + na
    w = [D + 1] + hid + [D]
    layers = [pkg.Dense(w[i], w[i + 1], act if i + 2 < len(w) else "identity") for i in range(len(w) - 1)]
    torch.manual_seed(1)
    X = torch.randn(B, nv, device=dev).t(); E = torch.randn(B, K * D, device=dev).t()
    res, P = {}, None
    for tag, env in (("default", "0"), ("layered", "1")):
        os.environ["CNF_GRAD_LAYERED"] = env; os.environ["CNF_COOP_GRAD"] = "0" if env == "1" else "1"; pkg.reload_tuning()
        ic = pkg.ICNF(nvariables=nv, naugments=na, nn=pkg.Chain(*layers), device=dev, steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.01 if na else 0.0,
                      nprobes=K, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=2))
        if P is None:
            ps, st = pkg.setup(torch.Generator().manual_seed(0), ic); P = ps.to(dev)
        m = pkg.TrainMode(True)
        l, g = pkg.loss_and_gradient(ic, m, X, P, st, eps=E)[:2]
        res[tag] = (g.double().cpu(), ic.grad_path(m, B=B, alg=1), float(l))
    a, b = res["default"][0], res["layered"][0]
    rel = float((a - b).norm() / b.norm()); n1 = w[0] * w[1]
    rel1 = float((a[:n1] - b[:n1]).norm() / b[:n1].norm())
    worst = max(worst, rel, rel1)
    print(f"{name:32s} path {res['default'][1]} vs {res['layered'][1]}: whole gradient {rel:.2e}, first layer {rel1:.2e}, loss {res['default'][2]:.6f} / {res['layered'][2]:.6f}", flush=True)
print("worst", worst)

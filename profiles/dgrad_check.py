# Dealt reverse sweep (cnf_coop_dgrad.hip) against the sweep of cnf_coop_grad.hip on the same plan, same checkpoints:
# gradient agreement and loss + gradient time, default architecture at several nvariables.
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
B = int(os.environ.get("DG_B", "32768"))
out = {}
for nv in [int(v) for v in os.environ.get("DG_NV", "16,20,24,28").split(",")]:
    icnf = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    X = torch.randn(B, nv, device=dev).t(); P = ps.to(dev)
    E = torch.randn(B, icnf.D, device=dev).t()
    m = pkg.TrainMode(True)
    res = {}
    for tag, env in (("old", "0"), ("dealt", os.environ.get("DG_FORCE", "1"))):
        os.environ["CNF_COOPD_GRAD"] = env; pkg.reload_tuning()
        for _ in range(2): l, gr = pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)[:2]
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(3): l, gr = pkg.loss_and_gradient(icnf, m, X, P, st, eps=E)[:2]
        t1.record(); torch.cuda.synchronize()
        res[tag] = (float(l), gr.double().cpu(), t0.elapsed_time(t1) / 3)
    go, gd = res["old"][1], res["dealt"][1]
    out[f"nv{nv}"] = dict(widths=icnf.nn.widths, loss_old=res["old"][0], loss_dealt=res["dealt"][0], ms_old=res["old"][2], ms_dealt=res["dealt"][2],
                          grad_rel=float((go - gd).norm() / go.norm()), grad_maxabs=float((go - gd).abs().max()), gnorm=float(go.norm()))
    print(json.dumps({f"nv{nv}": out[f"nv{nv}"]}), flush=True)

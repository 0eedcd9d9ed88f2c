"""Loss + gradient under an adaptive solver (cnf_loss_grad_adaptive) with the checkpoints of the frozen-grid sweep written by the
adaptive solve itself (default) against the sweep's own step-by-step forward pass (CNF_ADAPTIVE_CKPT=0): the reference's benchmark flow
at 2^10 samples and cfg2's flow at 1024 ... 65 536 samples (the last one beyond the one-launch kernel: the library's host loop), tolerance 1e-4.  Best of three runs of 20 calls."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
def timed(fn):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / 20)
    return round(best, 4)
for env in ("1", "0"):
    os.environ["CNF_ADAPTIVE_CKPT"] = env; pkg.reload_tuning()
    res = {}
    r = torch.distributions.Beta(2.0, 4.0).sample((1, 1024)).float().to(dev)
    icnf = pkg.ICNF(nvariables=1, device=dev)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf); ps = ps.to(dev)
    res["pkgbenchmark_1024"] = dict(ms=timed(lambda: pkg.loss_and_gradient(icnf, pkg.TrainMode(True), r, ps, st)), steps=icnf.last_solve_stats["naccept"])
    for B in (1024, 8192, 32768, 65536):
        spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
        p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
        X = torch.tensor(xs.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev); E = torch.tensor(eps.T.copy(), device=dev).t()
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        ic = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(), steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0,
                      device=dev, sol_kwargs=dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))
        res[f"d8_3x64_{B}"] = dict(ms=timed(lambda: pkg.loss_and_gradient(ic, pkg.TrainMode(False), X, P, {}, eps=E)), steps=ic.last_solve_stats["naccept"])
    # default architecture at nvariables = 10 (slab-accumulator gradient: only the separate loss solve over the grid goes)
    os.environ["CNF_COOP_GRAD_MID"] = "0"; pkg.reload_tuning()     # (the slab kernel itself: at this size the default is the auxiliary cooperative sweep)
    r10 = torch.randn(10, 8192, device=dev)
    ic10 = pkg.ICNF(nvariables=10, device=dev, sol_kwargs=dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))
    ps10, st10 = pkg.setup(torch.Generator().manual_seed(0), ic10); ps10 = ps10.to(dev)
    res["default_nv10_8192"] = dict(ms=timed(lambda: pkg.loss_and_gradient(ic10, pkg.TrainMode(True), r10, ps10, st10)), steps=ic10.last_solve_stats["naccept"],
                                    grad_path=ic10.grad_path(pkg.TrainMode(True), B=8192, alg=1, on_grid=True))
    os.environ.pop("CNF_COOP_GRAD_MID", None); pkg.reload_tuning()
    out["solve_writes_checkpoints" if env == "1" else "own_forward_pass"] = res
os.environ.pop("CNF_ADAPTIVE_CKPT", None)
print(json.dumps(out))

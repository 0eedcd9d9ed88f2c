# The cooperative gradient's second form (stage store + second-order sweep + products over tiles, CNF_COOP_GRAD3=1)
# against the recomputing sweeps (CNF_COOP_GRAD3=0) on the same handle configuration: per-layer gradient agreement, the fp64 oracle
# at a small batch, and loss + gradient time at full size (CNF_COOP_GRAD3=2: the second form with the one-workgroup-per-CU sweep for every shape).   G3_CASE = cfg4 | nv16 | nv20 | nv24 ..., G3_B, G3_ORACLE_B
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")


def spec_for(case):
    if case == "cfg4":
        return o64.make_spec(nvars=32, hidden=[256, 256, 256]), 0, (0.0, 0.0, 0.0)
    if case == "cfg4r":
        return o64.make_spec(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 0, (0.01, 0.01, 0.0)
    if case.startswith("nv"):
        nv = int(case[2:])
        D = 2 * nv + 1
        H = 4 * (D + 1)
        return o64.make_spec(nvars=nv, naug=nv + 1, hidden=[H, H], act=2, reg_z=True, reg_j=True, reg_aug=True), 1, (0.01, 0.01, 0.01)
    raise SystemExit(f"unknown case {case}")


def make_icnf(spec, alg, nsteps, lam):
    acts = {0: "identity", 1: "tanh", 2: "softplus"}
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], acts[spec.acts[i]]) for i in range(len(spec.acts))]
    return pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nconditions=spec.ncond, autonomous=spec.autonomous, nn=pkg.Chain(*layers),
                    compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=0), steer_rate=0.0, lambda1=lam[0] if spec.reg_z else 0.0,
                    lambda2=lam[1] if spec.reg_j else 0.0, lambda3=lam[2] if spec.reg_aug else 0.0, nprobes=1, device="cuda:0",
                    sol_kwargs=dict(alg=pkg.Tsit5() if alg == 1 else pkg.RK4(), adaptive=False, nsteps=nsteps))


def t(a):
    return torch.tensor(np.asarray(a, dtype=np.float32), device=dev)


def layer_slices(spec):
    out, off = [], 0
    for l in range(len(spec.acts)):
        wi, wo = spec.widths[l], spec.widths[l + 1]
        out.append((f"W{l + 1}", off, off + wi * wo)); off += wi * wo
        out.append((f"b{l + 1}", off, off + wo)); off += wo
    return out


def run(case):
    spec, alg, lam = spec_for(case)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    res = {"case": case, "widths": list(spec.widths)}
    # ---- small batch against the fp64 oracle ----
    Bo, nso = int(os.environ.get("G3_ORACLE_B", "100")), 2
    p, xs, eps, ys = o64.synth_inputs(spec, Bo, 321, bias_scale=0.2)
    Lr, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nso, alg, eps, ys, lam, wrt_x=True)
    got = {}
    for tag, sw in (("new", "1"), ("old", "0")):
        os.environ["CNF_COOP_GRAD3"] = sw
        pkg.reload_tuning()
        icnf = make_icnf(spec, alg, nso, lam)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, t(xs), t(p), {}, eps=t(eps), wrt_x=True)
        got[tag] = (float(val), g.double().cpu().numpy(), gx.double().cpu().numpy(), icnf.grad_path(mode))
    scale = np.abs(gref).max()
    res["oracle"] = {tag: dict(loss_err=abs(v[0] - Lr), grad_err_rel=float(np.abs(v[1] - gref).max() / scale),
                               gx_err_rel=float(np.abs(v[2] - gxref).max() / np.abs(gxref).max()), path=v[3]) for tag, v in got.items()}
    res["per_layer_new_vs_oracle"] = {nm: float(np.abs(got["new"][1][a:b] - gref[a:b]).max() / (np.abs(gref[a:b]).max() + 1e-30)) for nm, a, b in layer_slices(spec)}
    print(json.dumps(res), flush=True)
    # ---- full size: agreement and time ----
    B = int(os.environ.get("G3_B", "32768"))
    if B <= 0:
        return
    ns = 40
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn(spec.nvars, B, generator=g).to(dev)
    E = torch.randn(spec.nvars + spec.naug, B, generator=g).to(dev)
    P = t(p)
    full = {}
    for tag, sw in (("new", "1"), ("one_per_cu", "2"), ("old", "0")):
        os.environ["CNF_COOP_GRAD3"] = sw
        pkg.reload_tuning()
        icnf = make_icnf(spec, alg, ns, lam)
        for _ in range(2):
            val, gr = pkg.loss_and_gradient(icnf, mode, X, P, {}, eps=E)[:2]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            val, gr = pkg.loss_and_gradient(icnf, mode, X, P, {}, eps=E)[:2]
        e1.record()
        torch.cuda.synchronize()
        full[tag] = (float(val), gr.double().cpu().numpy(), e0.elapsed_time(e1) / 3)
    gn, go = full["new"][1], full["old"][1]
    g1 = full["one_per_cu"][1]
    out = dict(case=case, B=B, ms_new=full["new"][2], ms_one_per_cu=full["one_per_cu"][2], ms_old=full["old"][2],
               grad_maxabs_rel_new_vs_one_per_cu=float(np.abs(gn - g1).max() / np.abs(g1).max()), loss_new=full["new"][0], loss_old=full["old"][0],
               grad_rel=float(np.linalg.norm(gn - go) / np.linalg.norm(go)), grad_maxabs_rel=float(np.abs(gn - go).max() / np.abs(go).max()),
               per_layer={nm: float(np.abs(gn[a:b] - go[a:b]).max() / (np.abs(go[a:b]).max() + 1e-30)) for nm, a, b in layer_slices(spec)})
    print(json.dumps(out), flush=True)


for case in os.environ.get("G3_CASE", "cfg4,cfg4r,nv16,nv20,nv24,nv28").split(","):
    run(case)

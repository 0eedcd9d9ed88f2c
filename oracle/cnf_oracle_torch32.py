"""TEST/BENCH INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 2) - never on the product path.

A float32 torch-CPU restatement of the Hutchinson-VJP solve with the reference's unfused structure (per dynamics
call: whole-batch forward through the Dense chain, pullback, RK stage loop outside; src/core/icnf.jl:517-536,
src/core/utils.jl:150-159), every product a library GEMM (MKL / oneDNN through torch.mm).  SURVEY.md section 8(d) asks
for it as a CPU-favourable cross-check of the C restatement's timing: bench.py reports its rate beside
`cpu_baseline.value`.  Checked against the fp64 oracle in tests/test_oracle_kat.py."""
import math

import numpy as np
import torch

from cnf_oracle64 import MODE_HUTCH_VJP, Spec, tableau


def _act(a, kind):
    if kind == 1:
        h = torch.tanh(a)
        return h, 1.0 - h * h
    if kind == 2:
        return torch.nn.functional.softplus(a), torch.sigmoid(a)
    return a, torch.ones_like(a)


def inference_fixed(spec: Spec, p, xs, t0, t1, nsteps, alg, eps, ys=None):
    """Returns logp (B,) float32.  Column-per-sample data is held as (rows, B) tensors."""
    assert spec.mode == MODE_HUTCH_VJP
    D, K = spec.D, spec.nprobes
    w_off, b_off, _ = spec.param_offsets()
    pt = torch.as_tensor(np.asarray(p, dtype=np.float32))
    Ws, bs = [], []
    for l in range(len(spec.acts)):
        fin, fout = spec.widths[l], spec.widths[l + 1]
        Ws.append(pt[w_off[l]:w_off[l] + fin * fout].reshape(fin, fout).t().contiguous())
        bs.append(pt[b_off[l]:b_off[l] + fout][:, None])
    x = torch.as_tensor(np.asarray(xs, dtype=np.float32))
    B = x.shape[1]
    e = torch.as_tensor(np.asarray(eps, dtype=np.float32))
    y = None if ys is None else torch.as_tensor(np.asarray(ys, dtype=np.float32))

    def f(u, t):
        z = u[:D]
        rows = [z] + ([] if spec.autonomous else [torch.full((1, B), t, dtype=torch.float32)]) + ([] if y is None else [y])
        h = torch.cat(rows, 0)
        ds = []
        for W, b, kind in zip(Ws, bs, spec.acts):
            h, d = _act(torch.mm(W, h) + b, kind)
            ds.append(d)
        ld = torch.zeros(B)
        nd = torch.zeros(B)
        for k in range(K):
            ek = e[k * D:(k + 1) * D]
            dl = ek * ds[-1]
            for l in range(len(Ws) - 1, 0, -1):
                dl = torch.mm(Ws[l].t(), dl) * ds[l - 1]
            g = torch.mm(Ws[0][:, :D].t(), dl)
            ld -= (g * ek).sum(0) / K
            if spec.reg_j:
                nd += g.norm(dim=0) / K
        ed = h.norm(dim=0) if spec.reg_z else torch.zeros(B)
        return torch.cat([h, ld[None], ed[None], nd[None]], 0)

    c, a, b = tableau(alg)
    dt = (t1 - t0) / nsteps
    u = torch.cat([x, torch.zeros(spec.naug + 3, B)], 0)
    with torch.no_grad():
        for n in range(nsteps):
            tn = t0 + n * dt
            ks = []
            for i in range(len(c)):
                ui = u
                for j, aij in enumerate(a[i]):
                    if aij != 0.0:
                        ui = ui + (dt * aij) * ks[j]
                ks.append(f(ui, tn + c[i] * dt))
            for bi, ki in zip(b, ks):
                u = u + (dt * bi) * ki
    z = u[:D]
    return (-0.5 * D * math.log(2.0 * math.pi) - 0.5 * (z * z).sum(0) - u[D]).numpy()

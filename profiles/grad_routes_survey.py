# Which gradient implementation each named workload takes: cnf_grad_path_for / cnf_grad_form_for for the BASELINE
# configurations and the reference's default architecture ICNF(nvariables = nv), at the bench batch sizes (one 32-column call binds
# the parameters first: the auxiliary cooperative plans of the mid widths are packed with them).
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")


class A:
    batch = 0; path = 0; arith = "f32"; mode = "grad"


out = {}
for name in ("cfg1", "cfg2", "cfg2p", "cfg3", "cfg4", "cfg4r", "cfg5", "nv20"):
    w = bench.make_workload(pkg, o64, name, A(), 0, dev, torch, grad=True, batch=min(bench.CONFIGS[name][2], 4096 * 8))
    ic, m, B, alg = w["icnf"], w["mode"], bench.CONFIGS[name][2], w["alg"]
    pkg.loss_and_gradient(ic, m, *[a[:, :32] if (hasattr(a, "dim") and a.dim() == 2) else a for a in w["args"]], eps=w["E"][:, :32])   # binds the parameters (auxiliary plans are packed with them)
    out[name] = dict(B=B, path=ic.grad_path(m, B=B, alg=alg), form=ic.grad_form(m, B, alg, 40), family=ic.kernel_family(m, B=B))
for nv in (1, 3, 6, 7, 8, 10, 11, 12, 15, 16, 20, 21, 22, 29, 30, 40, 47, 48):
    ic = pkg.ICNF(nvariables=nv, device=dev, steer_rate=0.0, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
    m = pkg.TrainMode(True)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), ic)
    pkg.loss_and_gradient(ic, m, torch.randn(nv, 32, device=dev), ps.to(dev), st)      # binds the parameters
    row = {}
    for B in (1024, 32768):
        row[f"B{B}"] = dict(path=ic.grad_path(m, B=B, alg=1), form=ic.grad_form(m, B, 1, 40), family=ic.kernel_family(m, B=B))
    out[f"default_nv{nv}"] = dict(widths=ic.nn.widths, **row)
print(json.dumps(out, indent=1))

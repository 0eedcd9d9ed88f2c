"""rocprofv3 target: VCABM solves (tolerance 1e-6, so the order climbs) at cfg2 scale (S x B = 11 x 65 536) and at cfg4's
state size (35 x 262 144) - the three elementwise passes of csrc/cnf_vcabm.hip next to the dynamics kernels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
for nvars, hidden, B in ((8, [64, 64, 64], 65536), (32, [256, 256, 256], 262144)):
    spec = o64.make_spec(nvars=nvars, hidden=hidden)
    p, xs, eps, _ = o64.synth_inputs(spec, 1024, 3)
    rep = B // 1024
    X = torch.tensor(np.tile(xs, (1, rep)).T.copy(), device=dev).t()
    E = torch.tensor(np.tile(eps, (1, rep)).T.copy(), device=dev).t()
    P = torch.tensor(p, device=dev)
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], "tanh" if i < 3 else "identity") for i in range(4)]
    icnf = pkg.ICNF(nvariables=nvars, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0,
                    device=dev, sol_kwargs=dict(alg=pkg.VCABM(), reltol=1e-6, abstol=1e-6))
    for _ in range(3):
        lp = pkg.inference(icnf, pkg.TrainMode(False), X, P, {}, eps=E)[0]
    torch.cuda.synchronize()
    st = icnf.last_solve_stats
    print(nvars, B, st["naccept"], st["nreject"], st["orders"], float(lp.mean()))

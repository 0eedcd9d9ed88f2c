"""rocprofv3 target: the training step under the reference's default sol_kwargs at cfg2 scale (one cnf_loss_grad_adaptive call:
adaptive Tsit5 solve -> frozen grid -> checkpointing forward pass -> fused reverse sweep), 10 calls."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
B = 65536
p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
X = torch.tensor(xs.T.copy(), device=dev).t(); E = torch.tensor(eps.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], "tanh" if i < 3 else "identity") for i in range(4)]
icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev)
for _ in range(10):
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), X, P, {}, eps=E)
torch.cuda.synchronize()
print(float(val), icnf.last_solve_stats["naccept"], type(icnf.sol_kwargs["alg"]).__name__)

"""Wall time of a training step at cfg2 when ps changes every step (Adam on the device): parameter
repack (cnf_set_params on the device pointer) + loss + gradient + optimiser update."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(4)]
icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=2),
                steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev,
                sol_kwargs=dict(alg=pkg.RK4(), adaptive=False, nsteps=40))
B = 65536
p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
X = torch.tensor(xs.T.copy(), device=dev).t(); E = torch.tensor(eps.T.copy(), device=dev).t()
P = torch.tensor(p, device=dev)
opt = torch.optim.Adam([P], lr=1e-4)
m = pkg.TrainMode(False)
def step():
    val, g = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
    P.grad = g
    opt.step()
    return val
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): v = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(json.dumps({"train_step_ms": 1e3 * (t1 - t0) / n, "repack_on_device": icnf.repack_on_device(m), "loss": float(v)}))

import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
for name, kw in (("d8_6x64", dict(nvars=8, hidden=[64] * 6)), ("d3_5x24_softplus", dict(nvars=3, hidden=[24] * 5, act=2)), ("d20_2x32", dict(nvars=20, hidden=[32, 32]))):
  for B in (8, 64, 256):
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
    X = torch.tensor(xs.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
    E = torch.tensor(eps.T.copy(), device=dev).t()
    r = {}
    for path, pname in ((3, "layered"), (1, "simt")):
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=pkg.Chain(*layers), compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=path), steer_rate=0.0,
                        lambda1=0.0, lambda2=0.0, lambda3=0.0, device=dev, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        m = pkg.TrainMode(False)
        fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E)
        fn(); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); fn(); t1.record(); torch.cuda.synchronize()
        r[pname] = round(t0.elapsed_time(t1) / 2, 2)
    out[f"{name}_B{B}"] = r
print(json.dumps(out))

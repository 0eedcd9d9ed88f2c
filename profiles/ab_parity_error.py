import sys, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); o64, oc = e.load_oracle()
import os
for name, kw, B, alg in [("cfg2p", dict(nvars=8, hidden=[64,64,64]), 65536, 1), ("cfg2", dict(nvars=8, hidden=[64,64,64]), 65536, 0),
                         ("cfg3", dict(nvars=8, hidden=[64,64,64], nprobes=4, reg_z=True, reg_j=True), 65536, 1)]:
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 20240700 + len(name))
    for scale in (1.0, 3.0):
        pp = (p*scale).astype(np.float32)
        layers = [pkg.Dense(spec.widths[i], spec.widths[i+1], {0:"identity",1:"tanh"}[spec.acts[i]]) for i in range(len(spec.acts))]
        icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.01 if spec.reg_z else 0., lambda2=0.01 if spec.reg_j else 0., lambda3=0., nprobes=spec.nprobes,
                        device="cuda:0", sol_kwargs=dict(alg=pkg.Tsit5() if alg==1 else pkg.RK4(), adaptive=False, nsteps=40))
        mode = pkg.TrainMode(bool(spec.reg_z))
        dev = lambda a: torch.tensor(a, device="cuda:0")
        logp = pkg.inference(icnf, mode, dev(xs), dev(pp), {}, eps=dev(eps))[0].cpu().numpy()
        idx = np.arange(0, B, 16)
        ref = o64_ref = None
        nt = min(os.cpu_count(), oc.max_threads())
        ref = oc.inference_fixed(spec, pp, np.ascontiguousarray(xs[:, idx]), 0.0, 1.0, 40, alg, np.ascontiguousarray(eps[:, idx]), None, nthreads=nt)[0]
        # fp64 reference on a small subset
        sub = idx[:64]
        r64 = o64.inference_fixed(spec, pp, xs[:, sub], 0.0, 1.0, 40, alg, eps[:, sub], None)[0]
        print(name, "wscale", scale, "max|hip-c32| %.2e" % np.max(np.abs(logp[idx]-ref)), "max|hip-f64| %.2e mean %.2e" % (np.max(np.abs(logp[sub]-r64)), np.mean(np.abs(logp[sub]-r64))),
              "max|c32-f64| %.2e" % np.max(np.abs(ref[:64]-r64)))

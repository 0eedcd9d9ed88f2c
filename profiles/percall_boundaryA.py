"""Boundary A (one launch per dynamics call, the integrator outside the library — how an
OrdinaryDiffEq-driven Julia host would use cnf_aug_f): per-call time and the per-call HBM model
(read u 4S + eps 4KD, write du 4S bytes per sample)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
out = {}
for name, kw, B in (("cfg2", dict(nvars=8, hidden=[64, 64, 64]), 65536),
                    ("cfg4", dict(nvars=32, hidden=[256, 256, 256]), 32768)):
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 1)
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], "tanh" if spec.acts[i] == 1 else "identity")
              for i in range(len(spec.acts))]
    icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.0,
                    lambda2=0.0, lambda3=0.0, device=dev,
                    sol_kwargs=dict(alg=pkg.RK4(), adaptive=False, nsteps=40))
    mode = pkg.TrainMode(False)
    h = icnf._handle(mode)
    icnf._bind_params(h, torch.tensor(p, device=dev))
    S = spec.S
    u = torch.randn(B, S, device=dev)
    du = torch.empty_like(u)
    e = torch.tensor(eps.T.copy(), device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    call = lambda: pkg._lib.check(h.lib.cnf_aug_f(h.ptr, C.c_void_p(du.data_ptr()), C.c_void_p(u.data_ptr()), 0.3,
                                                  C.c_void_p(e.data_ptr()), None, B, st))
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    n = 160
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        call()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / n
    byts = (2 * 4 * S + 4 * spec.D) * B
    flop = {"cfg2": 36992, "cfg4": 590336}[name] * B
    out[name] = dict(us_per_call=us, GBps_per_call_abi=byts / us / 1e3, frac_of_8TBps=byts / us / 1e3 / 8000,
                     TFLOPs=flop / us / 1e6, frac_of_f32_mfma_peak=flop / us / 1e6 / 157.3,
                     launches_back_to_back=n, B=B)
print(json.dumps(out, indent=1))

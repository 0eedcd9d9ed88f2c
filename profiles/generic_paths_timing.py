"""Generic families on shapes no fused instance covers: layer-wise GEMM path (CNF_PATH_LAYERED) against the
thread-per-sample kernels (CNF_PATH_SIMT), whole solves (Tsit5 x 40 unless noted)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); o64, _ = entry.load_oracle()
dev = torch.device("cuda:0")
cases = {
    "d8_6x64_vjp_B65536": (dict(nvars=8, hidden=[64] * 6), 65536),
    "d64_3x512_vjp_B16384": (dict(nvars=64, hidden=[512] * 3), 16384),
    "d8_3x64_jvp_k2_B65536": (dict(nvars=8, hidden=[64] * 3, mode=1, nprobes=2), 65536),
    "d12_3x300_exact_B8192": (dict(nvars=12, hidden=[300] * 3, mode=2), 8192),
    "d8_6x64_vjp_B1024": (dict(nvars=8, hidden=[64] * 6), 1024),
}
out = {}
for name, (kw, B) in cases.items():
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 3)
    X = torch.tensor(xs.T.copy(), device=dev).t(); P = torch.tensor(p, device=dev)
    E = torch.tensor(eps.T.copy(), device=dev).t() if eps is not None else None
    r = {}
    for path, pname in ((3, "layered"), (1, "simt")):
        layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ["identity", "tanh", "softplus"][spec.acts[i]]) for i in range(len(spec.acts))]
        cm = (pkg.HIPJacVecMatrixMode if spec.mode == 1 else pkg.HIPVecJacMatrixMode)(kernel_path=path)
        icnf = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=pkg.Chain(*layers), compute_mode=cm, steer_rate=0.0,
                        lambda1=0.0, lambda2=0.0, lambda3=0.0, nprobes=spec.nprobes, device=dev,
                        sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=40))
        m = pkg.TestMode() if spec.mode == 2 else pkg.TrainMode(False)
        fn = lambda: pkg.inference(icnf, m, X, P, {}, eps=E)
        fn(); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); fn(); t1.record(); torch.cuda.synchronize()
        r[pname + "_ms"] = t0.elapsed_time(t1) / 2
    r["speedup"] = r["simt_ms"] / r["layered_ms"]
    out[name] = r
print(json.dumps(out, indent=1))

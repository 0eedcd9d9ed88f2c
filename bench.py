#!/usr/bin/env python3
"""bench.py — throughput of the batched log-density hot path on MI355X.

A "step" is one pass of the hot path over one batch: `loss(icnf, mode, xs, ps, st)` =
cnf_inference_fixed (fused fixed-step solve + log-density epilogue) + cnf_loss_mean (N = 1; cnf_loss_sums + the
RCCL all-reduce of the loss scalars when N > 1), with inputs already resident in HBM.
Metric (BASELINE.json): log-density evaluations counted as samples·steps per second.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg2p|cfg3|cfg4|cfg4r|cfg5|nv20] [--mode infer|grad]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`;
the batch columns are sharded by rank (weak scaling: 65 536 columns per GPU), no data-path
collective, one all-reduce of five scalars per step.

Timing protocol (the 63 ms window of 20 launches ran at boost clock and the rocprof
summaries could not reproduce it):
  1. W warm-up steps (untimed);
  2. an untimed PRE-ROLL of the same step until >= --preroll-seconds (default 2 s) of back-to-back
     launches have run, so DVFS has settled to the clock the chip sustains under this load; every
     launch is bracketed by HIP events and every 16th is followed by a shader-clock probe
     (profiles/ubench/clockprobe.hip), so the drift is visible: `sustained.first_quarter_ms`,
     `last_quarter_ms`, `median_ms`, `clock_mhz`;
  3. EXACTLY K timed steps between barrier + synchronize on both sides -> `value`, `ms_per_step`;
     the solve kernel's mean / median launch duration over those K steps -> `roofline`.
`rocprofv3 --kernel-trace --stats` of the same command (profiles/collect.sh) therefore averages over
the same steady state, and `roofline.frac` = flop_per_launch / (CSV average) reproduces.

The JSON line also carries
  roofline     — the fused solve kernel against the f32 MFMA peak (the path is compute-bound:
                 ≥ 97 flop/B); its HBM figure is reported beside it.
  secondary    — the north_star's target configuration (cfg2p: the same flow under Tsit5 x 40)
                 measured by the same protocol in the same process, with its own roofline.
  secondaries  — (default one-GPU line only) every other BASELINE configuration, the loss + gradient of cfg2, cfg3 and cfg4 and the
                 reference's default architecture at nvariables = 20 (inference and loss + gradient), compactly, same protocol.
  small_batch  — (default one-GPU line only) wall time of one call at the reference's own batch size, host side included: its
                 PkgBenchmark scenario (ICNF(nvariables = 1), VCABM, 2^10 samples; `loss` in TrainMode and TestMode) and cfg1.
  cpu_baseline — CPU fp32 restatements of the same algorithm on this host's cores (rank 0, N = 1):
                 the C port (oracle/cnf_oracle.c, cache-blocked register-tiled products, AVX-512 when
                 the CPU has it) and a whole-batch BLAS leg (torch.mm on every host thread at the full
                 batch — the structure of the reference's CPU path); `value` is the faster one.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# workload table: SURVEY.md §8(d).  flop / bytes are ALGORITHMIC figures per sample·step.
CONFIGS = {
    # name: (make_spec kwargs, alg, B per GPU, flop per sample·step, per-call-ABI bytes per sample·step,
    #        fused bytes per sample per solve, description)
    "cfg1": (dict(nvars=2, hidden=[32, 32]), 1, 1024, 28032, 288, 20,
             "FFJORD nvars=2, MLP 2x32 tanh, Tsit5 40 fixed steps, batch=1024, Hutchinson(1)"),
    "cfg2": (dict(nvars=8, hidden=[64, 64, 64]), 0, 65536, 147968, 480, 68,
             "FFJORD nvars=8, MLP 3x64 tanh, RK4 40 steps, batch=65536, Hutchinson(1)"),
    "cfg2p": (dict(nvars=8, hidden=[64, 64, 64]), 1, 65536, 221952, 720, 68,
              "FFJORD nvars=8, MLP 3x64 tanh, Tsit5 40 steps, batch=65536, Hutchinson(1)"),
    "cfg3": (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), 1, 65536,
             553728, 1296, 164,
             "RNODE nvars=8, MLP 3x64 tanh, Tsit5 40 steps, batch=65536, Hutchinson(4)"),
    "cfg4": (dict(nvars=32, hidden=[256, 256, 256]), 0, 32768, 2361344, 1632, 260,
             "FFJORD nvars=32, MLP 3x256 tanh, RK4 40 steps, batch=32768 per GPU, Hutchinson(1)"),
    # (not a BASELINE configuration: cfg4's shape under the reference's default regularised objective, for the gradient timing)
    "cfg4r": (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 0, 32768, 2361344, 1632, 260,
              "RNODE nvars=32, MLP 3x256 tanh, RK4 40 steps, batch=32768 per GPU, Hutchinson(1), |zdot| and |eps^T J| regularisers"),
    "cfg5": (dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=2), 0, 16384, 2393088, 480, 68,
             "CondFFJORD nvars=8+8 cond, MLP 3x128 tanh, exact trace, RK4 40 steps, batch=16384"),
    # (not a BASELINE configuration: what `ICNF(; nvariables = 20)` builds with no further arguments, src/core/icnf.jl:53-103 -
    # naugments = nvariables + 1, two softplus layers of 4 (D + 1) units, lambda_1 = lambda_2 = lambda_3 = 0.01 - on the dealt
    # cooperative kernels.  flop: forward 2 (42 168 + 168^2 + 168 41) + VJP 2 (41 168 + 168^2 + 168 41), 6 stages)
    "nv20": (dict(nvars=20, naug=21, hidden=[168, 168], act=2, reg_z=True, reg_j=True, reg_aug=True), 1, 32768, 1010016, 3096, 248,
             "the reference's DEFAULT architecture at nvariables=20: D=41, MLP 2x168 softplus, TrainMode{true} with the default "
             "lambdas, Tsit5 40 steps, batch=32768, Hutchinson(1)"),
}
NSTEPS = 40


def traffic_is_stale():
    """True when a kernel source changed after profiles/pmc_traffic.json was collected (its `_csrc_sha16` = sha256 over csrc/*.hip, *.h
    at collection time, profiles/make_pmc_traffic.py): the committed traffic figures then describe an older build."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import hashlib, glob
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            want = json.load(f).get("_csrc_sha16")
        h = hashlib.sha256()
        d = os.path.join(ROOT, "continuousnormalizingflows.jl_amd", "csrc")
        for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
            h.update(os.path.basename(fn).encode())
            h.update(open(fn, "rb").read())
        return want is None or h.hexdigest()[:16] != want
    except Exception:
        return True


def measured_traffic(name):
    """HBM bytes per launch of the solve kernel from the PMC passes of profiles/collect.sh
    (FETCH_SIZE and WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md §HBM).  bench.py cannot run rocprofv3 on itself, so the figure is read
    from the committed summary; null when the configuration has not been profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(name)
    except Exception:
        return None, None
    if isinstance(e, dict):
        return e.get("bytes"), e.get("source")
    return e, None


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = f32 vector rate
HBM_PEAK_GBS = 8000.0


def grad_mfma_per_stage(spec):
    """MFMA instructions one wave executes per RK stage for loss + gradient (forward solve kernel + reverse
    sweep), counted from the kernels' loops (csrc/cnf_mfma_kernel.h; csrc/cnf_grad2.hip, the barrier-free
    register-accumulator sweep of round 5): a product with MT output tiles and KS k-steps is MT*KS instructions;
    a cotangent block of MT x NT tiles over the tile's 16 samples is 4*MT*NT (biases are VALU row sums)."""
    H, L, K = max(spec.widths[1:-1]), len(spec.acts) - 1, spec.nprobes
    HT, ZR, CR = -(-H // 16), -(-spec.D // 4), -(-spec.ncond // 4)
    DT = -(-ZR // 4)
    hid, first, last = (L - 1) * 4 * HT * HT, HT * (ZR + CR), DT * 4 * HT
    # forward solve kernel: chain, then the pullback per probe (one probe: c = W_N^T eps hoisted, and
    # without |eps^T J| the last product is a dot with the hoisted q = W_1 eps)
    fwd = first + hid + last
    fwd += hid + (last if spec.reg_j else 0) if K == 1 else K * (HT * ZR + hid + last)
    chain = first + hid + (last if spec.reg_z else 0)          # recompute of h_l (+ zdot for |zdot|)
    top = HT * ZR + hid + last                                 # W_N^T kbar, W_l^T abar_l, W_1^T abar_1
    cot_h = (L - 1) * 4 * HT * HT                              # one term of every hidden cotangent
    small = 4 * HT                                             # one term of Wbar_N / Wbar_1
    # per probe: [c_k = W_N^T eps_k (several probes)], pullback, [g], dbar_1 = W_1 gbar, bottom-up + delta ubar^T, eps cbar^T, delta_1 gbar^T
    per_probe = (HT * ZR if K > 1 else 0) + hid + (last if spec.reg_j else 0) + HT * ZR + hid + cot_h + 2 * small
    rev = chain + K * per_probe + top + cot_h + 2 * small + (small if CR else 0)   # + abar h^T, kbar h_L^T, abar_1 [z; t; 1]^T (+ y^T)
    return fwd + rev


def layered_flop_per_stage(spec):
    """GEMM flop per sample per RK stage of the layer-wise gradient path (csrc/cnf_layered.hip): forward
    chain (recompute at the stage point; the stage derivatives themselves are checkpointed), per probe the
    pullback and its bottom-up reverse with the probe's weight cotangents, then the top-down pass with
    weight cotangents.  Plus the forward sweep (one chain per stage) and the fused solve for the loss."""
    w, K, D = spec.widths, spec.nprobes, spec.D
    N = len(w) - 1
    fwd = sum(2 * w[l + 1] * (w[l] + 1) for l in range(N))
    hid = sum(2 * w[l + 1] * w[l] for l in range(1, N))
    pull = hid + 2 * D * w[1]
    per_probe = pull + 2 * (2 * w[1] * D) + 2 * hid
    top = fwd + pull
    solve = sum(2 * w[l + 1] * w[l] for l in range(N)) + (K * pull)       # the regular fused solve for the loss
    return 2 * fwd + K * per_probe + top + solve      # forward sweep + recompute = 2 chains per stage


def coop_grad_flop_per_stage(spec):
    """Product flop per sample per RK stage of the cooperative gradient (csrc/cnf_coop_grad.hip): the checkpointing forward solve
    (chain + pullback), the reverse sweep's recomputed chain, pullback, bottom-up and top-down passes, and the deferred weight
    cotangents (two terms per matrix, bias columns included)."""
    w, D = spec.widths, spec.D
    N = len(w) - 1
    fwd = sum(2 * w[l + 1] * w[l] for l in range(N))
    hid = sum(2 * w[l + 1] * w[l] for l in range(1, N - 1))            # H x H products of one pass
    pull = hid + 2 * D * w[1]                                          # + W_N^T eps
    solve = fwd + pull + 2 * D * w[1]                                  # forward solve: chain, pullback, g = W_1^T delta_1
    chain = (fwd - 2 * w[N] * w[N - 1]) + pull + (2 * D * w[1] + hid) + (2 * D * w[1] + hid + 2 * D * w[1])
    wgrad = sum(2 * 2 * w[l + 1] * (w[l] + 1) for l in range(N))
    return solve + chain + wgrad


def coop_grad3_flop_per_stage(spec):
    """Product flop per sample per RK stage of the cooperative gradient's second form: the checkpointing forward
    solve (chain + pullback + g), the sweep's second-order chains alone (dbar_1 = W_1 gbar, the H x H products up and down,
    W_N^T kbar, Zbar = W_1^T sbar_1) and the weight cotangents over tiles (two terms per matrix)."""
    w, D = spec.widths, spec.D
    N = len(w) - 1
    fwd = sum(2 * w[l + 1] * w[l] for l in range(N))
    hid = sum(2 * w[l + 1] * w[l] for l in range(1, N - 1))
    solve = fwd + hid + 2 * D * w[1] + 2 * D * w[1]                    # chain, pullback (c hoisted or not: counted), g
    sweep = 2 * hid + 4 * (2 * D * w[1])                               # up, down; dbar_1, W_N^T kbar, Zbar (+ nothing else)
    wgrad = sum(2 * 2 * w[l + 1] * w[l] for l in range(N))
    return solve + sweep + wgrad


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override columns per GPU")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 SIMT, 2 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--preroll-seconds", type=float, default=2.0,
                    help="untimed back-to-back launches before the timed steps so the clock has settled (0 = none)")
    ap.add_argument("--secondary", default="auto",
                    help="second workload measured by the same protocol and reported under 'secondary' "
                         "(auto = cfg2p, the north_star's Tsit5 target, when --config cfg2 --mode infer; none = off)")
    ap.add_argument("--secondaries", default="auto",
                    help="further workloads measured by the same protocol in the same process and reported compactly under "
                         "'secondaries' ({value, ms_per_step, roofline: {kernel_ms, frac}, loss}): a comma list of "
                         "name[:grad] (e.g. cfg3,cfg4,cfg5,cfg2:grad,cfg4:grad,nv20,nv20:grad), 'none', or 'auto' = exactly that list on the "
                         "default one-GPU cfg2 line, so that every BASELINE configuration and the gradient are on the driver-run line")
    ap.add_argument("--mode", default="infer", choices=["infer", "grad"],
                    help="infer: loss (default, the BASELINE metric); grad: loss_and_gradient — forward with "
                         "checkpoints + reverse sweep + all-reduce of nparams floats")
    ap.add_argument("--arith", default="f32", choices=["f32", "bf16x6"],
                    help="hidden-product arithmetic: exact f32 MFMA (default) or split-bf16 (opt-in)")
    ap.add_argument("--collective", default="abi", choices=["abi", "torch"],
                    help="N > 1 on the nccl backend: 'abi' = the library's own RCCL communicator (cnf_comm_init / "
                         "cnf_allreduce_loss, include/cnf.h; falls back to torch.distributed if it cannot be formed), "
                         "'torch' = torch.distributed all_reduce")
    ap.add_argument("--bf16x6-secondary", default="auto", choices=["auto", "on", "off"],
                    help="also measure the opt-in split-bf16 arithmetic of the same workload and report it as 'secondary_bf16x6' "
                         "with its own error against the fp64 oracle (auto: the default cfg2 line on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (process group, RCCL communicator, loss all-reduce inside the timed "
                         "loop, teardown) even with one rank: the rehearsal of a multi-GPU launch on a 1-GPU box")
    ap.add_argument("--comm-timeout", type=float, default=90.0,
                    help="seconds to wait for ncclCommInitRank of the library's communicator before giving up (exit 3)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for "
                    "the single-GPU launch-contract test, where all ranks share device 0)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# CPU baseline legs (rank 0, N = 1): test infrastructure timed beside the GPU path, never on it
# ---------------------------------------------------------------------------------------------------
def _cpu_leg_c_port(oc, spec, alg, p, xs, eps, ys, target_s):
    """oracle/cnf_oracle.c on every host thread, a bounded column sample, one full 40-step solve."""
    nt = os.cpu_count() or 1
    nt = min(nt, oc.max_threads()) if oc.max_threads() > 0 else nt
    oc.set_fast_tanh(True)   # the arithmetic Lux's CPU path runs (NNlib.tanh_fast); vectorises
    ncols = xs.shape[1]
    B0 = min(64 * nt, ncols)        # (cfg1 has 1024 columns: fewer than 64 per thread of a 128-core host)
    t = time.perf_counter()
    oc.inference_fixed(spec, p, xs[:, :B0], 0.0, 1.0, NSTEPS, alg, eps[:, :B0],
                       None if ys is None else ys[:, :B0], nthreads=nt)
    dt0 = time.perf_counter() - t
    rate0 = B0 * NSTEPS / dt0
    Bs = int(min(ncols, max(B0, rate0 * target_s / NSTEPS)))
    Bs = min(ncols, max(B0, Bs // (64 * nt) * (64 * nt)))
    # a fast host finishes the whole batch in well under the target: repeat the solve until the sample is ~target_s long
    reps = int(max(1, min(64, round(rate0 * target_s / (Bs * NSTEPS)))))
    run = lambda: oc.inference_fixed(spec, p, xs[:, :Bs], 0.0, 1.0, NSTEPS, alg, eps[:, :Bs],
                                     None if ys is None else ys[:, :Bs], nthreads=nt)
    run()                                                    # page in the sample, spin the threads up
    t = time.perf_counter()
    for _ in range(reps):
        run()
    dt = time.perf_counter() - t
    oc.set_fast_tanh(False)
    return dict(value=reps * Bs * NSTEPS / dt, threads=nt, isa=oc.isa() if hasattr(oc, "isa") else "avx2",
                sample=f"{Bs} of the workload's columns, {reps} full {NSTEPS}-step solve(s), {dt:.1f} s, "
                       f"oracle/cnf_oracle.c (gcc -O3 -fopenmp, tanh_fast)")


def _cpu_leg_blas(spec, alg, p, xs, eps, ys, target_s):
    """The reference's CPU structure: per dynamics call whole-batch sgemm per Dense layer (forward, pullback)
    on every host thread — torch.mm (MKL / oneDNN), float32, the FULL batch; the bounded sample is a number
    of RK steps of the 40 (the rate per step does not depend on the step index).  Hutchinson-VJP only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnf_oracle_torch32 as t32
    nthr0 = torch.get_num_threads()
    nt = os.cpu_count() or nthr0
    best = None
    try:
        for threads in sorted({nt, max(1, nt // 2)}, reverse=True):   # SMT siblings rarely help a GEMM
            torch.set_num_threads(threads)
            B = xs.shape[1]
            t = time.perf_counter()
            t32.inference_fixed(spec, p, xs, 0.0, 1.0 / NSTEPS, 1, alg, eps, ys)     # one step: warm-up + estimate
            dt1 = time.perf_counter() - t
            ns = int(max(1, min(NSTEPS, (target_s / 2) / max(dt1, 1e-3))))
            t = time.perf_counter()
            t32.inference_fixed(spec, p, xs, 0.0, ns / NSTEPS, ns, alg, eps, ys)
            dt = time.perf_counter() - t
            leg = dict(value=B * ns / dt, threads=threads,
                       sample=f"all {B} columns, {ns} of the {NSTEPS} steps, {dt:.1f} s, "
                              f"oracle/cnf_oracle_torch32.py (torch.mm float32)")
            if best is None or leg["value"] > best["value"]:
                best = leg
    finally:
        torch.set_num_threads(nthr0)
    return best


def cpu_baseline(o64, oc, spec, alg, p, xs, eps, ys, target_s):
    legs = {"c_port": _cpu_leg_c_port(oc, spec, alg, p, xs, eps, ys, target_s)}
    if spec.mode == 0:
        try:
            legs["blas_whole_batch"] = _cpu_leg_blas(spec, alg, p, xs, eps, ys, target_s)
        except Exception as ex:  # pragma: no cover
            legs["blas_whole_batch"] = dict(error=str(ex)[:200])
    name, bestleg = max(((k, v) for k, v in legs.items() if "value" in v), key=lambda kv: kv[1]["value"])
    return dict(value=bestleg["value"], unit="samples*steps/s", cores=bestleg["threads"], kind="port",
                sample=f"[{name}] " + bestleg["sample"], host_threads=os.cpu_count(),
                julia_available=bool(shutil.which("julia")), legs=legs)


# ---------------------------------------------------------------------------------------------------
# shader-clock probe (measurement helper, profiles/ubench/clockprobe.hip)
# ---------------------------------------------------------------------------------------------------
class ClockProbe:
    def __init__(self, torch, dev):
        self.ok = False
        path = os.path.join(ROOT, "profiles", "ubench", "libclockprobe.so")
        if not os.path.exists(path):
            return
        try:
            self.lib = C.CDLL(path)
            self.lib.clockprobe_launch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
            khz = int(self.lib.clockprobe_ref_khz(dev.index or 0))
            self.ref_mhz = khz / 1e3 if khz > 0 else 100.0
            self.torch, self.dev, self.bufs = torch, dev, []
            self.ok = True
        except Exception:
            self.ok = False

    def sample(self):
        """Enqueue one ~20 us probe behind whatever was launched last on the current stream."""
        if not self.ok:
            return
        buf = self.torch.zeros(2, dtype=self.torch.int64, device=self.dev)
        self.lib.clockprobe_launch(int(20e-6 * self.ref_mhz * 1e6), C.c_void_p(buf.data_ptr()),
                                   C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream))
        self.bufs.append(buf)

    def drain(self):
        """MHz of every probe since the last drain (call after a synchronize)."""
        out = []
        for b in self.bufs:
            c, r = (int(v) for v in b.tolist())
            if r > 0:
                out.append(c / r * self.ref_mhz)
        self.bufs = []
        return out


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
def host_inputs(o64, name, rank, B):
    """The synthetic inputs of a workload on the host: weights shared by all ranks; the batch generated per global
    column block so that an N-GPU run evaluates N different shards (weak scaling)."""
    spec = o64.make_spec(**CONFIGS[name][0])
    p = o64.glorot_params(spec, np.random.default_rng(20240612))
    rng = np.random.default_rng(20240612 + 1000 * (rank + 1))
    xs = rng.standard_normal((spec.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((spec.nprobes * spec.D, B)).astype(np.float32)
    ys = rng.standard_normal((spec.ncond, B)).astype(np.float32) if spec.ncond else None
    return spec, p, xs, eps, ys


def make_workload(pkg, o64, name, a, rank, dev, torch, arith=None, grad=None, batch=None):
    kw, alg, Bdef, flop_ss, bytes_call_ss, bytes_fused, desc = CONFIGS[name]
    B = batch or a.batch or Bdef
    spec, p, xs, eps, ys = host_inputs(o64, name, rank, B)

    acts = {0: "identity", 1: "tanh", 2: "softplus"}
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], acts[spec.acts[i]])
              for i in range(len(spec.acts))]
    reg = bool(spec.reg_z or spec.reg_j)
    icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nconditions=spec.ncond,
                    nn=pkg.Chain(*layers),
                    compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=a.path, arith=1 if (arith or a.arith) == "bf16x6" else 0),
                    steer_rate=0.0, lambda1=0.01 if spec.reg_z else 0.0,
                    lambda2=0.01 if spec.reg_j else 0.0, lambda3=0.01 if (spec.reg_aug and spec.naug > 0) else 0.0, nprobes=spec.nprobes,
                    device=dev, sol_kwargs=dict(alg=pkg.Tsit5() if alg == 1 else pkg.RK4(),
                                                adaptive=False, nsteps=NSTEPS))
    mode = pkg.TestMode() if spec.mode == 2 else pkg.TrainMode(reg)
    # inputs resident in HBM, already in the column-major layout the ABI takes
    X = torch.tensor(xs.T.copy(), device=dev).t()
    E = torch.tensor(eps.T.copy(), device=dev).t()
    Y = torch.tensor(ys.T.copy(), device=dev).t() if ys is not None else None
    P = torch.tensor(p, device=dev)
    args = (X,) + ((Y,) if Y is not None else ()) + (P, {})
    return dict(name=name, spec=spec, alg=alg, B=B, flop_ss=flop_ss, bytes_call_ss=bytes_call_ss,
                bytes_fused=bytes_fused, desc=desc, icnf=icnf, mode=mode, args=args, E=E, nparams=int(len(p)),
                host=(p, xs, eps, ys), grad=(a.mode == "grad") if grad is None else bool(grad))


def small_batch_latency(pkg, o64, a, dev, torch):
    """Wall time of one call at the reference's own batch size (2^10 samples), host side included - what its PkgBenchmark suite
    measures (benchmark/benchmarks.jl:11-19: ICNF(; nvariables = 1), every default: the default net, VCABM at 1e-4, default lambdas):
    `loss` in TrainMode{true} and TestMode; and BASELINE's CPU-runnable configuration cfg1 (D = 2, 2 x 32, Tsit5 x 40) as one
    `inference` call.  The adaptive solves synchronise the stream (the host reads the step count), so calls do not pipeline."""
    import time
    out = {"what": "ms per call at B = 1024, Python + library + kernels; PkgBenchmark scenario = ICNF(nvariables = 1) defaults (VCABM, 1e-4)", "batch": 1024}
    r = torch.distributions.Beta(2.0, 4.0).sample((1, 1024)).float().to(dev)
    icnf = pkg.ICNF(nvariables=1, device=dev)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    ps = ps.to(dev)
    for key, mode in (("pkgbenchmark_loss_train_ms", pkg.TrainMode(True)), ("pkgbenchmark_loss_test_ms", pkg.TestMode())):
        for _ in range(20):
            pkg.loss(icnf, mode, r, ps, st)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            pkg.loss(icnf, mode, r, ps, st)
        torch.cuda.synchronize()
        out[key] = round(1e3 * (time.perf_counter() - t0) / n, 4)
        out[key.replace("_ms", "_steps")] = int(icnf.last_solve_stats["naccept"])
    w = make_workload(pkg, o64, "cfg1", a, 0, dev, torch, grad=False, batch=1024)
    fn = lambda: pkg.inference(w["icnf"], w["mode"], *w["args"], eps=w["E"])
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    out["cfg1_inference_ms"] = round(1e3 * (time.perf_counter() - t0) / n, 4)
    out["cfg1_kernel"] = w["icnf"].kernel_name(w["mode"])
    return out


def _quarters(ms):
    n = len(ms)
    q = max(1, n // 4)
    return float(np.mean(ms[:q])), float(np.mean(ms[-q:])), float(np.median(ms))


def measure(w, a, steps, warmup, preroll_s, pkg, torch, dist, world, dev, probe):
    """Warm-up, pre-roll to the sustained clock, then exactly `steps` timed steps.  Returns timings.
    `sharded` (N > 1, or --force-dist with one rank) selects the column-shard form of the step: cnf_loss_sums + the
    all-reduce of the loss scalars instead of cnf_loss_mean, and the barrier / max-over-ranks timing."""
    icnf, mode, args, E, B = w["icnf"], w["mode"], w["args"], w["E"], w["B"]
    grad = w["grad"]
    sharded = world > 1 or a.force_dist
    state = {}

    def launch(ev=None):
        """One step; the event pair brackets the dominant kernel's launch (the solve; for --mode grad the
        whole loss + gradient) on the launching stream."""
        if ev is not None:
            ev[0].record()
        if grad:
            state["loss"], _ = pkg.loss_and_gradient(icnf, mode, *args, eps=E)
            if ev is not None:
                ev[1].record()
            return
        logp, regs = pkg.inference(icnf, mode, *args, eps=E, _raw=True)
        if ev is not None:
            ev[1].record()
        if not sharded:    # one process: the mean comes out of the library's two reduction kernels (cnf_loss_mean), as in pkg.loss
            state["loss"] = pkg.loss_mean(icnf, mode, logp, regs)
        else:
            sums = pkg.loss_sums(icnf, mode, logp, regs)
            state["loss"] = pkg.reduce_loss(sums, B, (icnf.lambda1, icnf.lambda2, icnf.lambda3))

    def sync():
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def events(n):
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]

    for _ in range(max(1, warmup)):
        launch()
    sync()
    # estimate the step time from a short synchronised burst, then pre-roll
    t = time.perf_counter()
    for _ in range(3):
        launch()
    sync()
    est = max((time.perf_counter() - t) / 3, 1e-5)
    n_pre = int(min(20000, max(0, np.ceil(preroll_s / est)))) if preroll_s > 0 else 0
    if sharded:     # every rank must run the same number of steps (the loss all-reduce is a collective)
        tt = torch.tensor([n_pre], device=dev, dtype=torch.int64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        n_pre = int(tt.item())
    # Everything the timed region needs is allocated BEFORE the pre-roll, and the pre-roll's statistics are read out AFTER the
    # timed region: between the pre-roll's synchronize and the first timed launch the GPU idles for one host round trip only.
    # (Reading several hundred event pairs and clock probes in between left it idle for milliseconds - long enough for DVFS
    # to drop the clock: the first of K = 20 timed launches then took 3.4 ms instead of 2.96 and moved the mean by 1.5 %.)
    ev_pre = events(n_pre) if n_pre > 0 else []
    ev = events(steps)
    pre_wall = 0.0
    if n_pre > 0:
        t = time.perf_counter()
        for i in range(n_pre):
            launch(ev_pre[i])
            if i % 16 == 15:
                probe.sample()
        sync()
        pre_wall = time.perf_counter() - t
    else:
        sync()
    # ---- the timed region: exactly `steps` steps ----
    t0 = time.perf_counter()
    for i in range(steps):
        launch(ev[i])
    sync()
    elapsed = time.perf_counter() - t0
    if sharded:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    sustained = None
    if n_pre > 0:
        ms_pre = [s.elapsed_time(e) for s, e in ev_pre]
        fq, lq, med = _quarters(ms_pre)
        mhz = probe.drain()
        sustained = dict(launches=n_pre, seconds=pre_wall, first_quarter_ms=fq, last_quarter_ms=lq, median_ms=med,
                         clock_mhz=(dict(first_quarter=float(np.mean(mhz[:max(1, len(mhz) // 4)])),
                                         last_quarter=float(np.mean(mhz[-max(1, len(mhz) // 4):])),
                                         min=float(np.min(mhz)), max=float(np.max(mhz)), probes=len(mhz))
                                    if mhz else None))
    ms = [s.elapsed_time(e) for s, e in ev]
    probe.sample()
    torch.cuda.synchronize(dev)
    mhz_end = probe.drain()
    return dict(elapsed=elapsed, kern_ms=float(np.mean(ms)), kern_ms_median=float(np.median(ms)),
                kern_ms_min=float(np.min(ms)), kern_ms_max=float(np.max(ms)), sustained=sustained,
                clock_mhz_after_timed=(mhz_end[0] if mhz_end else None), loss=float(state["loss"]))


def report(w, m, a, steps, warmup, world):
    """The JSON fields of one measured workload."""
    spec, alg, B, icnf, mode = w["spec"], w["alg"], w["B"], w["icnf"], w["mode"]
    flop_ss = w["flop_ss"]
    stages = 4 if alg == 0 else 6
    value = world * B * NSTEPS * steps / m["elapsed"]
    path = icnf.kernel_path(mode)
    extra = {}
    grad = w["grad"]
    if grad:
        # (i) executed MFMA work of forward + reverse sweep per sample*step: v_mfma_f32_16x16x4_f32
        # instructions (2048 flop) per stage per 16-sample tile, recomputation included; (ii) the algorithmic figure:
        # reverse mode of a function costing F is 2F on top of F (each product once forwards, twice backwards) = 3 F
        gpath = icnf.grad_path(mode, B=B, alg=alg)      # the implementation THIS call took (cnf_grad_path_for), not the handle's hint
        gform = icnf.grad_form(mode, B, alg, NSTEPS) if gpath == 3 else 0
        extra["gradient_form"] = gform
        exec_ss = (grad_mfma_per_stage(spec) * 2048 / 16 if gpath == 1 else
                   (coop_grad3_flop_per_stage(spec) if gform == 2 else coop_grad_flop_per_stage(spec)) if gpath == 3
                   else layered_flop_per_stage(spec)) * stages
        extra["executed_flop_per_sample_step"] = exec_ss
        extra["executed_frac"] = exec_ss * B * NSTEPS / (m["kern_ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS
        flop_ss = 3 * flop_ss
    flops_launch = float(flop_ss) * B * NSTEPS
    ach = flops_launch / (m["kern_ms"] * 1e-3) / 1e12
    roof = {
        "bound": "mfma", "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": measured_traffic(w["name"] + (":grad" if grad else ""))[0],
        "traffic_source": (f"committed PMC pass {measured_traffic(w['name'] + (':grad' if grad else ''))[1]} (HBM bytes per step of the "
                           "whole workload; not collected by this run: bench.py cannot run rocprofv3 on itself)"
                           if measured_traffic(w["name"] + (":grad" if grad else ""))[1] else None),
        # algorithmic bytes of a step: inputs read once, outputs written once (a gradient step also writes dloss/dps)
        "algorithmic_bytes": w["bytes_fused"] * B + (4 * w["nparams"] if grad else 0),
        "kernel_ms": m["kern_ms"], "kernel_ms_median": m["kern_ms_median"], "kernel_ms_min": m["kern_ms_min"],
        "kernel_ms_max": m["kern_ms_max"], "flop_per_sample_step": flop_ss, "flop_per_launch": flops_launch,
        **extra,
        "hbm_model": {
            "fused_bytes_per_launch": w["bytes_fused"] * B,
            "fused_GBps": w["bytes_fused"] * B / (m["kern_ms"] * 1e-3) / 1e9,
            "per_call_abi_bytes_per_launch": w["bytes_call_ss"] * B * NSTEPS,
            "per_call_abi_GBps": w["bytes_call_ss"] * B * NSTEPS / (m["kern_ms"] * 1e-3) / 1e9,
            "per_call_abi_frac_of_8TBps": w["bytes_call_ss"] * B * NSTEPS / (m["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "stages_per_step": stages},
    }
    roof["traffic_over_algorithmic"] = (roof["traffic"] / roof["algorithmic_bytes"]) if roof["traffic"] else None
    s = m["sustained"]
    if s:
        s = dict(s)
        s["frac_last_quarter"] = flops_launch / (s["last_quarter_ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS
        s["frac_first_quarter"] = flops_launch / (s["first_quarter_ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS
        if s.get("clock_mhz"):
            # the f32 MFMA peak at the clock the chip actually held: 256 CUs x 256 flop/cycle
            pk = 256 * 256 * s["clock_mhz"]["last_quarter"] * 1e6 / 1e12
            s["peak_at_measured_clock_TFLOPs"] = pk
            s["frac_of_peak_at_measured_clock"] = flops_launch / (s["last_quarter_ms"] * 1e-3) / 1e12 / pk
        roof["sustained"] = s
    roof["clock_mhz_after_timed"] = m["clock_mhz_after_timed"]
    return {
        "value": value, "ms_per_step": 1e3 * m["elapsed"] / steps,
        "config": {"workload": w["desc"], "name": w["name"], "columns_per_gpu": B,
                   "global_columns": world * B, "nsteps": NSTEPS,
                   "integrator": "RK4" if alg == 0 else "Tsit5",
                   "kernel_path": {1: "simt", 2: "mfma", 3: "layered"}.get(path, str(path)),
                   "kernel_family": icnf.kernel_family(mode, B=B), "kernel": icnf.kernel_name(mode),
                   "mode": "grad" if grad else "infer",
                   **({"gradient_path": {1: "fused reverse-sweep kernel", 2: "layer-wise (hand-written MFMA product kernels)",
                                              3: "cooperative reverse sweep + deferred weight-cotangent products"}.get(
                       icnf.grad_path(mode, B=B, alg=alg), "none") +
                       (" (second form: the forward solve stores h_l / delta_l, second-order sweep, products over tiles)"
                        if extra.get("gradient_form") == 2 else "")} if grad else {}),
                   "collective": w.get("collective", ""),
                   "parallelism": f"column-shard x{world}, loss all-reduce (5 scalars)"
                   + (" + gradient all-reduce (nparams floats)" if grad else "")},
        "loss": m["loss"], "roofline": roof,
    }


def form_library_comm(pkg, torch, dist, dev, rank, world, timeout_s, fallback_name):
    """The library's own RCCL communicator (cnf_comm_init, include/cnf.h) over the ranks of the process group.
    Every torch.distributed collective of the set-up is issued by THIS (the main) thread, whose current device is the
    rank's: the unique id travels as a device tensor (no object collective, which would land on the calling thread's
    current device).  Only ncclCommInitRank itself runs in a helper thread - it takes the device as an argument
    (DeviceGuard in csrc/cnf_comm.hip) - so that a rendezvous that never completes becomes `exit 3` with a message after
    `timeout_s` instead of a job that hangs until the launcher's own timeout.  An init that FAILS (an error code, on any
    rank) falls back to torch.distributed's all_reduce on every rank, and the JSON line says so."""
    import threading
    n = pkg._lib.COMM_ID_BYTES
    uid = torch.zeros(n, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(pkg.Comm.unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    uid_bytes = bytes(uid.cpu().numpy().tobytes())
    box = {}

    def init_rank():
        try:
            torch.cuda.set_device(dev)         # the current device is per thread
            box["comm"] = pkg.Comm(rank, world, uid_bytes, dev)
        except Exception as ex:   # pragma: no cover
            box["err"] = str(ex)[:300]

    th = threading.Thread(target=init_rank, daemon=True)
    th.start()
    th.join(timeout_s)
    timed_out = th.is_alive()
    if timed_out:
        print(f"bench.py: rank {rank}: ncclCommInitRank of the library communicator did not return within {timeout_s:.0f} s; "
              f"re-run with --collective torch", file=sys.stderr, flush=True)
    # every rank learns the worst outcome on the process group's own communicator (the main thread's; the stuck helper thread
    # is inside a different one), so a timeout on ONE rank ends ALL ranks with exit 3 instead of leaving the others blocked in a
    # collective until the launcher kills them: -1 = some rank timed out, 0 = some rank's init failed, 1 = all formed
    ok = torch.tensor([-1 if timed_out else (1 if "comm" in box else 0)], device=dev, dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) < 0:
        if not timed_out:
            print(f"bench.py: rank {rank}: another rank's ncclCommInitRank timed out; exiting with it", file=sys.stderr, flush=True)
        os._exit(3)                            # a helper thread is inside RCCL: no orderly teardown is possible
    if int(ok.item()) == 1:
        pkg.set_comm(box["comm"])
        return "cnf_allreduce_loss: RCCL ncclAllReduce of 5 doubles through the C ABI (include/cnf.h)"
    if "comm" in box:
        box["comm"].destroy()
    print(f"bench.py: rank {rank}: cnf_comm_init unavailable ({box.get('err', 'failed on another rank')}); "
          f"using {fallback_name}", file=sys.stderr, flush=True)
    return fallback_name + f" [cnf_comm_init unavailable: {box.get('err', 'failed on another rank')}]"


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print(f"bench.py --gpus {a.gpus} must be launched with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if a.backend != "nccl":
        local = local % torch.cuda.device_count()   # contract test: ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sharded = world > 1 or a.force_dist
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")      # only reached by a bare `python bench.py --force-dist`
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    pkg = entry.load_package()
    o64, oc = entry.load_oracle()      # input generation and the cpu_baseline leg only (outside the timed regions)
    probe = ClockProbe(torch, dev)
    collective = "none (one GPU)"
    if sharded:
        collective = f"torch.distributed all_reduce ({a.backend})"
        if a.backend == "nccl" and a.collective == "abi":
            collective = form_library_comm(pkg, torch, dist, dev, rank, world, a.comm_timeout, collective)

    w = make_workload(pkg, o64, a.config, a, rank, dev, torch)
    w["collective"] = collective
    m = measure(w, a, a.steps, a.warmup, a.preroll_seconds, pkg, torch, dist, world, dev, probe)
    sec_name = ("cfg2p" if (a.config == "cfg2" and a.mode == "infer" and not a.batch) else "none") \
        if a.secondary == "auto" else a.secondary
    sec = None
    if sec_name != "none":
        w2 = make_workload(pkg, o64, sec_name, a, rank, dev, torch)
        w2["collective"] = collective
        m2 = measure(w2, a, a.steps, a.warmup, min(a.preroll_seconds, 1.5), pkg, torch, dist, world, dev, probe)
        sec = (w2, m2)

    # the opt-in split-bf16 arithmetic as a reported secondary (never the headline: exact f32 is the validated path)
    bf = None
    want_bf = a.bf16x6_secondary == "on" or (a.bf16x6_secondary == "auto" and a.config in ("cfg2", "cfg3") and a.mode == "infer" and
                                              a.arith == "f32" and world == 1 and not a.force_dist and not a.batch)
    if want_bf:
        try:
            w3 = make_workload(pkg, o64, a.config, a, rank, dev, torch, arith="bf16x6")
            w3["collective"] = collective
            m3 = measure(w3, a, a.steps, a.warmup, min(a.preroll_seconds, 1.0), pkg, torch, dist, world, dev, probe)
            # 64 columns of this very batch under both arithmetics; the fp64 oracle is run on them in the cpu_baseline leg
            idx = np.arange(0, w["B"], max(1, w["B"] // 64))[:64]
            errs = {"idx": idx}
            for tag, ww in (("f32", w), ("bf16x6", w3)):
                errs[tag] = pkg.inference(ww["icnf"], ww["mode"], *ww["args"], eps=ww["E"])[0].cpu().numpy()[idx]
            bf = (w3, m3, errs)
        except Exception as ex:  # pragma: no cover
            bf = ("error", str(ex)[:200])

    # every BASELINE configuration + the parameter gradient by the same protocol (shorter pre-roll, fewer timed steps for the
    # long ones), compactly: VERDICT r3 #1 - cfg3 / cfg4 / cfg5 and the gradient figures were builder-run claims only
    default_line = (a.config == "cfg2" and a.mode == "infer" and a.arith == "f32" and world == 1 and not a.force_dist and
                    not a.batch and a.path == 0)
    sec_list = ("cfg3,cfg4,cfg5,cfg2:grad,cfg3:grad,cfg4:grad,nv20,nv20:grad" if default_line else "none") if a.secondaries == "auto" else a.secondaries
    more = []
    if sec_list != "none":
        for item in [x.strip() for x in sec_list.split(",") if x.strip()]:
            nm, _, md = item.partition(":")
            g = md == "grad"
            try:
                wi = make_workload(pkg, o64, nm, a, rank, dev, torch, grad=g, batch=CONFIGS[nm][2])
                wi["collective"] = collective
                ki = min(a.steps, 20 if g else 50)
                mi = measure(wi, a, ki, min(a.warmup, 3), min(a.preroll_seconds, 1.0), pkg, torch, dist, world, dev, probe)
                more.append((item, wi, mi, ki))
            except Exception as ex:  # pragma: no cover
                more.append((item, None, str(ex)[:200], 0))
            finally:
                wi = None
                torch.cuda.empty_cache()

    small = None
    if default_line and a.secondaries == "auto":
        try:
            small = small_batch_latency(pkg, o64, a, dev, torch)
        except Exception as ex:  # pragma: no cover
            small = {"error": str(ex)[:200]}

    ranks_seen = None
    if sharded:
        # what every rank bound and what the communicator it reduced on reports: the driver's log then proves that RCCL saw N ranks
        c = pkg.get_comm()
        mine = torch.tensor([rank, dev.index or 0, c.size() if c is not None else -1, dist.get_world_size()], device=dev, dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = [dict(rank=int(t[0]), device=int(t[1]), cnf_comm_size=int(t[2]), process_group_size=int(t[3])) for t in allr]
        print(f"bench.py: rank {rank} bound cuda:{dev.index} ({torch.cuda.get_device_name(dev)}), "
              f"cnf_comm_size() = {c.size() if c is not None else 'n/a (torch.distributed collective)'}, world {world}", file=sys.stderr, flush=True)

    if rank == 0:
        r = report(w, m, a, a.steps, a.warmup, world)
        rf = r["roofline"]
        # ---- the full record of the run: ONE line on stderr (prefix "bench.py detail: "); the stdout line below is its digest ----
        detail = {"config": r["config"], "roofline": rf, "protocol": {"preroll_seconds": a.preroll_seconds}}

        def digest(ri, grad):
            q = ri["roofline"]
            d = {"ms": round(ri["ms_per_step"], 4), "frac": round(q["frac"], 4), "value": round(ri["value"], 1),
                 "family": ri["config"]["kernel_family"]}
            if "executed_frac" in q:
                d["exec"] = round(q["executed_frac"], 4)
            if q.get("traffic_over_algorithmic"):
                d["traffic_ratio"] = round(q["traffic_over_algorithmic"], 1)
            if grad:
                d["form"] = q.get("gradient_form", 0)
            return d

        out = {
            "metric": "log-density evals (samples*steps)/sec" if a.mode == "infer"
            else "training-step evals (samples*steps)/sec: loss + dloss/dp",
            "value": r["value"], "unit": "samples*steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.arith == "f32" else "f32 via 3-way bf16 split (6 bf16 MFMAs per hidden product)",
            "data": "synthetic",
            "config": {k: r["config"][k] for k in ("workload", "name", "columns_per_gpu", "global_columns", "nsteps", "integrator", "kernel_family",
                                                   "kernel", "mode", "parallelism") if k in r["config"]},
            "loss": r["loss"],
            "roofline": {"bound": "mfma", "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                         "traffic": rf["traffic"], "algorithmic_bytes": rf["algorithmic_bytes"], "traffic_over_algorithmic": rf["traffic_over_algorithmic"],
                         "kernel_ms": rf["kernel_ms"], "kernel_ms_median": rf["kernel_ms_median"], "flop_per_sample_step": rf["flop_per_sample_step"],
                         "per_call_abi_frac_of_8TBps": rf["hbm_model"]["per_call_abi_frac_of_8TBps"],
                         **({"sustained_frac_last_quarter": rf["sustained"]["frac_last_quarter"],
                             "clock_mhz": (rf["sustained"].get("clock_mhz") or {}).get("last_quarter")} if rf.get("sustained") else {}),
                         **({k: rf[k] for k in ("executed_frac", "gradient_form") if k in rf}),
                         "traffic_src": "profiles/pmc_traffic.json (committed PMC passes; bench.py cannot run rocprofv3 on itself)",
                         "traffic_stale": traffic_is_stale()},
        }
        if "gradient_path" in r["config"]:
            out["config"]["gradient_path"] = r["config"]["gradient_path"]
        if r["config"].get("collective"):
            out["config"]["collective"] = r["config"]["collective"]
        if ranks_seen is not None:
            out["ranks_seen"] = ranks_seen
        if more:
            out["secondaries"] = {"_": "ms per step; frac of the f32 MFMA peak (gradients: 3 F convention), exec = executed flops; traffic_ratio = HBM bytes / algorithmic; form 2 = stage-store gradient"}
            detail["secondaries"] = {}
            for item, wi, mi, ki in more:
                if wi is None:
                    out["secondaries"][item] = {"error": mi}
                    continue
                ri = report(wi, mi, a, ki, a.warmup, world)
                out["secondaries"][item] = digest(ri, wi["grad"])
                detail["secondaries"][item] = ri
        if small is not None:
            out["small_batch"] = {k: v for k, v in small.items() if k != "what"}
        if sec is not None:
            r2 = report(sec[0], sec[1], a, a.steps, a.warmup, world)
            out["secondary"] = {"name": sec[0]["name"], "why": "north_star's target (Tsit5 x 40), same protocol", "kernel_ms": r2["roofline"]["kernel_ms"],
                                **digest(r2, sec[0]["grad"])}
            detail["secondary"] = r2
        if bf is not None and bf[0] != "error":
            r3 = report(bf[0], bf[1], a, a.steps, a.warmup, world)
            out["secondary_bf16x6"] = {
                "what": "opt-in split-bf16 hidden products (cnf_config.arith = CNF_ARITH_BF16X6), not the headline",
                "value": r3["value"], "ms_per_step": r3["ms_per_step"], "speedup_vs_f32": r3["value"] / r["value"],
                "max_abs_dlogp_f32_vs_bf16x6": float(np.max(np.abs(bf[2]["f32"] - bf[2]["bf16x6"])))}
            detail["secondary_bf16x6"] = r3
        elif bf is not None:
            out["secondary_bf16x6"] = {"error": bf[1]}
        if world == 1 and not a.no_cpu_baseline and not a.force_dist:
            p, xs, eps, ys = w["host"]
            cb = cpu_baseline(o64, oc, w["spec"], w["alg"], p, xs, eps, ys, a.cpu_seconds)
            detail["cpu_baseline"] = cb
            out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "host_threads", "julia_available")}
            out["cpu_baseline"]["legs"] = {k: round(v["value"], 1) for k, v in cb["legs"].items() if "value" in v}
            out["gpu_over_cpu"] = out["value"] / cb["value"]
            # the legs BASELINE.md section 2 promises beside the headline: cfg2' (the north_star's target) and cfg1 (BASELINE config 1:
            # "CPU MatrixMode (reference, no GPU)") on the C port, bounded samples
            if default_line:
                for nm in ("cfg2p", "cfg1"):
                    try:
                        sp, pp, xx, ee, yy = host_inputs(o64, nm, rank, CONFIGS[nm][2])
                        leg = _cpu_leg_c_port(oc, sp, CONFIGS[nm][1], pp, xx, ee, yy, min(a.cpu_seconds, 5.0))
                        out["cpu_baseline"][nm] = {"value": round(leg["value"], 1), "cores": leg["threads"], "sample": leg["sample"]}
                    except Exception as ex:  # pragma: no cover
                        out["cpu_baseline"][nm] = {"error": str(ex)[:120]}
                if sec is not None and "value" in out["cpu_baseline"].get("cfg2p", {}):
                    out["secondary"]["gpu_over_cpu"] = out["secondary"]["value"] / out["cpu_baseline"]["cfg2p"]["value"]
            if bf is not None and bf[0] != "error":
                # the checker on the same 64 columns: the fp64 oracle (full 40-step solves), both arithmetics against it
                idx = bf[2]["idx"]
                ref64 = o64.inference_fixed(w["spec"], p, xs[:, idx], 0.0, 1.0, NSTEPS, w["alg"], eps[:, idx],
                                            None if ys is None else ys[:, idx])[0]
                out["secondary_bf16x6"]["max_abs_dlogp_vs_fp64"] = {
                    "columns": int(len(idx)), "f32": float(np.max(np.abs(bf[2]["f32"] - ref64))),
                    "bf16x6": float(np.max(np.abs(bf[2]["bf16x6"] - ref64))), "tolerance": 1e-4}
        print("bench.py detail: " + json.dumps(detail), file=sys.stderr, flush=True)
        print(json.dumps(out))
    if sharded:
        c = pkg.get_comm()
        pkg.set_comm(None)
        if c is not None:
            c.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

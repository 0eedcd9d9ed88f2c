#!/usr/bin/env python3
"""bench.py — throughput of the batched log-density hot path on MI355X.

A "step" is one pass of the hot path over one batch: `loss(icnf, mode, xs, ps, st)` =
cnf_inference_fixed (fused fixed-step solve + log-density epilogue) + cnf_loss_sums (+ the
RCCL all-reduce of the loss scalars when N > 1), with inputs already resident in HBM.
Metric (BASELINE.json): log-density evaluations counted as samples·steps per second.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg2p|cfg3|cfg5|cfg1]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`;
the batch columns are sharded by rank (weak scaling: 65 536 columns per GPU), no data-path
collective, one all-reduce of five scalars per step.

The JSON line also carries
  roofline     — the fused solve kernel against the f32 MFMA peak (the path is compute-bound:
                 ≥ 97 flop/B); its HBM figure is reported beside it.
  cpu_baseline — the CPU fp32 restatement (oracle/, "port") timed on this host's cores on a
                 bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
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
    "cfg5": (dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=2), 0, 16384, 2393088, 480, 68,
             "CondFFJORD nvars=8+8 cond, MLP 3x128 tanh, exact trace, RK4 40 steps, batch=16384"),
}
NSTEPS = 40


def measured_traffic(name):
    """HBM bytes per launch of the solve kernel from the PMC passes of profiles/collect.sh
    (FETCH_SIZE and WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md §HBM).  bench.py cannot run rocprofv3 on itself, so the figure is read
    from the committed summary; null when the configuration has not been profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(name)
    except Exception:
        return None


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = f32 vector rate
HBM_PEAK_GBS = 8000.0


def grad_mfma_per_stage(spec):
    """MFMA instructions one wave executes per RK stage for loss + gradient (forward solve kernel + reverse
    sweep), counted from the kernels' loops (csrc/cnf_mfma_kernel.h, cnf_grad.hip, cnf_grad_probes.hip):
    a product with MT output tiles and KS k-steps is MT*KS instructions; an outer-product update of one
    16x16 tile over the workgroup's 4 sample tiles is 4 x 4."""
    H, L, K = max(spec.widths[1:-1]), len(spec.acts) - 1, spec.nprobes
    HT, ZR, CR = -(-H // 16), -(-spec.D // 4), -(-spec.ncond // 4)
    DT = -(-ZR // 4)
    hid, first, last = (L - 1) * 4 * HT * HT, HT * (ZR + CR), DT * 4 * HT
    # forward solve kernel: chain, then the pullback per probe (one probe: c = W_N^T eps hoisted, and
    # without |eps^T J| the last product is a dot with the hoisted q = W_1 eps)
    fwd = first + hid + last
    fwd += hid + (last if spec.reg_j else 0) if K == 1 else K * (HT * ZR + hid + last)
    chain = first + hid + (last if spec.reg_z else 0)          # recompute of h_l, act'_l (+ zdot for |zdot|)
    top = HT * ZR + hid + last                                 # W_N^T kbar, W_l^T abar_l, W_1^T abar_1
    if K == 1:
        rev = chain + hid + ((last + HT * ZR) if spec.reg_j else 0) + hid + top
        rev += 4 * 8 + (L - 1) * 4 * (4 + 8 * HT) + 4 * (8 + (4 if CR else 0))
    else:
        per_probe = HT * ZR + hid + (last if spec.reg_j else 0) + HT * ZR + hid + (L - 1) * 16 * HT + 4 * 8
        rev = chain + K * per_probe + top
        rev += 4 * 4 + (L - 1) * 4 * (4 + 4 * HT) + 4 * (4 + (4 if CR else 0))
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override columns per GPU")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 SIMT, 2 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--mode", default="infer", choices=["infer", "grad"],
                    help="infer: loss (default, the BASELINE metric); grad: loss_and_gradient — forward with "
                         "checkpoints + reverse sweep + all-reduce of nparams floats")
    ap.add_argument("--arith", default="f32", choices=["f32", "bf16x6"],
                    help="hidden-product arithmetic: exact f32 MFMA (default) or split-bf16 (opt-in)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for "
                    "the single-GPU launch-contract test, where all ranks share device 0)")
    return ap.parse_args()


def cpu_baseline(o64, oc, spec, alg, p, xs, eps, ys, target_s):
    """Time the CPU restatement on all host cores on a bounded column sample."""
    nt = os.cpu_count() or 1
    nt = min(nt, oc.max_threads()) if oc.max_threads() > 0 else nt
    oc.set_fast_tanh(True)   # the arithmetic Lux's CPU path runs (NNlib.tanh_fast); vectorises
    B0 = 64 * nt
    t = time.perf_counter()
    oc.inference_fixed(spec, p, xs[:, :B0], 0.0, 1.0, NSTEPS, alg, eps[:, :B0],
                       None if ys is None else ys[:, :B0], nthreads=nt)
    dt0 = time.perf_counter() - t
    rate0 = B0 * NSTEPS / dt0
    Bs = int(min(xs.shape[1], max(B0, rate0 * target_s / NSTEPS)))
    Bs = max(B0, Bs // (64 * nt) * (64 * nt))
    t = time.perf_counter()
    oc.inference_fixed(spec, p, xs[:, :Bs], 0.0, 1.0, NSTEPS, alg, eps[:, :Bs],
                       None if ys is None else ys[:, :Bs], nthreads=nt)
    dt = time.perf_counter() - t
    out = dict(value=Bs * NSTEPS / dt, unit="samples*steps/s", cores=nt, kind="port",
               sample=f"{Bs} of the workload's columns, one full {NSTEPS}-step solve, "
                      f"{dt:.1f} s, oracle/cnf_oracle.c (gcc -O3 -march=x86-64-v3 -fopenmp, tanh_fast)")
    # CPU-favourable cross-check: the same unfused algorithm with every product a library
    # GEMM (torch.mm, MKL/oneDNN, float32), Hutchinson-VJP configurations only; a few seconds of work
    if spec.mode == 0:
        try:
            import torch
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cnf_oracle_torch32 as t32
            nthr0 = torch.get_num_threads()
            torch.set_num_threads(min(16, nthr0))   # small GEMMs: more threads than this only add overhead
            Bt = int(min(xs.shape[1], 8192))
            yt = None if ys is None else ys[:, :Bt]
            t32.inference_fixed(spec, p, xs[:, :256], 0.0, 1.0, 2, alg, eps[:, :256], None if ys is None else ys[:, :256])
            t = time.perf_counter()
            t32.inference_fixed(spec, p, xs[:, :Bt], 0.0, 1.0, NSTEPS, alg, eps[:, :Bt], yt)
            dtt = time.perf_counter() - t
            out["torch_f32_gemm"] = dict(value=Bt * NSTEPS / dtt, threads=torch.get_num_threads(),
                                         sample=f"{Bt} columns, {dtt:.1f} s, oracle/cnf_oracle_torch32.py")
            torch.set_num_threads(nthr0)
        except Exception as ex:  # pragma: no cover - the cross-check is optional
            out["torch_f32_gemm"] = dict(error=str(ex)[:200])
    return out


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    pkg = entry.load_package()
    o64, oc = entry.load_oracle()
    kw, alg, Bdef, flop_ss, bytes_call_ss, bytes_fused, desc = CONFIGS[a.config]
    B = a.batch or Bdef
    spec = o64.make_spec(**kw)
    # weights are shared by all ranks; the batch is generated per global column block so an
    # N-GPU run evaluates N different shards (weak scaling).
    p = o64.glorot_params(spec, np.random.default_rng(20240612))
    rng = np.random.default_rng(20240612 + 1000 * (rank + 1))
    xs = rng.standard_normal((spec.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((spec.nprobes * spec.D, B)).astype(np.float32)
    ys = rng.standard_normal((spec.ncond, B)).astype(np.float32) if spec.ncond else None

    acts = {0: "identity", 1: "tanh", 2: "softplus"}
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], acts[spec.acts[i]])
              for i in range(len(spec.acts))]
    reg = bool(spec.reg_z or spec.reg_j)
    icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nconditions=spec.ncond,
                    nn=pkg.Chain(*layers),
                    compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=a.path, arith=1 if a.arith == "bf16x6" else 0),
                    steer_rate=0.0, lambda1=0.01 if spec.reg_z else 0.0,
                    lambda2=0.01 if spec.reg_j else 0.0, lambda3=0.0, nprobes=spec.nprobes,
                    device=dev, sol_kwargs=dict(alg=pkg.Tsit5() if alg == 1 else pkg.RK4(),
                                                adaptive=False, nsteps=NSTEPS))
    mode = pkg.TestMode() if spec.mode == 2 else pkg.TrainMode(reg)
    # inputs resident in HBM, already in the column-major layout the ABI takes
    X = torch.tensor(xs.T.copy(), device=dev).t()
    E = torch.tensor(eps.T.copy(), device=dev).t()
    Y = torch.tensor(ys.T.copy(), device=dev).t() if ys is not None else None
    P = torch.tensor(p, device=dev)
    args = (X,) + ((Y,) if Y is not None else ()) + (P, {})

    def step():
        if a.mode == "grad":
            return pkg.loss_and_gradient(icnf, mode, *args, eps=E)[0]
        return pkg.loss(icnf, mode, *args, eps=E)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    # kernel-only timing with events on the launching stream (solve kernel = dominant kernel)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        if a.mode == "grad":
            lossv, gradv = pkg.loss_and_gradient(icnf, mode, *args, eps=E)
            ev[i][1].record()
            continue
        logp, regs = pkg.inference(icnf, mode, *args, eps=E, _raw=True)
        ev[i][1].record()
        sums = pkg.loss_sums(icnf, mode, logp, regs)
        lossv = pkg.reduce_loss(sums, B, (icnf.lambda1, icnf.lambda2, icnf.lambda3))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    if rank == 0:
        value = world * B * NSTEPS * a.steps / elapsed
        path = icnf.kernel_path(mode)
        flops_launch = float(flop_ss) * B * NSTEPS
        ach_tflops = flops_launch / (kern_ms * 1e-3) / 1e12
        stages = 4 if alg == 0 else 6
        if a.mode == "grad":
            # executed MFMA work of forward + reverse sweep per sample*step, not an
            # algorithmic figure: v_mfma_f32_16x16x4_f32 instructions (2048 flop) per stage per 16-sample tile
            gpath = icnf.grad_path(mode)
            flop_ss = (grad_mfma_per_stage(spec) * 2048 / 16 if gpath == 1 else layered_flop_per_stage(spec)) * stages
            flops_launch = float(flop_ss) * B * NSTEPS
            ach_tflops = flops_launch / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "log-density evals (samples*steps)/sec" if a.mode == "infer"
            else "training-step evals (samples*steps)/sec: loss + dloss/dp",
            "value": value, "unit": "samples*steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.arith == "f32" else "f32 via 3-way bf16 split (6 bf16 MFMAs per hidden product)",
            "data": "synthetic",
            "config": {"workload": desc, "name": a.config, "columns_per_gpu": B,
                       "global_columns": world * B, "nsteps": NSTEPS,
                       "integrator": "RK4" if alg == 0 else "Tsit5",
                       "kernel_path": {1: "simt", 2: "mfma"}.get(path, str(path)),
                       "mode": a.mode,
                       **({"gradient_path": {1: "fused reverse-sweep kernel", 2: "layer-wise (rocBLAS GEMMs)"}.get(
                           icnf.grad_path(mode), "none")} if a.mode == "grad" else {}),
                       "parallelism": f"column-shard x{world}, loss all-reduce (5 scalars)"
                       + (" + gradient all-reduce (nparams floats)" if a.mode == "grad" else "")},
            "loss": float(lossv),
            "roofline": {
                "bound": "mfma", "achieved": ach_tflops, "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach_tflops / F32_MFMA_PEAK_TFLOPS,
                "traffic": measured_traffic(a.config),
                "kernel_ms": kern_ms, "flop_per_sample_step": flop_ss,
                "hbm_model": {
                    "fused_bytes_per_launch": bytes_fused * B,
                    "fused_GBps": bytes_fused * B / (kern_ms * 1e-3) / 1e9,
                    "per_call_abi_bytes_per_launch": bytes_call_ss * B * NSTEPS,
                    "per_call_abi_GBps": bytes_call_ss * B * NSTEPS / (kern_ms * 1e-3) / 1e9,
                    "per_call_abi_frac_of_8TBps": bytes_call_ss * B * NSTEPS / (kern_ms * 1e-3) / 1e9
                    / HBM_PEAK_GBS,
                    "stages_per_step": stages},
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(o64, oc, spec, alg, p, xs, eps, ys, a.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

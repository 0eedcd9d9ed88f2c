# What cnf_loss_grad_adaptive differs from, measured in the fp64 oracle - no GPU.
#
# Under the reference's DEFAULT sol_kwargs (alg = VCABM(), reltol = abstol = 1e-4, sensealg = QuadratureAdjoint at 1e-4:
# src/core/icnf.jl:84-99) the gradient it trains with is a continuous adjoint of the loss on the VCABM solution.  The library
# returns the EXACT gradient of a different discrete object: the fixed-step Tsit5 solve on the accepted steps of an adaptive Tsit5
# solve at the same tolerance ("frozen grid", include/cnf.h: cnf_loss_grad_adaptive).  Both approximate the gradient of the
# loss of the exact flow.  This script measures, for the reference's default architecture at nvariables = 1 and 8:
#   fd_tight   central differences of the loss on the VCABM solution at tolerance 1e-10 along random unit directions v
#              (the gradient of the exact-flow loss, to ~1e-6 relative)
#   frozen     v . (frozen-grid gradient at tolerance 1e-4)       <- what the library returns
#   fd_1e-4    the same central differences with VCABM at the reference's own 1e-4 (what differentiating the adaptive solve
#              itself would see: step-size decisions change with p, so this is noisy - reported, not used)
# and prints  |frozen - fd_tight| / |fd_tight|  per direction and over all directions.
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import cnf_oracle64 as o64  # noqa: E402

LAM = (0.01, 0.01, 0.01)            # src/core/icnf.jl:73-75


def default_spec(nv):
    D = 2 * nv + 1                  # naugments = nvariables + 1 (icnf.jl:62)
    H = 4 * (D + 1)                 # n_hidden = 4 n_in, n_in = D + 1 (icnf.jl:64-66)
    return o64.make_spec(nvars=nv, naug=nv + 1, hidden=[H, H], act=2, reg_z=True, reg_j=True, reg_aug=True)


def loss_on(spec, p, xs, eps, u1):
    D = spec.D
    z, dlogp = u1[:D], u1[D]
    logp = o64.std_normal_logpdf(z) - dlogp
    A = np.sqrt((z[spec.nvars:] ** 2).sum(0))
    return float(np.mean(-logp + LAM[0] * u1[D + 1] + LAM[1] * u1[D + 2] + LAM[2] * A))


def vcabm_loss(spec, p, xs, eps, tol):
    B = xs.shape[1]
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
    u1, st = o64.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol, eps, None)
    return loss_on(spec, p, xs, eps, u1), st


def run(nv, scale=1.0, B=16, ndir=6, h=2e-4, seed=3):
    """scale: the Glorot-initialised weights multiplied by it (1: a fresh net, whose flow four Tsit5 steps resolve; 3: a stiffer,
    trained-like flow)"""
    spec = default_spec(nv)
    p, xs, eps, _ = o64.synth_inputs(spec, B, seed, bias_scale=0.1)
    p = p.astype(np.float64) * scale
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
    # the frozen grid of the library's gradient: accepted steps of adaptive Tsit5 at the reference's tolerance
    _, st = o64.integrate_adaptive_tsit5(spec, p, u0, 0.0, 1.0, 1e-4, 1e-4, eps, None)
    grid = np.concatenate([[0.0], np.cumsum(st["dts"])])
    grid[-1] = 1.0
    Lf, g = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, len(grid) - 1, o64.ALG_TSIT5, eps, None, LAM, tgrid=grid)
    Lv, stv = vcabm_loss(spec, p, xs, eps, 1e-4)
    Lt, stt = vcabm_loss(spec, p, xs, eps, 1e-10)
    rng = np.random.default_rng(100 + nv)
    rows = []
    for k in range(ndir):
        v = rng.standard_normal(p.size)
        v /= np.linalg.norm(v)
        fd_t = (vcabm_loss(spec, p + h * v, xs, eps, 1e-10)[0] - vcabm_loss(spec, p - h * v, xs, eps, 1e-10)[0]) / (2 * h)
        fd_r = (vcabm_loss(spec, p + h * v, xs, eps, 1e-4)[0] - vcabm_loss(spec, p - h * v, xs, eps, 1e-4)[0]) / (2 * h)
        rows.append(dict(frozen=float(v @ g), fd_tight=fd_t, fd_1e4=fd_r))
    fr = np.array([r["frozen"] for r in rows]); ft = np.array([r["fd_tight"] for r in rows]); f4 = np.array([r["fd_1e4"] for r in rows])
    return dict(nvariables=nv, weight_scale=scale, widths=list(spec.widths), B=B, nparams=int(p.size), tsit5_steps=len(grid) - 1, vcabm_steps_1e4=stv["naccept"],
                vcabm_steps_tight=stt["naccept"], loss_frozen_grid=Lf, loss_vcabm_1e4=Lv, loss_vcabm_tight=Lt,
                rel_loss_gap_frozen_vs_tight=abs(Lf - Lt) / abs(Lt), rel_loss_gap_vcabm1e4_vs_tight=abs(Lv - Lt) / abs(Lt),
                directions=rows,
                rel_gap_frozen_vs_tight=float(np.linalg.norm(fr - ft) / np.linalg.norm(ft)),
                rel_gap_per_direction=[float(abs(a - b) / abs(b)) for a, b in zip(fr, ft)],
                rel_gap_fd1e4_vs_tight=float(np.linalg.norm(f4 - ft) / np.linalg.norm(ft)),
                gnorm=float(np.linalg.norm(g)))


if __name__ == "__main__":
    out = [run(int(nv), float(sc)) for nv in os.environ.get("NV", "1,8").split(",") for sc in os.environ.get("SCALE", "1,3").split(",")]
    print(json.dumps(out, indent=1))

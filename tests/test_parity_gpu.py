"""GPU: parity of the HIP path (called through the C ABI) with the oracles.

Tolerances (float32 arithmetic, stated by north_star: log-density max abs error < 1e-4):
  * one dynamics call vs fp64 fixtures: 2e-5 relative-ish (|err| / (1+|ref|))
  * fixed-step solve vs fp64 fixtures / C restatement: 1e-4 absolute on logp, regs, state
  * structural properties (column independence, shard concatenation, SIMT-vs-MFMA agreement,
    determinism) are bit-exact where the same kernel runs on the same column."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_NAMES, load_golden

pytestmark = pytest.mark.gpu

ACTS = {0: "identity", 1: "tanh", 2: "softplus"}
TOL_CALL = 2e-5
TOL_SOLVE = 1e-4


def make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=(0.01, 0.01, 0.01)):
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]])
              for i in range(len(spec.acts))]
    cm = (pkg.HIPJacVecMatrixMode if spec.mode == 1 else pkg.HIPVecJacMatrixMode)(kernel_path=path)
    l1 = lambdas[0] if spec.reg_z else 0.0
    l2 = lambdas[1] if spec.reg_j else 0.0
    l3 = lambdas[2] if spec.reg_aug else 0.0
    return pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nconditions=spec.ncond,
                    autonomous=spec.autonomous, nn=pkg.Chain(*layers), compute_mode=cm,
                    steer_rate=0.0, lambda1=l1, lambda2=l2, lambda3=l3, nprobes=spec.nprobes,
                    device="cuda:0",
                    sol_kwargs=dict(alg=pkg.Tsit5() if alg == 1 else pkg.RK4(), adaptive=False,
                                    nsteps=nsteps))


def setsw(pkg, monkeypatch, var, val):
    """A CNF_* switch for this test: the environment (read by cnf_create for handles made from here on) AND the library's
    switchboard (include/cnf.h: cnf_tuning) for the handles that already exist."""
    monkeypatch.setenv(var, str(val))
    pkg.reload_tuning()


def delsw(pkg, monkeypatch, var):
    monkeypatch.delenv(var, raising=False)
    pkg.reload_tuning()


def mode_of(pkg, spec):
    if spec.mode == 2:
        return pkg.TestMode()
    return pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))


def dev(a):
    return None if a is None else torch.tensor(np.asarray(a, dtype=np.float32), device="cuda:0")


def run_inference(pkg, icnf, spec, p, xs, eps, ys, **kw):
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    return pkg.inference(icnf, mode_of(pkg, spec), *args, eps=dev(eps), **kw)


def paths_for(pkg, spec, alg, nsteps):
    """SIMT and the layer-wise GEMM path always; MFMA when the library says the configuration is covered."""
    out = [1, 3]
    try:
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2)
        icnf.kernel_path(mode_of(pkg, spec))
        out.append(2)
    except pkg._lib.CnfError as e:
        assert e.code == pkg._lib.ERR_UNSUPPORTED
    return out


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_aug_f_matches_golden(name, pkg):
    spec, meta, g = load_golden(name)
    for path in paths_for(pkg, spec, meta["alg"], meta["nsteps"]):
        icnf = make_icnf(pkg, spec, meta["alg"], meta["nsteps"], path)
        du = pkg.augmented_f(icnf, mode_of(pkg, spec), dev(g["u"]), dev(g["p"]), float(g["t"]),
                             dev(g["eps"]), dev(g["ys"])).cpu().numpy()
        err = np.max(np.abs(du - g["du"]) / (1.0 + np.abs(g["du"])))
        assert err < TOL_CALL, (name, path, err)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_inference_matches_golden(name, pkg):
    spec, meta, g = load_golden(name)
    for path in paths_for(pkg, spec, meta["alg"], meta["nsteps"]):
        icnf = make_icnf(pkg, spec, meta["alg"], meta["nsteps"], path)
        logp, (E, n, A), u1 = run_inference(pkg, icnf, spec, g["p"], g["xs"], g["eps"], g["ys"],
                                            return_state=True)
        assert icnf.kernel_path(mode_of(pkg, spec)) == path
        assert np.max(np.abs(logp.cpu().numpy() - g["logp"])) < TOL_SOLVE, (name, path)
        assert np.max(np.abs(E.cpu().numpy() - g["E"])) < TOL_SOLVE
        assert np.max(np.abs(n.cpu().numpy() - g["n"])) < TOL_SOLVE
        assert np.max(np.abs(A.cpu().numpy() - g["A"])) < TOL_SOLVE
        assert np.max(np.abs(u1.cpu().numpy() - g["u1"])) < TOL_SOLVE


CASES = [
    # (make_spec kwargs, B, alg, nsteps)
    (dict(nvars=8, hidden=[64, 64, 64]), 1000, 1, 40),                 # headline shape, ragged B
    (dict(nvars=8, hidden=[64, 64, 64]), 1, 0, 40),                    # single column
    (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), 300, 1, 40),
    (dict(nvars=2, hidden=[32, 32]), 1024, 1, 40),
    (dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=2), 130, 0, 40),
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, reg_z=True, reg_j=True, reg_aug=True), 257, 1, 20),
    (dict(nvars=3, hidden=[16, 16], autonomous=True, mode=1, reg_z=True, reg_j=True), 77, 0, 10),
    (dict(nvars=32, hidden=[256, 256, 256]), 48, 0, 10),
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 200, 1, 40),   # cfg4 shape, cooperative kernel
    (dict(nvars=8, hidden=[128, 128, 128]), 100, 1, 40),                            # 3x128 Hutchinson VJP (cooperative)
    (dict(nvars=15, naug=16, hidden=[128, 128], act=2, mode=2), 37, 1, 5),                  # ICNF(nvariables=15) in TestMode: two hidden layers, Q shortcut (fused: last-layer image read from global memory; layer-wise: one Q GEMM)
    # generic zero-padded MFMA instances (csrc/cnf_mfma_generic.hip)
    (dict(nvars=2, naug=3, hidden=[24, 24], act=2, reg_z=True, reg_j=True, reg_aug=True), 333, 1, 20),  # default net, nvariables=2
    (dict(nvars=4, ncond=3, hidden=[48, 48, 48], reg_z=True, reg_j=True), 130, 1, 20),     # conditioned Hutchinson VJP
    (dict(nvars=6, ncond=16, hidden=[64, 64], mode=1, reg_j=True), 90, 0, 20),             # conditioned JVP
    (dict(nvars=10, hidden=[96, 96], act=2, mode=2), 70, 0, 20),                           # exact trace, D=10, softplus
    (dict(nvars=16, hidden=[128, 128, 128], mode=2, autonomous=True), 40, 0, 10),          # exact trace, D=16, 3x128
    # zero-padded cooperative wide-layer instances (csrc/cnf_coop.hip)
    (dict(nvars=20, hidden=[192, 192], act=2, reg_z=True, reg_j=True), 70, 1, 8),          # D=20, 2x192 softplus
    (dict(nvars=6, naug=2, hidden=[256, 256, 256], reg_z=True, reg_j=True, reg_aug=True), 100, 0, 8),  # D=8, 3x256
    (dict(nvars=30, hidden=[128, 128], autonomous=True), 65, 1, 6),                         # D=30, 2x128 autonomous
    # 8 state k-steps (csrc/cnf_mfma_generic_zr8.hip): the reference's default nets for nvariables = 8 .. 15
    (dict(nvars=8, naug=9, hidden=[72, 72], act=2, reg_z=True, reg_j=True, reg_aug=True), 70, 1, 8),     # ICNF(nvariables=8): D=17, H=72 (5 tiles: its own instance since round 4)
    (dict(nvars=12, naug=13, hidden=[104, 104], act=2, reg_z=True, reg_j=True, reg_aug=True), 50, 0, 6),  # ICNF(nvariables=12): D=25, H=104
    (dict(nvars=15, naug=16, hidden=[128, 128], act=2, mode=1, reg_z=True, reg_j=True, reg_aug=True), 40, 1, 5),  # nvariables=15, JVP mode: D=31, H=128
    (dict(nvars=8, naug=9, ncond=4, hidden=[88, 88], act=2, reg_z=True, reg_j=True, reg_aug=True), 45, 1, 6),   # CondICNF default, 4 conditions
    (dict(nvars=8, naug=9, hidden=[72, 72], act=2, mode=2), 33, 0, 5),                      # the same net in TestMode (exact trace, 17 tangents)
    (dict(nvars=20, hidden=[32, 32, 32]), 90, 1, 6),                                        # D=20 with narrow tanh layers
    (dict(nvars=8, hidden=[72, 72, 72]), 60, 0, 6),                                         # 5 hidden tiles -> the 6-tile instance
]
GENERIC_MFMA_CASES = CASES[-15:]
# the cooperative kernel extended to conditions, several probes and the exact trace as unit probes (csrc/cnf_coop_x.hip):
# hidden widths 129 .. 256, which before round 3 ran layer-wise
COOPX_CASES = [
    (dict(nvars=8, ncond=8, hidden=[256, 256, 256], mode=2), 130, 0, 10),                   # cfg5's flow at H = 256: conditioned, exact trace (8 unit probes)
    (dict(nvars=8, ncond=8, hidden=[192, 192, 192], mode=2), 75, 1, 6),                     # ... at H = 192 (12 of 16 tiles), Tsit5
    (dict(nvars=8, ncond=8, hidden=[256, 256, 256], reg_z=True, reg_j=True), 100, 1, 8),    # conditioned RNODE, one probe
    (dict(nvars=12, hidden=[192, 192, 192], nprobes=3, reg_z=True, reg_j=True), 70, 0, 8),  # three Hutchinson probes, wide
    (dict(nvars=20, naug=4, ncond=5, hidden=[160, 160], act=2, nprobes=2, reg_z=True, reg_j=True, reg_aug=True), 45, 1, 6),   # softplus, two layers, augmented, 5 conditions
    (dict(nvars=30, hidden=[256, 224, 256], mode=2, autonomous=True), 33, 0, 4),             # exact trace, D = 30 (30 unit probes), ragged widths, autonomous
    (dict(nvars=9, ncond=3, hidden=[176, 176, 176], mode=1, reg_z=True, reg_j=True), 60, 1, 6),   # Hutchinson JVP (|J eps|), conditioned, tanh (pre-scaled images undone)
    (dict(nvars=24, hidden=[144, 144], act=2, mode=1, nprobes=2, reg_j=True), 40, 0, 5),     # Hutchinson JVP, softplus, two probes, two layers
    # 33 <= D <= 64 (16 state k-steps): the reference's default architecture for nvariables >= 16 (D = 2 nv + 1, H = 4 (D + 1), softplus)
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), 70, 1, 6),   # ICNF(nvariables = 16)
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), 40, 0, 6),   # ICNF(nvariables = 24): D = 49, H = 200
    (dict(nvars=30, naug=31, ncond=6, hidden=[248, 248], act=2, reg_z=True, reg_j=True, reg_aug=True), 35, 1, 4),   # D = 61, H = 248, conditioned: 16 hidden tiles x 16 state k-steps (one workgroup per CU)
    (dict(nvars=40, hidden=[192, 192, 192], reg_z=True, reg_j=True), 50, 0, 6),              # D = 40, tanh, three layers
    (dict(nvars=36, hidden=[128, 128, 128], mode=1, reg_j=True), 40, 1, 4),                  # D = 36, Hutchinson JVP
    # beyond 256 hidden units / 64 state rows: 20 / 24 hidden tiles x 24 state k-steps
    (dict(nvars=32, naug=33, hidden=[264, 264], act=2, reg_z=True, reg_j=True, reg_aug=True), 40, 1, 4),   # ICNF(nvariables = 32): D = 65, H = 264
    (dict(nvars=70, hidden=[352, 352, 352], reg_z=True, reg_j=True), 33, 0, 4),              # tanh, three layers, D = 70, H = 352
    # TestMode of two-hidden-layer flows: tr J = act'_2^T Q act'_1, one H x H product per evaluation
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, mode=2), 70, 1, 6),                   # ICNF(nvariables = 16), TestMode
    (dict(nvars=24, naug=25, ncond=4, hidden=[216, 216], act=2, mode=2), 40, 0, 5),          # conditioned, D = 49
    (dict(nvars=12, hidden=[192, 192], mode=2, autonomous=True), 50, 1, 4),                  # tanh, autonomous
    (dict(nvars=40, naug=41, hidden=[328, 328], act=2, mode=2), 33, 0, 3),                   # ICNF(nvariables = 40): 24 x 24 tiles
    (dict(nvars=6, naug=2, ncond=16, hidden=[200, 200], mode=2, autonomous=True), 40, 0, 4),  # 16 conditions, autonomous, tanh
]
CASES = CASES + COOPX_CASES


@pytest.mark.parametrize("kw,B,alg,nsteps", COOPX_CASES)
def test_wide_conditioned_probe_and_exact_flows_take_the_extended_cooperative_kernel(kw, B, alg, nsteps, pkg, oracles):
    """These shapes resolve to the fused path (kernel_path 2) on their own, a single dynamics call (boundary A) agrees with the
    fp64 oracle, and CNF_MFMA_COOPX=0 still serves them layer-wise (src/core/icnf.jl:297-339,517-559 on
    src/core/base_icnf.jl:272-296 inputs)."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0)
    mode = mode_of(pkg, spec)
    assert icnf.kernel_path(mode) == 2
    p, xs, eps, ys = o64.synth_inputs(spec, B, 99, bias_scale=0.2)
    u = np.concatenate([xs, 0.3 * np.ones((spec.naug, B), np.float32), 0.1 * np.ones((3, B), np.float32)], axis=0).astype(np.float32)
    du = pkg.augmented_f(icnf, mode, dev(u), dev(p), 0.41, dev(eps), dev(ys)).cpu().numpy()
    ref = o64.aug_f(spec, p, u, 0.41, eps, ys)
    assert np.max(np.abs(du - ref) / (1.0 + np.abs(ref))) < TOL_CALL
    os.environ["CNF_MFMA_COOPX"] = "0"
    pkg.reload_tuning()
    try:
        assert make_icnf(pkg, spec, alg, nsteps, path=0).kernel_path(mode) == 3
    finally:
        del os.environ["CNF_MFMA_COOPX"]
        pkg.reload_tuning()


# the cooperative kernel with its tiles dealt exactly over four owner waves (csrc/cnf_coop_d.hip): one-probe VJP solves of two-layer
# softplus flows with 8 .. 15 hidden tiles and D <= 64 - the reference's default architecture for nvariables = 16 .. 29.
# (H, D) chosen to hit every left-over tile count b = HT - 4 A in {0, 1, 2, 3}, every last-k-group length and both state sizes.
COOPD_CASES = [
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), 200, 1, 6),   # ICNF(nvariables = 16): 9 tiles (b = 1), H = 136: 2 k-steps in the last k-group, D = 33: 1
    (dict(nvars=18, naug=19, hidden=[152, 152], act=2, reg_z=True, reg_j=True, reg_aug=True), 77, 0, 6),    # nvariables = 18: 10 tiles (b = 2), RK4, ragged batch
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2), 130, 1, 5),                                          # nvariables = 20 as FFJORD: 11 tiles (b = 3), no regularisers
    (dict(nvars=40, hidden=[128, 128], act=2, reg_j=True), 64, 0, 5),                                         # 8 tiles (b = 0), D = 40 with H = 128: full last k-groups
    (dict(nvars=22, naug=23, hidden=[184, 184], act=2, reg_z=True, reg_j=True, reg_aug=True), 100, 1, 4),   # nvariables = 22: 12 tiles: A = 3, b = 0, D = 45
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), 90, 0, 5),    # nvariables = 24: 13 tiles (A = 3, b = 1), D = 49: 16 state registers
    (dict(nvars=29, naug=30, hidden=[240, 240], act=2, reg_z=True, reg_aug=True, autonomous=True), 70, 1, 4),   # nvariables = 29: 15 tiles (b = 3), D = 59, no time column
    (dict(nvars=35, hidden=[150, 230], act=2, reg_z=True, reg_j=True), 45, 1, 4),                             # unequal widths (padded to the widest), D = 35
    (dict(nvars=30, naug=3, hidden=[132, 132], act=2, reg_j=True), 1, 0, 3),                                  # one column
    # TestMode (exact trace) of the same flows: tr J = act'_2^T Q act'_1, one more H x H product instead of the pullback
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, mode=2), 150, 1, 5),                                   # ICNF(nvariables = 16), TestMode
    (dict(nvars=19, naug=20, hidden=[160, 160], act=2, mode=2), 70, 0, 5),                                    # 10 tiles (b = 2), RK4
    (dict(nvars=22, naug=23, hidden=[184, 184], act=2, mode=2, autonomous=True), 90, 1, 4),                   # 12 tiles: A = 3; no time column
    (dict(nvars=40, hidden=[176, 176], act=2, mode=2), 33, 0, 4),                                             # 11 tiles (b = 3), D = 40
    # other flows of these sizes: tanh (pre-scaled forward images), three hidden layers, D <= 32 (cooperative plans whose width
    # the cooperative kernel would pad to a whole quad of tiles)
    (dict(nvars=20, hidden=[200, 200, 200], reg_z=True, reg_j=True), 130, 1, 4),                              # 13 tiles, three tanh layers (two exchange buffers)
    (dict(nvars=10, hidden=[176, 176]), 77, 0, 5),                                                            # 11 tiles, two tanh layers
    (dict(nvars=30, naug=2, hidden=[144, 144, 144], reg_aug=True), 64, 1, 3),                                 # 9 tiles, D = 32 exactly
    (dict(nvars=12, hidden=[160, 160, 150], act=2, reg_j=True), 50, 0, 4),                                    # softplus, three layers, unequal widths
    (dict(nvars=40, hidden=[232, 232]), 45, 1, 4),                                                            # tanh, D = 40, 15 tiles (an extended-kernel plan)
    # 16 .. 24 hidden tiles: the 32-sample form (csrc/cnf_coop_d2.hip) - the default architecture at nvariables = 30 .. 47
    (dict(nvars=30, naug=31, hidden=[248, 248], act=2, reg_z=True, reg_j=True, reg_aug=True), 100, 1, 4),    # ICNF(nvariables = 30): 16 tiles (A = 4, b = 0), D = 61
    (dict(nvars=32, naug=33, hidden=[264, 264], act=2, reg_z=True, reg_j=True, reg_aug=True), 70, 0, 4),     # nvariables = 32: 17 tiles (one left-over tile: 2 units on waves 0, 1), D = 65
    (dict(nvars=36, naug=37, hidden=[296, 296], act=2, reg_z=True, reg_j=True, reg_aug=True), 45, 1, 3),     # 19 tiles (b = 3: six left-over units), D = 73
    (dict(nvars=43, naug=44, hidden=[352, 352], act=2, reg_j=True), 40, 0, 3),                                # nvariables = 43 (MiniBooNE's dimension): 22 tiles, D = 87: c per evaluation
    (dict(nvars=47, naug=48, hidden=[384, 384], act=2, reg_z=True, reg_j=True, reg_aug=True), 33, 1, 2),     # the largest: 24 x 24 tiles
    (dict(nvars=30, naug=31, ncond=6, hidden=[248, 248], act=2, reg_z=True, reg_j=True, reg_aug=True), 50, 1, 3),   # conditioned, 16 tiles (wider conditioned flows have no fused plan)
    (dict(nvars=48, hidden=[264, 264], reg_z=True, reg_j=True), 60, 1, 3),                                    # tanh on the 32-sample form: 17 tiles, D = 48
    (dict(nvars=30, naug=31, hidden=[248, 248], act=2, mode=2), 70, 1, 3),                                    # TestMode on the 32-sample form: ICNF(nvariables = 30), 16 tiles
    (dict(nvars=33, naug=34, hidden=[272, 272], act=2, mode=2), 50, 0, 3),                                    # 17 tiles, D = 67: 20 state registers, RK4
    (dict(nvars=38, naug=39, hidden=[312, 312], act=2, mode=2, autonomous=True), 40, 1, 2),                   # 20 tiles (A = 5), D = 77, no time column
    (dict(nvars=30, naug=31, ncond=4, hidden=[248, 248], act=2, mode=2), 33, 1, 2),                           # conditioned TestMode, 16 tiles
    (dict(nvars=42, naug=43, hidden=[344, 344], act=2, mode=2), 40, 1, 2),                                    # TestMode at 22 tiles, D = 85 (24 state registers): pre-activations parked in LDS
    (dict(nvars=60, naug=20, hidden=[340, 340], reg_aug=True), 40, 0, 3),                                     # tanh, 22 tiles, D = 80
    # conditioned flows (CondICNF: the condition rows of layer 1, src/layers/cond_layer.jl:7-31, src/core/base_icnf.jl:272-296)
    (dict(nvars=16, naug=17, ncond=5, hidden=[156, 156], act=2, reg_z=True, reg_j=True, reg_aug=True), 120, 1, 5),   # default architecture with 5 conditions: 10 tiles
    (dict(nvars=20, naug=21, ncond=16, hidden=[232, 232], act=2, autonomous=True), 64, 0, 4),                # 16 conditions, D = 41, 15 tiles, autonomous
    (dict(nvars=18, naug=19, ncond=3, hidden=[164, 164], act=2, mode=2), 70, 1, 4),                          # TestMode, 3 conditions
]


@pytest.mark.parametrize("kw,B,alg,nsteps", COOPD_CASES)
def test_dealt_cooperative_kernel_matches_the_oracles(kw, B, alg, nsteps, pkg, oracles, monkeypatch):
    """csrc/cnf_coop_d.hip (forced at these batch sizes with CNF_COOPD=2; on its own the 64-sample form takes over above 4096 columns, the 32-sample form serves every size) is the
    same augmented_f / solve (src/core/icnf.jl:517-559, src/core/base_icnf.jl:158-172): whole solves against the C restatement,
    logp / regularisers / final state; a single dynamics call against the fp64 oracle; and against the extended kernel
    (CNF_COOPD=0) on the same inputs - the two differ by summation order only."""
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    mode = mode_of(pkg, spec)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 500 + B, bias_scale=0.2)
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, nthreads=8)
    out = {}
    for tag, env in (("dealt", "2"), ("extended", "0")):
        setsw(pkg, monkeypatch, "CNF_COOPD", env)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2)
        assert icnf.kernel_family(mode, B=B) == ("coopd" if tag == "dealt" else icnf.kernel_family(mode))   # (else the plan's own: coopx / coop)
        assert icnf.kernel_family(mode) in ("coopx", "coop")
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        out[tag] = (logp.cpu().numpy(), [r.cpu().numpy() for r in regs], u1.cpu().numpy())
        assert np.max(np.abs(out[tag][0] - ref[0])) < TOL_SOLVE, tag
        for a_, b_ in zip(out[tag][1], ref[1]):
            assert np.max(np.abs(a_ - b_)) < TOL_SOLVE, tag
        u = np.concatenate([xs, 0.3 * np.ones((spec.naug, B), np.float32), 0.1 * np.ones((3, B), np.float32)], axis=0).astype(np.float32)
        du = pkg.augmented_f(icnf, mode, dev(u), dev(p), 0.37, dev(eps), dev(ys)).cpu().numpy()
        assert np.max(np.abs(du - o64.aug_f(spec, p, u, 0.37, eps, ys)) / (1.0 + np.abs(o64.aug_f(spec, p, u, 0.37, eps, ys)))) < TOL_CALL, tag
        # generate: the reversed solve from a given state (cnf_integrate_fixed: u0 in, u1 out)
        gen = (pkg.generate(icnf, mode, *(((dev(ys),) if spec.ncond else ()) + (dev(p), {}, B)), z0=dev(out[tag][2][:spec.D]), eps=dev(eps))
               if spec.mode == 0 else dev(xs))
        out[tag] += (gen.cpu().numpy(),)
    tol = 5e-5 + 2e-6 * float(np.abs(out["extended"][0]).max())       # (|logp| ~ 150 at D = 73 has a Float32 ulp of 1.5e-5)
    assert np.max(np.abs(out["dealt"][0] - out["extended"][0])) < tol
    assert np.max(np.abs(out["dealt"][2] - out["extended"][2])) < tol
    assert np.max(np.abs(out["dealt"][3] - out["extended"][3])) < 5e-5


def test_dealt_cooperative_kernel_at_full_size(pkg, oracles, monkeypatch):
    """ICNF(nvariables = 16) and (nvariables = 24) with every default (two softplus layers of 4 (D + 1), lambdas 0.01), B = 32 768,
    Tsit5 x 40: the dealt kernel serves it on its own, every column agrees with the extended kernel (summation order only) and a
    sample of columns with the C restatement; column shards concatenate bit-identically within the kernel."""
    o64, oc = oracles
    delsw(pkg, monkeypatch, "CNF_COOPD")
    # (nvariables = 32, 40: the 32-sample form with its Runge-Kutta sums in the plan's global ring - every workgroup walks four
    # super-tiles through the same ring slice; 40 on 16 384 columns to bound the C restatement's time)
    for nv, H in ((16, 136), (24, 200), (32, 264), (40, 328)):
        spec = o64.make_spec(nvars=nv, naug=nv + 1, hidden=[H, H], act=2, reg_z=True, reg_j=True, reg_aug=True)
        B = 32768 if nv < 40 else 16384
        p, xs, eps, _ = o64.synth_inputs(spec, B, 7 + nv, bias_scale=0.1)
        mode = mode_of(pkg, spec)
        icnf = make_icnf(pkg, spec, 1, 40)
        # (the 4096-column threshold belongs to the 64-sample form: the 32-sample one serves every batch size)
        assert icnf.kernel_family(mode) == "coopx" and icnf.kernel_family(mode, B=B) == "coopd" and icnf.kernel_family(mode, B=4096) == ("coopx" if nv < 30 else "coopd")
        logp, regs = run_inference(pkg, icnf, spec, p, xs, eps, None)
        logp = logp.cpu().numpy()
        setsw(pkg, monkeypatch, "CNF_COOPD", "0")
        lx, rx = run_inference(pkg, icnf, spec, p, xs, eps, None)
        delsw(pkg, monkeypatch, "CNF_COOPD")
        tol = 1e-4 + 2e-6 * float(np.abs(logp).max())      # (|logp| ~ 200 at D = 81 has a Float32 ulp of 1.5e-5)
        assert np.max(np.abs(logp - lx.cpu().numpy())) < tol
        for a_, b_ in zip(regs, rx):
            assert np.max(np.abs(a_.cpu().numpy() - b_.cpu().numpy())) < tol
        idx = np.arange(0, B, 257 if nv < 32 else 1031)
        ref = oc.inference_fixed(spec, p, xs[:, idx], 0.0, 1.0, 40, 1, eps[:, idx], None, nthreads=8)
        assert np.max(np.abs(logp[idx] - ref[0])) < max(TOL_SOLVE, tol)
        # two shards, both above the 4096-column threshold: the same kernel, the same bits
        cut = (12288 if nv < 40 else 6144) + 64 * 3 + 5
        la = run_inference(pkg, icnf, spec, p, xs[:, :cut], eps[:, :cut], None)[0].cpu().numpy()
        lb = run_inference(pkg, icnf, spec, p, xs[:, cut:], eps[:, cut:], None)[0].cpu().numpy()
        assert np.array_equal(np.concatenate([la, lb]), logp)


@pytest.mark.parametrize("solver", ["tsit5", "vcabm"])
def test_adaptive_solvers_on_the_dealt_kernel(solver, pkg, oracles, monkeypatch):
    """The reference's default configuration above 4096 columns: ICNF(nvariables = 16) under the adaptive solvers (default VCABM,
    or Tsit5, at 1e-4).  Every attempt is a single dynamics call (boundary A) of 5000 columns - on the dealt kernel; the loss, the
    accepted / rejected counts and the training step on the frozen grid agree with the same solves on the extended kernel
    (CNF_COOPD=0), which differ from it by summation order only."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True)
    B = 5000
    p, xs, eps, _ = o64.synth_inputs(spec, B, 1234, bias_scale=0.2)
    out = {}
    for tag, env in (("dealt", "1"), ("extended", "0")):
        setsw(pkg, monkeypatch, "CNF_COOPD", env)
        icnf = make_icnf(pkg, spec, 1, 1, lambdas=(0.01, 0.01, 0.01))
        icnf.sol_kwargs = dict(alg=pkg.VCABM() if solver == "vcabm" else pkg.Tsit5(), reltol=1e-4, abstol=1e-4)
        mode = pkg.TrainMode(True)
        assert icnf.kernel_family(mode, B=B, whole_solve=False) == ("coopd" if tag == "dealt" else "coopx")
        val = float(pkg.loss(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps)))
        stats = dict(icnf.last_solve_stats)
        gval, g = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps))
        out[tag] = (val, stats.get("naccept"), stats.get("nreject"), float(gval), g.cpu().numpy())
    assert abs(out["dealt"][0] - out["extended"][0]) < 1e-4
    assert out["dealt"][1:3] == out["extended"][1:3], (out["dealt"][1:3], out["extended"][1:3])
    assert abs(out["dealt"][3] - out["extended"][3]) < 1e-4
    assert np.max(np.abs(out["dealt"][4] - out["extended"][4])) < 1e-4 * np.abs(out["extended"][4]).max() + 1e-6


@pytest.mark.parametrize("kw,lam,B,alg,nsteps,grid", [
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 150, 1, 2, False),
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 70, 0, 3, False),
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 80, 1, 2, False),
    (dict(nvars=18, naug=19, hidden=[152, 152], act=2), (0.0, 0.0, 0.0), 100, 1, 3, True),      # on a non-uniform grid (device-resident step times)
    (dict(nvars=16, naug=17, ncond=5, hidden=[156, 156], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 90, 1, 2, False),   # conditioned
    (dict(nvars=20, hidden=[200, 200, 200], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 70, 0, 2, False),      # a cooperative (tanh, 3-layer) plan's forward on the dealt kernel
    (dict(nvars=14, naug=6, hidden=[152, 152], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 90, 1, 2, False),     # tanh, two layers, 10 tiles: the dealt sweep keeps h_1 too
    (dict(nvars=30, naug=11, hidden=[172, 172], reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.02, 0.03), 50, 0, 3, False),   # tanh, 11 tiles, 12 state registers
    (dict(nvars=32, naug=33, hidden=[264, 264], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 60, 1, 2, False),   # the 32-sample form (17 tiles)
    (dict(nvars=40, naug=41, hidden=[328, 328], act=2), (0.0, 0.0, 0.0), 40, 0, 2, True),                        # ... 21 tiles, on a grid
])
def test_cooperative_gradient_on_the_dealt_forward_solve(kw, lam, B, alg, nsteps, grid, pkg, oracles, monkeypatch):
    """The checkpointing forward half of the cooperative gradient on the dealt kernel (z per step, zdot and g = eps^T J per stage in
    the tile layout and stride the reverse sweep reads): loss, dloss/dps, dloss/dxs against fp64 autograd (CNF_COOPD=2 forces it
    at this batch size; src/core/icnf.jl:90-99)."""
    o64, _ = oracles
    setsw(pkg, monkeypatch, "CNF_COOPD", "2")
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 77 + B, bias_scale=0.2)
    mode = mode_of(pkg, spec)
    icnf = make_icnf(pkg, spec, alg, nsteps, lambdas=lam)
    tg = None
    if grid:
        tg = np.cumsum([0.0, 0.13, 0.31, 0.2, 0.36] if B > 50 else [0.0, 0.4, 0.6]).astype(np.float32)
        tg[-1] = 1.0
        nsteps = len(tg) - 1
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True, tgrid=None if tg is None else [float(v) for v in tg])
    if grid:
        import ctypes as C
        h = icnf._handle(mode)
        icnf._bind_params(h, dev(p))
        P = dev(p); icnf._bind_params(h, P)
        x, e = dev(xs).t().contiguous(), dev(eps).t().contiguous()
        g = torch.empty(P.numel(), device="cuda"); gx = torch.zeros(B, spec.nvars, device="cuda"); sums = torch.empty(4, device="cuda")
        pkg._lib.check(h.lib.cnf_loss_grad_grid(h.ptr, alg, nsteps, (C.c_float * len(tg))(*tg), x.data_ptr(), e.data_ptr(), None, B,
                                                (C.c_float * 3)(*lam), g.data_ptr(), gx.data_ptr(), sums.data_ptr(), None))
        torch.cuda.synchronize()
        assert int(h.lib.cnf_grad_path_for(h.ptr, B, alg, 1)) == 3
        val, g, gx = float(sums[0]) / B, g / B, (gx / B).t()
    else:
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
        val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
        assert icnf.grad_path(mode, B=B, alg=alg) == 3
    assert abs(float(val) - L) < 1e-4 + 2e-6 * abs(L)
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * np.abs(gref).max() + 1e-6
    assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7


@pytest.mark.parametrize("nv,alg", [(12, 1), (16, 1), (18, 0), (22, 1), (29, 1), (32, 1), (38, 0), (43, 1)])
def test_dealt_reverse_sweep_matches_the_cooperative_sweep(nv, alg, pkg, oracles, monkeypatch):
    """The dealt form of the reverse sweep (cnf_coop_dgrad.hip) against the sweep of cnf_coop_grad.hip on the same plan, the same
    checkpoints and the same operand arrays (CNF_COOPD_GRAD=0 selects the latter): default architecture at 7 (through the auxiliary
    cooperative plan: one shared tile per wave + six left-over units), 9, 10, 12 and 15 hidden
    tiles (left-over units 1, 2, 0, 3 of a wave's two slots) on 32-sample super-tiles, 17, 20 and 22 tiles on 16-sample ones (one, no,
    two left-over tiles), 3000 columns (ragged last super-tile), default lambdas.  The two differ by
    summation order only - and they do differ in the last bits, which is how the test knows both ran."""
    o64, _ = oracles
    D = 2 * nv + 1
    spec = o64.make_spec(nvars=nv, naug=nv + 1, hidden=[4 * (D + 1)] * 2, act=2, reg_z=True, reg_j=True, reg_aug=True)
    B = 3000 if nv >= 16 else 4500     # (7 .. 8 tiles take the cooperative route from 4096 columns on)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 99 + nv, bias_scale=0.2)
    out = {}
    for tag, env in (("dealt", "1"), ("coop", "0")):
        setsw(pkg, monkeypatch, "CNF_COOPD_GRAD", env)
        icnf = make_icnf(pkg, spec, alg, 3, lambdas=(0.01, 0.01, 0.01))
        mode = pkg.TrainMode(True)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert icnf.grad_path(mode, B=B, alg=alg) == 3
        out[tag] = (float(val), g.cpu().numpy(), gx.cpu().numpy())
    assert out["dealt"][0] == out["coop"][0]                      # the loss comes from the forward solve both share
    for k in (1, 2):
        a, b = out["dealt"][k], out["coop"][k]
        assert np.max(np.abs(a - b)) < 2e-5 * np.abs(b).max() + 1e-7
    assert not np.array_equal(out["dealt"][1], out["coop"][1])


@pytest.mark.parametrize("kw,B,alg,nsteps", GENERIC_MFMA_CASES)
def test_generic_instances_resolve_to_the_mfma_path(kw, B, alg, nsteps, pkg, oracles):
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    assert 2 in paths_for(pkg, spec, alg, nsteps)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0)          # AUTO must pick it too
    assert icnf.kernel_path(mode_of(pkg, spec)) == 2


@pytest.mark.parametrize("kw,B,alg,nsteps", CASES)
def test_inference_matches_c_restatement(kw, B, alg, nsteps, pkg, oracles):
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 20240612, bias_scale=0.1)
    ref_logp, ref_regs, ref_u = oc.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, nthreads=4)
    for path in paths_for(pkg, spec, alg, nsteps):
        icnf = make_icnf(pkg, spec, alg, nsteps, path)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        assert np.max(np.abs(logp.cpu().numpy() - ref_logp)) < TOL_SOLVE, (kw, path)
        for a, b in zip(regs, ref_regs):
            assert np.max(np.abs(a.cpu().numpy() - b)) < TOL_SOLVE
        assert np.max(np.abs(u1.cpu().numpy() - ref_u)) < TOL_SOLVE


def test_extended_cooperative_kernel_at_full_size(pkg, oracles, monkeypatch):
    """BASELINE cfg5's flow (CondFFJORD D = 8 + 8 conditions, exact trace, RK4 x 40, B = 16 384) at hidden width 256, where it
    runs on the extended cooperative kernel: every column against the layer-wise path (an independent implementation), 128
    columns against the C restatement, finiteness."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=8, ncond=8, hidden=[256, 256, 256], mode=2)
    B = 16384
    p, xs, eps, ys = o64.synth_inputs(spec, B, 20240616)
    icnf = make_icnf(pkg, spec, 0, 40, path=0)
    assert icnf.kernel_path(mode_of(pkg, spec)) == 2
    lp = run_inference(pkg, icnf, spec, p, xs, eps, ys)[0].cpu().numpy()
    assert np.all(np.isfinite(lp))
    setsw(pkg, monkeypatch, "CNF_MFMA_COOPX", "0")
    lay = make_icnf(pkg, spec, 0, 40, path=0)
    assert lay.kernel_path(mode_of(pkg, spec)) == 3
    lp2 = run_inference(pkg, lay, spec, p, xs, eps, ys)[0].cpu().numpy()
    assert np.max(np.abs(lp - lp2)) < 5e-5
    idx = np.random.default_rng(1).choice(B, 128, replace=False)
    ref = oc.inference_fixed(spec, p, xs[:, idx], 0.0, 1.0, 40, 0, eps[:, idx], ys[:, idx], nthreads=8)[0]
    assert np.max(np.abs(lp[idx] - ref)) < TOL_SOLVE


def test_cooperative_wide_layer_kernel_matches_per_wave_kernel(pkg, oracles, monkeypatch):
    """The workgroup-cooperative kernel (csrc/cnf_coop.hip, used when the operand images do not
    fit LDS) forced onto the headline shape must reproduce the per-wave kernel and the oracle."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    B = 1000                                        # ragged: 15 full super-tiles + 40 columns
    p, xs, eps, _ = o64.synth_inputs(spec, B, 77, bias_scale=0.1)
    base = run_inference(pkg, make_icnf(pkg, spec, 1, 40, path=2), spec, p, xs, eps, None, return_state=True)
    setsw(pkg, monkeypatch, "CNF_MFMA_COOP", "1")
    coop = run_inference(pkg, make_icnf(pkg, spec, 1, 40, path=2), spec, p, xs, eps, None, return_state=True)
    delsw(pkg, monkeypatch, "CNF_MFMA_COOP")
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 40, 1, eps, nthreads=4)
    assert np.max(np.abs(coop[0].cpu().numpy() - ref[0])) < TOL_SOLVE
    assert np.max(np.abs(coop[0].cpu().numpy() - base[0].cpu().numpy())) < 2e-5
    for a, b in zip(coop[1], base[1]):
        assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 2e-5
    assert np.max(np.abs(coop[2].cpu().numpy() - ref[2])) < TOL_SOLVE


@pytest.mark.parametrize("kw,alg", [
    (dict(nvars=8, hidden=[64, 64, 64]), 1),                                  # cfg2' shape: the per-wave instance's hoisting form (PRE = 2)
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 0),          # RNODE, one probe, RK4
    (dict(nvars=5, naug=2, hidden=[64, 64], act=2, reg_z=True, reg_aug=True), 1),   # softplus, two hidden layers, augmented
    (dict(nvars=11, hidden=[50, 64, 40]), 1),                                 # generic zero-padded instance (4 state k-steps, ragged widths)
])
def test_tile_split_kernel_for_small_batches(kw, alg, pkg, oracles):
    """Batches of at most one 16-sample tile per compute unit run on the tile-split form (cnf_coop.hip with one sample tile per
    workgroup: the hidden width over the four SIMDs of a CU, images in LDS) instead of one wave per tile.  It is the same
    augmented_f / solve (src/core/icnf.jl:517-559): checked against the C restatement, against the per-wave kernel on the
    same inputs (CNF_TILE_SPLIT=0), for ragged batches, the final state, and generate."""
    import os
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    nsteps = 12
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2)
    mode = mode_of(pkg, spec)
    old = os.environ.get("CNF_TILE_SPLIT")
    try:
        for B in (1, 16, 45, 1000, 4096):
            p, xs, eps, ys = o64.synth_inputs(spec, B, 300 + B, bias_scale=0.2)
            ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, nthreads=8)
            out = {}
            for tag, env in (("wave", "0"), ("split", "2")):
                os.environ["CNF_TILE_SPLIT"] = env
                pkg.reload_tuning()
                logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
                assert icnf.kernel_family(mode, B=B) == ("per_wave" if tag == "wave" else "tile_split")
                assert icnf.kernel_family(mode, B=B, whole_solve=False) == "per_wave"   # single dynamics calls never split
                out[tag] = (logp.cpu().numpy(), [r.cpu().numpy() for r in regs], u1.cpu().numpy())
                assert np.max(np.abs(out[tag][0] - ref[0])) < TOL_SOLVE, (tag, B)
                for a_, b_ in zip(out[tag][1], ref[1]):
                    assert np.max(np.abs(a_ - b_)) < TOL_SOLVE, (tag, B)
            assert np.max(np.abs(out["wave"][0] - out["split"][0])) < 2e-5
            assert np.max(np.abs(out["wave"][2] - out["split"][2])) < 2e-5
        # generate (the reversed solve) on the split form inverts the forward solve
        os.environ["CNF_TILE_SPLIT"] = "2"
        pkg.reload_tuning()
        if not spec.ncond:
            B = 333
            p, xs, eps, ys = o64.synth_inputs(spec, B, 77, bias_scale=0.2)
            _, _, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
            back = pkg.generate(icnf, mode, dev(p), {}, B, z0=u1[:spec.D].contiguous(), eps=dev(eps))
            assert np.max(np.abs(back.cpu().numpy() - xs)) < 5e-3          # 12 steps of RK4 / Tsit5 there and back
    finally:
        if old is None:
            os.environ.pop("CNF_TILE_SPLIT", None)
            pkg.reload_tuning()
        else:
            os.environ["CNF_TILE_SPLIT"] = old
            pkg.reload_tuning()


@pytest.mark.parametrize("alg,nsteps,kw", [
    (0, 7, dict(nvars=8, hidden=[64, 64, 64])),                           # cfg2's flow, RK4
    (1, 5, dict(nvars=8, hidden=[64, 64, 64])),                           # cfg2', Tsit5
    (1, 4, dict(nvars=6, naug=2, hidden=[64, 64, 64], reg_z=True, reg_j=True, reg_aug=True)),   # regularised, augmented
    (0, 3, dict(nvars=8, hidden=[64, 64, 64], autonomous=True)),
    (1, 6, dict(nvars=2, hidden=[32, 32])),                               # cfg1's flow: the (2 tiles, 2 layers, D <= 4) instance
    (0, 5, dict(nvars=3, naug=1, hidden=[32, 32], reg_z=True, reg_j=True, reg_aug=True)),
])
def test_hand_scheduled_solve_is_bit_identical_to_the_per_wave_kernel(alg, nsteps, kw, pkg, oracles, monkeypatch):
    """csrc/cnf_mfma2.hip (round 5) is mfma_solve_kernel's one-probe VJP solve with its instruction order laid out by hand: same
    fragments, same products in the same order, same elementwise expressions - so the same bits, at one and at two waves per SIMD,
    for the outputs, the final state and (through the checkpoints of its forward pass) the parameter gradient."""
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    B = 5003                                                             # above the tile-split threshold, ragged last tile
    p, xs, eps, ys = o64.synth_inputs(spec, B, 41, bias_scale=0.2)
    lam = (0.02, 0.03, 0.01)
    out = {}
    setsw(pkg, monkeypatch, "CNF_SOLVE2_PAIR", "0")                      # (the two-waves-per-tile form has its own test below)
    for tag, sw in (("old", "0"), ("one", "1"), ("two", "2")):
        setsw(pkg, monkeypatch, "CNF_SOLVE2", sw)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
        mode = mode_of(pkg, spec)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert ("mfma_solve2" in icnf.kernel_name(mode)) == (sw != "0"), icnf.kernel_name(mode)
        out[tag] = (logp, regs, u1, val, g, gx)
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, nthreads=8)
    assert np.max(np.abs(out["one"][0].cpu().numpy() - ref[0])) < TOL_SOLVE
    for tag in ("one", "two"):
        assert torch.equal(out[tag][0], out["old"][0]) and torch.equal(out[tag][2], out["old"][2]), tag
        for a_, b_ in zip(out[tag][1], out["old"][1]):
            assert torch.equal(a_, b_), tag
        assert float(out[tag][3]) == float(out["old"][3]) and torch.equal(out[tag][4], out["old"][4]) and torch.equal(out[tag][5], out["old"][5]), tag


@pytest.mark.parametrize("alg,nsteps,kw,B", [
    (1, 6, dict(nvars=2, hidden=[32, 32]), 1024),                                                  # cfg1's flow at its own batch
    (0, 5, dict(nvars=3, naug=1, hidden=[32, 32], reg_z=True, reg_j=True, reg_aug=True), 1000),    # regularised, augmented, ragged tile
    (1, 3, dict(nvars=8, hidden=[32, 32], autonomous=True), 8192),                                 # D = 8 (the zero-padded layout); two tiles per CU: the largest batch it takes
    (1, 4, dict(nvars=2, hidden=[32, 32], reg_z=True), 7),                                         # less than a tile
    (1, 4, dict(nvars=3, naug=4, hidden=[32, 32], act=2, reg_z=True, reg_j=True, reg_aug=True), 2048),   # softplus: the default architecture at nvariables = 3
])
def test_two_waves_per_tile_solve_is_bit_identical_to_the_per_wave_kernel(alg, nsteps, kw, B, pkg, oracles, monkeypatch):
    """mfma_solve2p_kernel (round 5): small batches of two-tile nets run the forward chain and the pullback of a tile on two waves, one
    stage apart - the same products and expressions as mfma_solve2_kernel / mfma_solve_kernel, so the same bits for the outputs, the
    final state and (through the checkpoints its forward wave writes) the parameter gradient; beyond two tiles per CU the per-wave
    form takes over."""
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 43, bias_scale=0.2)
    lam = (0.02, 0.03, 0.01)
    out = {}
    for tag, sw in (("pair", "1"), ("one", "0")):
        setsw(pkg, monkeypatch, "CNF_SOLVE2_PAIR", sw)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
        mode = mode_of(pkg, spec)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert ("mfma_solve2p" in icnf.kernel_name(mode)) == (sw == "1"), icnf.kernel_name(mode)
        out[tag] = (logp, regs, u1, val, g, gx)
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, nthreads=8)
    assert np.max(np.abs(out["pair"][0].cpu().numpy() - ref[0])) < TOL_SOLVE
    assert torch.equal(out["pair"][0], out["one"][0]) and torch.equal(out["pair"][2], out["one"][2])
    for a_, b_ in zip(out["pair"][1], out["one"][1]):
        assert torch.equal(a_, b_)
    assert float(out["pair"][3]) == float(out["one"][3]) and torch.equal(out["pair"][4], out["one"][4]) and torch.equal(out["pair"][5], out["one"][5])
    # a batch of more than two tiles per CU stays on the per-wave form and still agrees with it
    if B == 8192:
        setsw(pkg, monkeypatch, "CNF_SOLVE2_PAIR", "1")
        p2, xs2, eps2, ys2 = o64.synth_inputs(spec, 8192 + 16, 44, bias_scale=0.2)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
        a1 = run_inference(pkg, icnf, spec, p2, xs2, eps2, ys2)[0]
        setsw(pkg, monkeypatch, "CNF_SOLVE2_PAIR", "0")
        icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
        a0 = run_inference(pkg, icnf, spec, p2, xs2, eps2, ys2)[0]
        assert torch.equal(a1, a0)


def make_icnf_bf16x6(pkg, spec, alg, nsteps):
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2)
    icnf.compute_mode.arith = pkg._lib.ARITH_BF16X6
    return icnf


@pytest.mark.parametrize("name", ["cfg2_ffjord_d8_3x64_rk4", "cfg2p_ffjord_d8_3x64_tsit5",
                                  "cfg3_rnode_d8_3x64_tsit5_k4"])
def test_split_bf16_arithmetic_matches_golden(name, pkg):
    """Opt-in CNF_ARITH_BF16X6: hidden products as six bf16 MFMAs on an exact 3-way split of both
    operands.  Must meet the same tolerances as the exact-f32 kernel against the fp64 fixtures."""
    spec, meta, g = load_golden(name)
    icnf = make_icnf_bf16x6(pkg, spec, meta["alg"], meta["nsteps"])
    du = pkg.augmented_f(icnf, mode_of(pkg, spec), dev(g["u"]), dev(g["p"]), float(g["t"]),
                         dev(g["eps"]), None).cpu().numpy()
    assert np.max(np.abs(du - g["du"]) / (1.0 + np.abs(g["du"]))) < TOL_CALL
    logp, (E, n, A), u1 = run_inference(pkg, icnf, spec, g["p"], g["xs"], g["eps"], None, return_state=True)
    assert np.max(np.abs(logp.cpu().numpy() - g["logp"])) < TOL_SOLVE
    assert np.max(np.abs(n.cpu().numpy() - g["n"])) < TOL_SOLVE
    assert np.max(np.abs(u1.cpu().numpy() - g["u1"])) < TOL_SOLVE


def test_split_bf16_is_as_accurate_as_f32_on_stiffer_weights(pkg, oracles):
    """A 2-way split (3 products) loses 2 decimal digits on stiffer nets; the 3-way
    split must stay within a small factor of the exact-f32 kernel's own error against float64."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 512
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240614)
    p = (p * 2.5).astype(np.float32)
    ref = o64.inference_fixed(spec, p, xs[:, :64], 0.0, 1.0, 40, 1, eps[:, :64])[0]
    f32 = run_inference(pkg, make_icnf(pkg, spec, 1, 40, path=2), spec, p, xs, eps, None)[0].cpu().numpy()
    b16 = run_inference(pkg, make_icnf_bf16x6(pkg, spec, 1, 40), spec, p, xs, eps, None)[0].cpu().numpy()
    e32 = np.max(np.abs(f32[:64] - ref))
    e16 = np.max(np.abs(b16[:64] - ref))
    assert e16 < max(3.0 * e32, 2e-4), (e16, e32)
    assert np.max(np.abs(b16 - f32)) < 1e-3


def _forward_error_bound(spec, p, u, t, c_hidden, c_edge, c_act):
    """First-order, entry-wise bound on the error of zdot = MLP([z; t]) evaluated in finite precision, computed in
    float64: every product W_l h_{l-1} carries an error <= c_l (|W_l| |h_{l-1}|), every activation value a relative
    error <= c_act, and an error e in a layer's input reaches its output as <= |W_l| e (|tanh'| <= 1).  Returns
    (zdot64, bound), both (D, B)."""
    w_off, b_off, _ = spec.param_offsets()
    D, B = spec.D, u.shape[1]
    h = np.vstack([u[:D].astype(np.float64), np.full((1, B), float(t))])
    err = np.zeros_like(h)
    N = len(spec.acts)
    for l in range(N):
        fin, fout = spec.widths[l], spec.widths[l + 1]
        W = np.asarray(p[w_off[l]:w_off[l] + fin * fout], dtype=np.float64).reshape(fin, fout).T
        b = np.asarray(p[b_off[l]:b_off[l] + fout], dtype=np.float64)[:, None]
        pre = W @ h + b
        c = c_hidden if 0 < l < N - 1 else c_edge
        e_pre = np.abs(W) @ err + c * (np.abs(W) @ np.abs(h) + np.abs(b))
        if spec.acts[l] == 1:
            h = np.tanh(pre)
            err = e_pre + c_act * np.abs(h)          # |tanh'| <= 1
        else:
            h, err = pre, e_pre
    return h, err


def test_loss_mean_equals_the_mean_of_the_partial_sums(pkg, oracles):
    """cnf_loss_mean (the scalar loss of an unsharded batch out of the two reduction kernels) against cnf_loss_sums + the
    float64 combination a sharded caller performs after its all-reduce: the same bits; and against a float64 host mean."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    B = 777
    p, xs, eps, _ = o64.synth_inputs(spec, B, 31, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 1, 10)
    icnf.lambda1, icnf.lambda2 = 0.02, 0.03
    m = pkg.TrainMode(True)
    logp, regs = pkg.inference(icnf, m, dev(xs), dev(p), {}, eps=dev(eps), _raw=True)
    a = pkg.loss_mean(icnf, m, logp, regs)
    sums = pkg.loss_sums(icnf, m, logp, regs)
    b = pkg.reduce_loss(sums, B, (icnf.lambda1, icnf.lambda2, icnf.lambda3), group=False)
    assert a.shape == () and torch.equal(a, b)
    host = float((-logp.double() + 0.02 * regs[0].double() + 0.03 * regs[1].double()).mean())
    assert abs(float(a) - host) < 2e-6 * max(1.0, abs(host))
    assert abs(float(pkg.loss(icnf, m, dev(xs), dev(p), {}, eps=dev(eps))) - float(a)) == 0.0


def test_split_bf16_error_bound_holds_on_adversarial_products(pkg, oracles):
    """the six-term split-bf16 product obeys |err| <= (4 u + gamma_{6K}) sum_k |a_k| |b_k| (u = 2^-24, K the
    contraction length; gamma_n = n u / (1 - n u)) under ANY accumulation order inside v_mfma_f32_16x16x32_bf16, against
    gamma_K for the exact-f32 chain.  Adversarial hidden layers: rows of alternating +-c weights on nearly equal
    activations, so a pre-activation is a difference of terms 10^2..10^3 times its own size and the products' rounding
    errors are not masked - across magnitudes c, with weights on both sides of bf16's rounding boundaries.  Both
    arithmetics must stay inside their bound against the float64 evaluation; the measured constants are asserted too:
    the split arithmetic's observed error must be within 4x of the exact kernel's (it is usually smaller: bf16 x bf16
    products are exact and the instruction rounds less often than a 64-term fma chain)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    w_off, b_off, n = spec.param_offsets()
    u24 = 2.0 ** -24
    gam = lambda k: k * u24 / (1 - k * u24)
    worst = {}
    for trial, cmag in enumerate((1.0, 8.0, 40.0, 200.0, 1000.0, -0.05)):
        rng = np.random.default_rng(50 + trial)
        B = 256
        p, xs, eps, _ = o64.synth_inputs(spec, B, 900 + trial, bias_scale=0.3)
        p = p.astype(np.float32)
        # layer 1: nearly equal activations around 0.5 (small weights, common bias), closer together the larger c is, so that
        # layer 2's pre-activations - differences of 64 terms of size c / 2 - stay O(1) and nothing saturates
        exact_inputs = cmag < 0          # last case: layer 1 saturated to exactly +-1, so the product's inputs carry no error
        cmag = abs(cmag)
        jit = min(0.01, 0.08 / cmag)
        p[w_off[0]:b_off[0]] *= 2.0 * jit
        p[b_off[0]:b_off[0] + 64] = 0.55 + jit * rng.standard_normal(64)
        if exact_inputs:
            p[b_off[0]:b_off[0] + 64] = 15.0 * rng.choice([1.0, -1.0], 64)
        # layer 2 (hidden, split arithmetic): alternating-sign rows, magnitudes jittered in the last bf16-visible bits and below;
        # layer 3 (hidden, split arithmetic): benign Glorot weights; layer 4 as drawn
        W = (cmag * (1.0 + 2.0 ** -9 * rng.integers(-7, 8, (64, 64)) + 1e-6 * rng.standard_normal((64, 64)))
             * np.where(np.arange(64)[None, :] % 2 == 0, 1.0, -1.0) * rng.choice([1.0, -1.0], (64, 1)))
        p[w_off[1]:w_off[1] + 64 * 64] = W.T.reshape(-1).astype(np.float32)
        p[b_off[1]:b_off[1] + 64] = 0.05 * rng.standard_normal(64)
        u = np.vstack([xs.astype(np.float32), np.zeros((3, B), np.float32)])
        t = 0.37
        ref, _ = _forward_error_bound(spec, p, u, t, 0.0, 0.0, 0.0)
        res = {}
        for name, icnf, c_hidden in (("f32", make_icnf(pkg, spec, 1, 40, path=2), gam(64 + 1)),
                                     ("bf16x6", make_icnf_bf16x6(pkg, spec, 1, 40), 4 * u24 + gam(6 * 64 + 1))):
            du = pkg.augmented_f(icnf, mode_of(pkg, spec), dev(u), dev(p), t, dev(eps), None).cpu().numpy().astype(np.float64)
            _, bound = _forward_error_bound(spec, p, u, t, c_hidden, gam(64 + 1), 8 * u24)
            err = np.abs(du[:8] - ref)
            assert np.all(err <= bound + 1e-30), (name, cmag, float((err / bound).max()))
            res[name] = (float(err.max()), float((err / bound).max()))
        worst[-cmag if exact_inputs else cmag] = res
        # the cancellation is real: the bound is far above float32 resolution of the result itself
        assert res["bf16x6"][0] <= 4.0 * res["f32"][0] + 1e-7, (cmag, res)
    assert all(np.isfinite(v[k][0]) for v in worst.values() for k in v), worst
    print("split-bf16 vs f32, (max |err|, max err / bound) per weight magnitude:", worst)


def test_split_bf16_is_refused_where_not_implemented(pkg, oracles):
    o64, _ = oracles
    spec = o64.make_spec(nvars=32, hidden=[256, 256, 256])
    icnf = make_icnf_bf16x6(pkg, spec, 0, 10)
    with pytest.raises(pkg._lib.CnfError) as e:
        icnf.kernel_path(pkg.TrainMode(False))
    assert e.value.code == pkg._lib.ERR_UNSUPPORTED


def grad_icnf(pkg, spec, alg, nsteps):
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2)
    icnf.lambda1 = icnf.lambda2 = icnf.lambda3 = 0.0
    return icnf


def test_parameter_gradient_matches_golden_fixture(pkg, oracles):
    """dloss/dp of the discrete loss (reverse-sweep kernel, csrc/cnf_grad.hip) against the committed
    fp64 autograd fixture: cfg2 shape, B = 8, Tsit5 x 10.  Tolerance: float32 accumulation of a
    sum over 8 samples x 60 stages, relative to the gradient's scale."""
    import os
    from conftest import GOLDEN
    o64, _ = oracles
    f = np.load(os.path.join(GOLDEN, "grad_cfg2_d8_3x64_tsit5.npz"))
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    icnf = grad_icnf(pkg, spec, 1, int(f["nsteps"]))
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(f["xs"]), dev(f["p"]), {}, eps=dev(f["eps"]))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - float(f["loss"])) < 1e-4
    scale = np.abs(f["grad"]).max()
    assert np.max(np.abs(g - f["grad"])) < 2e-5 * scale + 1e-6, np.max(np.abs(g - f["grad"])) / scale


@pytest.mark.parametrize("D,H,B,alg,nsteps", [(8, 64, 100, 0, 6), (6, 56, 37, 1, 4)])
def test_parameter_gradient_matches_autograd_oracle(D, H, B, alg, nsteps, pkg, oracles):
    """Ragged batches (partial tiles), RK4 and Tsit5, padded D and H: gradient vs the fp64 oracle,
    plus the bias rows and the time column explicitly."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=D, hidden=[H, H, H])
    p, xs, eps, _ = o64.synth_inputs(spec, B, 5 + D, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps)
    icnf = grad_icnf(pkg, spec, alg, nsteps)
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs), dev(p), {}, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    err = np.abs(g - gref)
    assert err.max() < 5e-5 * scale + 1e-6, (err.max() / scale, int(err.argmax()))
    w_off, b_off, _ = spec.param_offsets()
    tcol = slice(w_off[0] + H * D, w_off[0] + H * (D + 1))       # d/dW_1[:, time column]
    assert np.abs(gref[tcol]).max() > 0 and err[tcol].max() < 5e-5 * scale + 1e-6
    for l in range(4):
        bs = slice(b_off[l], b_off[l] + spec.widths[l + 1])
        assert err[bs].max() < 5e-5 * scale + 1e-6


@pytest.mark.parametrize("alg,nsteps", [(0, 5), (1, 3)])
def test_parameter_gradient_of_the_regularised_objective(alg, nsteps, pkg, oracles):
    """TrainMode{true} objective mean(-logp + l1 |zdot| + l2 |eps^T J| + l3 |z_aug|) (src/core/icnf.jl:628-637)
    with augmented dimensions: gradient vs the fp64 autograd oracle."""
    o64, _ = oracles
    lam = (0.05, 0.07, 0.03)
    spec = o64.make_spec(nvars=6, naug=2, hidden=[64, 64, 64], reg_z=True, reg_j=True, reg_aug=True)
    B = 70
    p, xs, eps, _ = o64.synth_inputs(spec, B, 41, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, None, lam)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    assert np.max(np.abs(g - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(g - gref)) / scale
    # the regularisers matter: the plain FFJORD gradient is measurably different
    _, g0 = o64.loss_and_grad(o64.make_spec(nvars=6, naug=2, hidden=[64, 64, 64]), p, xs, 0.0, 1.0, nsteps, alg, eps)
    assert np.max(np.abs(g0 - gref)) > 1e-3 * scale


GRAD_SHAPES = [
    # (make_spec kwargs, lambdas, B, alg, nsteps)
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 50, 1, 4),  # reference default ICNF(nvariables=1), TrainMode{true}
    (dict(nvars=2, hidden=[32, 32]), (0.0, 0.0, 0.0), 33, 1, 5),                                # cfg1 shape
    (dict(nvars=2, naug=3, hidden=[24, 24], act=2, reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 20, 0, 4),  # default net, nvariables=2
    (dict(nvars=12, hidden=[48, 48, 48]), (0.0, 0.0, 0.0), 19, 0, 3),                           # D = 12 (padded state k-steps)
    (dict(nvars=5, hidden=[64, 64], autonomous=True, act=2), (0.0, 0.0, 0.0), 17, 1, 3),        # autonomous, softplus, L = 2
    (dict(nvars=8, ncond=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), (0.01, 0.01, 0.0), 40, 1, 3),   # conditioned RNODE
    (dict(nvars=2, ncond=2, naug=3, hidden=[32, 32], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 21, 0, 4),  # default CondICNF net (nvariables=2, nconditions=2)
    (dict(nvars=3, ncond=13, hidden=[16, 16, 16]), (0.0, 0.0, 0.0), 9, 0, 3),                   # many conditions, one hidden tile
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", GRAD_SHAPES)
def test_parameter_gradient_other_shapes_and_softplus(kw, lam, B, alg, nsteps, pkg, oracles):
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 77, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    assert np.max(np.abs(g - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(g - gref)) / scale


PROBE_GRAD_SHAPES = [
    # (make_spec kwargs, lambdas, B, alg, nsteps): several Hutchinson probes (csrc/cnf_grad2_probes.hip)
    (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), (0.01, 0.01, 0.0), 70, 1, 3),   # BASELINE cfg3 shape (RNODE, K = 4)
    (dict(nvars=8, hidden=[64, 64, 64], nprobes=2), (0.0, 0.0, 0.0), 37, 0, 4),                              # FFJORD, K = 2, no regularisers
    (dict(nvars=2, naug=3, hidden=[24, 24], act=2, nprobes=3, reg_z=True, reg_j=True, reg_aug=True), (0.02, 0.03, 0.01), 21, 1, 3),
    (dict(nvars=6, ncond=5, hidden=[40, 40, 40], nprobes=2, reg_j=True), (0.0, 0.05, 0.0), 18, 0, 3),        # conditioned, padded hidden width
    (dict(nvars=11, hidden=[16, 16], nprobes=5, autonomous=True), (0.0, 0.0, 0.0), 9, 1, 2),
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", PROBE_GRAD_SHAPES)
def test_parameter_gradient_with_several_probes(kw, lam, B, alg, nsteps, pkg, oracles):
    """K > 1 probes: ldot and ndot are probe means (1/K sum_k), the gradient kernel runs the pullback
    and its reverse once per probe.  Checked against the fp64 autograd oracle."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 91, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=2, lambdas=lam)
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), *args, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    assert np.max(np.abs(g - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(g - gref)) / scale
    # the probes matter: one probe alone gives a measurably different gradient
    kw1 = dict(kw, nprobes=1)
    _, g1 = o64.loss_and_grad(o64.make_spec(**kw1), p, xs, 0.0, 1.0, nsteps, alg, eps[:spec.D], ys, lam)
    assert np.max(np.abs(g1 - gref)) > 1e-3 * scale


def test_probe_gradient_kernel_agrees_with_the_single_probe_kernel(pkg, oracles):
    """K identical probes have the same loss and gradient as one probe: the two gradient kernels
    (cnf_grad2.hip compiled for one probe and, with a rolled probe loop, for several) must agree."""
    o64, _ = oracles
    kw = dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    lam = (0.02, 0.03, 0.0)
    s1, s3 = o64.make_spec(**kw), o64.make_spec(nprobes=3, **kw)
    B = 2000
    p, xs, eps, _ = o64.synth_inputs(s1, B, 17, bias_scale=0.1)
    i1 = make_icnf(pkg, s1, 1, 8, path=2, lambdas=lam)
    i3 = make_icnf(pkg, s3, 1, 8, path=2, lambdas=lam)
    v1, g1 = pkg.loss_and_gradient(i1, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
    v3, g3 = pkg.loss_and_gradient(i3, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(np.tile(eps, (3, 1))))
    assert abs(float(v1) - float(v3)) < 2e-5
    scale = float(g1.abs().max())
    assert float((g1 - g3).abs().max()) < 2e-5 * scale


def test_full_size_gradient_is_additive_over_column_shards(pkg, oracles):
    """Headline size (B = 65536, Tsit5 x 40): the summed gradient of the whole batch equals the sum of
    the gradients of two column shards — the identity the multi-GPU all-reduce relies on — and the
    training loss equals the inference loss."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 65536
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240621)
    icnf = grad_icnf(pkg, spec, 1, 40)
    m = pkg.TrainMode(False)
    X, E, P = dev(xs), dev(eps), dev(p)
    vf, gf = pkg.loss_and_gradient(icnf, m, X, P, {}, eps=E)
    h = 40000                                             # uneven split, partial tiles on neither side
    v1, g1 = pkg.loss_and_gradient(icnf, m, X[:, :h], P, {}, eps=E[:, :h])
    v2, g2 = pkg.loss_and_gradient(icnf, m, X[:, h:], P, {}, eps=E[:, h:])
    gs = (g1.double() * h + g2.double() * (B - h)) / B
    scale = float(gf.abs().max())
    assert float((gf.double() - gs).abs().max()) < 2e-5 * scale
    assert abs(float(vf) - (float(v1) * h + float(v2) * (B - h)) / B) < 1e-5
    assert abs(float(vf) - float(pkg.loss(icnf, m, X, P, {}, eps=E))) < 1e-5
    assert bool(torch.isfinite(gf).all())


def test_gradient_descent_on_the_gradient_kernel_reduces_the_loss(pkg, oracles):
    """End-to-end use of the training path: a few Adam steps driven by loss_and_gradient lower the
    NLL of a shifted, scaled Gaussian (the role MLJ `fit` plays around the reference's loss)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    rng = np.random.default_rng(3)
    p0 = o64.glorot_params(spec, rng)
    xs = (0.5 * rng.standard_normal((8, 4096)) + 1.0).astype(np.float32)
    eps = rng.standard_normal((8, 4096)).astype(np.float32)
    icnf = grad_icnf(pkg, spec, 0, 10)
    ps = dev(p0)
    opt = torch.optim.Adam([ps], lr=2e-3)
    losses = []
    for _ in range(25):
        val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs), ps, {}, eps=dev(eps))
        losses.append(float(val))
        ps.grad = g
        opt.step()
    assert losses[-1] < losses[0] - 0.5, losses
    assert all(np.isfinite(losses))
    # gradient twice on the same inputs is bit-identical (no atomics in the accumulation)
    a = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs), ps, {}, eps=dev(eps))[1]
    b = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs), ps, {}, eps=dev(eps))[1]
    assert torch.equal(a, b)


LAYERED_GRAD_SHAPES = [
    # (make_spec kwargs, lambdas, B, alg, nsteps): outside the fused gradient kernels -> layer-wise path (csrc/cnf_layered.hip)
    (dict(nvars=32, hidden=[256, 256, 256]), (0.0, 0.0, 0.0), 40, 0, 2),                                   # BASELINE cfg4 shape (cooperative forward kernel)
    (dict(nvars=8, ncond=8, hidden=[128, 128, 128], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 300, 1, 2),  # cfg5 widths, conditioned RNODE, chunked weight cotangents
    (dict(nvars=3, naug=2, hidden=[24, 40, 16, 32, 24], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.02, 0.03), 50, 1, 3),  # five unequal hidden layers (SIMT forward)
    (dict(nvars=20, hidden=[64, 64], nprobes=3, reg_j=True), (0.0, 0.05, 0.0), 33, 0, 3),                 # D = 20, three probes
    (dict(nvars=4, hidden=[96], autonomous=True), (0.0, 0.0, 0.0), 25, 1, 2),                              # one hidden layer
    (dict(nvars=48, naug=49, hidden=[392, 392], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 40, 1, 2),   # ICNF(nvariables = 48): the first default net past the fused kernels (25 hidden tiles; products of K = 392 on lg_gemm2's wide-K instances)
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", LAYERED_GRAD_SHAPES)
def test_parameter_gradient_layerwise_path(kw, lam, B, alg, nsteps, pkg, oracles):
    """Every Hutchinson-VJP configuration has a gradient: what the fused reverse-sweep kernels do not
    cover runs layer-wise on the library's own MFMA product kernels (csrc/cnf_lgemm.hip; the cfg4 shape on the cooperative
    reverse sweep).  Checked against the fp64 autograd oracle."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 123, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    assert np.max(np.abs(g - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(g - gref)) / scale


COOP_GRAD_SHAPES = [
    # (make_spec kwargs, B, alg, nsteps, env): wide tanh FFJORD nets whose forward solve runs on the cooperative kernel
    (dict(nvars=32, hidden=[256, 256, 256]), 40, 0, 2, {}),                       # BASELINE cfg4 shape, RK4, one ragged super-tile
    (dict(nvars=32, hidden=[256, 256, 256]), 333, 1, 2, {}),                      # Tsit5 (6 stages), several super-tiles
    (dict(nvars=20, naug=5, hidden=[200, 200, 200], reg_aug=True), 150, 0, 3, {}),   # zero-padded width (13 of 16 tiles), augmented, l3 |z_aug|
    (dict(nvars=8, hidden=[128, 128, 128]), 200, 1, 2, {}),                       # 3 x 128 (8 hidden tiles, 2 state k-steps)
    (dict(nvars=8, hidden=[64, 64, 64]), 130, 1, 3, {"CNF_MFMA_COOP": "1", "CNF_GRAD_LAYERED": "1"}),   # 3 x 64 forced onto it
    (dict(nvars=7, hidden=[128, 128, 128], autonomous=True), 90, 0, 2, {}),       # no time column
    # the regularised objective (the reference's default lambdas are non-zero, src/core/icnf.jl:73-75): |zdot| and |eps^T J|
    # cotangents from the forward solve's stage checkpoints
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 77, 0, 2, {"lam": (0.02, 0.03, 0.0)}),
    (dict(nvars=20, naug=5, hidden=[200, 200, 200], reg_z=True, reg_j=True, reg_aug=True), 100, 1, 2, {"lam": (0.01, 0.01, 0.01)}),   # the reference's defaults
    (dict(nvars=8, hidden=[128, 128, 128], reg_z=True), 70, 1, 2, {"lam": (0.05, 0.0, 0.0)}),        # |zdot| alone
    (dict(nvars=8, hidden=[128, 128, 128], reg_j=True), 70, 0, 2, {"lam": (0.0, 0.05, 0.0)}),        # |eps^T J| alone
    # forward solve on the extended cooperative kernel in its checkpointing form: softplus, two hidden layers, 33 <= D <= 64 -
    # the reference's default architecture for nvariables >= 16 under its default objective
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), 70, 1, 2, {"lam": (0.01, 0.01, 0.01)}),   # ICNF(nvariables = 16)
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), 45, 0, 2, {"lam": (0.01, 0.01, 0.01)}),   # ICNF(nvariables = 24): 16 x 16 tiles
    (dict(nvars=40, hidden=[192, 192, 192]), 50, 0, 2, {}),                       # tanh, D = 40, three layers
    (dict(nvars=12, hidden=[160, 160], act=2), 60, 1, 2, {}),                     # softplus, two layers, D <= 32
    (dict(nvars=10, hidden=[176, 176]), 40, 0, 3, {}),                            # tanh, two layers
    (dict(nvars=32, naug=33, hidden=[264, 264], act=2, reg_z=True, reg_j=True, reg_aug=True), 40, 1, 2, {"lam": (0.01, 0.01, 0.01)}),   # ICNF(nvariables = 32): 20 x 24 tiles
    (dict(nvars=80, hidden=[384, 384, 384]), 36, 0, 2, {}),                       # tanh, D = 80, H = 384 (24 x 24 tiles)
    # conditioned flows (CondICNF): the condition rows of layer 1 and of its cotangent
    (dict(nvars=8, ncond=8, hidden=[256, 256, 256], reg_z=True, reg_j=True), 70, 1, 2, {"lam": (0.01, 0.01, 0.0)}),   # conditioned RNODE, tanh
    (dict(nvars=16, naug=17, ncond=5, hidden=[156, 156], act=2, reg_z=True, reg_j=True, reg_aug=True), 50, 0, 2, {"lam": (0.01, 0.01, 0.01)}),   # default architecture (nvariables = 16) with 5 conditions
    (dict(nvars=20, naug=21, ncond=16, hidden=[232, 232], act=2, autonomous=True), 40, 1, 2, {}),   # 16 conditions, D = 41, autonomous
    # edges: one column and one step; the largest instance (24 x 24 tiles) filled exactly
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 1, 0, 1, {"lam": (0.01, 0.01, 0.0)}),
    (dict(nvars=47, naug=48, hidden=[384, 384], act=2, reg_z=True, reg_j=True, reg_aug=True), 20, 1, 2, {"lam": (0.01, 0.01, 0.01)}),   # ICNF(nvariables = 47)
]


@pytest.mark.parametrize("kw,B,alg,nsteps,env", COOP_GRAD_SHAPES)
def test_parameter_gradient_cooperative_reverse_sweep(kw, B, alg, nsteps, env, pkg, oracles, monkeypatch):
    """Wide hidden layers (the cooperative forward kernel's shapes): checkpointing forward solve + one cooperative reverse-sweep
    launch per RK step + deferred weight-cotangent products (csrc/cnf_coop_grad.hip) - dloss/dps, dloss/dxs and the loss
    against fp64 autograd through the same discrete solve (src/core/icnf.jl:90-99 differentiates `loss` through the solve),
    and against the layer-wise path on the same inputs (CNF_COOP_GRAD=0)."""
    o64, _ = oracles
    env = dict(env)
    lam = env.pop("lam", None)
    for k, v in env.items():
        setsw(pkg, monkeypatch, k, v)
    spec = o64.make_spec(**kw)
    if lam is None:
        lam = (0.0, 0.0, 0.03 if spec.reg_aug else 0.0)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 321, bias_scale=0.2)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True)
    mode = mode_of(pkg, spec)
    out = {}
    for tag, flag in (("coop", "1"), ("layered", "0")):
        setsw(pkg, monkeypatch, "CNF_COOP_GRAD", flag)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
        val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
        assert icnf.grad_path(mode) == (3 if tag == "coop" else 2), (tag, icnf.grad_path(mode))
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
        assert abs(out[tag][0] - L) < 1e-4 + 2e-6 * abs(L), tag
        scale = np.abs(gref).max()
        assert np.max(np.abs(out[tag][1] - gref)) < 5e-5 * scale + 1e-6, (tag, np.max(np.abs(out[tag][1] - gref)) / scale)
        assert np.max(np.abs(out[tag][2] - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7, tag
    assert np.max(np.abs(out["coop"][1] - out["layered"][1])) < 2e-5 * np.abs(gref).max() + 1e-6


@pytest.mark.parametrize("kw", [
    dict(nvars=15, naug=16, hidden=[128, 128], act=2, reg_z=True, reg_j=True, reg_aug=True),    # ICNF(nvariables = 15): 8 hidden tiles
    dict(nvars=12, naug=13, hidden=[104, 104], act=2, reg_z=True, reg_j=True, reg_aug=True),    # ICNF(nvariables = 12): 7
    dict(nvars=6, hidden=[112, 112]),                                                            # tanh FFJORD
])
def test_mid_width_gradient_takes_the_cooperative_sweep_at_large_batches(kw, pkg, oracles, monkeypatch):
    """Two-hidden-layer nets of 7 - 8 hidden tiles keep their per-wave forward plan; from 4096 columns on their gradient runs on
    the cooperative reverse sweep through an auxiliary cooperative plan and image (cnf_handle::plan_cg) - against fp64 autograd
    and against the slab-accumulator kernel they take below that size (CNF_COOP_GRAD_MID=0 keeps them there)."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    B, nsteps, alg = 4096, 2, 1
    lam = (0.01, 0.01, 0.01) if spec.reg_z else (0.0, 0.0, 0.0)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 77, bias_scale=0.2)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True)
    mode = mode_of(pkg, spec)
    out = {}
    for tag, flag in (("aux", "1"), ("slab", "0")):
        setsw(pkg, monkeypatch, "CNF_COOP_GRAD_MID", flag)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert icnf.kernel_path(mode) == 2 and icnf.grad_path(mode) == 1
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
        assert abs(out[tag][0] - L) < 1e-4 + 2e-6 * abs(L), tag
        assert np.max(np.abs(out[tag][1] - gref)) < 5e-5 * np.abs(gref).max() + 1e-6, tag
        assert np.max(np.abs(out[tag][2] - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7, tag
    assert np.max(np.abs(out["aux"][1] - out["slab"][1])) > 0.0      # two implementations, two summation orders
    # below a threshold given in columns (round 5: the default is every batch size) the same handle serves the slab kernel
    setsw(pkg, monkeypatch, "CNF_COOP_GRAD_MID", "4096")
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
    val, g = pkg.loss_and_gradient(icnf, mode, dev(xs[:, :300]), dev(p), {}, eps=dev(eps[:, :300]))
    setsw(pkg, monkeypatch, "CNF_COOP_GRAD_MID", "0")
    icnf0 = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
    val0, g0 = pkg.loss_and_gradient(icnf0, mode, dev(xs[:, :300]), dev(p), {}, eps=dev(eps[:, :300]))
    assert float(val) == float(val0) and torch.equal(g, g0)


@pytest.mark.parametrize("kw,B", [
    (dict(nvars=15, naug=16, hidden=[128, 128], act=2, reg_z=True, reg_j=True, reg_aug=True), 8192),   # ICNF(nvariables = 15): 8 hidden tiles
    (dict(nvars=8, hidden=[128, 128, 128], reg_z=True, reg_j=True), 8192),                                  # 3 x 128 tanh
    (dict(nvars=16, hidden=[192, 192, 192], reg_z=True, reg_j=True), 16384),                                # 3 x 192 tanh: 12 tiles
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2, reg_z=True, reg_j=True, reg_aug=True), 16384),     # ICNF(nvariables = 20)
])
def test_fused_gradient_routes_agree_when_a_launch_has_more_workgroups_than_compute_units(kw, B, pkg, oracles, monkeypatch):
    """The gradient of a batch that gives the sweep kernels more workgroups than the chip has compute units (two workgroups resident
    per CU) against the route the configuration takes with the cooperative sweep switched off (CNF_COOP_GRAD=0: layer-wise, or the
    slab-accumulator kernel).  Added when the 8-tile instances of cnf_coop_grad.hip were found to drop entries of X_1 in exactly this
    regime (layer-1 cotangent off by 1e-3 relative; every parity case until then ran at most one workgroup per CU): a 128-bit
    buffer store whose fourth data register the next VALU instruction overwrote before the store had read it - a wait-state hazard the
    compiler does not cover on gfx950 (CNF_STORE_DATA_HAZARD, csrc/cnf_coop_dev.h).  Every route has to agree here to summation order."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 4242, bias_scale=0.2)
    lam = (0.01, 0.01, 0.01 if spec.reg_aug else 0.0)
    out = {}
    for tag, env in (("default", "1"), ("no_coop", "0")):
        setsw(pkg, monkeypatch, "CNF_COOP_GRAD", env)
        icnf = make_icnf(pkg, spec, 1, 2, lambdas=lam)
        val, g = pkg.loss_and_gradient(icnf, mode_of(pkg, spec), dev(xs), dev(p), {}, eps=dev(eps))
        out[tag] = (float(val), g.double().cpu().numpy())
    a, b = out["default"][1], out["no_coop"][1]
    assert abs(out["default"][0] - out["no_coop"][0]) < 1e-5 * abs(out["no_coop"][0]) + 1e-6
    assert np.linalg.norm(a - b) < 2e-6 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
    W = spec.widths
    assert np.linalg.norm(a[:W[0] * W[1]] - b[:W[0] * W[1]]) < 5e-6 * np.linalg.norm(b[:W[0] * W[1]])     # the first layer's weights on their own


def test_cooperative_gradient_at_full_size_agrees_with_the_layerwise_path(pkg, oracles, monkeypatch):
    """BASELINE cfg4's shard (D = 32, 3 x 256, RK4 x 40, B = 32 768): loss, dloss/dps and dloss/dxs of the cooperative reverse
    sweep against the layer-wise path on the same inputs - two independent implementations of the same discrete adjoint
    (different products, different summation orders), both pinned to fp64 autograd at small sizes."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=32, hidden=[256, 256, 256])
    B = 32768
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240615)
    out = {}
    for tag, flag in (("coop", "1"), ("layered", "0")):
        setsw(pkg, monkeypatch, "CNF_COOP_GRAD", flag)
        icnf = make_icnf(pkg, spec, 0, 40, path=0, lambdas=(0.0, 0.0, 0.0))
        val, g, gx = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert icnf.grad_path(pkg.TrainMode(False)) == (3 if tag == "coop" else 2)
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
    a, b = out["coop"], out["layered"]
    assert abs(a[0] - b[0]) < 2e-6 * abs(b[0]) + 1e-6
    assert np.max(np.abs(a[1] - b[1])) < 2e-5 * np.abs(b[1]).max()
    assert np.max(np.abs(a[2] - b[2])) < 2e-5 * np.abs(b[2]).max()
    assert np.all(np.isfinite(a[1])) and np.all(np.isfinite(a[2]))


STAGE_STORE_GRAD_SHAPES = [
    # (make_spec kwargs, B, alg, nsteps, lam, force the dealt forward kernel at this batch): the cooperative gradient's SECOND FORM
    (dict(nvars=32, hidden=[256, 256, 256]), 40, 0, 2, None, False),                                   # BASELINE cfg4's shape, RK4, one ragged super-tile
    (dict(nvars=32, hidden=[256, 256, 256]), 333, 1, 2, None, False),                                  # Tsit5, several super-tiles, tiles behind the batch
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 77, 0, 2, (0.02, 0.03, 0.0), False),   # |zdot| and |eps^T J| cotangents
    (dict(nvars=30, naug=2, hidden=[252, 252, 252], autonomous=True, reg_aug=True), 65, 1, 2, (0.0, 0.0, 0.02), False),   # zero-padded width (rows of four), no time row, |z_aug|
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), 1, 0, 1, (0.01, 0.01, 0.0), False),    # one column, one step
    # the forward solve on the dealt kernel (8 .. 15 hidden tiles; CNF_COOPD=2 takes it below 4096 columns): the real tiles are stored
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), 70, 1, 2, (0.01, 0.01, 0.01), True),   # ICNF(nvariables = 16): 9 tiles (A = 2, one left-over), 12 state registers
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2, reg_z=True, reg_j=True, reg_aug=True), 130, 1, 2, (0.01, 0.01, 0.01), True),  # ICNF(nvariables = 20): 11 tiles (three left-over)
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), 45, 0, 2, (0.01, 0.01, 0.01), True),   # ICNF(nvariables = 24): 13 tiles (A = 3), 16 state registers
    (dict(nvars=12, hidden=[160, 160], act=2), 60, 1, 2, None, True),                                   # softplus, 10 tiles, D <= 32
    (dict(nvars=48, hidden=[168, 168], act=2), 37, 1, 2, None, True),                                   # D = 48: the time row opens a fourth 16-row group of the input side (DTZ > DT)
    (dict(nvars=20, naug=4, hidden=[136, 136], autonomous=True, reg_aug=True), 100, 0, 2, (0.0, 0.0, 0.02), True),   # tanh, two layers, autonomous (no time row), RK4, |z_aug|
    (dict(nvars=28, naug=29, hidden=[232, 232], act=2, reg_z=True, reg_j=True, reg_aug=True), 33, 1, 1, (0.01, 0.01, 0.01), True),   # ICNF(nvariables = 28): 15 tiles (A = 3, three left-over), one step
    (dict(nvars=10, hidden=[176, 176]), 40, 0, 3, None, True),                                          # tanh, two layers
    (dict(nvars=20, naug=5, hidden=[200, 200, 200], reg_aug=True), 150, 0, 3, None, True),              # tanh, three layers, 13 of 16 tiles
    (dict(nvars=8, hidden=[136, 136, 136], act=2), 90, 1, 2, None, True),                               # softplus, three layers (A = 2)
]


@pytest.mark.parametrize("kw,B,alg,nsteps,lam,dealt", STAGE_STORE_GRAD_SHAPES)
def test_cooperative_gradient_second_form(kw, B, alg, nsteps, lam, dealt, pkg, oracles, monkeypatch):
    """Round 6: the checkpointing forward solve stores h_l and delta_l of every stage as tiles, the sweep of
    csrc/cnf_coop_grad3.hip runs the second-order chains alone and the weight cotangents are products over tiles
    (csrc/cnf_wgrad_tiles.hip).  dloss/dps, dloss/dxs and the loss against fp64 autograd through the same discrete solve
    (src/core/icnf.jl:90-99 differentiates `loss` through the solve), and against the sweeps that recompute both first-order
    chains (CNF_COOP_GRAD3=0) on the same handle configuration; cnf_grad_form_for says which one a call takes.  Two hidden layers
    run the sweep with two workgroups per CU (csrc/cnf_coop_grad3w.hip; CNF_COOP_GRAD3=3 takes it at these small batches too);
    CNF_COOP_GRAD3=2 keeps the one-per-CU sweep for every shape: same chains, same summation order - the two must agree bit for bit."""
    o64, _ = oracles
    if dealt:
        setsw(pkg, monkeypatch, "CNF_COOPD", "2")
    spec = o64.make_spec(**kw)
    if lam is None:
        lam = (0.0, 0.0, 0.03 if spec.reg_aug else 0.0)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 654, bias_scale=0.2)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True)
    mode = mode_of(pkg, spec)
    out = {}
    for tag, flag in (("store", "3"), ("store_one_per_cu", "2"), ("recompute", "0")):
        setsw(pkg, monkeypatch, "CNF_COOP_GRAD3", flag)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        assert icnf.grad_path(mode, B=B, alg=alg) == 3
        assert icnf.grad_form(mode, B, alg, nsteps) == (1 if tag == "recompute" else 2), (tag, icnf.grad_form(mode, B, alg, nsteps))
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
        assert abs(out[tag][0] - L) < 1e-4 + 2e-6 * abs(L), tag
        scale = np.abs(gref).max()
        assert np.max(np.abs(out[tag][1] - gref)) < 5e-5 * scale + 1e-6, (tag, np.max(np.abs(out[tag][1] - gref)) / scale)
        assert np.max(np.abs(out[tag][2] - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7, tag
    assert np.max(np.abs(out["store"][1] - out["recompute"][1])) < 2e-5 * np.abs(gref).max() + 1e-6
    assert np.array_equal(out["store"][1], out["store_one_per_cu"][1]) and np.array_equal(out["store"][2], out["store_one_per_cu"][2])


def test_second_form_is_deterministic_and_falls_back_when_the_store_does_not_fit(pkg, oracles, monkeypatch):
    """The stage store is bounded (cnf_tuning.coop_grad3_gib): a call whose store would not fit takes the recomputing sweeps, and
    says so through cnf_grad_form_for.  Two calls of the second form on one handle return the same bits (fixed summation order)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=32, hidden=[256, 256, 256])
    B, nsteps, alg = 2048, 3, 0
    p, xs, eps, _ = o64.synth_inputs(spec, B, 11, bias_scale=0.2)
    mode = pkg.TrainMode(False)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=(0.0, 0.0, 0.0))
    assert icnf.grad_form(mode, B, alg, nsteps) == 2
    v1, g1 = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps))[:2]
    v2, g2 = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps))[:2]
    assert float(v1) == float(v2) and torch.equal(g1, g2)
    # 2 kinds x 3 layers x nsteps x 4 stages x 128 tiles x 16 KB = 302 MB: does not fit 0 GiB
    old = pkg.set_tuning(coop_grad3_gib=0)
    try:
        assert icnf.grad_form(mode, B, alg, nsteps) == 1
        v3, g3 = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps))[:2]
    finally:
        pkg.set_tuning(**old)
    assert abs(float(v3) - float(v1)) < 1e-6 * abs(float(v1)) + 1e-6
    assert float((g3 - g1).abs().max()) < 2e-5 * float(g1.abs().max())


SLAB_GRAD_SHAPES = [
    # two hidden layers, 4..7 hidden tiles: tile-fused reverse sweep with slab accumulators (csrc/cnf_grad_slab.hip)
    (dict(nvars=7, naug=8, hidden=[64, 64], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 45, 1, 3),   # ICNF(nvariables=7): D=15, two input tiles
    (dict(nvars=8, naug=9, hidden=[72, 72], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 70, 1, 3),   # ICNF(nvariables=8): D=17, H=72
    (dict(nvars=10, naug=11, hidden=[88, 88], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.02, 0.03), 33, 0, 3), # ICNF(nvariables=10): D=21, H=88
    (dict(nvars=10, hidden=[80, 96]), (0.0, 0.0, 0.0), 50, 1, 2),                                                         # FFJORD, unequal widths, one input tile
    (dict(nvars=15, hidden=[112, 100], autonomous=True, reg_j=True), (0.0, 0.05, 0.0), 21, 0, 2),                         # 7 tiles, autonomous, D=15 in one tile
    (dict(nvars=12, naug=13, hidden=[104, 104], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 40, 1, 2), # ICNF(nvariables=12): 7 tiles, D-sized images from global memory
    (dict(nvars=14, naug=15, hidden=[120, 120], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.02, 0.01), 19, 0, 2), # ICNF(nvariables=14): D=29, 8 tiles
    (dict(nvars=6, hidden=[128, 128], reg_z=True), (0.03, 0.0, 0.0), 35, 1, 2),                                           # 8 tiles, D=6
    (dict(nvars=15, naug=16, hidden=[128, 128], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 22, 1, 2), # ICNF(nvariables=15): D=31 + time = 32 input columns
    (dict(nvars=15, hidden=[64, 48]), (0.0, 0.0, 0.0), 30, 0, 2),                                                         # D=15 + time = 16 columns in one input tile
    (dict(nvars=8, naug=9, ncond=4, hidden=[88, 88], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 41, 1, 2), # CondICNF(nvariables=8, nconditions=4) default net
    (dict(nvars=6, ncond=16, hidden=[128, 112], reg_j=True), (0.0, 0.04, 0.0), 23, 0, 2),                                 # 16 conditions, 8 tiles (D-sized images from global memory)
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", SLAB_GRAD_SHAPES)
def test_parameter_gradient_slab_kernel(kw, lam, B, alg, nsteps, pkg, oracles, monkeypatch):
    """The mid-width two-hidden-layer nets (the reference's default architecture for 7..11 variables): parameters and
    data gradients against fp64 autograd, and against the layer-wise path on the same inputs."""
    o64, _ = oracles
    setsw(pkg, monkeypatch, "CNF_COOP_GRAD_MID", "0")       # (7 - 8 hidden tiles otherwise take the auxiliary cooperative sweep: round 5)
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 222, bias_scale=0.2)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    assert icnf.grad_path(mode) == 1 and icnf.grad_path(mode, B=B, alg=alg) == 1
    cargs = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g, gx = pkg.loss_and_gradient(icnf, mode, *cargs, eps=dev(eps), wrt_x=True)
    assert abs(float(val) - L) < 1e-4
    sc = np.abs(gref).max()
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * sc + 1e-6, np.max(np.abs(g.cpu().numpy() - gref)) / sc
    assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7
    again = pkg.loss_and_gradient(icnf, mode, *cargs, eps=dev(eps))[1]
    assert torch.equal(g, again)                                                                      # no atomics
    setsw(pkg, monkeypatch, "CNF_GRAD_LAYERED", "1")
    g2 = pkg.loss_and_gradient(icnf, mode, *cargs, eps=dev(eps))[1]
    delsw(pkg, monkeypatch, "CNF_GRAD_LAYERED")
    assert float((g - g2).abs().max()) < 5e-5 * float(g.abs().max())


def test_layerwise_gradient_agrees_with_the_fused_kernel(pkg, oracles, monkeypatch):
    """cfg2 shape, B = 5000 (several column chunks + a ragged tail): the layer-wise path forced with
    CNF_GRAD_LAYERED=1 against the fused reverse-sweep kernel - two independent implementations."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    lam = (0.02, 0.03, 0.0)
    B = 5000
    p, xs, eps, _ = o64.synth_inputs(spec, B, 9, bias_scale=0.1)
    icnf = make_icnf(pkg, spec, 1, 6, path=2, lambdas=lam)
    v1, g1 = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
    setsw(pkg, monkeypatch, "CNF_GRAD_LAYERED", "1")
    v2, g2 = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
    g2b = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))[1]
    delsw(pkg, monkeypatch, "CNF_GRAD_LAYERED")
    assert abs(float(v1) - float(v2)) < 1e-5
    scale = float(g1.abs().max())
    assert float((g1 - g2).abs().max()) < 2e-5 * scale
    assert torch.equal(g2, g2b)                                    # chunk slabs, fixed order: reproducible


JVP_GRAD_SHAPES = [
    # Hutchinson JVP mode (LuxJacVecMatrixMode): ldot = -<eps, J eps>/K, ndot = |J eps|/K
    (dict(nvars=8, hidden=[64, 64, 64], mode=1), (0.0, 0.0, 0.0), 60, 1, 3),
    (dict(nvars=8, hidden=[64, 64, 64], mode=1, reg_z=True, reg_j=True), (0.02, 0.05, 0.0), 60, 0, 3),
    (dict(nvars=5, naug=2, ncond=4, hidden=[48, 96], act=2, mode=1, nprobes=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.03, 0.02), 41, 1, 2),
    (dict(nvars=32, hidden=[256, 256, 256], mode=1, reg_j=True), (0.0, 0.04, 0.0), 24, 0, 2),
    (dict(nvars=32, hidden=[256, 256, 256], mode=1, reg_z=True), (0.03, 0.0, 0.0), 24, 0, 2),      # no |J eps| term: the VJP twin's cooperative sweep
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2, mode=1, reg_z=True, reg_aug=True), (0.01, 0.0, 0.01), 40, 1, 2),   # default architecture, dealt sweep
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", JVP_GRAD_SHAPES)
def test_parameter_gradient_in_jvp_mode(kw, lam, B, alg, nsteps, pkg, oracles):
    """The gradient in Hutchinson JVP mode against the fp64 autograd oracle.  With the |J eps| regulariser: the layer-wise path
    (pushforward, its reverse, shared top-down pass).  Without it eps^T (J eps) = (eps^T J) eps, the loss is the VJP mode's and the
    library serves the gradient from the VJP mode's fused reverse sweeps (cnf_handle::grad_twin, round 5)."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 321, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam)
    icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    assert icnf.grad_path(mode) == (2 if spec.reg_j else (1 if max(spec.widths[1:-1]) <= 64 else 3))
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps))
    g = g.cpu().numpy().astype(np.float64)
    assert abs(float(val) - L) < 1e-4
    scale = np.abs(gref).max()
    assert np.max(np.abs(g - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(g - gref)) / scale


def test_jvp_mode_gradient_through_the_vjp_twin_agrees_with_its_own_layerwise_gradient(pkg, oracles, monkeypatch):
    """JVP mode without the Jacobian regulariser: the fused VJP-twin gradient against the mode's own layer-wise gradient
    (CNF_JVP_GRAD_TWIN=0), same inputs - and the speed-up that motivates the route, at a batch that fills the chip."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], mode=1, reg_z=True)
    B, alg, nsteps, lam = 4096, 1, 4, (0.02, 0.0, 0.0)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 55, bias_scale=0.2)
    out = {}
    for tag, env in (("twin", "1"), ("own", "0")):
        setsw(pkg, monkeypatch, "CNF_JVP_GRAD_TWIN", env)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        mode = pkg.TrainMode(True)
        assert icnf.grad_path(mode, B=B, alg=alg) == (1 if tag == "twin" else 2)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
    assert abs(out["twin"][0] - out["own"][0]) < 2e-5 * (1 + abs(out["own"][0]))
    for k in (1, 2):
        a, b = out["twin"][k], out["own"][k]
        assert np.max(np.abs(a - b)) < 5e-5 * np.abs(b).max() + 1e-6


def test_several_probes_on_the_frozen_grid_of_an_adaptive_solve(pkg, oracles, monkeypatch):
    """The training step under the reference's default sol_kwargs with K = 2 probes on the default architecture: the adaptive solve runs
    with both probes (their trace estimates enter the error norm), its accepted steps are frozen, and the gradient on that grid is the
    two one-probe dealt sweeps (cnf_grad_path_for(.., on_grid = 1) = 3) - against the layer-wise gradient on the same grid."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=16, naug=17, hidden=[136, 136], act=2, nprobes=2, reg_z=True, reg_j=True, reg_aug=True)
    B, lam = 4100, (0.01, 0.01, 0.01)
    p, xs, eps, _ = o64.synth_inputs(spec, B, 12, bias_scale=0.2)
    out = {}
    for tag, env in (("loop", "1"), ("own", "0")):
        setsw(pkg, monkeypatch, "CNF_PROBE_GRAD_TWIN", env)
        icnf = make_icnf(pkg, spec, 1, 4, path=0, lambdas=lam)
        icnf.sol_kwargs = dict(alg=pkg.Tsit5(), reltol=1e-3, abstol=1e-3)
        mode = pkg.TrainMode(True)
        h = icnf._handle(mode)
        assert h.lib.cnf_grad_path_for(h.ptr, B, 1, 1) == (3 if tag == "loop" else 2)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64), list(icnf.last_solve_stats["tgrid"]))
    assert out["loop"][3] == out["own"][3] and len(out["loop"][3]) >= 3          # the same frozen grid
    assert abs(out["loop"][0] - out["own"][0]) < 2e-5 * (1 + abs(out["own"][0]))
    for k in (1, 2):
        a, b = out["loop"][k], out["own"][k]
        assert np.max(np.abs(a - b)) < 5e-5 * np.abs(b).max() + 1e-6


@pytest.mark.parametrize("solver", ["vcabm", "tsit5"])
@pytest.mark.parametrize("kw,B", [
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, reg_z=True, reg_j=True, reg_aug=True), 1024),      # the reference's PkgBenchmark scenario
    (dict(nvars=8, hidden=[64, 64, 64], mode=2), 777),                                                # TestMode: exact trace
    (dict(nvars=5, ncond=3, hidden=[32, 48, 32, 16], act=2, reg_z=True), 300),                         # generic family, host-loop controller
])
def test_loss_under_an_adaptive_solver_in_one_library_call_is_the_four_calls(solver, kw, B, pkg, oracles):
    """cnf_loss_adaptive (round 5): `loss` of an unsharded batch under VCABM / adaptive Tsit5 as one library call - the same kernels
    in the same order as cnf_assemble_u0 + cnf_solve_* + cnf_epilogue + cnf_loss_mean, so the same bits, the same step record."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 91, bias_scale=0.2)
    icnf = make_icnf(pkg, spec, 1, 8, path=0, lambdas=(0.01, 0.01, 0.01))
    icnf.sol_kwargs = dict(alg=pkg.VCABM() if solver == "vcabm" else pkg.Tsit5(), reltol=1e-4, abstol=1e-4)
    mode = mode_of(pkg, spec)
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val = pkg.loss(icnf, mode, *args, eps=dev(eps))
    st_one = dict(icnf.last_solve_stats)
    logp, regs = pkg.inference(icnf, mode, *args, eps=dev(eps), _raw=True)
    st_four = dict(icnf.last_solve_stats)
    ref = pkg.loss_mean(icnf, mode, logp, regs)
    assert icnf.adaptive and torch.equal(val, ref), (float(val), float(ref))
    for k in ("naccept", "nreject", "nf", "dts", "alg_used", "controller"):
        assert st_one[k] == st_four[k], k
    if solver == "vcabm":
        assert st_one["orders"] == st_four["orders"]


PROBE_LOOP_SHAPES = [
    # several probes on shapes whose one-probe gradient runs on the cooperative / dealt reverse sweep
    (dict(nvars=32, hidden=[256, 256, 256], nprobes=3, reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 40, 0, 2),
    (dict(nvars=20, naug=21, hidden=[168, 168], act=2, nprobes=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 37, 1, 2),   # default architecture
    (dict(nvars=12, hidden=[192, 192, 192], nprobes=4, mode=1, reg_z=True), (0.02, 0.0, 0.0), 33, 1, 2),                           # JVP -> K-probe VJP -> one-probe twin
]


@pytest.mark.parametrize("kw,lam,B,alg,nsteps", PROBE_LOOP_SHAPES)
def test_several_probes_train_probe_by_probe_on_the_cooperative_sweep(kw, lam, B, alg, nsteps, pkg, oracles, monkeypatch):
    """K > 1 probes on wide nets (round 5): the loss is the mean over the probes of the one-probe losses, so the gradient is K
    calls of the one-probe configuration's cooperative reverse sweep (cnf_handle::grad_twin), averaged in probe order - against fp64
    autograd of the K-probe loss and against the configuration's own layer-wise gradient (CNF_PROBE_GRAD_TWIN=0)."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 77, bias_scale=0.2)
    L, gref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam)
    out = {}
    two_hidden = len(spec.widths) == 4
    for tag, env in (("default", "1"), ("loop", "2"), ("own", "0")):
        setsw(pkg, monkeypatch, "CNF_PROBE_GRAD_TWIN", env)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
        # by default two hidden layers take the loop, and three where the one-probe call runs the cooperative gradient's second form
        # (cfg4's 3 x 256: 1.22 x the layer-wise path at K = 4, B = 32 768; with the recomputing sweeps - 3 x 192 - 0.96 x: layer-wise)
        second_form = spec.widths[1:-1] == [256, 256, 256]
        assert icnf.grad_path(mode, B=B, alg=alg) == (3 if tag == "loop" or (tag == "default" and (two_hidden or second_form)) else 2)
        if tag == "default":
            continue
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64))
    scale = np.abs(gref).max()
    assert abs(out["loop"][0] - L) < 1e-4
    assert np.max(np.abs(out["loop"][1] - gref)) < 5e-5 * scale + 1e-6, np.max(np.abs(out["loop"][1] - gref)) / scale
    assert abs(out["loop"][0] - out["own"][0]) < 2e-5 * (1 + abs(L))
    for k in (1, 2):
        a, b = out["loop"][k], out["own"][k]
        assert np.max(np.abs(a - b)) < 5e-5 * np.abs(b).max() + 1e-6


@pytest.mark.parametrize("kw,lam", [
    (dict(nvars=8, hidden=[64, 64, 64]), (0.0, 0.0, 0.0)),                                                  # fused kernel, one probe
    (dict(nvars=6, naug=2, hidden=[64, 64, 64], nprobes=3, reg_z=True, reg_j=True, reg_aug=True), (0.02, 0.03, 0.01)),   # fused probes kernel, augmented
    (dict(nvars=5, ncond=3, hidden=[32, 48, 32, 16], act=2, reg_z=True), (0.02, 0.0, 0.0)),                 # layer-wise path
    (dict(nvars=4, hidden=[40, 40], mode=1, reg_j=True), (0.0, 0.04, 0.0)),                                 # layer-wise path, JVP mode
])
def test_gradient_with_respect_to_the_data(kw, lam, pkg, oracles):
    """dloss/dxs (DI.gradient wrt x in the reference's smoke tests): the costate at t0, written by the same
    reverse sweep — all four gradient implementations against fp64 autograd, ragged batch."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    B = 37
    p, xs, eps, ys = o64.synth_inputs(spec, B, 55, bias_scale=0.2)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, 3, 1, eps, ys, lam, wrt_x=True)
    icnf = make_icnf(pkg, spec, 1, 3, path=0, lambdas=lam)
    mode = pkg.TrainMode(bool(spec.reg_z or spec.reg_j or spec.reg_aug))
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
    assert gx.shape == (spec.nvars, B)
    assert abs(float(val) - L) < 1e-4
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * np.abs(gref).max() + 1e-6
    sx = np.abs(gxref).max()
    assert np.max(np.abs(gx.cpu().numpy().astype(np.float64) - gxref)) < 5e-5 * sx + 1e-7, np.max(np.abs(gx.cpu().numpy() - gxref)) / sx


def test_randomised_gradient_shapes(pkg, oracles):
    """18 random configurations through loss_and_gradient (whatever implementation the library picks:
    fused one-probe / several-probe kernels or the layer-wise path, VJP and JVP modes) against fp64
    autograd, parameters and data gradients."""
    import os
    o64, _ = oracles
    rng = np.random.default_rng(int(os.environ.get("CNF_FUZZ_SEED", 20240711)))
    seen = set()
    for it in range(18):
        D = int(rng.integers(1, 21))
        naug = int(rng.integers(0, min(3, D)))
        C = int(rng.choice([0, 0, 4, 17]))
        L = int(rng.integers(1, 5))
        H = int(rng.choice([8, 24, 48, 64, 96]))
        hidden = [H] * L if rng.integers(0, 2) else [int(rng.choice([8, 16, 40, 64, 80])) for _ in range(L)]
        mode = int(rng.choice([0, 0, 0, 1]))
        K = int(rng.choice([1, 1, 2, 4]))
        reg = bool(rng.integers(0, 2))
        kw = dict(nvars=D - naug, naug=naug, ncond=C, hidden=hidden, act=int(rng.choice([1, 2])), mode=mode, nprobes=K,
                  autonomous=bool(rng.integers(0, 4) == 0), reg_z=reg, reg_j=reg, reg_aug=reg and naug > 0)
        lam = (0.02, 0.03, 0.01) if reg else (0.0, 0.0, 0.0)
        spec = o64.make_spec(**kw)
        alg, nsteps, B = int(rng.integers(0, 2)), int(rng.integers(1, 4)), int(rng.integers(1, 60))
        p, xs, eps, ys = o64.synth_inputs(spec, B, 5000 + it, bias_scale=0.2)
        L64, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, eps, ys, lam, wrt_x=True)
        icnf = make_icnf(pkg, spec, alg, nsteps, path=0, lambdas=lam)
        tm = pkg.TrainMode(reg)
        seen.add(icnf.grad_path(tm))
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
        val, g, gx = pkg.loss_and_gradient(icnf, tm, *args, eps=dev(eps), wrt_x=True)
        assert abs(float(val) - L64) < 1e-4, kw
        sc = np.abs(gref).max()
        assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * sc + 1e-6, (kw, alg, nsteps, B)
        sx = np.abs(gxref).max()
        assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * sx + 1e-7, (kw, alg, nsteps, B)
    if "CNF_FUZZ_SEED" not in os.environ:                              # the default seed exercises both implementations
        assert seen == {1, 2}, seen


def test_randomised_shapes_fused_vs_generic_kernels(pkg, oracles):
    """90 random configurations (D, C, H, L in 1..4, activation, trace mode, regularisers, integrator, ragged
    B): wherever the library picks a fused MFMA instance, its result must agree with the generic
    SIMT kernels — two independent GPU implementations of the same math."""
    o64, _ = oracles
    import os
    rng = np.random.default_rng(int(os.environ.get("CNF_FUZZ_SEED", 20240620)))   # CNF_FUZZ_SEED: extra sweeps by hand
    rng_k = np.random.default_rng(7)                          # probe counts from their own stream (shapes unchanged)
    checked = probes_checked = 0
    for it in range(90):
        D = int(rng.integers(1, 15))
        naug = int(rng.integers(0, min(3, D)))
        C = int(rng.choice([0, 0, 0, 3, 8, 13]))
        H = int(rng.choice([8, 16, 24, 32, 40, 48, 64, 72, 96, 128]))
        L = int(rng.choice([1, 2, 2, 3, 3, 4]))
        act = int(rng.choice([1, 2]))
        mode = int(rng.choice([0, 0, 1, 2]))
        reg = bool(rng.integers(0, 2)) and mode != 2
        hidden = [H] * L
        if rng.integers(0, 3) == 0:                           # non-uniform widths: padded to the widest layer
            hidden = [int(rng.choice([8, 12, 16, 24, 32, 40, 48, 64])) for _ in range(L)]
        K = int(rng_k.choice([1, 1, 2, 3, 4, 8])) if mode == 0 else 1
        kw = dict(nvars=D - naug, naug=naug, ncond=C, hidden=hidden, act=act, mode=mode, nprobes=K,
                  autonomous=bool(rng.integers(0, 4) == 0), reg_z=reg, reg_j=reg, reg_aug=reg and naug > 0)
        spec = o64.make_spec(**kw)
        alg, nsteps, B = int(rng.integers(0, 2)), int(rng.integers(2, 6)), int(rng.integers(1, 90))
        if 2 not in paths_for(pkg, spec, alg, nsteps):
            continue
        p, xs, eps, ys = o64.synth_inputs(spec, B, 1000 + it, bias_scale=0.2)
        a = run_inference(pkg, make_icnf(pkg, spec, alg, nsteps, path=2), spec, p, xs, eps, ys, return_state=True)
        b = run_inference(pkg, make_icnf(pkg, spec, alg, nsteps, path=1), spec, p, xs, eps, ys, return_state=True)
        err = float((a[0] - b[0]).abs().max())
        # two float32 implementations with different summation orders over 2..5 coarse steps: a few ulp of the largest |logp|,
        # and never more than the north_star's 1e-4 (extra seeds reach 5.3e-5 on a 2x8 softplus net)
        tol = 1e-4 * max(1.0, float(b[0].abs().max()) / 128.0)
        assert err < tol, (kw, alg, nsteps, B, err, tol)
        assert float((a[2] - b[2]).abs().max()) < 5e-5 * max(1.0, float(b[2].abs().max()) / 64.0), kw
        for u, v in zip(a[1], b[1]):
            assert float((u - v).abs().max()) < 5e-5 * max(1.0, float(v.abs().max()) / 64.0), kw
        checked += 1
        probes_checked += K > 1
    assert checked >= 45 and probes_checked >= 8, (checked, probes_checked)


@pytest.mark.parametrize("kw", [dict(nvars=8, hidden=[64, 64, 64]),                       # exact-shape instance + gradient image
                                dict(nvars=5, ncond=3, hidden=[40, 24, 40], act=2),        # padded generic instance
                                dict(nvars=32, hidden=[256, 256, 256]),                    # cooperative kernel
                                dict(nvars=3, hidden=[16, 16], mode=2)])                   # tangent engine
def test_device_side_repack_equals_host_repack(kw, pkg, oracles):
    """cnf_set_params on a device pointer runs a gather kernel (no host round trip); on a host pointer
    the same gather after one copy.  Both must give the bits the host packer gives, here observed
    through the solve: device-ps, host-ps and the C oracle's answer."""
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    B = 33
    p, xs, eps, ys = o64.synth_inputs(spec, B, 12, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 0, 4, path=2)
    mode = mode_of(pkg, spec)
    args_d = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    args_h = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (torch.tensor(p), {})
    a = pkg.inference(icnf, mode, *args_d, eps=dev(eps))[0]
    b = pkg.inference(icnf, mode, *args_h, eps=dev(eps))[0]
    assert torch.equal(a, b)
    assert icnf.repack_on_device(mode)
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 4, 0, eps, ys)[0]
    assert np.max(np.abs(a.cpu().numpy() - ref)) < TOL_SOLVE
    # in-place update on the device is picked up (stream-ordered, no synchronisation needed)
    P = dev(p)
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (P, {})
    pkg.inference(icnf, mode, *args, eps=dev(eps))
    P.mul_(0.5)
    c = pkg.inference(icnf, mode, *args, eps=dev(eps))[0]
    ref2 = oc.inference_fixed(spec, (p * np.float32(0.5)).astype(np.float32), xs, 0.0, 1.0, 4, 0, eps, ys)[0]
    assert np.max(np.abs(c.cpu().numpy() - ref2)) < TOL_SOLVE
    assert not torch.equal(a, c)


def test_randomised_shapes_layerwise_vs_generic_kernels(pkg, oracles):
    """The layer-wise GEMM path (csrc/cnf_layered.hip, CNF_PATH_LAYERED) against the thread-per-sample kernels
    on 40 random configurations, including everything the fused kernels do not take: up to 6 hidden layers
    of unequal width up to 300, D up to 40, several JVP probes, all three trace modes, conditions."""
    o64, _ = oracles
    import os
    rng = np.random.default_rng(int(os.environ.get("CNF_FUZZ_SEED", 20240702)))
    for it in range(40):
        D = int(rng.integers(1, 41))
        naug = int(rng.integers(0, min(3, D)))
        C = int(rng.choice([0, 0, 5, 19]))
        L = int(rng.integers(1, 7))
        hidden = [int(rng.choice([7, 16, 33, 64, 100, 128, 200, 300])) for _ in range(L)]
        mode = int(rng.choice([0, 0, 1, 2]))
        if mode == 2:
            D = min(D, 12)                                        # exact trace: D tangents per evaluation
            naug = min(naug, D - 1)
        K = int(rng.choice([1, 1, 2, 3])) if mode != 2 else 1
        reg = bool(rng.integers(0, 2)) and mode != 2
        kw = dict(nvars=D - naug, naug=naug, ncond=C, hidden=hidden, act=int(rng.choice([1, 2])), mode=mode, nprobes=K,
                  autonomous=bool(rng.integers(0, 4) == 0), reg_z=reg, reg_j=reg, reg_aug=reg and naug > 0)
        spec = o64.make_spec(**kw)
        alg, nsteps, B = int(rng.integers(0, 2)), int(rng.integers(1, 4)), int(rng.integers(1, 70))
        p, xs, eps, ys = o64.synth_inputs(spec, B, 3000 + it, bias_scale=0.2)
        a = run_inference(pkg, make_icnf(pkg, spec, alg, nsteps, path=3), spec, p, xs, eps, ys, return_state=True)
        b = run_inference(pkg, make_icnf(pkg, spec, alg, nsteps, path=1), spec, p, xs, eps, ys, return_state=True)
        # (two float32 implementations, different contraction orders: 1e-4, or a few ulp of the largest magnitude where that
        # is larger - an extra seed reached 1.03e-4 at |logp| = 57 on a 300-wide exact-trace net)
        lscale = max(1.0, float(b[0].abs().max()) / 32.0)
        assert float((a[0] - b[0]).abs().max()) < 1e-4 * lscale, (kw, alg, nsteps, B)
        assert float((a[2] - b[2]).abs().max()) < 1e-4 * max(1.0, float(b[2].abs().max()) / 32.0), kw
        for u, v in zip(a[1], b[1]):
            assert float((u - v).abs().max()) < 1e-4, kw


@pytest.mark.parametrize("nv,alg,B", [(48, 1, 70), (56, 0, 33), (63, 1, 45)])
def test_default_architecture_past_the_fused_kernels_runs_layerwise(nv, alg, B, pkg, oracles):
    """ICNF(nvariables >= 48) - two softplus layers of 4 (D + 1) >= 392 units, more than the 24 hidden tiles the cooperative kernels
    hold - runs layer-wise (src/core/icnf.jl:53-103, 517-559): whole solves against the C restatement, TrainMode and TestMode."""
    o64, oc = oracles
    D = 2 * nv + 1
    for kw in (dict(reg_z=True, reg_j=True, reg_aug=True), dict(mode=2)):
        spec = o64.make_spec(nvars=nv, naug=nv + 1, hidden=[4 * (D + 1)] * 2, act=2, **kw)
        p, xs, eps, ys = o64.synth_inputs(spec, B, 900 + nv, bias_scale=0.2)
        icnf = make_icnf(pkg, spec, alg, 3, path=0)
        assert icnf.kernel_family(mode_of(pkg, spec), B=B) == "layered"
        ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 3, alg, eps, ys, nthreads=8)
        logp, regs, _ = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        assert np.max(np.abs(logp.cpu().numpy() - ref[0])) < TOL_SOLVE * max(1.0, float(np.abs(ref[0]).max()) / 32.0)
        for a_, b_ in zip(regs, ref[1]):
            assert np.max(np.abs(a_.cpu().numpy() - b_)) < TOL_SOLVE


def test_auto_path_prefers_fused_then_layerwise(pkg, oracles):
    """AUTO: fused MFMA instance when one covers the configuration, else the layer-wise GEMM path."""
    o64, oc = oracles
    fused = make_icnf(pkg, o64.make_spec(nvars=8, hidden=[64, 64, 64]), 0, 4, path=0)
    assert fused.kernel_path(pkg.TrainMode(False)) == pkg._lib.PATH_MFMA
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64, 64, 64, 64])            # six hidden layers: no fused instance
    icnf = make_icnf(pkg, spec, 1, 3, path=0)
    assert icnf.kernel_path(pkg.TrainMode(False)) == pkg._lib.PATH_LAYERED
    B = 2500
    p, xs, eps, _ = o64.synth_inputs(spec, B, 8, bias_scale=0.2)
    big = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    small = run_inference(pkg, icnf, spec, p, xs[:, :100], eps[:, :100], None)[0].cpu().numpy()
    ref = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 3, 1, eps, None)[0]
    assert np.max(np.abs(big - ref)) < TOL_SOLVE
    assert np.max(np.abs(small - ref[:100])) < TOL_SOLVE


def test_kernel_family_is_reported_by_the_library(pkg, oracles, monkeypatch):
    """cnf_kernel_family / cnf_kernel_family_for / cnf_kernel_name: a host asks the library which kernel organisation serves a
    handle and a call instead of inferring it from CNF_* environment variables."""
    o64, _ = oracles
    delsw(pkg, monkeypatch, "CNF_TILE_SPLIT")
    T = pkg.TrainMode(False)
    cases = [
        (dict(nvars=8, hidden=[64, 64, 64]), 0, T, "per_wave", "mfma_vjp<HT=4"),
        (dict(nvars=32, hidden=[256, 256, 256]), 0, T, "coop", "coop_vjp<HT=16"),
        (dict(nvars=8, ncond=8, hidden=[256, 256, 256], mode=2), 0, pkg.TestMode(), "coopx", "coopx<HT=16"),
        (dict(nvars=16, naug=17, hidden=[136, 136], act=2), 0, T, "coopx", "coopx<HT=12"),
        (dict(nvars=8, hidden=[64] * 6), 0, T, "layered", "layered"),
        (dict(nvars=8, hidden=[64, 64, 64]), 1, T, "simt", "simt"),
    ]
    for kw, path, mode, fam, name in cases:
        icnf = make_icnf(pkg, o64.make_spec(**kw), 0, 4, path=path)
        assert icnf.kernel_family(mode) == fam, kw
        assert icnf.kernel_name(mode).startswith(name), icnf.kernel_name(mode)
        # (one-probe VJP solves of the default architecture's shapes leave the extended kernel for the dealt one above 4096 columns)
        assert icnf.kernel_family(mode, B=65536) == ("coopd" if kw.get("act") == 2 and fam == "coopx" else fam)
        assert icnf.kernel_family(mode, B=4096) == ("tile_split" if fam == "per_wave" and path == 0 else fam)
    small = make_icnf(pkg, o64.make_spec(nvars=8, hidden=[64, 64, 64]), 0, 4)
    assert small.kernel_family(T, B=4096) == "tile_split" and small.kernel_family(T, B=4097) == "per_wave"
    assert small.kernel_family(T, B=4096, whole_solve=False) == "per_wave"
    setsw(pkg, monkeypatch, "CNF_TILE_SPLIT", "0")
    assert small.kernel_family(T, B=4096) == "per_wave"


@pytest.mark.parametrize("kw,B,alg", [
    (dict(nvars=8, hidden=[64, 64, 64]), 20000, 0),                         # per-wave kernel (cfg2's)
    (dict(nvars=8, hidden=[64, 64, 64]), 3000, 1),                          # tile-split kernel
    (dict(nvars=32, hidden=[256, 256, 256]), 2000, 0),                      # cooperative kernel (cfg4's)
    (dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=2), 1500, 0),      # tangent engine, conditioned (cfg5's)
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True), 1000, 1),   # extended cooperative kernel
    (dict(nvars=8, hidden=[64] * 5), 1200, 1),                              # layer-wise path
    (dict(nvars=24, naug=25, hidden=[200, 200], act=2, reg_z=True, reg_j=True, reg_aug=True), 4500, 1),   # dealt kernel + dealt sweep, Runge-Kutta sums in the plan's global ring
    (dict(nvars=32, naug=33, hidden=[264, 264], act=2, reg_z=True, reg_j=True, reg_aug=True), 4200, 1),   # ... the 32-sample form, the 16-sample sweep
])
def test_hot_path_is_hip_graph_capturable_on_a_side_stream(kw, B, alg, pkg, oracles):
    """after the first call has sized the workspaces, every entry point only enqueues work on the caller's
    stream.  cnf_inference_fixed + cnf_loss_mean (and cnf_loss_grad_fixed) are captured into a HIP graph on a NON-default
    stream, the outputs are wiped, the graph is replayed - twice - and must reproduce the eager results bit for bit."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    nsteps = 6
    icnf = make_icnf(pkg, spec, alg, nsteps)
    mode = mode_of(pkg, spec)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 4242, bias_scale=0.2)
    X, E, P = dev(xs), dev(eps), dev(p)
    args = (X,) + ((dev(ys),) if spec.ncond else ()) + (P, {})
    with_grad = spec.mode == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def step():
        logp, regs = pkg.inference(icnf, mode, *args, eps=E, _raw=True)
        out = [logp, regs, pkg.loss_mean(icnf, mode, logp, regs)]
        if with_grad:
            val, g = pkg.loss_and_gradient(icnf, mode, *args, eps=E)
            out += [val, g]
        return out

    with torch.cuda.stream(side):
        step()                                   # first use: workspaces, function attributes, parameter images
        eager = [t.clone() for t in step()]
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        captured = step()
    for rep in range(2):
        for t in captured:
            t.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for a_, b_ in zip(captured, eager):
            assert torch.equal(a_, b_), (rep, kw)
    assert torch.isfinite(eager[2]).item()


def test_two_handles_on_two_streams_run_concurrently(pkg, oracles):
    """Two flows (two cnf_handles, different kernel families) driven from two streams at the same time: no state is shared
    between handles (workspaces, operand images and the tile queue word are per handle), so the interleaved results are the
    serial results bit for bit, for solves and gradients."""
    o64, _ = oracles
    flows = []
    for kw, B, alg in ((dict(nvars=8, hidden=[64, 64, 64]), 30000, 1), (dict(nvars=32, hidden=[256, 256, 256]), 3000, 0),
                       (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), 9000, 1)):
        spec = o64.make_spec(**kw)
        p, xs, eps, ys = o64.synth_inputs(spec, B, 99 + B, bias_scale=0.2)
        flows.append((make_icnf(pkg, spec, alg, 5), mode_of(pkg, spec), (dev(xs), dev(p), {}), dev(eps), torch.cuda.Stream()))
    torch.cuda.synchronize()

    def run(f):
        icnf, mode, args, E, _ = f
        logp, regs = pkg.inference(icnf, mode, *args, eps=E, _raw=True)
        val, g = pkg.loss_and_gradient(icnf, mode, *args, eps=E)
        return logp, regs, val, g

    serial = [[t.clone() for t in run(f)] for f in flows]
    torch.cuda.synchronize()
    for _ in range(6):
        outs = []
        for f in flows:                      # enqueue on every stream before any of them is waited for
            with torch.cuda.stream(f[4]):
                outs.append(run(f))
        torch.cuda.synchronize()
        for o, sres in zip(outs, serial):
            for a_, b_ in zip(o, sres):
                assert torch.equal(a_, b_)


def test_million_column_batch(pkg, oracles):
    """B = 1 000 003 (64-bit column indexing, a ragged last tile, several tile rounds per CU): head, middle and
    tail columns against the C restatement; the gradient at B = 300 007 is additive over a split."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 1_000_003
    rng = np.random.default_rng(12)
    p = o64.glorot_params(spec, rng)
    xs = rng.standard_normal((8, B)).astype(np.float32)
    eps = rng.standard_normal((8, B)).astype(np.float32)
    icnf = make_icnf(pkg, spec, 1, 10)
    logp = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    assert np.isfinite(logp).all()
    for lo in (0, 499_968, B - 67):
        sl = slice(lo, lo + 67)
        ref = oc.inference_fixed(spec, p, xs[:, sl], 0.0, 1.0, 10, 1, eps[:, sl], None)[0]
        assert np.max(np.abs(logp[sl] - ref)) < TOL_SOLVE, lo
    Bg = 300_007
    g_icnf = grad_icnf(pkg, spec, 0, 4)
    X, E, P = dev(xs[:, :Bg]), dev(eps[:, :Bg]), dev(p)
    vf, gf = pkg.loss_and_gradient(g_icnf, pkg.TrainMode(False), X, P, {}, eps=E)
    h = 123_457
    v1, g1 = pkg.loss_and_gradient(g_icnf, pkg.TrainMode(False), X[:, :h], P, {}, eps=E[:, :h])
    v2, g2 = pkg.loss_and_gradient(g_icnf, pkg.TrainMode(False), X[:, h:], P, {}, eps=E[:, h:])
    gs = (g1.double() * h + g2.double() * (Bg - h)) / Bg
    assert float((gf.double() - gs).abs().max()) < 3e-5 * float(gf.abs().max())
    assert abs(float(vf) - (float(v1) * h + float(v2) * (Bg - h)) / Bg) < 2e-5


def test_empty_batch_is_a_no_op(pkg, oracles):
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 4, 1)
    icnf = make_icnf(pkg, spec, 1, 40)
    logp, regs = run_inference(pkg, icnf, spec, p, xs[:, :0], eps[:, :0], None)
    assert logp.shape == (0,) and all(r.shape == (0,) for r in regs)
    for kw in (dict(), dict(alg=pkg.Tsit5())):               # the default solver (VCABM) and adaptive Tsit5: empty and one-column batches
        icnf = make_icnf(pkg, spec, 1, 40)
        icnf.sol_kwargs = dict(reltol=1e-5, abstol=1e-5, **kw)
        logp, regs = run_inference(pkg, icnf, spec, p, xs[:, :0], eps[:, :0], None)
        assert logp.shape == (0,) and all(r.shape == (0,) for r in regs)
        val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(False), dev(xs[:, :0]), dev(p), {}, eps=dev(eps[:, :0]))
        assert g.shape == (p.size,) and bool(torch.isnan(g).all()) and bool(torch.isnan(val))   # a mean over zero columns (Statistics.mean of nothing)
        one = run_inference(pkg, icnf, spec, p, xs[:, :1], eps[:, :1], None)[0]
        ref = run_inference(pkg, make_icnf(pkg, spec, 1, 200), spec, p, xs[:, :1], eps[:, :1], None)[0]
        assert one.shape == (1,) and abs(float(one - ref)) < 1e-3


def test_shard_concatenation_is_bit_identical_and_deterministic(pkg, oracles):
    """Sharding invariance ("bit-for-bit (fixed-step, same kernel)") at the headline shape: evaluating
    column blocks separately - as the ranks of a multi-GPU run do - reproduces the unsharded bits when the shards run on the
    kernel the full batch runs on: (i) shards above the tile-split threshold (more 16-sample tiles than compute units - every
    BASELINE shard size), (ii) any shard size with the tile-split form switched off.  Shards small enough for the tile-split
    kernel (<= 4096 columns) differ from the per-wave kernel's bits by rounding only (the D-row products are summed in a
    different order): equal to 2e-5, and deterministic."""
    import os
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    icnf = make_icnf(pkg, spec, 1, 40)
    # (i) two shards of more than 4096 columns each
    B = 2 * 4112 + 37
    p, xs, eps, _ = o64.synth_inputs(spec, B, 5)
    full = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    again = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    assert np.array_equal(full, again)
    parts = []
    for r in range(2):
        lo, hi = pkg.shard_columns(B, r, 2)
        assert hi - lo > 4096
        parts.append(run_inference(pkg, icnf, spec, p, xs[:, lo:hi], eps[:, lo:hi], None)[0].cpu().numpy())
    assert np.array_equal(np.concatenate(parts), full)
    # (ii) eight small shards
    B = 4096 + 37
    p, xs, eps, _ = o64.synth_inputs(spec, B, 5)
    full = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    old = os.environ.get("CNF_TILE_SPLIT")
    try:
        for env, exact in (("0", True), ("1", False)):
            os.environ["CNF_TILE_SPLIT"] = env
            pkg.reload_tuning()
            parts = []
            for r in range(8):
                lo, hi = pkg.shard_columns(B, r, 8)
                parts.append(run_inference(pkg, icnf, spec, p, xs[:, lo:hi], eps[:, lo:hi], None)[0].cpu().numpy())
            got = np.concatenate(parts)
            if exact:
                assert np.array_equal(got, full)
            else:
                assert np.max(np.abs(got - full)) < 2e-5
                lo, hi = pkg.shard_columns(B, 3, 8)
                assert np.array_equal(parts[3], run_inference(pkg, icnf, spec, p, xs[:, lo:hi], eps[:, lo:hi], None)[0].cpu().numpy())
    finally:
        if old is None:
            os.environ.pop("CNF_TILE_SPLIT", None)
            pkg.reload_tuning()
        else:
            os.environ["CNF_TILE_SPLIT"] = old
            pkg.reload_tuning()


def test_full_size_headline_properties(pkg, oracles):
    """BASELINE headline size (D=8, 3x64, Tsit5x40, B=65536): checks that do not need a CPU
    reference at that size — agreement with the C restatement on a random column subset,
    finiteness, and the loss reduction against a float64 host sum."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 65536
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240614)
    icnf = make_icnf(pkg, spec, 1, 40)
    logp, regs = run_inference(pkg, icnf, spec, p, xs, eps, None)
    lp = logp.cpu().numpy()
    assert np.all(np.isfinite(lp))
    idx = np.random.default_rng(0).choice(B, 512, replace=False)
    ref = oc.inference_fixed(spec, p, xs[:, idx], 0.0, 1.0, 40, 1, eps[:, idx], nthreads=4)[0]
    assert np.max(np.abs(lp[idx] - ref)) < TOL_SOLVE
    sums = pkg.loss_sums(icnf, pkg.TrainMode(False), logp, torch.stack(list(regs))).cpu().numpy()
    assert abs(sums[0] + lp.astype(np.float64).sum()) < 1e-6 * abs(lp.astype(np.float64).sum()) + 1e-2
    loss = float(pkg.loss(icnf, pkg.TrainMode(False), dev(xs), dev(p), {}, eps=dev(eps)))
    assert abs(loss + lp.astype(np.float64).mean()) < 1e-4


def test_full_size_cfg3_k_probe_result_is_mean_of_single_probe_solves(pkg, oracles):
    """cfg3 at full size (RNODE, K=4, B=65536): l̇ and ṅ are linear in the per-probe terms while z's
    trajectory is probe-independent, so the K=4 solve must equal the mean of the
    four single-probe solves — a size-independent check of the K-probe extension."""
    o64, _ = oracles
    B, D, K = 65536, 8, 4
    sk = o64.make_spec(nvars=D, hidden=[64, 64, 64], nprobes=K, reg_z=True, reg_j=True)
    s1 = o64.make_spec(nvars=D, hidden=[64, 64, 64], nprobes=1, reg_z=True, reg_j=True)
    p, xs, eps, _ = o64.synth_inputs(sk, B, 20240615)
    lk, (Ek, nk, _) = run_inference(pkg, make_icnf(pkg, sk, 1, 40), sk, p, xs, eps, None)
    i1 = make_icnf(pkg, s1, 1, 40)
    runs = [run_inference(pkg, i1, s1, p, xs, eps[k * D:(k + 1) * D], None) for k in range(K)]
    lm = torch.stack([r[0] for r in runs]).mean(0)
    nm = torch.stack([r[1][1] for r in runs]).mean(0)
    assert float((lk - lm).abs().max()) < 5e-5
    assert float((nk - nm).abs().max()) < 5e-5
    assert float((Ek - runs[0][1][0]).abs().max()) < 5e-5
    assert bool(torch.isfinite(lk).all())


def test_full_size_cfg4_shards_concatenate_bit_identically(pkg, oracles):
    """cfg4 per-GPU size (D=32, 3x256, RK4x40, B=32768, cooperative kernel): two half-batches —
    what two ranks would evaluate — reproduce the unsharded bits; a column subset agrees with the
    C restatement."""
    o64, oc = oracles
    spec = o64.make_spec(nvars=32, hidden=[256, 256, 256])
    B = 32768
    p, xs, eps, _ = o64.synth_inputs(spec, B, 20240616)
    icnf = make_icnf(pkg, spec, 0, 40)
    assert icnf.kernel_path(pkg.TrainMode(False)) == 2
    full = run_inference(pkg, icnf, spec, p, xs, eps, None)[0]
    h = B // 2
    a = run_inference(pkg, icnf, spec, p, xs[:, :h], eps[:, :h], None)[0]
    b = run_inference(pkg, icnf, spec, p, xs[:, h:], eps[:, h:], None)[0]
    assert torch.equal(torch.cat([a, b]), full)
    idx = np.random.default_rng(1).choice(B, 96, replace=False)
    ref = oc.inference_fixed(spec, p, xs[:, idx], 0.0, 1.0, 40, 0, eps[:, idx], nthreads=4)[0]
    assert np.max(np.abs(full.cpu().numpy()[idx] - ref)) < TOL_SOLVE


def test_full_size_cfg5_exact_trace_equals_unit_probe_hutchinson(pkg, oracles):
    """cfg5 at full size (cond D=8+8, 3x128, exact trace, B=16384): the exact trace is the sum of
    the D unit-vector probes, i.e. D times the K=D Hutchinson mean with probes e_1..e_D — checked
    through two different code paths (tangent engine, exact seeds vs generic SIMT K-probe VJP) on a
    slice, plus the C restatement on a column subset."""
    o64, oc = oracles
    D, C, B = 8, 8, 16384
    se = o64.make_spec(nvars=D, ncond=C, hidden=[128, 128, 128], mode=2)
    p, xs, _, ys = o64.synth_inputs(se, B, 20240617)
    icnf = make_icnf(pkg, se, 0, 40)
    assert icnf.kernel_path(pkg.TestMode()) == 2
    full = run_inference(pkg, icnf, se, p, xs, None, ys)[0].cpu().numpy()
    assert np.all(np.isfinite(full))
    idx = np.random.default_rng(2).choice(B, 128, replace=False)
    ref = oc.inference_fixed(se, p, xs[:, idx], 0.0, 1.0, 40, 0, None, ys[:, idx], nthreads=4)[0]
    assert np.max(np.abs(full[idx] - ref)) < TOL_SOLVE
    # K = D unit probes through the Hutchinson VJP path: dlogp_exact = D * dlogp_K
    sk = o64.make_spec(nvars=D, ncond=C, hidden=[128, 128, 128], nprobes=D)
    n = 256
    onehot = np.tile(np.eye(D, dtype=np.float32).reshape(D * D, 1), (1, n))
    ik = make_icnf(pkg, sk, 0, 40)
    _, _, uk = run_inference(pkg, ik, sk, p, xs[:, :n], onehot, ys[:, :n], return_state=True)
    _, _, ue = run_inference(pkg, icnf, se, p, xs[:, :n], None, ys[:, :n], return_state=True)
    assert float((ue[:D] - uk[:D]).abs().max()) < 2e-5
    assert float((ue[D] - D * uk[D]).abs().max()) < 1e-4


FULL_SIZE = [
    # BASELINE.json configs at their full batch sizes: (name, make_spec kwargs, B, alg)
    ("cfg2", dict(nvars=8, hidden=[64, 64, 64]), 65536, 0),          # the benched instantiation: RK4 x 40 at B = 65 536
    ("cfg2p", dict(nvars=8, hidden=[64, 64, 64]), 65536, 1),
    ("cfg3", dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), 65536, 1),
    ("cfg4", dict(nvars=32, hidden=[256, 256, 256]), 32768, 0),
    ("cfg4full", dict(nvars=32, hidden=[256, 256, 256]), 262144, 0),  # all 262 144 columns of BASELINE config 4 on one GPU
    ("cfg5", dict(nvars=8, ncond=8, hidden=[128, 128, 128], mode=2), 16384, 0),
]


@pytest.mark.parametrize("name,kw,B,alg", FULL_SIZE)
def test_full_size_every_column_matches_the_c_restatement(name, kw, B, alg, pkg, oracles):
    """All B columns of the full-size solve (40 steps) against oracle/cnf_oracle.c on every host core:
    max |dlogp| < 1e-4 (north_star), plus the regulariser rows and the final state."""
    import os
    o64, oc = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 20240700 + len(name))
    icnf = make_icnf(pkg, spec, alg, 40)
    logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
    assert icnf.kernel_path(mode_of(pkg, spec)) == 2
    nt = max(1, min(os.cpu_count() or 1, oc.max_threads()))
    # every column when the host has the cores for it (at most ~20 s of CPU work per config); otherwise a
    # regular subsample of the columns, so the suite stays bounded on small hosts
    cost = {"cfg2": 0.7, "cfg2p": 1.0, "cfg3": 2.5, "cfg4": 10.0, "cfg4full": 10.0, "cfg5": 10.0}[name]       # relative CPU cost per sample*step
    budget = int(45.0 * 2.5e3 * nt / (40 * cost))
    stride = max(1, -(-B // max(budget, 256)))
    idx = np.arange(0, B, stride)
    sub = lambda a: None if a is None else np.ascontiguousarray(a[:, idx])
    ref_logp, ref_regs, ref_u = oc.inference_fixed(spec, p, sub(xs), 0.0, 1.0, 40, alg, sub(eps), sub(ys), nthreads=nt)
    err = np.abs(logp.cpu().numpy()[idx] - ref_logp)
    assert err.max() < TOL_SOLVE, (name, float(err.max()), int(err.argmax()))
    for a, b in zip(regs, ref_regs):
        assert np.max(np.abs(a.cpu().numpy()[idx] - b)) < TOL_SOLVE
    assert np.max(np.abs(u1.cpu().numpy()[:, idx] - ref_u)) < TOL_SOLVE
    print(f"{name}: B={B} columns checked={idx.size} max|dlogp|={err.max():.2e} mean={err.mean():.2e} ({nt} CPU threads)")


def _fixture_flow(pkg, o64, name, sol_kwargs, tspan=(0.0, 1.0), path=0):
    import json, os
    from conftest import GOLDEN
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    spec = o64.make_spec(**json.loads(str(d["spec"])))
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
    icnf = pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nconditions=spec.ncond, nn=pkg.Chain(*layers),
                    compute_mode=pkg.HIPVecJacMatrixMode(kernel_path=path), steer_rate=0.0, tspan=tspan,
                    lambda1=0.01 if spec.reg_z else 0.0, lambda2=0.01 if spec.reg_j else 0.0,
                    lambda3=0.01 if spec.reg_aug else 0.0, device="cuda:0", sol_kwargs=sol_kwargs)
    return d, spec, icnf


GENERATE_FIXTURES = ["generate_cfg2p", "generate_cfg5_cond_exact", "generate_aug_softplus"]


@pytest.mark.parametrize("name", GENERATE_FIXTURES)
def test_generate_matches_the_oracle_fixture(name, pkg, oracles):
    """`generate` (src/core/base_icnf.jl:351-404,185-194) = the reversed-tspan solve from a base sample z0: the samples
    against the committed fp64 fixture, and the whole final state through cnf_integrate_fixed(t1 -> t0) on every kernel
    family that serves the shape, against the fixture and against the oracle run now (so far only a round
    trip with the path's own forward pass)."""
    import ctypes as C
    o64, _ = oracles
    alg_of = lambda d: pkg.Tsit5() if int(d["alg"]) == 1 else pkg.RK4()
    d, spec, icnf = _fixture_flow(pkg, o64, name, None)
    icnf.sol_kwargs = dict(alg=alg_of(d), adaptive=False, nsteps=int(d["nsteps"]))
    mode = mode_of(pkg, spec)
    B = d["z0"].shape[1]
    ys = d.get("ys")
    args = ((dev(ys),) if spec.ncond else ()) + (dev(d["p"]), {}, B)
    x = pkg.generate(icnf, mode, *args, z0=dev(d["z0"]), eps=dev(d["eps"]))
    assert x.shape == (spec.nvars, B)
    assert np.max(np.abs(x.cpu().numpy() - d["u1"][:spec.nvars])) < TOL_SOLVE
    u0 = np.concatenate([d["z0"].astype(np.float64), np.zeros((3, B))], 0)
    ref = o64.integrate_fixed(spec, d["p"], u0, 1.0, 0.0, int(d["nsteps"]), int(d["alg"]), d["eps"], ys)
    assert np.max(np.abs(ref - d["u1"])) < 1e-12                      # the fixture is what the oracle computes today
    for path in paths_for(pkg, spec, int(d["alg"]), int(d["nsteps"])):
        _, _, ic = _fixture_flow(pkg, o64, name, dict(alg=alg_of(d), adaptive=False, nsteps=int(d["nsteps"])), path=path)
        h = ic._handle(mode)
        ic._bind_params(h, dev(d["p"]))
        U0 = torch.tensor(u0.T.astype(np.float32).copy(), device="cuda:0")          # (B, S): column-major S x B
        U1 = torch.empty_like(U0)
        E = dev(d["eps"]).t().contiguous()
        Y = dev(ys).t().contiguous() if ys is not None else None
        pkg._lib.check(h.lib.cnf_integrate_fixed(h.ptr, int(d["alg"]), int(d["nsteps"]), 1.0, 0.0, pkg._lib.ptr(U0), pkg._lib.ptr(E),
                                                 pkg._lib.ptr(Y), B, pkg._lib.ptr(U1), pkg._lib.stream_ptr(torch.device("cuda:0"))))
        err = np.max(np.abs(U1.t().cpu().numpy() - d["u1"]))
        assert err < TOL_SOLVE, (name, path, err)


@pytest.mark.parametrize("name", GENERATE_FIXTURES)
def test_fixed_dt_stepping_takes_a_shorter_last_step(name, pkg, oracles):
    """sol_kwargs = (alg, adaptive = false, dt) on a span that is not a multiple of dt - what STEER's drawn end time meets
    (src/core/base_icnf.jl:23-43): steps of dt and a shorter last step onto t1, as OrdinaryDiffEq takes them
    (cnf_inference_fixed_dt / cnf_integrate_fixed_dt), on every kernel family; forwards against the fixture, backwards
    (generate) against the oracle run now (round 1 took round(span / dt) EQUAL steps instead)."""
    o64, _ = oracles
    d0, spec, _ = _fixture_flow(pkg, o64, name, None)
    alg, nsteps, t1 = int(d0["alg"]), int(d0["nsteps"]), float(d0["t1_steer"])
    mode = mode_of(pkg, spec)
    ys = d0.get("ys")
    B = d0["z0"].shape[1]
    grid = o64.fixed_dt_grid(0.0, t1, 1.0 / nsteps)
    assert abs((grid[-1] - grid[-2]) - (grid[1] - grid[0])) > 1e-3      # a genuinely shorter last step
    assert np.allclose(pkg.ICNF.fixed_dt_grid(0.0, t1, 1.0 / nsteps), grid, rtol=0, atol=1e-12)   # the host's plan = the oracle's
    for path in paths_for(pkg, spec, alg, nsteps):
        d, _, icnf = _fixture_flow(pkg, o64, name, dict(alg=pkg.Tsit5() if alg == 1 else pkg.RK4(), adaptive=False, dt=1.0 / nsteps),
                                   tspan=(0.0, t1), path=path)
        xs = d["z0"][:spec.nvars]
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(d["p"]), {})
        logp, (E, n, A), u1 = pkg.inference(icnf, mode, *args, eps=dev(d["eps"]), return_state=True)
        assert np.max(np.abs(logp.cpu().numpy() - d["logp_steer"])) < TOL_SOLVE, (name, path)
        assert np.max(np.abs(u1.cpu().numpy() - d["u1_steer"])) < TOL_SOLVE
        for got, key in ((E, "E_steer"), (n, "n_steer"), (A, "A_steer")):
            assert np.max(np.abs(got.cpu().numpy() - d[key])) < TOL_SOLVE
        # (equal steps to the same t1 - round 1's behaviour - is a different discretisation of the same ODE; on these smooth
        # random-init fields both are converged to ~1e-13 in fp64, so the state cannot tell them apart: the step sequence can)
        # backwards with the same dt: generate over the reversed span
        gargs = ((dev(ys),) if spec.ncond else ()) + (dev(d["p"]), {}, B)
        x = pkg.generate(icnf, mode, *gargs, z0=dev(d["z0"]), eps=dev(d["eps"]))
        u0 = np.concatenate([d["z0"].astype(np.float64), np.zeros((3, B))], 0)
        ref = o64.integrate_fixed_dt(spec, d["p"], u0, t1, 0.0, 1.0 / nsteps, alg, d["eps"], ys)
        assert np.max(np.abs(x.cpu().numpy() - ref[:spec.nvars])) < TOL_SOLVE, (name, path)


def test_steer_draws_t1_once_and_steps_with_dt(pkg, oracles):
    """STEER end to end (steer_rate != 0, TrainMode{true}; src/core/base_icnf.jl:23-43): t1 <- t1 + |t1 - t0| U(-rate, rate) drawn
    once per call for the whole batch, then fixed-dt steps with a shorter last one; loss and its gradient are those of the
    oracle on that grid.  TrainMode{false} and TestMode leave tspan alone."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    B, rate, nsteps = 48, 0.25, 10
    p, xs, eps, _ = o64.synth_inputs(spec, B, 77, bias_scale=0.1)
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
    icnf = pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), steer_rate=rate, lambda1=0.02, lambda2=0.03, lambda3=0.0,
                    device="cuda:0", sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, dt=1.0 / nsteps))
    lam = (0.02, 0.03, 0.0)
    for seed in (3, 4):
        g = torch.Generator(device="cpu").manual_seed(seed)
        r = (torch.rand((), generator=g, dtype=torch.float32).item() * 2.0 - 1.0) * rate
        t1 = 1.0 + r
        icnf.steer_rng.manual_seed(seed)
        val = float(pkg.loss(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps)))
        logp, (E, n, A), _ = o64.inference_fixed(spec, p, xs, 0.0, t1, 0, 1, eps, None, dt=1.0 / nsteps)
        ref = float(np.mean(-logp + lam[0] * E + lam[1] * n))
        assert abs(val - ref) < 1e-4, (seed, t1, val, ref)
        icnf.steer_rng.manual_seed(seed)
        v2, grad = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
        grid = o64.fixed_dt_grid(0.0, t1, 1.0 / nsteps)
        Lr, gr = o64.loss_and_grad(spec, p, xs, 0.0, t1, len(grid) - 1, 1, eps, None, lambdas=lam, tgrid=grid)
        assert abs(float(v2) - Lr) < 1e-4 and abs(float(v2) - val) < 1e-5
        assert np.max(np.abs(grad.cpu().numpy() - gr)) < 5e-5 * max(1.0, float(np.max(np.abs(gr))))
    # no steering outside TrainMode{true}
    icnf.steer_rng.manual_seed(3)
    a = float(pkg.loss(icnf, pkg.TrainMode(False), dev(xs), dev(p), {}, eps=dev(eps)))
    logp = o64.inference_fixed(spec, p, xs, 0.0, 1.0, nsteps, 1, eps, None)[0]
    assert abs(a + float(np.mean(logp))) < 1e-4


def test_gradient_under_the_default_solver_states_its_discretisation(pkg, oracles):
    """ADVICE r1 / VERDICT r1 #7: with the default solver (VCABM) `loss()` integrates with VCABM while `loss_and_gradient()`
    differentiates an adaptive Tsit5 solve with frozen steps.  The substitution is recorded in last_solve_stats and the two
    values agree to the solver tolerance; with a fixed-step solver they are the same discretisation."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    p, xs, eps, _ = o64.synth_inputs(spec, 512, 78, bias_scale=0.1)
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], ACTS[spec.acts[i]]) for i in range(len(spec.acts))]
    mk = lambda kw: pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(*layers), steer_rate=0.0, lambda1=0.01, lambda2=0.01,
                             lambda3=0.0, device="cuda:0", sol_kwargs=kw)
    m = pkg.TrainMode(True)
    dflt = mk(dict())                                               # all defaults: VCABM, reltol = abstol = 1e-4
    lv = float(pkg.loss(dflt, m, dev(xs), dev(p), {}, eps=dev(eps)))
    assert dflt.last_solve_stats["alg_used"] == "VCABM"
    gv, _ = pkg.loss_and_gradient(dflt, m, dev(xs), dev(p), {}, eps=dev(eps))
    st = dflt.last_solve_stats
    assert st["alg_used"] == "Tsit5" and "frozen" in st["gradient_of"]
    assert abs(lv - float(gv)) < 2e-2                                # two adaptive discretisations at tolerance 1e-4
    fine = mk(dict(alg=pkg.Tsit5(), adaptive=False, nsteps=80))
    lf = float(pkg.loss(fine, m, dev(xs), dev(p), {}, eps=dev(eps)))
    assert abs(lv - lf) < 2e-2 and abs(float(gv) - lf) < 2e-3        # both within tolerance of a fine fixed-step solve
    gf, _ = pkg.loss_and_gradient(fine, m, dev(xs), dev(p), {}, eps=dev(eps))
    assert abs(float(gf) - lf) < 1e-6 and "exact discrete adjoint" in fine.last_solve_stats["gradient_of"]


def test_params_binding_contract(pkg, oracles):
    """writes that bypass the tensor's version counter need invalidate_params(); a ps on another device is refused;
    TestMode gradients take one probe's worth of eps like TestMode inference."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=4, hidden=[32, 32])
    p, xs, eps, _ = o64.synth_inputs(spec, 64, 79, bias_scale=0.1)
    icnf = make_icnf(pkg, spec, 1, 8)
    P = dev(p)
    m = pkg.TrainMode(False)
    a = pkg.inference(icnf, m, dev(xs), P, {}, eps=dev(eps))[0].clone()
    P.data.mul_(1.5)                                                 # no version bump through .data
    stale = pkg.inference(icnf, m, dev(xs), P, {}, eps=dev(eps))[0]
    assert torch.equal(stale, a)                                     # the documented contract: the handle still holds the old weights
    icnf.invalidate_params()
    fresh = pkg.inference(icnf, m, dev(xs), P, {}, eps=dev(eps))[0]
    ref = o64.inference_fixed(spec, (p * 1.5).astype(np.float32), xs, 0.0, 1.0, 8, 1, eps)[0]
    assert np.max(np.abs(fresh.cpu().numpy() - ref)) < TOL_SOLVE and not torch.equal(fresh, a)
    if torch.cuda.device_count() > 1:
        with pytest.raises(ValueError):
            pkg.inference(icnf, m, dev(xs), P.to("cuda:1"), {}, eps=dev(eps))
    icnf.nprobes = 3                                                 # TrainMode uses 3 probes, TestMode ignores them
    v, g = pkg.loss_and_gradient(icnf, pkg.TestMode(), dev(xs), P, {}, eps=dev(eps))       # eps of ONE probe is accepted
    lt = pkg.loss(icnf, pkg.TestMode(), dev(xs), P, {}, eps=dev(eps))
    assert abs(float(v) - float(lt)) < 1e-5 and bool(torch.isfinite(g).all())


def test_generate_inverts_inference(pkg, oracles):
    """generate integrates the reversed tspan (src/core/base_icnf.jl:372): pushing x forward
    to z and pulling z back must return x (integrator error only)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 256, 9, bias_scale=0.1)
    icnf = make_icnf(pkg, spec, 1, 40)
    _, _, u1 = run_inference(pkg, icnf, spec, p, xs, eps, None, return_state=True)
    back = pkg.generate(icnf, pkg.TrainMode(False), dev(p), {}, 256, z0=u1[:8].contiguous(), eps=dev(eps))
    assert np.max(np.abs(back.cpu().numpy() - xs)) < 1e-4
    samples = pkg.generate(icnf, pkg.TrainMode(False), dev(p), {}, 100)
    assert samples.shape == (8, 100) and bool(torch.isfinite(samples).all())


def test_layer_call_and_drawn_probes(pkg, oracles):
    """(icnf)(xs, ps, st) uses TrainMode{false} and draws eps from icnf.rng
    (src/core/base_icnf.jl:509-515, 258-259): two calls give different Hutchinson estimates
    whose mean error against the exact trace shrinks; TestMode is deterministic."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=2, hidden=[32, 32])
    p, xs, _, _ = o64.synth_inputs(spec, 64, 3, bias_scale=0.1)
    icnf = make_icnf(pkg, spec, 1, 20)
    a, st = icnf(dev(xs), dev(p), {})
    b, _ = icnf(dev(xs), dev(p), {})
    assert st == {} and not torch.equal(a, b)
    e1 = pkg.inference(icnf, pkg.TestMode(), dev(xs), dev(p), {})[0]
    e2 = pkg.inference(icnf, pkg.TestMode(), dev(xs), dev(p), {})[0]
    assert torch.equal(e1, e2)
    assert float((a - e1).abs().max()) < 5.0


@pytest.mark.parametrize("use_bias,conditioned", [(True, False), (False, False), (True, True)])
def test_planar_layer_net_matches_the_equivalent_dense_chain(use_bias, conditioned, pkg, oracles):
    """The reference's alternative nn (test/ci_tests/smoke_tests.jl:32-46): ICNF(nn = Chain(PlanarLayer(...))),
    inference in TrainMode{true} and TestMode, against the oracle run on the equivalent Dense chain."""
    o64, oc = oracles
    nv, C = 2, (2 if conditioned else 0)
    D = 2 * nv + 1                                           # nvariables + naugments (default nvariables + 1)
    n_in = D + 1 + C
    rng = np.random.default_rng(5)
    u, w = rng.uniform(-0.7, 0.7, D), rng.uniform(-0.7, 0.7, n_in)
    b = rng.uniform(-0.3, 0.3, 1) if use_bias else np.zeros(0)
    ps = np.concatenate([u, w, b]).astype(np.float32)
    B = 50
    xs = rng.standard_normal((nv, B)).astype(np.float32)
    eps = rng.standard_normal((D, B)).astype(np.float32)
    ys = rng.standard_normal((C, B)).astype(np.float32) if C else None
    for mode_id, mode in ((0, pkg.TrainMode(True)), (2, pkg.TestMode())):
        spec = o64.Spec(nvars=nv, naug=nv + 1, ncond=C, widths=[n_in, 1, D], acts=[1, 0], mode=mode_id,
                        reg_z=mode_id == 0, reg_j=mode_id == 0, reg_aug=mode_id == 0)
        p_dense = np.concatenate([w, b if use_bias else np.zeros(1), u, np.zeros(D)]).astype(np.float32)
        ref = oc.inference_fixed(spec, p_dense, xs, 0.0, 1.0, 20, 1, eps, ys, nthreads=2)
        icnf = pkg.ICNF(nvariables=nv, nconditions=C, nn=pkg.Chain(pkg.PlanarLayer(n_in, D, pkg.tanh, use_bias=use_bias)),
                        steer_rate=0.0, device="cuda:0", sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=20))
        args = (dev(xs),) + ((dev(ys),) if C else ()) + (dev(ps), {})
        logp, (E, n, A) = pkg.inference(icnf, mode, *args, eps=dev(eps))
        assert np.max(np.abs(logp.cpu().numpy() - ref[0])) < TOL_SOLVE
        for a_, b_ in zip((E, n, A), ref[1]):
            assert np.max(np.abs(a_.cpu().numpy() - b_)) < TOL_SOLVE
    samples = pkg.generate(icnf, pkg.TestMode(), *(((dev(ys),) if C else ()) + (dev(ps), {}, B)))
    assert samples.shape == (nv, B) and bool(torch.isfinite(samples).all())


@pytest.mark.parametrize("mode,config,batch", [("infer", "cfg2", 4096), ("grad", "cfg2", 4096),
                                               ("grad", "cfg4", 512)])    # cfg4: cooperative reverse sweep + reduce_gradient
def test_bench_contract_with_two_ranks_on_one_gpu(mode, config, batch):
    """The N > 1 path of bench.py end to end, as the driver launches it (torch.distributed.run, one process
    per rank, barrier + max-over-ranks timing, one JSON line from rank 0) — here two ranks sharing the one
    GPU with the gloo backend, since RCCL refuses two ranks on one device.  Checks the contract fields
    and that the job-wide loss is the mean over both ranks' different column blocks."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-cpu-baseline", "--batch", str(batch), "--mode", mode, "--config", config, "--preroll-seconds", "0.2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_columns"] == 2 * batch and out["config"]["columns_per_gpu"] == batch and out["config"]["name"] == config
    assert out["value"] > 0 and abs(out["value"] - 2 * batch * 40 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    # every rank reported what it bound and the group it reduced on (gloo: no library communicator, so cnf_comm_size = -1)
    assert [r["rank"] for r in out["ranks_seen"]] == [0, 1] and all(r["process_group_size"] == 2 for r in out["ranks_seen"])
    assert "secondaries" not in out      # the extra workloads belong to the default one-GPU line only
    if mode == "grad":
        assert ("cooperative" in out["config"]["gradient_path"]) == (config == "cfg4")
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                             "--batch", str(batch), "--mode", mode, "--config", config, "--preroll-seconds", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    one = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][0])
    # rank 0's block is the single-process block; the two-rank mean differs from it (rank 1 holds other columns)
    assert abs(out["loss"] - one["loss"]) > 1e-6 and abs(out["loss"] - one["loss"]) < 0.5


def _fit_icnf(pkg, nvars, ncond=0, naug=0):
    return pkg.ICNF(nvariables=nvars, naugments=naug, nconditions=ncond, steer_rate=0.0, lambda1=0.0, lambda2=0.0, lambda3=0.0,
                    device="cuda:0", sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=8))


def test_mlj_style_fit_transform_and_distribution_wrapper(pkg):
    """ICNFModel.fit (shuffled mini-batches, WeightDecay + Adam on loss(TrainMode{true}), every step one
    loss_and_gradient call) learns a shifted, scaled 1-D Gaussian; transform returns densities; the
    ICNFDist wrapper's pdf integrates to 1 in TestMode (exact trace: a CNF is a normalised density) and its
    samples have the data's moments (src/exts/mlj_ext/core_icnf.jl, src/exts/dist_ext/core_icnf.jl)."""
    g = torch.Generator().manual_seed(5)
    X = 2.0 + 0.5 * torch.randn(4096, 1, generator=g)
    icnf = _fit_icnf(pkg, 1)
    seen = []
    model = pkg.ICNFModel(icnf=icnf, batchsize=1024, epochs=60, eta=5e-3, callback=lambda it, l: seen.append(l) or False,
                          shuffle_rng=torch.Generator().manual_seed(1), init_rng=torch.Generator().manual_seed(2))
    fitresult, cache, report = model.fit(X)
    assert cache is None and report["stats"]["iterations"] == 60 * 4 == len(seen)
    entropy = 0.5 * np.log(2 * np.pi * np.e * 0.25)                      # NLL of the true density: 0.726
    assert seen[0] > np.mean(seen[-8:]) + 0.3 and np.mean(seen[-8:]) < entropy + 0.08, (seen[0], np.mean(seen[-8:]))
    px = model.transform(fitresult, X[:100])
    assert list(px.columns) == ["px"] and len(px) == 100 and (px["px"] > 0).all()
    d = pkg.ICNFDist.from_fit(model, fitresult, pkg.TestMode())
    assert len(d) == 1
    grid = torch.linspace(-2.0, 6.0, 4001)[None, :]
    p = d.pdf(grid).double().cpu()
    integral = float(torch.trapezoid(p, grid[0].double()))
    assert abs(integral - 1.0) < 2e-3, integral
    assert abs(float(d.logpdf(torch.tensor([2.0]))) - float(d.logpdf(torch.tensor([[2.0]]))[0])) < 1e-6
    s = d.rand(20000)
    assert s.shape == (1, 20000) and d.rand().shape == (1,)
    assert abs(float(s.mean()) - 2.0) < 0.05 and abs(float(s.std()) - 0.5) < 0.05
    assert set(model.fitted_params(fitresult)) == {"learned_parameters", "states"}


def test_conditioned_fit_and_distribution_wrapper(pkg):
    """CondICNFModel: x | y ~ N(y, 0.3^2); after a short fit the conditional density peaks near y, and
    CondICNFDist uses the first n condition columns for n points (src/exts/dist_ext/core_cond_icnf.jl:45)."""
    g = torch.Generator().manual_seed(6)
    Y = torch.rand(4096, 1, generator=g) * 4 - 2
    X = Y + 0.3 * torch.randn(4096, 1, generator=g)
    icnf = _fit_icnf(pkg, 1, ncond=1)
    model = pkg.CondICNFModel(icnf=icnf, batchsize=0, epochs=150, eta=5e-3, callback=None, init_rng=torch.Generator().manual_seed(3))
    fitresult, _, report = model.fit((X, Y))
    assert report["stats"]["iterations"] == 150                            # batchsize 0 = full batch
    px = model.transform(fitresult, (X[:64], Y[:64]))
    assert (px["px"] > 0).all()
    ys = torch.tensor([[-1.0, 1.0]])
    d = pkg.CondICNFDist.from_fit(model, fitresult, pkg.TestMode(), ys)
    at_own = d.logpdf(torch.tensor([[-1.0, 1.0]]))                         # x = y for both columns
    swapped = d.logpdf(torch.tensor([[1.0, -1.0]]))                        # x = -y
    assert float(at_own.min()) > float(swapped.max()) + 2.0, (at_own, swapped)
    assert d.rand(2).shape == (1, 2)
    with pytest.raises(IndexError):
        d.logpdf(torch.zeros(1, 3))


def _adapter_flow(pkg, o64, nvars, naug=0, ncond=0, hidden=(24, 24), act="softplus", nsteps=10, lambdas=(0.01, 0.01, 0.01), epsdist=None):
    """A small flow for the adapter tests, its Spec for the fp64 oracle (TrainMode: Hutchinson VJP with the regularisers the
    lambdas switch on; TestMode: exact trace) and the parameters LuxCore.setup would give it from a fixed generator."""
    D = nvars + naug
    widths = [D + 1 + ncond] + list(hidden) + [D]
    layers = [pkg.Dense(widths[i], widths[i + 1], act if i + 1 < len(widths) - 1 else "identity") for i in range(len(widths) - 1)]
    icnf = pkg.ICNF(nvariables=nvars, naugments=naug, nconditions=ncond, nn=pkg.Chain(*layers), steer_rate=0.0,
                    lambda1=lambdas[0], lambda2=lambdas[1], lambda3=lambdas[2], device="cuda:0", epsdist=epsdist,
                    sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=nsteps) if nsteps else None)
    kw = dict(nvars=nvars, naug=naug, ncond=ncond, hidden=list(hidden), act=2 if act == "softplus" else 1)
    train = o64.make_spec(reg_z=lambdas[0] != 0, reg_j=lambdas[1] != 0, reg_aug=lambdas[2] != 0 and naug > 0, **kw)
    test = o64.make_spec(mode=2, **kw)
    return icnf, train, test


def test_distribution_wrappers_logpdf_against_the_oracle(pkg, oracles):
    """ICNFDist / CondICNFDist (src/exts/dist_ext/core_icnf.jl:36-75, core_cond_icnf.jl:45): `logpdf(d, A)` is `inference` on
    the wrapped flow - TestMode (exact trace) against the fp64 oracle on the same parameters, fixed-step and under the
    reference's DEFAULT sol_kwargs (VCABM at 1e-4: the restatement of oracle/cnf_oracle64.py and a fine fixed-step solve);
    a point given as a vector is the one-column matrix; the conditioned wrapper takes the FIRST n condition columns."""
    o64, _ = oracles
    rng = np.random.default_rng(31)
    # ---- unconditioned, augmented, fixed-step ----
    icnf, _, spec = _adapter_flow(pkg, o64, nvars=3, naug=2)
    ps, st = pkg.setup(torch.Generator().manual_seed(7), icnf)
    p = ps.numpy().astype(np.float32)
    A = rng.standard_normal((3, 37)).astype(np.float32)
    d = pkg.ICNFDist(icnf, pkg.TestMode(), ps.to("cuda:0"), st)
    got = d.logpdf(torch.tensor(A)).cpu().numpy()
    ref = o64.inference_fixed(spec, p, A, 0.0, 1.0, 10, 1, None)[0]
    assert len(d) == 3 and got.shape == (37,) and np.max(np.abs(got - ref)) < TOL_SOLVE, np.max(np.abs(got - ref))
    assert abs(float(d.logpdf(torch.tensor(A[:, 5]))) - ref[5]) < TOL_SOLVE                 # a vector: one point
    assert np.max(np.abs(d.pdf(torch.tensor(A)).cpu().numpy() - np.exp(ref))) < 2e-4 * np.exp(ref).max()
    # ---- the reference's default sol_kwargs: VCABM, reltol = abstol = 1e-4 (src/core/icnf.jl:84-89) ----
    icnf_d, _, spec_d = _adapter_flow(pkg, o64, nvars=2, naug=3, nsteps=0)
    ps_d, st_d = pkg.setup(torch.Generator().manual_seed(8), icnf_d)
    p_d = ps_d.numpy().astype(np.float32)
    A2 = rng.standard_normal((2, 24)).astype(np.float32)
    got_d = pkg.ICNFDist(icnf_d, pkg.TestMode(), ps_d.to("cuda:0"), st_d).logpdf(torch.tensor(A2)).cpu().numpy()
    assert icnf_d.adaptive and isinstance(icnf_d.sol_kwargs["alg"], pkg.VCABM)
    u0 = np.vstack([A2.astype(np.float64), np.zeros((3 + 3, 24))])
    u1, _ = o64.integrate_vcabm(spec_d, p_d, u0, 0.0, 1.0, 1e-4, 1e-4, None, None)
    ref_v = o64.std_normal_logpdf(u1[:5]) - u1[5]
    ref_fine = o64.inference_fixed(spec_d, p_d, A2, 0.0, 1.0, 200, 1, None)[0]
    assert np.max(np.abs(got_d - ref_v)) < 200 * 1e-4 and np.max(np.abs(got_d - ref_fine)) < 100 * 1e-4, (np.max(np.abs(got_d - ref_v)), np.max(np.abs(got_d - ref_fine)))
    # ---- conditioned: CondICNFDist(icnf, mode, ys, ps, st) uses ys[:, 1:n] ----
    icnf_c, _, spec_c = _adapter_flow(pkg, o64, nvars=2, ncond=3, hidden=(32, 32), act="tanh")
    ps_c, st_c = pkg.setup(torch.Generator().manual_seed(9), icnf_c)
    p_c = ps_c.numpy().astype(np.float32)
    Y = rng.standard_normal((3, 20)).astype(np.float32)
    A3 = rng.standard_normal((2, 12)).astype(np.float32)
    dc = pkg.CondICNFDist(icnf_c, pkg.TestMode(), torch.tensor(Y), ps_c.to("cuda:0"), st_c)
    ref_c = o64.inference_fixed(spec_c, p_c, A3, 0.0, 1.0, 10, 1, None, Y[:, :12])[0]
    assert np.max(np.abs(dc.logpdf(torch.tensor(A3)).cpu().numpy() - ref_c)) < TOL_SOLVE


def test_mlj_model_fit_steps_and_transform_against_the_oracle(pkg, oracles):
    """ICNFModel.fit (src/exts/mlj_ext/core_icnf.jl:32-57): two full-batch steps of OptimiserChain(WeightDecay, Adam) on
    loss(icnf, TrainMode{true}(), xs, ps, st) - the parameters after them against the SAME optimiser arithmetic in numpy driven by
    the fp64 oracle's loss_and_grad (probes pinned through `epsdist`, the shuffle replayed from the same generator); and
    `transform` (core_icnf.jl:59-68) = exp.(logp̂x) in TestMode against the oracle on the fitted parameters."""
    o64, _ = oracles
    n, nvars, naug, nsteps = 48, 2, 1, 6
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, nvars)).astype(np.float32) * 0.7 + 0.3
    probes = [rng.standard_normal((nvars + naug, n)).astype(np.float32) for _ in range(2)]
    calls = []

    def epsdist(gen, shape, device):                                    # rand!(rng, epsdist, eps): call k returns the k-th pinned array
        k = len(calls)                                                  # (the library asks for the Julia memory layout: (B, D) contiguous)
        calls.append(shape)
        if k >= len(probes):                                            # transform / TestMode draws probes too and ignores them (base_icnf.jl:258-259)
            return torch.zeros(shape, device=device)
        e = probes[k]
        assert shape == (e.shape[1], e.shape[0])
        return torch.tensor(np.ascontiguousarray(e.T), device=device)

    lam = (0.01, 0.02, 0.03)
    icnf, spec, spec_test = _adapter_flow(pkg, o64, nvars, naug, nsteps=nsteps, lambdas=lam, epsdist=epsdist)
    eta, wd, b1, b2, ee = 2e-3, 1e-4, 0.9, 0.999, 1e-8
    model = pkg.ICNFModel(icnf=icnf, batchsize=0, epochs=2, eta=eta, weight_decay=wd, callback=None,
                          shuffle_rng=torch.Generator().manual_seed(4), init_rng=torch.Generator().manual_seed(5))
    (ps_fit, st), _, report = model.fit(X)
    assert report["stats"]["iterations"] == 2 and len(calls) == 2          # one draw per optimiser step
    # ---- the same two steps in numpy on the oracle's gradients ----
    p = pkg.setup(torch.Generator().manual_seed(5), icnf)[0].numpy().astype(np.float64)
    g_sh = torch.Generator().manual_seed(4)
    m = np.zeros_like(p); v = np.zeros_like(p)
    sens = []
    for k in range(2):
        idx = torch.randperm(n, generator=g_sh).numpy()                  # MLUtils.DataLoader(shuffle = true): a fresh permutation per epoch
        L, g = o64.loss_and_grad(spec, p.astype(np.float32), X.T[:, idx], 0.0, 1.0, nsteps, 1, probes[k], None, lam)
        g = g + wd * p                                                   # WeightDecay in front of Adam
        m = b1 * m + (1 - b1) * g; v = b2 * v + (1 - b2) * g * g
        step = eta * (m / (1 - b1 ** (k + 1))) / (np.sqrt(v / (1 - b2 ** (k + 1))) + ee)
        sens.append(np.abs(g) > 1e-3 * np.abs(g).max())                  # where the update is not the sign of rounding noise
        p = p - step
    got = ps_fit.cpu().numpy().astype(np.float64)
    ok = sens[0] & sens[1]
    assert ok.mean() > 0.8, ok.mean()
    assert np.max(np.abs(got[ok] - p[ok])) < 2e-6, np.max(np.abs(got[ok] - p[ok]))          # two steps of 2e-3 each
    assert np.max(np.abs(got - p)) <= 2.001 * 2 * eta                                       # the rest moved by at most a step per epoch
    assert abs(report["stats"]["final_loss"] - L) < 1e-4 + 2e-6 * abs(L)
    # ---- transform: exp.(logp̂x), TestMode, on the fitted parameters ----
    Xn = rng.standard_normal((9, nvars)).astype(np.float32)
    px = np.asarray(model.transform((ps_fit, st), Xn)["px"])
    ref = np.exp(o64.inference_fixed(spec_test, got.astype(np.float32), Xn.T, 0.0, 1.0, nsteps, 1, None)[0])
    assert np.max(np.abs(px - ref)) < 2e-4 * ref.max()


def test_custom_base_and_probe_distributions(pkg, oracles):
    """ICNF(; basedist, epsdist) (src/core/icnf.jl:76-83): a non-default base density is evaluated on the host
    from the final state (logp̂x = logpdf(basedist, z) - Δlogp, base_icnf.jl:168) and sampled in generate;
    Rademacher / callable probe distributions feed the same kernels."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=3, hidden=[32, 32])
    p, xs, eps, _ = o64.synth_inputs(spec, 50, 4, bias_scale=0.2)
    std = make_icnf(pkg, spec, 1, 6)
    loc = torch.tensor([0.5, -1.0, 2.0], device="cuda:0")
    base = torch.distributions.MultivariateNormal(loc, covariance_matrix=torch.diag(torch.tensor([0.25, 1.0, 4.0], device="cuda:0")))
    cus = make_icnf(pkg, spec, 1, 6)
    cus.basedist = base
    m = pkg.TestMode()
    lp0, _, u1 = pkg.inference(std, m, dev(xs), dev(p), {}, return_state=True)
    lp1 = pkg.inference(cus, m, dev(xs), dev(p), {})[0]
    z = u1[:3].t()
    expect = base.log_prob(z) - u1[3]
    assert float((lp1 - expect).abs().max()) < 1e-5
    assert float((lp1 - lp0).abs().max()) > 0.1
    assert abs(float(pkg.loss(cus, m, dev(xs), dev(p), {})) + float(lp1.mean())) < 1e-5
    tight = torch.distributions.MultivariateNormal(loc, covariance_matrix=1e-10 * torch.eye(3, device="cuda:0"))
    cus.basedist = tight
    g = pkg.generate(cus, m, dev(p), {}, 5)
    one = pkg.generate(std, m, dev(p), {}, 1, z0=loc[:, None])
    assert float((g - one).abs().max()) < 1e-3                       # every sample starts at loc
    with pytest.raises(NotImplementedError):
        pkg.loss_and_gradient(cus, pkg.TrainMode(False), dev(xs), dev(p), {})
    rad = make_icnf(pkg, spec, 1, 6)
    rad.epsdist = "rademacher"
    from importlib import import_module
    draw = import_module(pkg.__name__ + ".icnf")._draw_eps
    e = draw(rad, 2, 1000)
    assert e.shape == (1000, 6) and set(e.unique().tolist()) == {-1.0, 1.0}
    a = pkg.inference(rad, pkg.TrainMode(False), dev(xs), dev(p), {})[0]
    assert bool(torch.isfinite(a).all())
    rad.epsdist = lambda gen, shape, device: torch.full(shape, 0.5, device=device)
    b = pkg.inference(rad, pkg.TrainMode(False), dev(xs), dev(p), {})[0]
    c = pkg.inference(std, pkg.TrainMode(False), dev(xs), dev(p), {}, eps=torch.full((3, 50), 0.5, device="cuda:0"))[0]
    assert torch.equal(b, c)


def _adaptive_icnf(pkg, spec, tol, path=0, **kw):
    icnf = make_icnf(pkg, spec, 1, 1, path=path)
    icnf.sol_kwargs = dict(alg=pkg.Tsit5(), reltol=tol, abstol=tol, **kw)       # adaptive by default, as in OrdinaryDiffEq
    return icnf


@pytest.mark.parametrize("kw,tol", [
    (dict(nvars=8, hidden=[64, 64, 64]), 1e-4),                                             # fused kernel, the reference's default tolerances
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 1e-6),
    (dict(nvars=3, ncond=2, hidden=[24, 24], act=2, mode=2), 1e-5),                         # exact trace, conditioned
    (dict(nvars=4, hidden=[32, 32, 32, 32, 32]), 1e-5),                                     # layer-wise path (five hidden layers)
])
def test_adaptive_tsit5_follows_the_oracle_restatement(kw, tol, pkg, oracles):
    """Adaptive Tsit5 (PI controller on the host, one cnf_step_embedded attempt per step) against the fp64 oracle's
    restatement of the same algorithm: same accepted / rejected counts, the same step sizes to 1e-3, the same
    final state; and against a fine fixed-step solve to within the tolerance."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    B = 40
    p, xs, eps, ys = o64.synth_inputs(spec, B, 77, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)                                                        # stiffer field: a dozen steps
    icnf = _adaptive_icnf(pkg, spec, tol)
    logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
    st = icnf.last_solve_stats
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
    uref, sref = o64.integrate_adaptive_tsit5(spec, p, u0, 0.0, 1.0, tol, tol, eps, ys)
    # the controller sees a float32 error estimate: at the reference's tolerance (1e-4) the step sequence is the
    # oracle's; near float32's noise floor (1e-6) an accept/size decision may differ by a step
    # (the embedded estimate is a small difference of O(1) stage derivatives, so below ~1e-5 its float32 rounding
    # noise moves the PI controller's step sizes; the solution stays within the tolerance)
    assert abs(st["naccept"] - sref["naccept"]) <= (1 if tol >= 1e-4 else 3), (st, sref)
    assert abs(st["nreject"] - sref["nreject"]) <= (1 if tol >= 1e-4 else 3), (st, sref)
    assert st["naccept"] >= 5
    if tol >= 1e-4 and len(st["dts"]) == len(sref["dts"]):
        assert np.allclose(st["dts"], sref["dts"], rtol=1e-2), (st["dts"], sref["dts"])
    else:
        assert abs(st["dts"][0] - sref["dts"][0]) < 1e-2 * sref["dts"][0]                    # Hairer's initial step
    natt = st["naccept"] + st["nreject"]
    # first-same-as-last / retry reuse.  Host loop: the fused attempt re-evaluates its first stage once (2 + 7 + 6 (n - 1));
    # device-side controller (per-wave kernels, unconditioned): the derivative at the new state is carried over (2 + 6 n)
    assert st["nf"] in (2 + 7 + 6 * (natt - 1), 2 + 6 * natt)
    assert np.max(np.abs(u1.cpu().numpy() - uref)) < 2e-4
    fine = o64.integrate_fixed(spec, p, u0, 0.0, 1.0, 40, 1, eps, ys)
    assert np.max(np.abs(u1.cpu().numpy() - fine)) < 50 * tol + 2e-4
    z = uref[:spec.D]
    lp = -0.5 * spec.D * np.log(2 * np.pi) - 0.5 * (z * z).sum(0) - uref[spec.D]
    assert np.max(np.abs(logp.cpu().numpy() - lp)) < 2e-4


@pytest.mark.parametrize("kw,lam,gpath", [
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 1),     # register-accumulator kernel
    (dict(nvars=8, hidden=[64, 64, 64], nprobes=3, reg_j=True), (0.0, 0.03, 0.0), 1),       # several-probe kernel
    (dict(nvars=4, naug=5, ncond=2, hidden=[40, 40], act=2, reg_z=True), (0.02, 0.0, 0.0), 1),   # default-style net, conditioned
    (dict(nvars=10, hidden=[72, 72], act=2, reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 1),  # slab shape (at this batch size: its auxiliary cooperative sweep)
    (dict(nvars=3, naug=2, ncond=2, hidden=[24, 48, 24], act=2, nprobes=2, reg_aug=True), (0.0, 0.0, 0.05), 2),   # layer-wise path
    # the cooperative gradient on a frozen grid: the extended kernel checkpoints on the caller's step times
    (dict(nvars=16, naug=17, hidden=[136, 136], act=2, reg_z=True, reg_j=True, reg_aug=True), (0.01, 0.01, 0.01), 3),   # ICNF(nvariables = 16), default lambdas
    (dict(nvars=32, hidden=[256, 256, 256], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 3),   # cfg4's shape (its uniform-grid forward has a checkpointing instance of its own)
    (dict(nvars=8, ncond=8, hidden=[192, 192], reg_z=True), (0.05, 0.0, 0.0), 3),            # conditioned
    # a cooperative plan with TWO state k-steps (D <= 8, 3 x 128 tanh) used to checkpoint on a grid through the
    # extended kernel's 8-k-step instance - another layout's offsets and checkpoint stride on this plan's image and buffers
    (dict(nvars=8, hidden=[128, 128, 128]), (0.0, 0.0, 0.0), 3),
    (dict(nvars=7, hidden=[120, 128, 100], reg_z=True, reg_j=True), (0.02, 0.03, 0.0), 2),   # unequal widths: layer-wise
    (dict(nvars=5, naug=2, hidden=[128, 128, 128], reg_z=True, reg_j=True, reg_aug=True), (0.02, 0.03, 0.04), 3),
])
def test_gradient_of_the_adaptive_solve_on_its_frozen_grid(kw, lam, gpath, pkg, oracles):
    """loss_and_gradient with the adaptive solver: the accepted steps are frozen and the discrete solve on that
    non-uniform grid is reversed (cnf_loss_grad_grid; the fused reverse-sweep kernels read the step times from device
    memory, the layer-wise path takes them from the host) - against fp64 autograd on the same grid."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    if gpath != 3:   # (the cooperative path answers once parameters are bound: checked behind the call below)
        assert _adaptive_icnf(pkg, spec, 1e-4).grad_path(pkg.TrainMode(True)) == gpath
    B = 45
    p, xs, eps, ys = o64.synth_inputs(spec, B, 88, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    icnf = _adaptive_icnf(pkg, spec, 1e-4)
    icnf.lambda1, icnf.lambda2, icnf.lambda3 = lam
    mode = pkg.TrainMode(True)
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
    assert icnf.grad_path(mode) == gpath
    ts = icnf.last_solve_stats["tgrid"]
    # cnf_grad_path_for: what this call took (the slab shape of 5 hidden tiles: the auxiliary cooperative sweep up to 8192 columns, round 5)
    assert icnf.grad_path(mode, B=B, alg=1, on_grid=True) == (3 if kw["hidden"] == [72, 72] else gpath)
    assert len(ts) >= 5 and ts[0] == 0.0 and ts[-1] == 1.0 and len(set(np.round(np.diff(ts), 6))) > 1   # non-uniform
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, len(ts) - 1, 1, eps, ys, lam, wrt_x=True, tgrid=ts)
    assert abs(float(val) - L) < 1e-4 + 2e-6 * abs(L)       # (a loss of 300 has a Float32 ulp of 3e-5)
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * np.abs(gref).max() + 1e-6
    assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7
    # and the loss of the grid solve is the loss the adaptive inference reports
    # (the grid solve steps by the float32 differences of the recorded times, the adaptive solve by the steps themselves)
    assert abs(float(val) - float(pkg.loss(icnf, mode, *args, eps=dev(eps)))) < 1e-4 + 2e-6 * abs(float(val))


def test_adaptive_solve_couples_the_batch_and_round_trips(pkg, oracles):
    """The error norm runs over the whole S x B state (OrdinaryDiffEq's default norm), so the step sequence - and the
    last digits of every column - depend on the batch composition, unlike the fixed-step solve; generate with the
    adaptive solver inverts inference to the tolerance."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 3000, 5, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    xs[:, 2000:] *= 3.0                                                                     # harder columns at the end
    icnf = _adaptive_icnf(pkg, spec, 1e-5)
    full = run_inference(pkg, icnf, spec, p, xs, eps, None, return_state=True)
    steps_full = list(icnf.last_solve_stats["dts"])
    head = run_inference(pkg, icnf, spec, p, xs[:, :2000], eps[:, :2000], None)
    steps_head = list(icnf.last_solve_stats["dts"])
    assert steps_full != steps_head
    assert float((full[0][:2000] - head[0]).abs().max()) < 5e-3                             # same solution to the tolerance (RMS-controlled)
    m = pkg.TrainMode(False)
    z1 = full[2][:8]
    back = pkg.generate(icnf, m, dev(p), {}, 3000, z0=z1, eps=dev(eps))
    assert float((back - dev(xs)).abs().max()) < 5e-3


def test_errors_surface_as_exceptions(pkg, oracles):
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, _ = o64.synth_inputs(spec, 8, 1)
    icnf = make_icnf(pkg, spec, 1, 40)
    with pytest.raises(ValueError, match="DimensionMismatch"):
        pkg.inference(icnf, pkg.TrainMode(), dev(xs[:5]), dev(p), {})
    with pytest.raises(ValueError, match="DimensionMismatch"):
        pkg.inference(icnf, pkg.TrainMode(), dev(xs), dev(p[:-1]), {})
    with pytest.raises(ValueError):
        pkg.inference(icnf, pkg.TrainMode(), torch.tensor(xs), dev(p), {})      # host tensor


def test_abi_level_errors_and_grid_gradient(pkg, oracles):
    """Direct C-ABI calls: status codes of the stepping / gradient entry points, and cnf_loss_grad_grid on a UNIFORM
    RK4 grid against cnf_loss_grad_fixed (layer-wise grid path vs fused reverse-sweep kernel)."""
    import ctypes as C
    o64, _ = oracles
    L = pkg._lib
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 300
    p, xs, eps, _ = o64.synth_inputs(spec, B, 2, bias_scale=0.2)
    icnf = make_icnf(pkg, spec, 0, 5, path=2, lambdas=(0.0, 0.0, 0.0))
    mode = pkg.TrainMode(False)
    h = icnf._handle(mode)
    P = dev(p)
    icnf._bind_params(h, P)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    X = dev(xs.T.copy()); E = dev(eps.T.copy())                         # (B, rows) contiguous = column-major (rows, B)
    lam = (C.c_float * 3)(0.0, 0.0, 0.0)
    g1 = torch.empty(p.size, device="cuda:0"); g2 = torch.empty_like(g1)
    s1 = torch.empty(4, device="cuda:0"); s2 = torch.empty_like(s1)
    L.check(h.lib.cnf_loss_grad_fixed(h.ptr, 0, 5, 0.0, 1.0, ptr(X), ptr(E), None, B, lam, ptr(g1), None, ptr(s1), None))
    grid = (C.c_float * 6)(*[i / 5 for i in range(6)])
    L.check(h.lib.cnf_loss_grad_grid(h.ptr, 0, 5, grid, ptr(X), ptr(E), None, B, lam, ptr(g2), None, ptr(s2), None))
    assert float((g1 - g2).abs().max()) < 2e-5 * float(g1.abs().max())
    assert float((s1 - s2).abs().max()) < 1e-3 * float(s1.abs().max())
    # status codes
    u = torch.zeros(B, 11, device="cuda:0"); un = torch.empty_like(u); err = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    step = lambda alg, dtv, uu, unn, at, rt: h.lib.cnf_step_embedded(h.ptr, alg, 0, 0.0, dtv, ptr(uu), ptr(E), None, B, at, rt,
                                                                   ptr(unn), ptr(err), None)
    assert step(1, 0.1, u, un, 1e-4, 1e-4) == L.OK
    assert step(0, 0.1, u, un, 1e-4, 1e-4) == L.ERR_INVALID              # RK4 has no embedded pair here
    assert step(1, 0.1, u, u, 1e-4, 1e-4) == L.ERR_INVALID               # aliasing
    assert step(1, 0.1, u, un, 0.0, 0.0) == L.ERR_INVALID                # both tolerances zero
    assert b"tolerances" in h.lib.cnf_last_error()
    assert h.lib.cnf_loss_grad_grid(h.ptr, 1, 0, grid, ptr(X), ptr(E), None, B, lam, ptr(g2), None, None, None) == L.ERR_INVALID
    assert h.lib.cnf_loss_grad_grid(h.ptr, 1, 5, None, ptr(X), ptr(E), None, B, lam, ptr(g2), None, None, None) == L.ERR_INVALID
    assert h.lib.cnf_epilogue(h.ptr, None, B, ptr(g2), None, None) == L.ERR_INVALID
    assert h.lib.cnf_assemble_u0(h.ptr, ptr(X), -1, ptr(u), None) == L.ERR_INVALID
    assert h.lib.cnf_grad_path(None) == L.ERR_INVALID and h.lib.cnf_repack_on_device(h.ptr) == 1


def test_plain_cpp_host_on_the_c_abi(pkg, oracles, tmp_path):
    """examples/abi_demo.cpp: a C++/HIP host with no Python or torch in the process drives the library through
    include/cnf.h (the call sequence of INTEGRATION.md's Julia glue) and reproduces the Python mirror's log-densities."""
    import os, shutil, struct, subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "continuousnormalizingflows.jl_amd")
    exe = str(tmp_path / "abi_demo")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_demo.cpp"),
                    "-o", exe, "-L", libdir, "-lcnf_hip", f"-Wl,-rpath,{libdir}"], check=True, timeout=300)
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B, nsteps = 5000, 20
    p, xs, eps, _ = o64.synth_inputs(spec, B, 14, bias_scale=0.2)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("6i", 8, 64, 3, B, nsteps, 1))
        f.write(p.astype(np.float32).tobytes())
        f.write(np.ascontiguousarray(xs.T, dtype=np.float32).tobytes())       # column-major (rows, B)
        f.write(np.ascontiguousarray(eps.T, dtype=np.float32).tobytes())
    r = subprocess.run([exe, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "kernel_path=2" in r.stdout, r.stdout
    got = np.fromfile(fout, dtype=np.float32)
    icnf = make_icnf(pkg, spec, 1, nsteps)
    ref = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    assert got.shape == (2 * B,) and np.array_equal(got[:B], ref)          # same library, same inputs: same bits
    icnf.sol_kwargs = {}                                                    # the reference's defaults: VCABM at 1e-4
    ref_v = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].cpu().numpy()
    st = icnf.last_solve_stats
    assert np.array_equal(got[B:], ref_v)
    assert f"vcabm naccept={st['naccept']} nreject={st['nreject']} nf={st['nf']} max_order={max(st['orders'])}" in r.stdout, r.stdout


@pytest.mark.parametrize("planar", [False, True])
@pytest.mark.parametrize("conditioned", [False, True])
def test_reference_smoke_flow_with_all_defaults(conditioned, planar, pkg):
    """test/ci_tests/smoke_tests.jl:69-156 with nothing but the defaults: ICNF(; nvariables, [nconditions]) - default net,
    default (adaptive) solver at the reference's tolerances, default lambdas and steer rate - through inference, generate,
    loss, the layer call, the gradients with respect to ps and x, and the MLJ / Distributions adapters.  The reference only
    checks `!isnothing`; here also shapes, finiteness and that a short fit lowers the loss."""
    import warnings
    g = torch.Generator().manual_seed(11)
    r = torch.distributions.Beta(2.0, 4.0).sample((2, 64)).float()           # Beta(2, 4) data, ndimensions = 2 (smoke_tests.jl:10-13)
    r2 = torch.distributions.Beta(2.0, 4.0).sample((2, 64)).float()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        nnkw = {}
        if planar:   # smoke_tests.jl:32-46
            nnkw["nn"] = pkg.Chain(pkg.PlanarLayer(2 * (3 if conditioned else 2) + 2, 2 * 2 + 1, pkg.tanh))
        icnf = pkg.ICNF(nvariables=2, nconditions=2 if conditioned else 0, device="cuda:0", **nnkw)
        assert isinstance(icnf.sol_kwargs.get("alg", pkg.VCABM()), pkg.VCABM)                  # the reference's default solver
        ps, st = pkg.setup(g, icnf)
        ps = ps.cuda()
        cond = (r2.cuda(),) if conditioned else ()
        for mode in (pkg.TrainMode(True), pkg.TestMode()):
            logp, (E, n, A) = pkg.inference(icnf, mode, r.cuda(), *cond, ps, st)
            assert logp.shape == (64,) and bool(torch.isfinite(logp).all()) and E.shape == n.shape == A.shape == (64,)
            x = pkg.generate(icnf, mode, *cond, ps, st, 64)
            assert x.shape == (2, 64) and bool(torch.isfinite(x).all())
            assert np.isfinite(float(pkg.loss(icnf, mode, r.cuda(), *cond, ps, st)))
            out = icnf((r.cuda(),) + cond if conditioned else r.cuda(), ps, st)
            assert out is not None
        for mode in (pkg.TrainMode(True), pkg.TestMode()):                                     # DI.gradient of diff_loss / diff2_loss, both omodes
            val, gps, gx = pkg.loss_and_gradient(icnf, mode, r.cuda(), *cond, ps, st, wrt_x=True)
            assert gps.shape == ps.shape and gx.shape == (2, 64) and bool(torch.isfinite(gps).all()) and bool(torch.isfinite(gx).all())
            assert float(gps.abs().max()) > 0 and float(gx.abs().max()) > 0
        Model = pkg.CondICNFModel if conditioned else pkg.ICNFModel
        model = Model(icnf=icnf, batchsize=32, epochs=8, callback=None, init_rng=torch.Generator().manual_seed(3))
        data = (r.t(), r2.t()) if conditioned else r.t()
        l0 = float(pkg.loss(icnf, pkg.TestMode(), r.cuda(), *cond, pkg.setup(torch.Generator().manual_seed(3), icnf)[0].cuda(), st))
        fitresult, _, report = model.fit(data)
        assert report["stats"]["iterations"] == 16
        l1 = float(pkg.loss(icnf, pkg.TestMode(), r.cuda(), *cond, fitresult[0], st))
        assert l1 < l0
        px = model.transform(fitresult, data)
        assert len(px) == 64 and (px["px"] > 0).all()
        if conditioned:
            d = pkg.CondICNFDist.from_fit(model, fitresult, pkg.TestMode(), r2)
        else:
            d = pkg.ICNFDist.from_fit(model, fitresult, pkg.TestMode())
        assert d.logpdf(r).shape == (64,) and d.pdf(r[:, 0]).dim() == 0 and d.rand(5).shape == (2, 5) and d.rand().shape == (2,)


# ---- the reference's default solver: VCABM (variable-order, variable-step Adams PECE) ----

def _vcabm_handle(pkg, icnf, spec, p):
    h = icnf._handle(mode_of(pkg, spec))
    icnf._bind_params(h, dev(p))
    return h


@pytest.mark.parametrize("kw,path,B,rand", [
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 0, 37, None),              # fused single-call kernel; S B odd: scalar passes
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 0, 36, None),              # S B a multiple of 4: 16-byte passes
    (dict(nvars=3, ncond=2, hidden=[24, 24], act=2), 1, 38, None),                          # SIMT family, conditioned
    (dict(nvars=8, hidden=[64, 64, 64]), 0, 64, (1, +1.0)),                                 # random orders and step sizes
    (dict(nvars=8, hidden=[64, 64, 64]), 0, 52, (2, -1.0)),                                 # ... stepping backwards (generate)
    (dict(nvars=4, naug=2, hidden=[32, 32, 32, 32], act=2), 3, 40, (3, +1.0)),              # ... layer-wise dynamics
])
def test_vcabm_passes_follow_a_scripted_order_and_step_sequence(kw, path, B, rand, pkg, oracles):
    """cnf_vcabm_begin / _attempt / _accept / _state driven with a fixed script of (order, dt) - orders up to 12, steps
    growing, shrinking, a rejected (repeated) attempt, the order k+1 estimate where the history allows it - against the
    fp64 stepper on the same script: u_{n+1} after every attempt and all four error sums.  No controller involved, so
    every pass (differences, beta / g coefficients, predictor, corrector, error estimates) is compared one to one."""
    import ctypes as C
    o64, _ = oracles
    L = pkg._lib
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 21, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 1, 1, path=path)
    h = _vcabm_handle(pkg, icnf, spec, p)
    S = spec.D + 3
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
    abstol, reltol = 1e-5, 1e-4
    f = lambda u, t: o64.aug_f(spec, p, u, t, eps, ys).astype(np.float64)
    ref = o64.VcabmStepper(f, u0, 0.1, abstol, reltol)
    colmajor = lambda a: dev(np.ascontiguousarray(a.T))                                     # (rows, B) -> B x rows = column-major rows x B
    d_u0, d_eps, d_ys = colmajor(u0), colmajor(eps), (colmajor(ys) if spec.ncond else None)
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(0)
    L.check(h.lib.cnf_vcabm_begin(h.ptr, 0.1, ptr(d_u0), ptr(d_eps), ptr(d_ys), B, st))
    err3 = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    errp = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    out = torch.empty(B, S, device="cuda:0")
    script = [(1, 0.01), (2, 0.012), (3, 0.02), (3, 0.02), (4, 0.015), (5, 0.03), (6, 0.03), (7, 0.025), (8, 0.04),
              (9, 0.03), (10, 0.03), (11, 0.035), (12, 0.03), (12, 0.05), (11, 0.02), (7, 0.06), (3, 0.03), (4, 0.03)]
    if rand is not None:
        rs = np.random.default_rng(rand[0])
        script, avail = [], 0
        for i in range(24):   # any order the stored differences allow: down freely, up by at most one per accepted step
            ok = [k for k in range(1, min(12, i + 1) + 1) if min(k, i) <= avail]
            k = int(rs.choice(ok[-3:])) if rs.uniform() < 0.7 else int(rs.choice(ok))
            script.append((k, rand[1] * float(rs.uniform(0.005, 0.05))))
            avail = min(k + 1, i + 1)
    worst = 0.0
    for i, (k, dt) in enumerate(script):
        if i == 6:   # a rejected attempt: tried with a larger step first, then repeated - the state must be untouched
            L.check(h.lib.cnf_vcabm_attempt(h.ptr, k, 3 * dt, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st))
        L.check(h.lib.cnf_vcabm_attempt(h.ptr, k, dt, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st))
        un, errs = ref.attempt(k, dt)
        want_up = k < 12 and len(ref.hist) >= k
        L.check(h.lib.cnf_vcabm_accept(h.ptr, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(errp) if want_up else None, st))
        up = ref.accept(want_up)
        tt = C.c_double(0)
        L.check(h.lib.cnf_vcabm_state(h.ptr, B, ptr(out), C.byref(tt), st))
        torch.cuda.synchronize()
        assert abs(tt.value - ref.t) < 1e-6
        worst = max(worst, float(np.max(np.abs(out.cpu().numpy().T - un))))
        got = err3.cpu().numpy()
        # the estimates are high-order differences of float32 derivatives: compare relative to the sum's own size with
        # the float32 noise floor of a k-th difference (~2^k ulp of |f|) scaled like the estimate itself
        for j, (a, b) in enumerate(zip(got, errs)):
            if j < k:   # orders k, k-1, k-2 exist
                noise = ref.n * (abs(dt) * 2.0 ** (k - j) * 1e-6 / abstol) ** 2
                assert abs(a - b) <= 2e-2 * b + noise, (i, k, j, a, b, noise)
            else:
                assert a == 0.0
        if want_up:
            noise = ref.n * (abs(dt) * 2.0 ** (k + 1) * 1e-6 / abstol) ** 2
            assert abs(float(errp) - up) <= 2e-2 * up + noise, (i, k, float(errp), up, noise)
    assert worst < 2e-5, worst
    # misuse surfaces as status codes
    assert h.lib.cnf_vcabm_accept(h.ptr, ptr(d_eps), ptr(d_ys), B, abstol, reltol, None, st) == L.ERR_INVALID      # no pending attempt
    assert h.lib.cnf_vcabm_attempt(h.ptr, 13, 0.01, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st) == L.ERR_INVALID
    assert h.lib.cnf_vcabm_attempt(h.ptr, 3, 0.01, ptr(d_eps), ptr(d_ys), B + 1, abstol, reltol, ptr(err3), st) == L.ERR_INVALID
    L.check(h.lib.cnf_vcabm_begin(h.ptr, 0.0, ptr(d_u0), ptr(d_eps), ptr(d_ys), B, st))
    assert h.lib.cnf_vcabm_attempt(h.ptr, 2, 0.01, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st) == L.ERR_INVALID  # no history yet
    for k in (1, 1, 1):                                                                    # three steps at order 1 ...
        L.check(h.lib.cnf_vcabm_attempt(h.ptr, k, 0.01, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st))
        L.check(h.lib.cnf_vcabm_accept(h.ptr, ptr(d_eps), ptr(d_ys), B, abstol, reltol, None, st))
    assert h.lib.cnf_vcabm_attempt(h.ptr, 3, 0.01, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st) == L.ERR_INVALID  # ... then 3: two orders up
    assert b"at most one" in h.lib.cnf_last_error()
    assert h.lib.cnf_vcabm_attempt(h.ptr, 2, 0.01, ptr(d_eps), ptr(d_ys), B, abstol, reltol, ptr(err3), st) == L.OK


@pytest.mark.parametrize("kw,tol", [
    (dict(nvars=8, hidden=[64, 64, 64]), 1e-4),                                             # the reference's default tolerances
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 1e-6),
    (dict(nvars=3, ncond=2, hidden=[24, 24], act=2, mode=2), 1e-5),                         # exact trace, conditioned
])
def test_vcabm_solve_follows_the_oracle_restatement(kw, tol, pkg, oracles):
    """The default solver end to end (`ICNF` without sol_kwargs.alg): host controller + order selection over the device
    passes against the fp64 restatement - accepted / rejected counts, the order history, the final state - and against
    a fine fixed-step solve to within a multiple of the tolerance."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    B = 40
    p, xs, eps, ys = o64.synth_inputs(spec, B, 78, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    icnf = make_icnf(pkg, spec, 1, 1)
    icnf.sol_kwargs = dict(reltol=tol, abstol=tol)                                          # alg defaults to VCABM()
    # the policy exists twice - the host loop of icnf.py (what a sharded solve runs) and inside cnf_solve_vcabm (the default for one
    # process): from the same initial step they must take bit-identical decisions, steps and states
    icnf.sol_kwargs["dt"] = 2.0 ** -7          # exactly representable in float32: the step controller amplifies a 1e-8 difference 1000-fold over a solve
    runs = {}
    for pol in ("python", "library"):
        icnf.adaptive_policy = pol
        runs[pol] = (run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)[2], dict(icnf.last_solve_stats))
    (u_py, st_py), (u_lib, st_lib) = runs["python"], runs["library"]
    assert st_lib["orders"] == st_py["orders"] and (st_lib["naccept"], st_lib["nreject"]) == (st_py["naccept"], st_py["nreject"])
    assert np.array_equal(np.float32(st_py["dts"]), np.float32(st_lib["dts"])) and torch.equal(u_py, u_lib)
    u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
    # ... and, from that common initial step, the decisions of the fp64 restatement (orders exactly at the reference's tolerance)
    uref0, sref0 = o64.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol, eps, ys, dt0=2.0 ** -7)
    slack0 = 1 if tol >= 1e-4 else 4
    assert abs(st_lib["naccept"] - sref0["naccept"]) <= slack0 and abs(st_lib["nreject"] - sref0["nreject"]) <= slack0, (st_lib, sref0)
    if tol >= 1e-4:
        assert st_lib["orders"] == sref0["orders"], (st_lib["orders"], sref0["orders"])
        assert np.allclose(st_lib["dts"], sref0["dts"], rtol=2e-2), (st_lib["dts"], sref0["dts"])
    assert np.max(np.abs(u_lib.cpu().numpy() - uref0)) < (20 * tol + 2e-5 if tol >= 1e-4 else 100 * tol)
    del icnf.sol_kwargs["dt"]
    logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)      # library policy, Hairer's initial step
    assert isinstance(icnf.sol_kwargs["alg"], pkg.VCABM) and icnf.adaptive
    st = icnf.last_solve_stats
    uref, sref = o64.integrate_vcabm(spec, p, u0, 0.0, 1.0, tol, tol, eps, ys)
    # Hairer's initial step with the exponent 1 / (current order = 1) is tiny (1e-7 .. 1e-6): the first steps grow tenfold each and
    # their error estimates sit at the Float32 noise floor, so later order decisions may differ from the fp64 run by a few steps
    slack = max(4, sref["naccept"] // 5)
    assert abs(st["naccept"] - sref["naccept"]) <= slack and abs(st["nreject"] - sref["nreject"]) <= slack, (st, sref)
    assert st["orders"][:4] == [1, 2, 3, 3] and max(st["orders"]) >= 4
    assert np.allclose(st["dts"][:4], sref["dts"][:4], rtol=2e-2), (st["dts"][:4], sref["dts"][:4])
    assert abs(st["dts"][0] - sref["dts"][0]) < 1e-2 * sref["dts"][0]                        # Hairer's initial step
    assert st["nf"] == 2 + 2 * st["naccept"] + st["nreject"]                                # PECE: two evaluations per accepted step
    fine = o64.integrate_fixed(spec, p, u0, 0.0, 1.0, 400, 1, eps, ys)
    assert np.max(np.abs(uref - fine)) < 100 * tol                                          # the restatement itself
    assert np.max(np.abs(u1.cpu().numpy() - uref)) < 200 * tol        # two different step sequences, each within 100 tol of the fine solve (the tight bound is the common-dt run above)
    assert np.max(np.abs(u1.cpu().numpy() - fine)) < 100 * tol
    z = u1.cpu().numpy()[:spec.D]
    lp = -0.5 * spec.D * np.log(2 * np.pi) - 0.5 * (z * z).sum(0) - u1.cpu().numpy()[spec.D]
    assert np.max(np.abs(logp.cpu().numpy() - lp) / (1.0 + np.abs(lp))) < 1e-6               # the epilogue on that state (float32 rounding of logp)


def test_vcabm_round_trip_large_batch_and_training(pkg, oracles):
    """generate inverts inference under the default solver (negative steps), a 65 536-column solve agrees with Tsit5 at
    the same tolerance, and loss_and_gradient under VCABM differentiates the adaptive Tsit5 discretisation (value
    within the tolerance of the VCABM loss)."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    B = 65536
    p, xs, eps, _ = o64.synth_inputs(spec, B, 6, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    icnf = make_icnf(pkg, spec, 1, 1)
    icnf.sol_kwargs = dict(alg=pkg.VCABM(), reltol=1e-5, abstol=1e-5)
    full = run_inference(pkg, icnf, spec, p, xs, eps, None, return_state=True)
    st = dict(icnf.last_solve_stats)
    assert st["naccept"] >= 6 and max(st["orders"]) >= 4
    ts = _adaptive_icnf(pkg, spec, 1e-5)
    other = run_inference(pkg, ts, spec, p, xs, eps, None)
    assert float((full[0] - other[0]).abs().max()) < 5e-3
    m = pkg.TrainMode(False)
    back = pkg.generate(icnf, m, dev(p), {}, B, z0=full[2][:8], eps=dev(eps))
    assert icnf.last_solve_stats["dts"][0] < 0
    assert float((back - dev(xs)).abs().max()) < 5e-3
    val, g = pkg.loss_and_gradient(icnf, pkg.TrainMode(True), dev(xs), dev(p), {}, eps=dev(eps))
    assert bool(torch.isfinite(g).all()) and abs(float(val) - float(-full[0].mean())) < 1e-3


# ---- the remaining cells of the reference's smoke matrix for the gradient: TestMode and PlanarLayer nets ----

@pytest.mark.parametrize("kw,alg,nsteps,gpath", [
    (dict(nvars=3, ncond=2, hidden=[24, 24], act=2, mode=2), 1, 8, 1),                      # default-style softplus net, conditioned
    (dict(nvars=8, hidden=[64, 64, 64], mode=2), 0, 6, 1),                                  # 8 unit probes, three hidden layers
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, mode=2), 1, 8, 1),                       # ICNF(; nvariables = 1): the reference's benchmark net
    (dict(nvars=1, hidden=[16, 16], act=1, mode=2), 1, 8, 1),                               # D = 1: the one-probe kernel
    (dict(nvars=4, naug=5, hidden=[40, 40], act=2, mode=2), 1, 6, 1),                       # ICNF(; nvariables = 4): 9 unit probes
    (dict(nvars=13, hidden=[56, 56], act=1, mode=2), 1, 5, 1),                              # 13 unit probes, 4 hidden tiles
    (dict(nvars=15, hidden=[32, 32], act=1, mode=2), 1, 5, 2),                              # state + time column beyond one input tile: layer-wise path
    (dict(nvars=2, naug=3, hidden=[20], act=1, mode=2), 1, 8, 2),                           # augmented, one hidden layer: layer-wise path
])
def test_gradient_of_the_test_mode_loss(kw, alg, nsteps, gpath, pkg, oracles):
    """`DI.gradient` of `loss(icnf, TestMode(), ...)` with respect to ps and to xs (test/ci_tests/smoke_tests.jl:85-90 with
    omode = TestMode()): the exact trace -tr J = -sum_k e_k^T J e_k reversed with the D unit vectors as probes on the
    several-probe fused kernel (D <= 8) or the layer-wise path, against fp64 autograd through the exact-trace solve."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    B = 41
    p, xs, eps, ys = o64.synth_inputs(spec, B, 52, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, alg, nsteps)
    mode = pkg.TestMode()
    assert icnf.grad_path(mode) == gpath
    args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
    val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, wrt_x=True)
    L, gref, gxref = o64.loss_and_grad(spec, p, xs, 0.0, 1.0, nsteps, alg, None, ys, (0.0, 0.0, 0.0), wrt_x=True)
    assert abs(float(val) - L) < 2e-5 * (1 + abs(L))
    assert abs(float(val) - float(pkg.loss(icnf, mode, *args))) < 1e-5 * (1 + abs(L))
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * np.abs(gref).max() + 1e-6
    assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7


@pytest.mark.parametrize("use_bias", [True, False])
@pytest.mark.parametrize("conditioned", [False, True])
def test_gradient_of_a_planar_layer_net(use_bias, conditioned, pkg, oracles):
    """planar = true in the reference's smoke matrix (test/ci_tests/smoke_tests.jl:32-46, 130-134): the gradient with respect
    to the PlanarLayer's own parameter vector (u, w, b) and to xs, TrainMode{true} and TestMode, against fp64 autograd on the
    equivalent Dense chain (the pinned zero bias carries no gradient entry)."""
    o64, _ = oracles
    nv, C = 2, (2 if conditioned else 0)
    D = 2 * nv + 1
    n_in = D + 1 + C
    rng = np.random.default_rng(9)
    u, w = rng.uniform(-0.7, 0.7, D), rng.uniform(-0.7, 0.7, n_in)
    b = rng.uniform(-0.3, 0.3, 1) if use_bias else np.zeros(0)
    ps = np.concatenate([u, w, b]).astype(np.float32)
    B = 33
    xs = rng.standard_normal((nv, B)).astype(np.float32)
    eps = rng.standard_normal((D, B)).astype(np.float32)
    ys = rng.standard_normal((C, B)).astype(np.float32) if C else None
    lam = (0.01, 0.01, 0.01)
    for mode_id, mode in ((0, pkg.TrainMode(True)), (2, pkg.TestMode())):
        tr = mode_id == 0
        spec = o64.Spec(nvars=nv, naug=nv + 1, ncond=C, widths=[n_in, 1, D], acts=[1, 0], mode=mode_id,
                        reg_z=tr, reg_j=tr, reg_aug=tr)
        p_dense = np.concatenate([w, b if use_bias else np.zeros(1), u, np.zeros(D)])
        L, gd, gxref = o64.loss_and_grad(spec, p_dense, xs, 0.0, 1.0, 12, 1, eps, ys, lam if tr else (0.0, 0.0, 0.0), wrt_x=True)
        gw, gb, gu = gd[:n_in], gd[n_in:n_in + 1], gd[n_in + 1:n_in + 1 + D]
        gref = np.concatenate([gu, gw, gb if use_bias else np.zeros(0)])
        icnf = pkg.ICNF(nvariables=nv, nconditions=C, nn=pkg.Chain(pkg.PlanarLayer(n_in, D, pkg.tanh, use_bias=use_bias)),
                        steer_rate=0.0, device="cuda:0", sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, nsteps=12))
        args = (dev(xs),) + ((dev(ys),) if C else ()) + (dev(ps), {})
        val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
        assert g.shape == (ps.size,)
        assert abs(float(val) - L) < 2e-5 * (1 + abs(L))
        assert np.max(np.abs(g.cpu().numpy() - gref)) < 5e-5 * np.abs(gref).max() + 1e-6, (mode_id, g.cpu().numpy(), gref)
        assert np.max(np.abs(gx.cpu().numpy() - gxref)) < 5e-5 * np.abs(gxref).max() + 1e-7
    # and the MLJ-style fit runs on it, as in the reference's smoke test
    model = pkg.ICNFModel(icnf=icnf, batchsize=16, epochs=3, callback=None) if not conditioned else \
        pkg.CondICNFModel(icnf=icnf, batchsize=16, epochs=3, callback=None)
    data = (xs.T, ys.T) if conditioned else xs.T
    fitresult, _, report = model.fit(data)
    assert report["stats"]["iterations"] == 9 and np.isfinite(report["stats"]["final_loss"])


@pytest.mark.parametrize("alg,policy", [("tsit5", None), ("vcabm", "library"), ("vcabm", "python")])
def test_adaptive_solves_fail_loudly(alg, policy, pkg, oracles):
    """SciML reports maxiters / instability through the retcode, which the reference ignores (src/core/base_icnf.jl:138-139);
    here they are exceptions, the same ones from the host loops and from the library's own policy loop."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=3, hidden=[16, 16])
    p, xs, eps, _ = o64.synth_inputs(spec, 20, 4, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 1, 1)
    icnf.sol_kwargs = dict(alg=pkg.Tsit5() if alg == "tsit5" else pkg.VCABM(), reltol=1e-6, abstol=1e-6, maxiters=3)
    if policy:
        icnf.adaptive_policy = policy
    with pytest.raises(RuntimeError, match="maxiters"):
        run_inference(pkg, icnf, spec, p, xs, eps, None)
    icnf.sol_kwargs["maxiters"] = 100000
    bad = p.copy()
    bad[-3:] = np.inf                                                                       # a bias at infinity: non-finite dynamics
    with pytest.raises(FloatingPointError, match="non-finite"):
        run_inference(pkg, icnf, spec, bad, xs, eps, None)


def test_parameters_are_rebound_when_a_new_tensor_reuses_the_address(pkg, oracles):
    """The handle skips the repack only for the same tensor object at the same version: a fresh tensor that the caching
    allocator places at a freed tensor's address must not be mistaken for it."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=3, hidden=[16, 16])
    p, xs, eps, _ = o64.synth_inputs(spec, 20, 4, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 1, 10)
    a = run_inference(pkg, icnf, spec, p, xs, eps, None)[0].clone()
    ptrs = set()
    for k in range(4):                                   # same size, allocated right after the previous one was freed
        t = dev(p * (1.0 + 0.1 * (k + 1)))
        ptrs.add(t.data_ptr())
        b = pkg.inference(icnf, mode_of(pkg, spec), dev(xs), t, {}, eps=dev(eps))[0]
        ref = run_inference(pkg, make_icnf(pkg, spec, 1, 10), spec, p * (1.0 + 0.1 * (k + 1)), xs, eps, None)[0]
        assert torch.equal(b, ref) and not torch.equal(a, b)
        del t
    t = dev(p)
    v0 = pkg.inference(icnf, mode_of(pkg, spec), dev(xs), t, {}, eps=dev(eps))[0]
    t.mul_(1.5)                                          # in-place update of the same object: the version changes
    v1 = pkg.inference(icnf, mode_of(pkg, spec), dev(xs), t, {}, eps=dev(eps))[0]
    assert torch.equal(v0, a) and not torch.equal(v1, v0)


def test_reassigned_fields_reach_the_library(pkg, oracles):
    """The regulariser switches are baked into a handle (NORM_Z / NORM_J are type parameters in the reference,
    src/core/icnf.jl:109-115); reassigning lambda / nprobes on the Python object selects another handle."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=3, hidden=[16, 16])
    p, xs, eps, _ = o64.synth_inputs(spec, 20, 4, bias_scale=0.3)
    icnf = make_icnf(pkg, spec, 1, 10)
    m = pkg.TrainMode(True)
    E0 = pkg.inference(icnf, m, dev(xs), dev(p), {}, eps=dev(eps))[1][0]
    assert float(E0.abs().max()) == 0.0                                     # lambda1 = 0: no kinetic term
    icnf.lambda1 = 0.05
    lp, (E1, n1, _) = pkg.inference(icnf, m, dev(xs), dev(p), {}, eps=dev(eps))
    spec_z = o64.make_spec(nvars=3, hidden=[16, 16], reg_z=True)
    ref = o64.inference_fixed(spec_z, p, xs, 0.0, 1.0, 10, 1, eps)
    assert float(n1.abs().max()) == 0.0 and np.max(np.abs(E1.cpu().numpy() - ref[1][0])) < TOL_SOLVE


def test_adaptive_tsit5_library_policy_is_the_host_loop(pkg, oracles, monkeypatch):
    """cnf_solve_tsit5 / cnf_loss_grad_adaptive (the PI controller inside the library, one call per solve / training step)
    against the Python host loop that sharded solves run: from a common, exactly representable initial step the same
    accepted steps, state, loss and gradient bit for bit; with Hairer's initial step the same counts and a gradient
    within 1e-5 (the two initial-step computations round differently).  (The library's HOST loop: the device-side
    controller, which this batch would otherwise get, is compared with it in the next test.)"""
    setsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER", "0")
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True)
    B = 48
    p, xs, eps, _ = o64.synth_inputs(spec, B, 91, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    icnf = _adaptive_icnf(pkg, spec, 1e-5, dt=2.0 ** -6)
    icnf.lambda1, icnf.lambda2 = 0.02, 0.03
    mode = pkg.TrainMode(True)
    out = {}
    # (library0: the gradient's own forward pass on the frozen grid, as the Python loop runs it - bit for bit; library: the default, the
    # solve's fused attempts fill the sweep's checkpoints themselves, an ulp of t and dt per step away)
    for pol in ("python", "library0", "library"):
        icnf.adaptive_policy = pol.rstrip("0")
        setsw(pkg, monkeypatch, "CNF_ADAPTIVE_CKPT", "0" if pol == "library0" else "1")
        lp, _, u1 = pkg.inference(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), return_state=True)
        st = dict(icnf.last_solve_stats)
        val, g, gx = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps), wrt_x=True)
        out[pol] = (u1, st, float(val), g, gx, list(icnf.last_solve_stats["tgrid"]))
    a, b, c = out["python"], out["library0"], out["library"]
    assert (a[1]["naccept"], a[1]["nreject"], a[1]["nf"]) == (b[1]["naccept"], b[1]["nreject"], b[1]["nf"]) and a[1]["naccept"] >= 6
    assert np.array_equal(np.float32(a[1]["dts"]), np.float32(b[1]["dts"])) and torch.equal(a[0], b[0])
    assert np.array_equal(np.float32(a[5]), np.float32(b[5])) and a[2] == b[2] and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert np.array_equal(np.float32(b[5]), np.float32(c[5])) and abs(b[2] - c[2]) < 2e-6 * abs(b[2])
    assert float((b[3] - c[3]).abs().max()) < 1e-5 * float(b[3].abs().max()) and float((b[4] - c[4]).abs().max()) < 1e-5 * float(b[4].abs().max())
    del icnf.sol_kwargs["dt"]
    res = {}
    for pol in ("python", "library"):
        icnf.adaptive_policy = pol
        val, g = pkg.loss_and_gradient(icnf, mode, dev(xs), dev(p), {}, eps=dev(eps))
        res[pol] = (float(val), g, dict(icnf.last_solve_stats))
    assert res["python"][2]["naccept"] == res["library"][2]["naccept"]
    assert abs(res["python"][0] - res["library"][0]) < 1e-5
    assert float((res["python"][1] - res["library"][1]).abs().max()) < 1e-5 * float(res["python"][1].abs().max()) + 1e-7


@pytest.mark.parametrize("kw,B,tol,dt0", [
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 48, 1e-5, 2.0 ** -6),      # cfg2's kernel, RNODE rows live
    (dict(nvars=8, hidden=[64, 64, 64]), 1000, 1e-4, None),                                 # ragged last tile, Hairer's initial step
    (dict(nvars=2, hidden=[32, 32]), 4096, 1e-4, None),                                     # cfg1's net, 256 tiles
    (dict(nvars=5, naug=2, hidden=[24, 24], act=2), 16, 1e-6, None),                        # generic zero-padded instance, one tile
    (dict(nvars=5, naug=2, hidden=[24, 24], act=2, autonomous=True, reg_z=True), 300, 1e-5, None),
    (dict(nvars=3, hidden=[24, 24], act=2, mode=2), 200, 1e-5, None),                       # exact trace (TestMode)
    (dict(nvars=8, hidden=[64, 64, 64], autonomous=True), 32763, 1e-4, None),               # every wave slot of the chip, ragged last tile
    (dict(nvars=9, naug=10, hidden=[80, 80], act=2, autonomous=True), 100, 1e-4, None),     # reference default net for nvariables = 9
    (dict(nvars=2, naug=3, ncond=2, hidden=[32, 32], act=2, autonomous=True), 300, 1e-4, None),   # conditioned default-style net
    (dict(nvars=2, hidden=[32, 32], autonomous=True), 6000, 1e-4, None),                    # 375 tiles on 256 workgroups: one or two of the eight waves own a tile
    (dict(nvars=2, hidden=[32, 32], autonomous=True), 16384, 1e-4, None),                   # four of the eight
])
def test_device_side_step_controller_is_the_host_loop(kw, B, tol, dt0, pkg, oracles, monkeypatch):
    """Adaptive Tsit5 in ONE launch (mfma_adaptive_kernel: grid-wide error norm, PI controller in every wave) against the
    library's host loop (3-4 launches and a round trip per attempt): the same accepted and rejected counts.  For an
    autonomous field the two differ only in the order the error norm is summed (in double): step sizes to 1e-6, states to
    a few ulp.  With a time input the device kernel carries the derivative at the new state over to the next attempt
    (first-same-as-last, evaluated at fl(t + dt)) where the host loop's fused attempt re-evaluates it at fl(t_new), t_new
    summed in double - an ulp apart; the embedded estimate is a small difference of O(1) derivatives, so that ulp moves
    the controller's step sizes by ~0.1 % (as float32 noise moves both away from the fp64 oracle), the states by << tol."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 5, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    extra = {} if dt0 is None else dict(dt=dt0)
    icnf = _adaptive_icnf(pkg, spec, tol, **extra)
    res = {}
    for ctl in ("1", "0"):
        setsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER", ctl)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        res[ctl] = (logp, u1, dict(icnf.last_solve_stats))
    dev_st, host_st = res["1"][2], res["0"][2]
    natt = dev_st["naccept"] + dev_st["nreject"]
    assert dev_st["nf"] == (1 if dt0 is not None else 2) + 6 * natt, dev_st                 # the one-launch path ran
    assert dev_st["controller"] == "device" and host_st["controller"] == "host"
    auto = bool(kw.get("autonomous", False))
    slack = 0 if (auto or tol >= 1e-4) else 2
    assert abs(dev_st["naccept"] - host_st["naccept"]) <= slack and abs(dev_st["nreject"] - host_st["nreject"]) <= slack, (dev_st, host_st)
    assert dev_st["naccept"] >= 3
    auto = bool(kw.get("autonomous", False))
    scale = max(1.0, float(res["0"][1].abs().max()))
    if auto:
        assert np.allclose(dev_st["dts"], host_st["dts"], rtol=1e-6), (dev_st["dts"], host_st["dts"])
    elif tol >= 1e-4:
        assert np.allclose(dev_st["dts"], host_st["dts"], rtol=1e-2), (dev_st["dts"], host_st["dts"])
    assert abs(sum(dev_st["dts"]) - 1.0) < 1e-5
    assert float((res["1"][1] - res["0"][1]).abs().max()) < (5e-6 if auto else 2 * tol + 5e-6) * scale
    lscale = max(1.0, float(res["0"][0].abs().max()))
    assert float((res["1"][0] - res["0"][0]).abs().max()) < (5e-6 if auto else 2 * tol + 5e-6) * lscale
    if B <= 1000:
        u0 = np.vstack([xs.astype(np.float64), np.zeros((spec.naug + 3, B))])
        uref, sref = o64.integrate_adaptive_tsit5(spec, p, u0, 0.0, 1.0, tol, tol, eps, ys, dt0=dt0)
        assert abs(dev_st["naccept"] - sref["naccept"]) <= (1 if tol >= 1e-4 else 3)
        assert np.max(np.abs(res["1"][1].cpu().numpy() - uref)) < 2e-4 * scale


@pytest.mark.parametrize("kw,B,tol,dt0", [
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 48, 1e-5, 2.0 ** -7),      # cfg2's kernel, RNODE rows live
    (dict(nvars=8, hidden=[64, 64, 64]), 1000, 1e-4, None),                                 # ragged last tile, Hairer's initial step
    (dict(nvars=1, hidden=[8, 8], act=2), 1024, 1e-4, None),                                # the reference's benchmark net (nvariables = 1)
    (dict(nvars=5, naug=2, hidden=[24, 24], act=2), 16, 1e-7, None),                        # orders up to 8+, one tile
    (dict(nvars=3, hidden=[24, 24], act=2, mode=2), 200, 1e-5, None),                       # exact trace (TestMode)
    (dict(nvars=8, hidden=[64, 64, 64], autonomous=True), 16379, 1e-4, None),               # every wave slot of the 256-thread kernel
    (dict(nvars=9, naug=10, hidden=[80, 80], act=2), 100, 1e-4, None),                      # reference default net for nvariables = 9 (8 state k-steps)
    (dict(nvars=2, naug=3, ncond=2, hidden=[32, 32], act=2), 300, 1e-4, None),              # conditioned default-style net
    (dict(nvars=2, naug=3, ncond=2, hidden=[32, 32], act=2, mode=2), 64, 1e-5, None),       # ... TestMode
    (dict(nvars=2, hidden=[32, 32]), 6000, 1e-4, None),                                     # 375 tiles on 256 workgroups: one or two of the four waves own a tile
    (dict(nvars=2, hidden=[32, 32]), 16384, 1e-4, None),                                    # a small net at the kernel's capacity at one workgroup per CU
    (dict(nvars=2, hidden=[32, 32]), 32768, 1e-4, None),                                    # ... and at two (nets of <= 2 hidden tiles, round 5)
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, reg_z=True, reg_j=True, reg_aug=True), 24001, 1e-4, None),   # default net, three of four waves own a tile
])
def test_default_solver_on_the_device_is_the_host_policy(kw, B, tol, dt0, pkg, oracles, monkeypatch):
    """The reference's default solver VCABM in ONE launch (mfma_vcabm_kernel: predictor, corrector and order-raising passes per
    tile in registers, grid-wide error norms, the step-size and order policy in every wave) against cnf_solve_vcabm's host loop
    over the device passes of cnf_vcabm.hip: the same evaluations at the same arguments and the same float32 arithmetic per
    element, so the same orders, accepted / rejected counts and step sizes (the error norms are summed in a different order, in
    double: decisions could differ only on a knife edge) and states equal to a few ulp."""
    o64, _ = oracles
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 6, bias_scale=0.3)
    p = (p * 2.0).astype(np.float32)
    icnf = make_icnf(pkg, spec, 1, 1)
    icnf.sol_kwargs = dict(reltol=tol, abstol=tol, **({} if dt0 is None else dict(dt=dt0)))   # alg defaults to VCABM()
    res = {}
    for ctl in ("1", "0"):
        setsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER", ctl)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        res[ctl] = (logp, u1, dict(icnf.last_solve_stats))
    d, h = res["1"][2], res["0"][2]
    assert d["controller"] == "device" and h["controller"] == "host" and d["alg_used"] == "VCABM"
    assert d["orders"] == h["orders"] and (d["naccept"], d["nreject"], d["nf"]) == (h["naccept"], h["nreject"], h["nf"]), (d, h)
    assert d["naccept"] >= 5 and np.allclose(d["dts"], h["dts"], rtol=1e-6), (d["dts"], h["dts"])
    scale = max(1.0, float(res["0"][1].abs().max()))
    assert float((res["1"][1] - res["0"][1]).abs().max()) < 5e-6 * scale
    assert float((res["1"][0] - res["0"][0]).abs().max()) < 5e-6 * max(1.0, float(res["0"][0].abs().max()))
    if tol <= 1e-7:
        assert max(d["orders"]) >= 6, d["orders"]


@pytest.mark.parametrize("kw,B,tol", [
    (dict(nvars=1, naug=2, hidden=[16, 16], act=2, reg_z=True, reg_j=True, reg_aug=True), 1024, 1e-4),      # the reference's benchmark flow
    (dict(nvars=8, hidden=[64, 64, 64], reg_z=True, reg_j=True), 1000, 1e-5),                               # cfg2's kernel, ragged tile
    (dict(nvars=8, hidden=[64, 64, 64], nprobes=4, reg_z=True, reg_j=True), 500, 1e-4),                     # cfg3: four probes
    (dict(nvars=3, hidden=[24, 24], act=2, mode=2), 200, 1e-5),                                             # TestMode (unit probes in the sweep)
    (dict(nvars=5, naug=2, ncond=3, hidden=[32, 32], act=2, reg_z=True), 300, 1e-4),                        # conditioned, zero-padded instance
    (dict(nvars=2, hidden=[32, 32]), 20000, 1e-4),                                                          # 1250 tiles: several waves per workgroup
    (dict(nvars=8, hidden=[64, 64, 64]), 40000, 1e-4),                                                      # beyond the one-launch kernel: the host loop's fused attempts fill the slots
    (dict(nvars=10, naug=11, hidden=[88, 88], act=2, reg_z=True, reg_j=True, reg_aug=True), 500, 1e-4),     # slab-accumulator kernel (default architecture, nvariables = 10): it reads the solve's checkpoints, 8 state k-steps wide
    (dict(nvars=8, hidden=[64, 64, 64], mode=1, reg_z=True), 700, 1e-4),                                    # JVP mode: the solve on the tangent engine, the sweep on the VJP twin
])
def test_adaptive_solve_writes_the_checkpoints_of_its_own_gradient(kw, B, tol, pkg, oracles, monkeypatch):
    """cnf_loss_grad_adaptive (round 5): where the frozen-grid gradient is the fused per-wave sweep, the one-launch adaptive Tsit5
    solve that finds the grid also writes the sweep's checkpoints - z_n and the six stage derivatives of every accepted step - so the
    gradient needs no forward pass of its own.  Against the same call with its own step-by-step forward pass (CNF_ADAPTIVE_CKPT=0):
    the same grid; the two forward passes differ by an ulp in t and dt per step, so loss and gradient agree to 1e-5."""
    o64, _ = oracles
    setsw(pkg, monkeypatch, "CNF_COOP_GRAD_MID", "0")       # (the slab case: 5 - 8 hidden tiles otherwise take the auxiliary cooperative sweep)
    spec = o64.make_spec(**kw)
    p, xs, eps, ys = o64.synth_inputs(spec, B, 17, bias_scale=0.2)
    out = {}
    for tag, env in (("solve", "1"), ("own", "0")):
        setsw(pkg, monkeypatch, "CNF_ADAPTIVE_CKPT", env)
        icnf = make_icnf(pkg, spec, 1, 4, path=0, lambdas=(0.01, 0.01, 0.01))
        icnf.sol_kwargs = dict(alg=pkg.Tsit5(), reltol=tol, abstol=tol)
        mode = mode_of(pkg, spec)
        assert icnf.grad_path(mode, B=B, alg=1, on_grid=True) == 1
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
        val, g, gx = pkg.loss_and_gradient(icnf, mode, *args, eps=dev(eps), wrt_x=True)
        st = icnf.last_solve_stats
        out[tag] = (float(val), g.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64), list(st["tgrid"]))
    assert out["solve"][3] == out["own"][3] and len(out["solve"][3]) >= 3
    assert abs(out["solve"][0] - out["own"][0]) < 1e-5 * (1 + abs(out["own"][0]))
    for k in (1, 2):
        a, b = out["solve"][k], out["own"][k]
        assert np.max(np.abs(a - b)) < 1e-5 * np.abs(b).max() + 1e-7, np.max(np.abs(a - b)) / np.abs(b).max()


@pytest.mark.parametrize("solver", ["tsit5", "vcabm"])
def test_one_launch_solves_longer_than_the_pinned_record(solver, pkg, oracles, monkeypatch):
    """The one-launch kernels write their status words and the first 120 accepted steps into pinned host memory (round 5); a longer
    solve's record comes from the device arrays.  A stiff small flow at 1e-6 takes 140 (Tsit5) / 420 (VCABM) steps: the record is
    complete, the steps add up to the span, and the solve is the library's host loop's to the first steps' last digit."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=2, hidden=[32, 32], autonomous=True)      # (autonomous: the two Tsit5 paths then differ in summation order only)
    p, xs, eps, ys = o64.synth_inputs(spec, 64, 5, bias_scale=0.3)
    p = (p * 8.0).astype(np.float32)
    icnf = make_icnf(pkg, spec, 1, 1)
    icnf.sol_kwargs = dict(alg=pkg.Tsit5() if solver == "tsit5" else pkg.VCABM(), reltol=1e-6, abstol=1e-6)
    res = {}
    for ctl in ("1", "0"):
        setsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER", ctl)
        logp, regs, u1 = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
        res[ctl] = (logp, u1, dict(icnf.last_solve_stats))
    d, h = res["1"][2], res["0"][2]
    assert d["controller"] == "device" and h["controller"] == "host"
    assert d["naccept"] > 130 and len(d["dts"]) == d["naccept"], (d["naccept"], len(d["dts"]))
    assert abs(sum(d["dts"]) - 1.0) < 1e-4
    if solver == "vcabm":
        assert len(d["orders"]) == d["naccept"] and 1 <= min(d["orders"]) and max(d["orders"]) <= 12
        assert d["orders"][:40] == h["orders"][:40]
    assert np.allclose(d["dts"][:40], h["dts"][:40], rtol=1e-4), (d["dts"][:40], h["dts"][:40])
    assert abs(d["naccept"] - h["naccept"]) <= 0.05 * h["naccept"], (d["naccept"], h["naccept"])
    scale = max(1.0, float(res["0"][0].abs().max()))
    assert float((res["1"][0] - res["0"][0]).abs().max()) < 1e-3 * scale


def test_device_side_step_controller_reports_failures(pkg, oracles):
    """The one-launch adaptive solve fails as loudly as the host loop: maxiters, a non-finite error estimate."""
    o64, _ = oracles
    spec = o64.make_spec(nvars=8, hidden=[64, 64, 64])
    p, xs, eps, ys = o64.synth_inputs(spec, 64, 5, bias_scale=0.3)
    icnf = _adaptive_icnf(pkg, spec, 1e-7, maxiters=3)
    with pytest.raises(Exception, match="maxiters"):
        run_inference(pkg, icnf, spec, (p * 2.0).astype(np.float32), xs, eps, ys)
    icnf = _adaptive_icnf(pkg, spec, 1e-4)
    bad = p.copy()
    bad[-3:] = np.inf                                                                       # a bias at infinity: non-finite dynamics
    with pytest.raises(FloatingPointError, match="non-finite"):
        run_inference(pkg, icnf, spec, bad, xs, eps, ys)


def test_usage_example_runs_end_to_end():
    """examples/usage.py - the reference's examples/usage.jl on the HIP path (data, ICNF, ICNFModel fit for 300 epochs with the
    default VCABM solver and STEER, ICNFDist pdf / rand) - as its own process."""
    import json, os, subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "usage.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    res = json.loads(lines[-1])
    assert lines[0].startswith("Iteration: 1 | Loss: ") and len(lines) == 6             # iterations 1, 65, 129, 193, 257
    first = float(lines[0].split("Loss: ")[1])
    assert res["fit_iterations"] == 300 and res["final_loss"] < first - 1.0
    assert abs(res["new_data_mean"] - res["true_mean"]) < 0.2 and all(np.isfinite(v) for v in res.values())


def test_randomised_shapes_under_the_adaptive_solvers(pkg, oracles, monkeypatch):
    """30 random configurations (state size, conditions, widths, depth, activation, trace mode, probes, regularisers, ragged B,
    either time direction) through the default solver VCABM, adaptive Tsit5 and - where a gradient exists - the one-call
    adaptive training step, each against a fine fixed-step solve / the frozen-grid gradient of the same library: every
    dynamics family (fused, cooperative-free generic, layer-wise, SIMT) under every solver loop."""
    import os
    o64, _ = oracles
    rng = np.random.default_rng(int(os.environ.get("CNF_FUZZ_SEED", 20240703)) + 17)
    tol = 1e-5
    kinds = set()
    for it in range(30):
        D = int(rng.integers(1, 12))
        naug = int(rng.integers(0, min(3, D)))
        C = int(rng.choice([0, 0, 2, 5]))
        L = int(rng.choice([1, 2, 2, 3, 4]))
        hidden = [int(rng.choice([8, 16, 24, 32, 48, 64]))] * L
        if rng.integers(0, 3) == 0:
            hidden = [int(rng.choice([8, 12, 16, 24, 32, 40])) for _ in range(L)]
        mode = int(rng.choice([0, 0, 1, 2]))
        reg = bool(rng.integers(0, 2)) and mode != 2
        K = int(rng.choice([1, 1, 2, 3])) if mode == 0 else 1
        kw = dict(nvars=D - naug, naug=naug, ncond=C, hidden=hidden, act=int(rng.choice([1, 2])), mode=mode, nprobes=K,
                  autonomous=bool(rng.integers(0, 4) == 0), reg_z=reg, reg_j=reg, reg_aug=reg and naug > 0)
        spec = o64.make_spec(**kw)
        B = int(rng.integers(1, 70))
        p, xs, eps, ys = o64.synth_inputs(spec, B, 3000 + it, bias_scale=0.3)
        p = (p * 1.5).astype(np.float32)
        fine = run_inference(pkg, make_icnf(pkg, spec, 1, 200), spec, p, xs, eps, ys, return_state=True)
        scale = max(1.0, float(fine[2].abs().max()))
        path = int(rng.choice([0, 0, 1, 3]))                   # the library's choice, or the SIMT / layer-wise family forced
        for alg in (pkg.VCABM(), pkg.Tsit5()):
            icnf = make_icnf(pkg, spec, 1, 1, path=path)
            icnf.sol_kwargs = dict(alg=alg, reltol=tol, abstol=tol)
            got = run_inference(pkg, icnf, spec, p, xs, eps, ys, return_state=True)
            st = icnf.last_solve_stats
            assert st["naccept"] >= 2 and float((got[2] - fine[2]).abs().max()) < 300 * tol * scale, (kw, B, type(alg).__name__, st)
            kinds.add((icnf.kernel_path(mode_of(pkg, spec)), type(alg).__name__))
        # the adaptive training step (Tsit5 grid): one library call vs the host loop + cnf_loss_grad_grid
        mode_obj = mode_of(pkg, spec)
        args = (dev(xs),) + ((dev(ys),) if spec.ncond else ()) + (dev(p), {})
        icnf.sol_kwargs["dt"] = 2.0 ** -5
        res = {}
        setsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER", "0")        # bit for bit: the library's HOST loop is the Python loop,
        setsw(pkg, monkeypatch, "CNF_ADAPTIVE_CKPT", "0")            # with the gradient's own forward pass on the frozen grid
        for pol in ("library", "python"):
            icnf.adaptive_policy = pol
            res[pol] = pkg.loss_and_gradient(icnf, mode_obj, *args, eps=dev(eps), wrt_x=True)
        delsw(pkg, monkeypatch, "CNF_ADAPTIVE_CKPT")
        icnf.adaptive_policy = "library"                             # the default: the solve's attempts fill the sweep's checkpoints
        res["ckpt"] = pkg.loss_and_gradient(icnf, mode_obj, *args, eps=dev(eps), wrt_x=True)
        delsw(pkg, monkeypatch, "CNF_DEVICE_CONTROLLER")
        for a, b in zip(res["library"], res["python"]):
            assert torch.equal(torch.as_tensor(a), torch.as_tensor(b)), (kw, B)
        for a, b in zip(res["ckpt"], res["library"]):
            a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
            assert float((a - b).abs().max()) < 2e-5 * float(b.abs().max()) + 1e-7, (kw, B)
        assert bool(torch.isfinite(res["library"][1]).all())
        icnf.adaptive_policy = "library"                        # and the one-launch solve under the same training step
        one = pkg.loss_and_gradient(icnf, mode_obj, *args, eps=dev(eps), wrt_x=True)
        assert abs(float(one[0]) - float(res["library"][0])) < 1e-3 * max(1.0, abs(float(res["library"][0])))
    assert len(kinds) >= 4, kinds


def test_vcabm_against_the_committed_fixture(pkg):
    """The default solver against tests/golden/vcabm_default_softplus_aug.npz (fp64 restatement, generated by
    tests/golden/make_golden.py).  From a common explicit initial step (tags a0 / b0): at the reference's tolerance the same
    accepted / rejected counts and order history, steps within 2 %, state within 20 tol; at 1e-6 counts within 4 and the state
    within 100 tol.  With Hairer's (tiny) initial step (tags a / b) the same first steps, counts within a fifth, and the state
    within 200 tol (two step sequences, each within 100 tol of the solution).  No oracle code runs here."""
    import os
    from conftest import GOLDEN
    f = np.load(os.path.join(GOLDEN, "vcabm_default_softplus_aug.npz"))
    nn = pkg.Chain(pkg.Dense(6, 24, pkg.softplus), pkg.Dense(24, 24, pkg.softplus), pkg.Dense(24, 5))
    for tag in ("a0", "b0", "a", "b"):
        tol = float(f[f"tol_{tag}"])
        kw = dict(reltol=tol, abstol=tol)
        if tag.endswith("0"):
            kw["dt"] = 2.0 ** -7
        icnf = pkg.ICNF(nvariables=2, naugments=3, nn=nn, steer_rate=0.0, lambda1=0.01, lambda2=0.01, lambda3=0.0, device="cuda:0",
                        sol_kwargs=kw)
        _, _, u1 = pkg.inference(icnf, pkg.TrainMode(True), dev(f["xs"]), dev(f["p"]), {}, eps=dev(f["eps"]), return_state=True)
        st = icnf.last_solve_stats
        err = float(np.max(np.abs(u1.cpu().numpy() - f[f"u1_{tag}"])))
        na, nr = int(f[f"naccept_{tag}"]), int(f[f"nreject_{tag}"])
        if tag == "a0":
            assert (st["naccept"], st["nreject"]) == (na, nr) and st["orders"] == f["orders_a0"].tolist(), (st, na, nr)
            assert np.allclose(st["dts"], f["dts_a0"], rtol=2e-2) and err < 20 * tol, err
        elif tag == "b0":
            assert abs(st["naccept"] - na) <= 4 and abs(st["nreject"] - nr) <= 4, st
            assert max(st["orders"]) >= 8 and err < 100 * tol, err
        else:
            assert abs(st["naccept"] - na) <= max(4, na // 5) and abs(st["nreject"] - nr) <= 4, (st, na, nr)
            assert np.allclose(st["dts"][:4], f[f"dts_{tag}"][:4], rtol=2e-2) and st["orders"][:4] == [1, 2, 3, 3]
            assert err < 200 * tol, err


# ---------------------------------------------------------------------------------------------------
# the collective behind the C ABI: RCCL executed on the MI355X with one rank
# ---------------------------------------------------------------------------------------------------
def test_rccl_allreduce_through_the_abi_with_one_rank(pkg):
    """cnf_comm_unique_id + cnf_comm_init (ncclCommInitRank) + cnf_allreduce_loss / cnf_allreduce_sum (ncclAllReduce) on
    the caller's stream.  A 1-GPU lease cannot form a larger communicator (RCCL refuses two ranks on one device), so this
    pins the dlopen, the init and the enqueue order; N = 2 semantics are covered by the gloo tests."""
    dev = torch.device("cuda:0")
    comm = pkg.Comm(0, 1, pkg.Comm.unique_id(), dev)
    assert comm.lib.cnf_comm_rank(comm.ptr) == 0 and comm.lib.cnf_comm_size(comm.ptr) == 1
    sums = torch.tensor([3.5, 0.25, -1.0, 2.0], device=dev)
    out5 = comm.allreduce_loss(sums, 1234)
    torch.cuda.synchronize()
    assert out5.dtype == torch.float64 and out5.tolist() == [3.5, 0.25, -1.0, 2.0, 1234.0]
    # stream order: the reduction is enqueued behind the kernel that produces sums4 on the same (non-default) stream
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        big = torch.randn(1 << 22, device=dev)
        sums2 = torch.stack([big.sum(), big.abs().sum(), (big * big).sum(), big.max()])
        out = comm.allreduce_loss(sums2, 7).clone()
    s.synchronize()
    assert torch.allclose(out[:4], sums2.double()) and out[4].item() == 7.0
    g32 = torch.arange(9480, device=dev, dtype=torch.float32)
    g64 = torch.tensor([1e-30, 2.0, 3e30], device=dev, dtype=torch.float64)
    comm.allreduce_sum(g32)
    comm.allreduce_sum(g64)
    torch.cuda.synchronize()
    assert torch.equal(g32, torch.arange(9480, device=dev, dtype=torch.float32)) and g64.tolist() == [1e-30, 2.0, 3e30]
    with pytest.raises(ValueError):
        comm.allreduce_sum(torch.zeros(4, device=dev, dtype=torch.int32))
    # the host reductions route through the installed communicator and agree with the local formula
    pkg.set_comm(comm)
    try:
        lam = (0.01, 0.02, 0.03)
        got = float(pkg.reduce_loss(sums, 100, lam))
        assert abs(got - (3.5 + 0.01 * 0.25 - 0.02 * 1.0 + 0.03 * 2.0) / 100) < 1e-7
        gr = pkg.reduce_gradient(torch.full((64,), 50.0, device=dev), 100)
        assert torch.allclose(gr, torch.full((64,), 0.5, device=dev))
        # group=False opts out even though a communicator is installed
        assert abs(float(pkg.reduce_loss(sums, 100, lam, group=False)) - got) < 1e-7
    finally:
        pkg.set_comm(None)
        comm.destroy()


def test_comm_init_all_with_one_device_and_grouped_reductions(pkg):
    """cnf_comm_init_all (ncclCommInitAll: one host process driving several devices - the shape a Julia host without
    MPI uses, julia/hip_ext/comm.jl) with the one device a lease has, and the grouped form of the reductions
    (cnf_comm_group_start / _end around the per-device calls, which RCCL requires when one thread drives several ranks)."""
    import ctypes as C
    lib = pkg._lib.load()
    dev = torch.device("cuda:0")
    comms = (C.c_void_p * 1)()
    devs = (C.c_int * 1)(0)
    pkg._lib.check(lib.cnf_comm_init_all(comms, 1, devs))
    c = C.c_void_p(comms[0])
    try:
        assert lib.cnf_comm_rank(c) == 0 and lib.cnf_comm_size(c) == 1
        sums = torch.tensor([1.5, -2.0, 0.125, 8.0], device=dev)
        out5 = torch.zeros(5, dtype=torch.float64, device=dev)
        g = torch.arange(1000, device=dev, dtype=torch.float32)
        st = pkg._lib.stream_ptr(dev)
        pkg._lib.check(lib.cnf_comm_group_start())
        pkg._lib.check(lib.cnf_allreduce_loss(c, pkg._lib.ptr(sums), 321, pkg._lib.ptr(out5), st))
        pkg._lib.check(lib.cnf_allreduce_sum(c, pkg._lib.ptr(g), g.numel(), pkg._lib.DTYPE_F32, st))
        pkg._lib.check(lib.cnf_comm_group_end())
        torch.cuda.synchronize()
        assert out5.tolist() == [1.5, -2.0, 0.125, 8.0, 321.0]
        assert torch.equal(g, torch.arange(1000, device=dev, dtype=torch.float32))
    finally:
        lib.cnf_comm_destroy(c)
    bad = (C.c_int * 1)(99)
    assert lib.cnf_comm_init_all(comms, 1, bad) != 0          # a device that does not exist: a status code, not a crash
    assert lib.cnf_comm_init_all(comms, 0, devs) == pkg._lib.ERR_INVALID


@pytest.mark.parametrize("launcher", ["torchrun", "bare"])
@pytest.mark.parametrize("mode", ["infer", "grad"])
def test_bench_nccl_launch_path_rehearsed_with_one_rank(launcher, mode):
    """Every line a multi-GPU `bench.py --gpus N` run executes, executed on the one GPU a lease has: launched by
    torch.distributed.run exactly as the driver launches N > 1 (or bare, with --force-dist supplying the rendezvous),
    init_process_group("nccl"), the library's RCCL communicator formed over the group (unique id broadcast as a device tensor
    from the main thread; ncclCommInitRank in the watchdog thread), cnf_loss_sums + cnf_allreduce_loss INSIDE the timed loop,
    the all-reduced pre-roll count and max-over-ranks time, comm destroy + destroy_process_group.  The one exchange is
    the mean of src/core/icnf.jl:636."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--batch", "4096",
            "--mode", mode, "--preroll-seconds", "0.2", "--no-cpu-baseline"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
        env = dict(os.environ)
    else:
        cmd = [sys.executable] + tail
        env = dict(os.environ, MASTER_PORT=str(port))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["config"]["collective"].startswith("cnf_allreduce_loss"), out["config"]["collective"]
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    # the rank reports the device it bound and the size the RCCL communicator ITSELF reports (ncclCommCount through cnf_comm_size)
    assert out["ranks_seen"] == [dict(rank=0, device=0, cnf_comm_size=1, process_group_size=1)], out["ranks_seen"]
    assert "bound cuda:0" in r.stderr and "cnf_comm_size() = 1" in r.stderr
    # the same numbers as the unsharded step on the same columns (one rank: the all-reduce is the identity)
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                             "--batch", "4096", "--mode", mode, "--preroll-seconds", "0", "--secondary", "none"],
                            capture_output=True, text=True, timeout=600, cwd=ROOT)
    one = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][0])
    assert abs(out["loss"] - one["loss"]) < 2e-6 * max(1.0, abs(one["loss"])), (out["loss"], one["loss"])
    # and the torch.distributed transport, selected explicitly, on the same path
    r2 = subprocess.run(cmd + ["--collective", "torch", "--secondary", "none"], capture_output=True, text=True, timeout=600,
                        cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    out2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert out2["config"]["collective"].startswith("torch.distributed all_reduce (nccl)")
    assert abs(out2["loss"] - one["loss"]) < 2e-6 * max(1.0, abs(one["loss"]))


def test_loss_through_the_library_communicator_matches_the_local_loss(pkg, oracles):
    """loss / loss_and_gradient / the adaptive solvers with a Comm installed (one rank): same numbers as without."""
    o64, _ = oracles
    spec = o64.make_spec(8, [64, 64, 64], reg_z=True, reg_j=True)
    B = 777
    p, xs, eps, _ = o64.synth_inputs(spec, B, 91, bias_scale=0.2)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda:0")
    mk = lambda kw: pkg.ICNF(nvariables=8, naugments=0, nn=pkg.Chain(pkg.Dense(9, 64, "tanh"), pkg.Dense(64, 64, "tanh"),
                                                                     pkg.Dense(64, 64, "tanh"), pkg.Dense(64, 8)),
                             steer_rate=0.0, lambda1=0.02, lambda2=0.03, lambda3=0.0, device="cuda:0", sol_kwargs=kw)
    m = pkg.TrainMode(True)
    res = {}
    for tag in ("local", "comm"):
        comm = None
        if tag == "comm":
            comm = pkg.Comm(0, 1, pkg.Comm.unique_id(), torch.device("cuda:0"))
            pkg.set_comm(comm)
        try:
            fixed = mk(dict(alg=pkg.Tsit5(), adaptive=False, nsteps=10))
            v, g = pkg.loss_and_gradient(fixed, m, dev(xs), dev(p), {}, eps=dev(eps))
            lv = pkg.loss(fixed, m, dev(xs), dev(p), {}, eps=dev(eps))
            adap = mk(dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))
            la = pkg.inference(adap, m, dev(xs), dev(p), {}, eps=dev(eps))[0]
            dflt = mk(dict(reltol=1e-4, abstol=1e-4))
            lvc = pkg.inference(dflt, m, dev(xs), dev(p), {}, eps=dev(eps))[0]
            res[tag] = (float(v), g.cpu().numpy(), float(lv), la.cpu().numpy(), list(adap.last_solve_stats["dts"]),
                        lvc.cpu().numpy(), list(dflt.last_solve_stats["orders"]))
        finally:
            if comm is not None:
                pkg.set_comm(None)
                comm.destroy()
    a, b = res["local"], res["comm"]
    assert abs(a[0] - b[0]) < 1e-6 and abs(a[2] - b[2]) < 1e-6 and np.allclose(a[1], b[1], rtol=1e-6, atol=1e-8)
    assert np.allclose(a[4], b[4], rtol=1e-6) and np.max(np.abs(a[3] - b[3])) < 1e-5      # same adaptive steps
    assert a[6] == b[6] and np.max(np.abs(a[5] - b[5])) < 1e-4                              # VCABM: same orders

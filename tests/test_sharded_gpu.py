"""GPU, world_size 2 (gloo; both ranks share the one device, RCCL would refuse that): the sharded product path.
Each rank runs the HIP kernels on its contiguous column block; loss and gradient are all-reduced inside
`loss` / `loss_and_gradient`, and the adaptive solve all-reduces its error sum so both ranks take the steps of the
unsharded solve (Tsit5 and the reference's default VCABM, whose order selection reads three more all-reduced sums)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build(pkg, o64, spec, sol_kwargs):
    acts = ["identity", "tanh", "softplus"]
    layers = [pkg.Dense(spec.widths[i], spec.widths[i + 1], acts[spec.acts[i]]) for i in range(len(spec.acts))]
    return pkg.ICNF(nvariables=spec.nvars, naugments=spec.naug, nn=pkg.Chain(*layers), steer_rate=0.0,
                    lambda1=0.02 if spec.reg_z else 0.0, lambda2=0.03 if spec.reg_j else 0.0, lambda3=0.0,
                    device="cuda:0", sol_kwargs=sol_kwargs)


def _case(pkg, o64):
    spec = o64.make_spec(8, [64, 64, 64], reg_z=True, reg_j=True)
    B = 1001                                                   # ragged: 501 + 500
    p, xs, eps, _ = o64.synth_inputs(spec, B, 33, bias_scale=0.2)
    return spec, B, (p * 1.5).astype(np.float32), xs, eps


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    o64, _ = entry.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec, B, p, xs, eps = _case(pkg, o64)
        lo, hi = pkg.shard_columns(B, rank, world)
        dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda:0")
        m = pkg.TrainMode(True)
        fixed = _build(pkg, o64, spec, dict(alg=pkg.Tsit5(), adaptive=False, nsteps=10))
        val, g = pkg.loss_and_gradient(fixed, m, dev(xs[:, lo:hi]), dev(p), {}, eps=dev(eps[:, lo:hi]))
        lval = pkg.loss(fixed, m, dev(xs[:, lo:hi]), dev(p), {}, eps=dev(eps[:, lo:hi]))
        adap = _build(pkg, o64, spec, dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))
        logp = pkg.inference(adap, m, dev(xs[:, lo:hi]), dev(p), {}, eps=dev(eps[:, lo:hi]))[0]
        adap_dts = list(adap.last_solve_stats["dts"])
        dflt = _build(pkg, o64, spec, dict(reltol=1e-4, abstol=1e-4))          # the reference's default solver, VCABM
        logp_v = pkg.inference(dflt, m, dev(xs[:, lo:hi]), dev(p), {}, eps=dev(eps[:, lo:hi]))[0]
        vst = dflt.last_solve_stats
        # --- (a) STEER with rng-drawn probes: every rank must integrate the same drawn t1 although the
        # probe generators differ per rank; (b) an EMPTY shard (B = 1 over 2 ranks) must join every all-reduce of the
        # adaptive solvers instead of returning early; (c) group=False: a rank-local adaptive solve on ONE rank only
        steer = pkg.ICNF(nvariables=spec.nvars, naugments=0, nn=fixed.nn, steer_rate=0.3, lambda1=0.02, lambda2=0.03, lambda3=0.0,
                         device="cuda:0", sol_kwargs=dict(reltol=1e-4, abstol=1e-4))       # default solver (VCABM), STEER on
        t1s, seeds = [], int(steer.rng.initial_seed())
        for _ in range(3):
            pkg.loss(steer, m, dev(xs[:, lo:hi]), dev(p), {})                              # probes drawn from icnf.rng
            t1s.append(round(sum(steer.last_solve_stats["dts"]), 6))
        l1, h1 = pkg.shard_columns(1, rank, world)                                         # rank 0: one column, rank 1: none
        tiny = []
        for kw in (dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4), dict(reltol=1e-4, abstol=1e-4),
                   dict(alg=pkg.Tsit5(), adaptive=False, nsteps=6)):
            ic = _build(pkg, o64, spec, kw)
            tiny.append(float(pkg.loss(ic, m, dev(xs[:, l1:h1]), dev(p), {}, eps=dev(eps[:, l1:h1]))))
        local = None
        if rank == 0:
            local = pkg.inference(adap, m, dev(xs[:, :64]), dev(p), {}, eps=dev(eps[:, :64]), group=False)[0].cpu().numpy()
        dist.barrier()
        q.put((rank, float(val), float(lval), g.cpu().numpy(), logp.cpu().numpy(), adap_dts,
               logp_v.cpu().numpy(), list(vst["dts"]), list(vst["orders"]), t1s, seeds, tiny, local))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_the_gpu_reproduce_the_unsharded_results(pkg, oracles):
    o64, _ = oracles
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=500) for _ in procs), key=lambda r: r[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    # unsharded reference in this process (no process group here)
    spec, B, p, xs, eps = _case(pkg, o64)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda:0")
    m = pkg.TrainMode(True)
    fixed = _build(pkg, o64, spec, dict(alg=pkg.Tsit5(), adaptive=False, nsteps=10))
    val, g = pkg.loss_and_gradient(fixed, m, dev(xs), dev(p), {}, eps=dev(eps))
    adap = _build(pkg, o64, spec, dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4))
    adap.adaptive_policy = "python"                       # the host loop the ranks run (a single process would get the one-launch solve)
    logp = pkg.inference(adap, m, dev(xs), dev(p), {}, eps=dev(eps))[0].cpu().numpy()
    dts = list(adap.last_solve_stats["dts"])
    assert len(dts) >= 4
    dflt = _build(pkg, o64, spec, dict(reltol=1e-4, abstol=1e-4))
    dflt.adaptive_policy = "python"                       # the host loop the ranks run (the library's single-call policy is its twin)
    logp_v = pkg.inference(dflt, m, dev(xs), dev(p), {}, eps=dev(eps))[0].cpu().numpy()
    vdts, vorders = list(dflt.last_solve_stats["dts"]), list(dflt.last_solve_stats["orders"])
    assert isinstance(dflt.sol_kwargs["alg"], pkg.VCABM) and len(vdts) >= 5
    # the extra checks first (they do not need the unsharded numbers computed above)
    assert res[0][9] == res[1][9] and len(set(res[0][9])) == 3        # STEER: same t1 on both ranks, a new draw per call
    assert all(abs(t - 1.0) <= 0.3 + 1e-6 and abs(t - 1.0) > 1e-4 for t in res[0][9])
    assert res[0][10] != res[1][10]                                   # probe generators are seeded per rank
    assert res[0][11] == res[1][11]                                   # empty shard: both ranks return the global mean ...
    one = []
    for kw in (dict(alg=pkg.Tsit5(), reltol=1e-4, abstol=1e-4), dict(reltol=1e-4, abstol=1e-4), dict(alg=pkg.Tsit5(), adaptive=False, nsteps=6)):
        ic = _build(pkg, o64, spec, kw)
        one.append(float(pkg.loss(ic, m, dev(xs[:, :1]), dev(p), {}, eps=dev(eps[:, :1]))))
    assert np.allclose(res[0][11], one, rtol=1e-5, atol=1e-5), (res[0][11], one)   # ... of the one column
    adap.adaptive_policy = "library"                      # a rank-local solve is a single-process solve: the library's own policy
    loc = pkg.inference(adap, m, dev(xs[:, :64]), dev(p), {}, eps=dev(eps[:, :64]))[0].cpu().numpy()
    assert res[1][12] is None and np.max(np.abs(res[0][12] - loc)) < 1e-5          # rank-local solve while a group exists
    res = [r[:9] for r in res]
    for rank, v, lv, gr, lp, d, lpv, dv, ov in res:
        assert abs(v - float(val)) < 1e-5 and abs(lv - float(val)) < 1e-5          # global mean on every rank
        assert np.max(np.abs(gr - g.cpu().numpy())) < 2e-5 * float(g.abs().max())  # all-reduced gradient = unsharded gradient
        assert np.allclose(d, dts, rtol=1e-6), (d, dts)                            # the unsharded solve's steps
        lo, hi = pkg.shard_columns(B, rank, 2)
        assert np.max(np.abs(lp - logp[lo:hi])) < 1e-5
        assert ov == vorders and np.allclose(dv, vdts, rtol=1e-5), (dv, vdts, ov, vorders)   # VCABM: same orders, same steps
        assert np.max(np.abs(lpv - logp_v[lo:hi])) < 1e-4
    assert res[0][5] == res[1][5]                                                  # both ranks took identical steps

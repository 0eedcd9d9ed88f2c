"""CPU, world_size 2, gloo: the N>1 path.  Each rank evaluates its contiguous column shard
(with the CPU oracle standing in for the kernels, which need a GPU) and the loss reduction of
continuousnormalizingflows.jl_amd/sharding.py must reproduce the unsharded loss."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    o64, oc = entry.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = o64.make_spec(8, [64, 64, 64], nprobes=2, reg_z=True, reg_j=True)
        B = 101                                        # ragged: 51 + 50
        p, xs, eps, _ = o64.synth_inputs(spec, B, 21, bias_scale=0.1)
        lam = (0.01, 0.02, 0.0)
        lo, hi = pkg.shard_columns(B, rank, world)
        logp, (E, n, A), _ = oc.inference_fixed(spec, p, xs[:, lo:hi], 0.0, 1.0, 8, o64.ALG_TSIT5,
                                                eps[:, lo:hi])
        sums = torch.tensor([-logp.sum(dtype=np.float64), E.sum(dtype=np.float64),
                             n.sum(dtype=np.float64), A.sum(dtype=np.float64)], dtype=torch.float32)
        got = float(pkg.reduce_loss(sums, hi - lo, lam))
        # shard outputs are bit-identical to the same columns of the unsharded run
        full = oc.inference_fixed(spec, p, xs, 0.0, 1.0, 8, o64.ALG_TSIT5, eps)
        same = bool(np.array_equal(full[0][lo:hi], logp))
        ref = float(np.mean(-full[0].astype(np.float64) + lam[0] * full[1][0] + lam[1] * full[1][1]))
        # gradient path: per-shard summed gradients (fp64 autograd oracle standing in for the
        # reverse-sweep kernel) all-reduced to the gradient of the global mean loss
        s1 = o64.make_spec(4, [16, 16])
        p1, x1, e1, _ = o64.synth_inputs(s1, 11, 22, bias_scale=0.1)
        l1, h1 = pkg.shard_columns(11, rank, world)
        _, gs = o64.loss_and_grad(s1, p1, x1[:, l1:h1], 0.0, 1.0, 3, o64.ALG_RK4, e1[:, l1:h1])
        gred = pkg.reduce_gradient(torch.tensor(gs * (h1 - l1)), h1 - l1).numpy()
        _, gfull = o64.loss_and_grad(s1, p1, x1, 0.0, 1.0, 3, o64.ALG_RK4, e1)
        gerr = float(np.max(np.abs(gred - gfull)))
        q.put((rank, got, ref, same, gerr))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_loss_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, ref, same, gerr in res:
        assert same, "sharded columns differ from the unsharded run"
        assert abs(got - ref) < 1e-5 * max(1.0, abs(ref)), (rank, got, ref)
        assert gerr < 1e-12, gerr
    assert res[0][1] == res[1][1]                    # every rank returns the same global mean

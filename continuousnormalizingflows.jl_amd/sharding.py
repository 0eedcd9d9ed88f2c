"""Batch-column sharding of the hot path across the GPUs of one node.

Columns (samples) are independent under fixed-step integration (every operation of
augmented_f is column-wise, src/core/icnf.jl:530-535), so rank r of G evaluates the contiguous
column block [r*B/G, (r+1)*B/G) with a full replica of the (tiny) weights and no data-path
collective.  The only exchange is the mean in `loss` (src/core/icnf.jl:636): one all-reduce of
five scalars (four partial sums + the column count) — RCCL over xGMI when the process group
backend is "nccl", gloo in the CPU tests.  The message is 40 bytes, i.e. latency-bound.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch


def shard_columns(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced column range [lo, hi) of rank `rank` (first B % world ranks get one
    extra column).  Column-major storage makes each shard a contiguous byte range."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


_const_cache = {}


def _const(device, values) -> torch.Tensor:
    """Small float64 device constants, created once per (device, values) — not per step."""
    key = (str(device), tuple(float(v) for v in values))
    t = _const_cache.get(key)
    if t is None:
        t = torch.tensor(key[1], device=device, dtype=torch.float64)
        _const_cache[key] = t
    return t


def reduce_loss(sums4: torch.Tensor, B_local: int, lambdas: Sequence[float],
                group=None) -> torch.Tensor:
    """(Σ-logp, ΣĖ, Σṅ, ΣȦ) of this rank's columns -> global mean loss on every rank.
    Partial sums are combined in float64 so the result does not depend on the rank count
    beyond fp32 rounding of the per-rank sums."""
    import torch.distributed as dist
    lam = _const(sums4.device, (1.0, *lambdas))
    if dist.is_available() and dist.is_initialized():
        buf = torch.cat([sums4.to(torch.float64), _const(sums4.device, (float(B_local),))])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return (torch.dot(buf[:4], lam) / buf[4]).to(torch.float32)
    return (torch.dot(sums4.to(torch.float64), lam) / float(B_local)).to(torch.float32)


def reduce_gradient(grad_sum: torch.Tensor, B_local: int, group=None) -> torch.Tensor:
    """Summed per-shard gradient (nparams floats) -> gradient of the global mean loss on every
    rank: the one collective of the training path whose size is not a handful of scalars
    (37-580 KiB over xGMI; one all-reduce, no bucketing needed at this size)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return grad_sum / float(B_local)                  # single process: no device-side count, no collective
    cnt = torch.tensor([float(B_local)], device=grad_sum.device, dtype=torch.float64)
    dist.all_reduce(grad_sum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return grad_sum / cnt.to(grad_sum.dtype)

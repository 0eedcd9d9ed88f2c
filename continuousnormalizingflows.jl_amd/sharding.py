"""Batch-column sharding of the hot path across the GPUs of one node.

Columns (samples) are independent under fixed-step integration (every operation of
augmented_f is column-wise, src/core/icnf.jl:530-535), so rank r of G evaluates the contiguous
column block [r*B/G, (r+1)*B/G) with a full replica of the (tiny) weights and no data-path
collective.  The only exchange is the mean in `loss` (src/core/icnf.jl:636): one all-reduce of
five scalars (four partial sums + the column count) — RCCL over xGMI when the process group
backend is "nccl", gloo in the CPU tests.  The message is 40 bytes, i.e. latency-bound.

Two transports carry that all-reduce:
  * `Comm` — the library's own RCCL communicator behind the C ABI (`cnf_comm_init`, `cnf_allreduce_loss`,
    `cnf_allreduce_sum`; include/cnf.h): what a Julia host calls, and what `reduce_loss` / `reduce_gradient` use when a
    communicator has been installed with `set_comm` (bench.py does for N > 1 on the nccl backend);
  * `torch.distributed` (nccl = RCCL, or gloo in the CPU tests) otherwise.
`group=False` anywhere means "this call is rank-local": no collective is issued even though a process group exists.
"""
from __future__ import annotations

import ctypes as C
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


class Comm:
    """One rank of an RCCL communicator owned by libcnf_hip.so (cnf_comm, include/cnf.h)."""

    def __init__(self, rank: int, nranks: int, uid: bytes, device):
        from . import _lib
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rank, self.nranks = int(rank), int(nranks)
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError(f"the RCCL unique id is {_lib.COMM_ID_BYTES} bytes")
        self.ptr = C.c_void_p()
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        _lib.check(self.lib.cnf_comm_init(C.byref(self.ptr), self.rank, self.nranks, buf, self.device.index or 0))
        self._out5 = torch.empty(5, dtype=torch.float64, device=self.device)

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId through the ABI (rank 0 calls this and ships the bytes to the others)."""
        from . import _lib
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.load().cnf_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_process_group(cls, device, group=None) -> "Comm":
        """Build the communicator over the ranks of an initialised torch.distributed group: rank 0's unique id
        travels through the group (the only use of torch.distributed on this transport)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0], device)

    def allreduce_loss(self, sums4: torch.Tensor, B_local: int) -> torch.Tensor:
        """(4 float sums, local column count) -> 5 doubles summed over the ranks, on the current stream."""
        from . import _lib
        _lib.check(self.lib.cnf_allreduce_loss(self.ptr, _lib.ptr(sums4), int(B_local), _lib.ptr(self._out5),
                                               _lib.stream_ptr(self.device)))
        return self._out5

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a contiguous float32 / float64 device tensor."""
        from . import _lib
        if not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64) or t.device != self.device:
            raise ValueError("allreduce_sum needs a contiguous float32/float64 tensor on the communicator's device")
        _lib.check(self.lib.cnf_allreduce_sum(self.ptr, _lib.ptr(t), t.numel(),
                                              _lib.DTYPE_F32 if t.dtype == torch.float32 else _lib.DTYPE_F64,
                                              _lib.stream_ptr(self.device)))
        return t

    def size(self) -> int:
        """Ranks the RCCL communicator itself reports (cnf_comm_size -> ncclCommCount), not what the caller passed in."""
        return int(self.lib.cnf_comm_size(self.ptr))

    def comm_rank(self) -> int:
        return int(self.lib.cnf_comm_rank(self.ptr))

    def destroy(self):
        if self.ptr:
            self.lib.cnf_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_comm: Optional[Comm] = None


def set_comm(comm: Optional[Comm]) -> None:
    """Install (or remove, with None) the library communicator the reductions below use."""
    global _comm
    _comm = comm


def get_comm() -> Optional[Comm]:
    return _comm


def is_sharded(group=None) -> bool:
    """True when this call should issue collectives: a library communicator or an initialised process group, and the
    caller has not opted out with group=False."""
    if group is False:
        return False
    if _comm is not None:
        return True
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def allsum(vals: Sequence[float], device, group=None):
    """Sum a handful of host scalars over the ranks in float64 (the adaptive solvers' error sums)."""
    if not is_sharded(group):
        return [float(v) for v in vals]
    if _comm is not None and torch.device(device).type == "cuda":
        t = torch.tensor(list(vals), dtype=torch.float64, device=device)
        _comm.allreduce_sum(t)
        return [float(v) for v in t.tolist()]
    import torch.distributed as dist
    t = torch.tensor(list(vals), dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, group=group)
    return [float(v) for v in t.tolist()]


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
    lam = _const(sums4.device, (1.0, *lambdas))
    if not is_sharded(group):
        return (torch.dot(sums4.to(torch.float64), lam) / float(B_local)).to(torch.float32)
    if _comm is not None and sums4.is_cuda:
        buf = _comm.allreduce_loss(sums4, B_local)
        return (torch.dot(buf[:4], lam) / buf[4]).to(torch.float32)
    import torch.distributed as dist
    buf = torch.cat([sums4.to(torch.float64), _const(sums4.device, (float(B_local),))])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (torch.dot(buf[:4], lam) / buf[4]).to(torch.float32)


def reduce_gradient(grad_sum: torch.Tensor, B_local: int, group=None) -> torch.Tensor:
    """Summed per-shard gradient (nparams floats) -> gradient of the global mean loss on every
    rank: the one collective of the training path whose size is not a handful of scalars
    (37-580 KiB over xGMI; one all-reduce, no bucketing needed at this size)."""
    if not is_sharded(group):
        return grad_sum / float(B_local)                  # single process: no device-side count, no collective
    cnt = torch.tensor([float(B_local)], device=grad_sum.device, dtype=torch.float64)
    if _comm is not None and grad_sum.is_cuda:
        g = grad_sum if grad_sum.is_contiguous() else grad_sum.contiguous()
        _comm.allreduce_sum(g)
        _comm.allreduce_sum(cnt)
        return g / cnt.to(g.dtype)
    import torch.distributed as dist
    dist.all_reduce(grad_sum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return grad_sum / cnt.to(grad_sum.dtype)


def global_count(B_local: int, device, group=None) -> int:
    """Total number of columns over the ranks."""
    return int(round(allsum([float(B_local)], device, group)[0]))

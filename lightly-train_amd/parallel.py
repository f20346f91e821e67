"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The DINOv2 step shards by images: every rank runs the full step on its own B images, masks and RNG; the only
exchanges are:
  C1   gradient mean over ranks        -> GradSync (bucketed async all-reduce on the flat fp32 grad buffer)
  C3/4 DINO / iBOT center sums         -> async all-reduce, consumed at the next step (dinov2.py, _pending)
  C5/6 Sinkhorn-Knopp row sums         -> synchronous all-reduce inside the teacher path
Parameters: `DINOv2.__init__` broadcasts rank 0's flat student / teacher storage once at construction (what DDP does when it wraps a
module); after that nothing but gradients is exchanged -- identical students give identical EMA teachers.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def per_rank_batch(global_batch: int, world: int) -> int:
    """common_helpers.py:665-719 semantics: the global batch must divide evenly over ranks."""
    if global_batch % world != 0:
        raise ValueError(f"Batch size {global_batch} must be divisible by (num_nodes * devices) = {world}.")
    return global_batch // world


class AbiComm:
    """The library's own RCCL communicator (include/lt_amd.h: lt_comm_*) as the gradient all-reduce transport instead of
    torch.distributed's.  `from_torch_group()` bootstraps it from an initialised torch process group: rank 0 draws the unique id, a
    byte-tensor broadcast hands it out.  One communicator per process."""
    _instance: Optional["AbiComm"] = None

    def __init__(self, rank_: int, world: int, unique_id: bytes) -> None:
        from . import _lib

        self.lib = _lib.load()
        _lib.check(self.lib.lt_comm_init(rank_, world, unique_id, len(unique_id)), "lt_comm_init")
        self.world = world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from . import _lib

        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().lt_comm_unique_id(buf, 128), "lt_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_group(cls, device: torch.device) -> "AbiComm":
        if cls._instance is None:
            r, w = rank(), world_size()
            t = torch.zeros(128, dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
            if r == 0:
                t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
            if w > 1:
                dist.broadcast(t, src=0)
            cls._instance = cls(r, w, bytes(t.cpu().tolist()))
        return cls._instance

    def all_reduce(self, t: Tensor) -> None:
        """In-place sum on the communicator's stream, ordered after the current stream's work (no host wait)."""
        from . import _lib, ops

        assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.lt_comm_allreduce_f32(t.data_ptr(), t.numel(), ops._stream()), "lt_comm_allreduce_f32")

    def wait(self) -> None:
        """The current stream waits for every collective enqueued so far."""
        from . import _lib, ops

        _lib.check(self.lib.lt_comm_wait(ops._stream()), "lt_comm_wait")

    def destroy(self) -> None:
        self.lib.lt_comm_destroy()
        AbiComm._instance = None


def bucket_ranges(numel: int, bucket_elems: int) -> List[Tuple[int, int]]:
    return [(o, min(o + bucket_elems, numel)) for o in range(0, numel, bucket_elems)]


class GradSync:
    """Mean-all-reduce of a flat gradient buffer.  xGMI is point-to-point (7 links/GPU), so large messages (up to 64 MiB
    fp32 per call) keep RCCL on its bandwidth-optimal algorithms; calls are issued async and back-to-back so RCCL
    pipelines them on its own stream while the caller continues.

    `start(lo, hi)` may be called during backward for a range whose gradients are final (enqueued on the CURRENT stream:
    the collective is ordered after it); `finish()` reduces whatever no `start` has covered, waits for everything on the
    current stream and applies the 1/world scale.  Every rank must issue the same ranges in the same order.

    LT_GRAD_COMM_MODE=rsag ("for the large message prefer a direct reduce-scatter / all-gather"): every call of at least
    `RSAG_MIN_ELEMS` elements runs as reduce_scatter_tensor + all_gather_into_tensor on the part divisible by the world size (the
    remainder, and short calls, stay all-reduces).  Each element is summed once, by the rank that owns its shard, and gathered: the
    replicas stay bit-identical.  Unmeasured on hardware like every N > 1 path here; the default is the plain all-reduce."""

    RSAG_MIN_ELEMS = 1 << 16

    def __init__(self, flat_grad: Tensor, bucket_bytes: int = 64 << 20, comm: Optional[AbiComm] = None) -> None:
        self.g = flat_grad
        # LT_GRAD_COMM=abi: the all-reduces go through the library's own RCCL communicator (lt_comm_*) instead of torch.distributed
        if comm is None and os.environ.get("LT_GRAD_COMM") == "abi" and flat_grad.is_cuda and world_size() > 1:
            comm = AbiComm.from_torch_group(flat_grad.device)
        self.comm = comm
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.ranges = bucket_ranges(flat_grad.numel(), self.bucket_elems)
        self.handles: List = []
        self.covered: List[Tuple[int, int]] = []
        self.rsag = os.environ.get("LT_GRAD_COMM_MODE", "") == "rsag" and self.comm is None
        self._shards: List[Tensor] = []     # reduce-scatter outputs, alive until their all-gather has run

    def uncovered(self, lo: int, hi: int) -> List[Tuple[int, int]]:
        """Sub-ranges of [lo, hi) no earlier `start` of this step has reduced."""
        gaps, at = [], lo
        for a, b in sorted(self.covered):
            if b <= at or a >= hi:
                continue
            if a > at:
                gaps.append((at, a))
            at = max(at, b)
        if at < hi:
            gaps.append((at, hi))
        return gaps

    def start(self, lo: int = 0, hi: Optional[int] = None) -> None:
        """Launch the all-reduce of the not-yet-reduced part of [lo, hi) (call as gradients become final)."""
        if world_size() == 1:
            return
        hi = self.g.numel() if hi is None else hi
        for a, b in self.uncovered(lo, hi):
            for c, d in bucket_ranges(b - a, self.bucket_elems):
                if self.comm is not None:
                    self.comm.all_reduce(self.g[a + c:a + d])
                elif self.rsag and d - c >= self.RSAG_MIN_ELEMS:
                    self._reduce_scatter_all_gather(self.g[a + c:a + d])
                else:
                    self.handles.append(dist.all_reduce(self.g[a + c:a + d], op=dist.ReduceOp.SUM, async_op=True))
            self.covered.append((a, b))

    def _reduce_scatter_all_gather(self, t: Tensor) -> None:
        """In-place sum of `t` over the ranks as reduce-scatter + all-gather (the tail that does not divide by the world size: all-reduce).
        Both collectives go to the process group in this order; on the device they are stream-ordered by the backend, on gloo the
        gather is issued once the scatter has completed."""
        w = world_size()
        n = t.numel() // w * w
        if n:
            shard = torch.empty(n // w, dtype=t.dtype, device=t.device)
            h = dist.reduce_scatter_tensor(shard, t[:n], op=dist.ReduceOp.SUM, async_op=True)
            if not t.is_cuda:
                h.wait()
            else:
                self.handles.append(h)
            self.handles.append(dist.all_gather_into_tensor(t[:n], shard, async_op=True))
            self._shards.append(shard)
        if n < t.numel():
            self.handles.append(dist.all_reduce(t[n:], op=dist.ReduceOp.SUM, async_op=True))

    def reset(self) -> None:
        """Drop the bookkeeping of an unfinished step (waits for its collectives first)."""
        for h in self.handles:
            h.wait()
        if self.comm is not None:
            self.comm.wait()
        self.handles.clear()
        self.covered.clear()
        self._shards.clear()

    def finish(self) -> None:
        w = world_size()
        if w == 1:
            return
        self.start()
        for h in self.handles:
            h.wait()
        if self.comm is not None:
            self.comm.wait()
        self.handles.clear()
        self.covered.clear()
        self._shards.clear()
        if self.g.is_cuda:
            from . import ops

            ops.scale_f32(self.g, 1.0 / w)
        else:
            self.g.mul_(1.0 / w)


def coalesced_mean(values: "List[Tensor]") -> "List[Tensor]":
    """Mean over ranks of several scalars with ONE all-reduce (the reference logs `train_loss` and each entry of `log_dict`
    with `sync_dist=True`, i.e. one tiny all-reduce per scalar per step, LT/_methods/method.py:131-144 -- SURVEY.md 8(f).1)."""
    w = world_size()
    if w == 1 or not values:
        return list(values)
    buf = torch.stack([v.detach().reshape(()).float() for v in values])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    buf /= w
    return list(buf.unbind(0))

"""Collective engines behind one small interface.

* ``TorchComm``   — ``torch.distributed`` process group: ``gloo`` for the CPU
  plumbing configuration, ``nccl`` (== RCCL on ROCm) on MI355X.
* ``NativeComm``  — our C++ communicator (``csrc/comm``): TCP key-value
  bootstrap + ``ncclCommInitRank`` + collectives issued directly on a dedicated
  high-priority HIP stream.

Both expose stream-ordered, host-non-blocking calls on the GPU: ``all_reduce``
returns a handle whose ``wait()`` makes the *current stream* wait (no host sync).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class TorchComm:
    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.supports_avg = self.backend == "nccl"

    def all_reduce_mean(self, flat: torch.Tensor, async_op: bool = True):
        if self.world_size == 1:
            return _Done()
        if self.supports_avg:
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op) \
                or _Done()
        flat.div_(self.world_size)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op) \
            or _Done()

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        if self.world_size == 1:
            return _Done()
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op) or _Done()

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if self.world_size == 1:
            return _Done()
        return dist.broadcast(t, src=src, group=self.group, async_op=async_op) or _Done()

    def all_gather(self, out: torch.Tensor, t: torch.Tensor):
        if self.world_size == 1:
            out.view_as(t.unsqueeze(0) if out.dim() == t.dim() + 1 else t).copy_(t)
            return
        dist.all_gather_into_tensor(out, t, group=self.group) if self.backend == "nccl" else \
            dist.all_gather(list(out.unbind(0)), t, group=self.group)

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)


def default_comm(kind: str = "auto", group=None):
    """``auto``: native RCCL communicator when built and the tensors live on a GPU
    process group; ``torch.distributed`` otherwise."""
    from . import process_group as pg

    npg = pg.native()
    if npg is not None:                       # native process group: its communicators, nothing to negotiate
        return npg.gpu_comm if npg.gpu_comm is not None else npg.host_comm
    import os

    if kind == "peer" or (kind in ("auto", "native") and os.environ.get("DPA_COMM") == "peer"):
        # peer-memory communicator over the torch process group's store (no torch collective is ever issued)
        from .peer_comm import PeerNativeComm

        store = dist.distributed_c10d._get_default_store()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        seq = [0]

        def exchange(payload: bytes):
            seq[0] += 1
            base = "dpa/peer/x%d" % seq[0]
            store.set("%s/%d" % (base, rank), payload)
            return [payload if r == rank else bytes(store.get("%s/%d" % (base, r))) for r in range(world)]

        return PeerNativeComm.build(rank, world, torch.cuda.current_device(), exchange)
    if kind in ("auto", "native"):
        try:
            from .native_comm import NativeComm

            nc = NativeComm.from_env_or_group(group)
            if nc is not None:
                return nc
        except Exception as exc:  # noqa: BLE001
            if kind == "native":
                raise
            import warnings

            warnings.warn("native RCCL communicator unavailable (%s); using torch.distributed" % exc)
    return TorchComm(group)


def reduce_tensor(t: torch.Tensor, world_size: Optional[int] = None) -> torch.Tensor:
    """Mean over ranks of a small metric tensor (reference ``reduce_tensor``,
    ``imagenet_ddp_apex.py:562-566``).  Callers pack loss/top1/top5 into ONE tensor."""
    from . import process_group as pg

    if not pg.is_initialized():
        return t.clone()
    rt = pg.all_reduce_sum(t.clone())
    rt /= (world_size or pg.get_world_size())
    return rt

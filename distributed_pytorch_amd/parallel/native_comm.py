"""NativeComm — the C++ RCCL communicator (``csrc/comm/comm.cpp``) behind the ``TorchComm`` interface.

The RCCL unique id is created by rank 0 in C++ (``ncclGetUniqueId``) and published through the
rendezvous store of the already-initialised ``torch.distributed`` process group (TCPStore for
``tcp://``, the env store for ``env://`` -- same rendezvous flags as the reference,
``nd_imagenet.py:130-131``); every rank then calls ``ncclCommInitRank`` itself.  Collectives run on
the communicator's own high-priority HIP stream; ``wait()`` handles order the *current* stream
after them with an event -- no host synchronisation.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import backend

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int64: 3, torch.uint8: 4}
_COUNTER = [0]


class _StreamWait:
    def __init__(self, comm):
        self.comm = comm

    def wait(self):
        self.comm.wait_comm(torch.cuda.current_stream().cuda_stream)
        return True


class NativeComm:
    backend = "rccl-native"
    supports_avg = True

    def __init__(self, raw, group=None):
        self.raw = raw
        self.group = group
        self.world_size = raw.world_size
        self.rank = raw.rank

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_env_or_group(cls, group=None) -> Optional["NativeComm"]:
        if os.environ.get("DPA_COMM", "auto") == "torch":
            return None
        if not torch.cuda.is_available() or not (dist.is_available() and dist.is_initialized()):
            return None
        if dist.get_backend(group) != "nccl":
            return None
        ext = backend.require()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        device = torch.cuda.current_device()
        _COUNTER[0] += 1
        # exchange the unique id through the process group itself (works for every rendezvous kind)
        holder = [ext.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(holder, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
        raw = ext.Communicator(holder[0], rank, world, device)
        comm = cls(raw, group)
        comm._self_test()
        return comm

    def _self_test(self) -> None:
        t = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        self.all_reduce_sum(t).wait()
        torch.cuda.current_stream().synchronize()
        want = self.world_size * (self.world_size + 1) / 2.0
        got = float(t[0].item())
        if abs(got - want) > 1e-3 or float(t[-1].item()) != got:
            raise RuntimeError("native RCCL self-test failed: got %r, want %r" % (got, want))

    # ------------------------------------------------------------------ collectives
    def _order_after_current(self) -> None:
        self.raw.comm_waits_for(torch.cuda.current_stream().cuda_stream)

    def all_reduce_mean(self, flat: torch.Tensor, async_op: bool = True):
        self._order_after_current()
        self.raw.all_reduce(flat.data_ptr(), flat.numel(), _DT[flat.dtype], True, 0)
        h = _StreamWait(self.raw)
        if not async_op:
            h.wait()
        return h

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = True):
        self._order_after_current()
        self.raw.all_reduce(t.data_ptr(), t.numel(), _DT[t.dtype], False, 0)
        h = _StreamWait(self.raw)
        if not async_op:
            h.wait()
        return h

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if not t.is_contiguous():
            raise ValueError("broadcast needs a contiguous tensor")
        self._order_after_current()
        self.raw.broadcast(t.data_ptr(), t.numel() * t.element_size(), src, 0)
        h = _StreamWait(self.raw)
        if not async_op:
            h.wait()
        return h

    def all_gather(self, out: torch.Tensor, t: torch.Tensor):
        self._order_after_current()
        self.raw.all_gather(t.data_ptr(), out.data_ptr(), t.numel(), _DT[t.dtype], 0)
        _StreamWait(self.raw).wait()

    def barrier(self):
        t = torch.zeros(1, dtype=torch.float32, device="cuda")
        self.all_reduce_sum(t).wait()
        torch.cuda.current_stream().synchronize()

    def make_reducer(self, flats, param_bucket):
        ext = backend.require()
        return ext.BucketReducer(self.raw, [f.data_ptr() for f in flats], [f.numel() for f in flats],
                                 [_DT[f.dtype] for f in flats], list(param_bucket))

"""NativeComm — the C++ RCCL communicator (``csrc/comm/comm.cpp``) behind the ``TorchComm`` interface.

The RCCL unique id is created by rank 0 in C++ (``ncclGetUniqueId``) and published through the
rendezvous STORE of the already-initialised ``torch.distributed`` process group (TCPStore for
``tcp://``, the env store for ``env://`` -- same rendezvous flags as the reference,
``nd_imagenet.py:130-131``); every rank then calls ``ncclCommInitRank`` itself.  Only the store is
used, never a torch collective, so torch's own (lazily created) RCCL communicator is not built and the
process holds ONE communicator.  Collectives run on the communicator's own HIP stream (normal priority,
see ``csrc/comm/comm.cpp``); ``wait()`` handles order the *current* stream after them with an event --
no host synchronisation.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import backend

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int64: 3, torch.uint8: 4}
_COUNTER = [0]
# Small fp32 all-reduces (SyncBN statistics: 2C+1 floats, packed metrics) skip RCCL: one kernel on the caller's
# stream reads every peer's slot over xGMI (csrc/comm/p2p.hip).  DPA_P2P=0 keeps everything on RCCL.
P2P_CAPACITY_FLOATS = 16384
_shared = [None]            # the process's NativeComm, for callers that only hold a tensor (process_group facade)


def shared() -> Optional["NativeComm"]:
    return _shared[0]


class _OnCurrentStream:
    def wait(self):
        return True


class _StreamWait:
    def __init__(self, comm):
        self.comm = comm

    def wait(self):
        self.comm.wait_comm(torch.cuda.current_stream().cuda_stream)
        return True


class NativeComm:
    backend = "rccl-native"
    supports_avg = True

    def __init__(self, raw, group=None, exchange=None):
        self.raw = raw
        self.group = group
        self.world_size = raw.world_size
        self.rank = raw.rank
        self._exchange = exchange       # collective: bytes of this rank -> list of every rank's bytes
        self._p2p = None                # PeerAllReduce, False (unavailable) or None (not set up yet)
        self.p2p_calls = 0
        _shared[0] = self
        backend.register_stream("comm")           # the communicator's own HIP stream (csrc/comm/comm.cpp)

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
        gen = _COUNTER[0]
        store = dist.distributed_c10d._get_default_store()
        if group is not None:
            store = dist.PrefixStore("dpa_group_%d" % hash(tuple(dist.get_process_group_ranks(group))), store)
        seq = [0]

        def exchange(payload: bytes):
            """Collective over the store: bytes of this rank -> list of every rank's bytes."""
            seq[0] += 1
            base = "dpa/native/%d/x%d" % (gen, seq[0])
            store.set("%s/%d" % (base, rank), payload)
            return [payload if r == rank else bytes(store.get("%s/%d" % (base, r))) for r in range(world)]

        # exchange the unique id through the process group's store (works for every rendezvous kind)
        uid_key = "dpa/native/%d/uid" % gen
        if rank == 0:
            store.set(uid_key, bytes(ext.rccl_unique_id()))
        uid = bytes(store.get(uid_key))
        # build + verify, then agree: either every rank uses the native communicator or none does (a rank that
        # failed alone would leave the others waiting inside their first collective)
        comm, err = None, ""
        try:
            comm = cls(ext.Communicator(uid, rank, world, device), group, exchange)
            comm._self_test()
        except Exception as exc:  # noqa: BLE001
            err = "rank %d: %r" % (rank, exc)
        errs = [e.decode() for e in exchange(err.encode()) if e]
        if errs:
            if comm is not None:
                _shared[0] = None
                try:
                    comm.raw.abort()
                except Exception:  # noqa: BLE001
                    pass
            raise RuntimeError("native RCCL communicator unavailable on at least one rank: " + "; ".join(errs))
        return comm

    # ------------------------------------------------------------------ small-message peer path
    def _agree(self, ok: bool) -> bool:
        """True only if every rank says ok (through RCCL, so all ranks take the same branch)."""
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device="cuda")
        self._order_after_current()
        self.raw.all_reduce(flag.data_ptr(), 1, 0, False, 0)
        _StreamWait(self.raw).wait()
        return float(flag.item()) == float(self.world_size)

    def _setup_p2p(self):
        """Collective.  Allocates this rank's fine-grained slot buffer, swaps hipIpc handles, maps the peers and
        checks three known sums (both slot parities) before the path is used."""
        if os.environ.get("DPA_P2P", "1") == "0" or self._exchange is None or not 2 <= self.world_size <= 16:
            return False
        ext = backend.require()
        peer, handle = None, b"-"
        try:
            peer = ext.PeerAllReduce(self.rank, self.world_size, torch.cuda.current_device(), P2P_CAPACITY_FLOATS,
                                     float(os.environ.get("DPA_P2P_TIMEOUT_S", "30")))
            handle = bytes(peer.handle())
        except Exception:  # noqa: BLE001
            peer = None
        handles = self._exchange(handle)
        ok = peer is not None and all(len(h) == 64 for h in handles)
        if ok:
            try:
                peer.open_peers([bytes(h) for h in handles])
            except Exception:  # noqa: BLE001
                ok = False
        if not self._agree(ok):
            return False
        base = torch.arange(1, 1025, dtype=torch.float32, device="cuda")
        want = base * (self.world_size * (self.world_size + 1) / 2.0)
        good = True
        for n in (1024, 3, 1024):
            t = (base * float(self.rank + 1))[:n].contiguous()
            peer.all_reduce(t.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().synchronize()
            good = good and peer.status() == 0 and bool(torch.equal(t, want[:n]))
        if not self._agree(good):
            return False
        return peer

    def peer_path(self):
        if self._p2p is None:
            self._p2p = self._setup_p2p()
        return self._p2p or None

    def check_health(self) -> None:
        """Raises when an earlier one-shot peer all-reduce timed out (its output was filled with NaN on the device
        and the ranks' slot protocol is out of step: the job has to stop).  Cheap -- one read of a host-mapped
        word -- and called at every optimizer step, barrier and checkpoint."""
        peer = self._p2p
        if peer and peer.status() != 0:
            raise RuntimeError("one-shot peer all-reduce %d timed out: a peer rank did not arrive (dead or hung); "
                               "its result was poisoned with NaN -- aborting this rank" % peer.status())

    def _small(self, t: torch.Tensor) -> bool:
        if t.dtype != torch.float32 or not t.is_cuda or t.numel() > P2P_CAPACITY_FLOATS or not t.is_contiguous() \
                or self._p2p is False:
            return False
        peer = self.peer_path()
        if peer is None:
            return False
        self.check_health()
        peer.all_reduce(t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        self.p2p_calls += 1
        return True

    def _self_test(self) -> None:
        t = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        self._rccl_sum(t).wait()
        torch.cuda.current_stream().synchronize()
        want = self.world_size * (self.world_size + 1) / 2.0
        got = float(t[0].item())
        if abs(got - want) > 1e-3 or float(t[-1].item()) != got:
            raise RuntimeError("native RCCL self-test failed: got %r, want %r" % (got, want))

    # ------------------------------------------------------------------ collectives
    def _order_after_current(self) -> None:
        self.raw.comm_waits_for(torch.cuda.current_stream().cuda_stream)

    def all_reduce_mean(self, flat: torch.Tensor, async_op: bool = True, wire: int = -1):
        self._order_after_current()
        self.raw.all_reduce(flat.data_ptr(), flat.numel(), _DT[flat.dtype], True, 0)
        h = _StreamWait(self.raw)
        if not async_op:
            h.wait()
        return h

    def _rccl_sum(self, t: torch.Tensor):
        self._order_after_current()
        self.raw.all_reduce(t.data_ptr(), t.numel(), _DT[t.dtype], False, 0)
        return _StreamWait(self.raw)

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = True):
        if self._small(t):
            return _OnCurrentStream()          # already ordered on the caller's stream
        h = self._rccl_sum(t)
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

    def all_reduce_max(self, t: torch.Tensor):
        self._order_after_current()
        self.raw.all_reduce_max(t.data_ptr(), t.numel(), _DT[t.dtype], 0)
        h = _StreamWait(self.raw)
        h.wait()
        return h

    def barrier(self):
        t = torch.zeros(1, dtype=torch.float32, device="cuda")
        self._rccl_sum(t).wait()
        torch.cuda.current_stream().synchronize()

    def make_reducer(self, flats, param_bucket, chunk_bytes: int = 128 << 20, timeout_s: float = 0.0,
                     exit_on_timeout: bool = True, grace_s: float = 10.0, wire_dtype: int = -1):
        """The C++ BucketReducer on this communicator's stream (``csrc/comm/reducer_core.cpp`` + StreamBackend).
        ``wire_dtype``: dtype code fp32 buckets travel in (1 bf16, 2 fp16, -1 as stored)."""
        ext = backend.require()
        return ext.BucketReducer(ext.StreamBackend(self.raw), [f.data_ptr() for f in flats],
                                 [f.numel() for f in flats], [_DT[f.dtype] for f in flats], list(param_bucket),
                                 int(chunk_bytes), float(timeout_s or 0.0), bool(exit_on_timeout), float(grace_s),
                                 int(wire_dtype))

"""PeerNativeComm — the peer-memory communicator (``csrc/comm/peer.hip``) behind the ``NativeComm`` interface.

Selected with ``DPA_COMM=peer`` (``bench.py --comm peer``, ``--dist-backend native``).  Every collective is built from
ordinary kernels that read the other ranks' hipIpc-mapped windows directly: a two-shot (reduce-scatter + all-gather)
all-reduce that keeps all 7 xGMI links of a GPU busy at once, byte-copy broadcast / all-gather, and a one-workgroup
barrier with a wall-clock deadline.  No RCCL communicator exists in the process.

Because hipIpc does not care whether a peer's window lives on another GPU or in another PROCESS ON THE SAME GPU, this
is also how more than one rank runs through the whole GPU data-parallel path (arena views written by kernels, events,
re-planning, SyncBN, hipGraph capture) on a machine with a single device: ``bench.py --gpus 2 --share-gpu``,
``tests/test_gpu_distributed.py``.

Reference parity: the gradient all-reduce behind ``torch.nn.parallel.DistributedDataParallel`` (imagenet_ddp.py:127)
and Apex DDP (imagenet_ddp_apex.py:183); SURVEY.md 2.5 (xGMI sizing), N-C1 / N-C2.
"""
from __future__ import annotations

import os

import torch

from .. import backend
from .native_comm import _DT, NativeComm, _StreamWait

_SUM, _MAX = 0, 1


class PeerNativeComm(NativeComm):
    backend = "peer-native"
    supports_avg = True

    # ------------------------------------------------------------------ construction
    @classmethod
    def build(cls, rank: int, world: int, device: int, exchange):
        """Collective.  ``exchange``: bytes of this rank -> list of every rank's bytes (the rendezvous store)."""
        ext = backend.require()
        window = int(float(os.environ.get("DPA_PEER_WINDOW_MB", "128")) * (1 << 20))
        timeout = float(os.environ.get("DPA_PEER_TIMEOUT_S", "120"))
        coarse = os.environ.get("DPA_PEER_WINDOW_COARSE", "0") == "1"
        raw, handle, err = None, b"-", ""
        try:
            raw = ext.PeerCommunicator(rank, world, device, window, timeout, coarse)
            handle = bytes(raw.handle())
        except Exception as exc:  # noqa: BLE001
            err = "rank %d: %r" % (rank, exc)
        handles = exchange(handle)
        if not err and world > 1:
            try:
                raw.open_peers([bytes(h) for h in handles])
            except Exception as exc:  # noqa: BLE001
                err = "rank %d: %r" % (rank, exc)
        # all-or-none: a rank that failed alone would leave the others inside their first barrier until the deadline
        errs = [e.decode() for e in exchange(err.encode()) if e]
        if errs:
            raise RuntimeError("peer-memory communicator could not be built: " + "; ".join(errs))
        comm = cls(raw, None, exchange)
        try:
            comm._self_test()
        except Exception as exc:  # noqa: BLE001
            err = "rank %d: %r" % (rank, exc)
        errs = [e.decode() for e in exchange(err.encode()) if e]
        if errs:
            raise RuntimeError("peer-memory communicator failed its self-test: " + "; ".join(errs))
        return comm

    def _self_test(self) -> None:
        """Three known results through every kernel of the path (vector body + ragged tail, both wire widths)."""
        w = self.world_size
        for n, wire in ((1024 + 37, -1), (5, -1), (4096, 1)):
            # k * (rank + 1) and k * W(W+1)/2 with k <= 8 need at most 8 significant bits up to W = 16: exact in bf16
            t = (torch.arange(n, dtype=torch.float32, device="cuda") % 8 + 1.0) * float(self.rank + 1)
            self._order_after_current()
            self.raw.all_reduce(t.data_ptr(), n, 0, _SUM, False, wire, 0)
            _StreamWait(self.raw).wait()
            torch.cuda.current_stream().synchronize()
            want = (torch.arange(n, dtype=torch.float32, device="cuda") % 8 + 1.0) * (w * (w + 1) / 2.0)
            if self.raw.status() != 0 or not torch.equal(t, want):
                raise RuntimeError("peer all-reduce self-test failed (n=%d, wire=%d)" % (n, wire))

    # ------------------------------------------------------------------ collectives (all on the communicator's stream)
    def _agree(self, ok: bool) -> bool:
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device="cuda")
        self._order_after_current()
        self.raw.all_reduce(flag.data_ptr(), 1, 0, _SUM, False, -1, 0)
        _StreamWait(self.raw).wait()
        return float(flag.item()) == float(self.world_size)

    def check_health(self) -> None:
        super().check_health()
        err = self.raw.async_error()
        if err:
            raise RuntimeError(err + " -- its result was poisoned with NaN; aborting this rank")

    def all_reduce_mean(self, flat: torch.Tensor, async_op: bool = True, wire: int = -1):
        self._order_after_current()
        self.raw.all_reduce(flat.data_ptr(), flat.numel(), _DT[flat.dtype], _SUM, True, wire, 0)
        h = _StreamWait(self.raw)
        if not async_op:
            h.wait()
        return h

    def _rccl_sum(self, t: torch.Tensor):       # name kept: NativeComm's callers
        self._order_after_current()
        self.raw.all_reduce(t.data_ptr(), t.numel(), _DT[t.dtype], _SUM, False, -1, 0)
        return _StreamWait(self.raw)

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
        if not (t.is_contiguous() and out.is_contiguous()):
            raise ValueError("all_gather needs contiguous tensors")
        self._order_after_current()
        self.raw.all_gather(t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), 0)
        _StreamWait(self.raw).wait()

    def all_reduce_max(self, t: torch.Tensor):
        self._order_after_current()
        self.raw.all_reduce(t.data_ptr(), t.numel(), _DT[t.dtype], _MAX, False, -1, 0)
        h = _StreamWait(self.raw)
        h.wait()
        return h

    def barrier(self):
        self._order_after_current()
        self.raw.barrier(0)
        _StreamWait(self.raw).wait()
        torch.cuda.current_stream().synchronize()
        self.check_health()

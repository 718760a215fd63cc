"""DistributedDataParallel — bucketed, backward-overlapped gradient all-reduce.

Capability parity with what the reference gets from
``torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu])``
(``imagenet_ddp.py:127``, ``nd_imagenet.py:154``) and from
``apex.parallel.DistributedDataParallel(model, delay_allreduce=True)``
(``imagenet_ddp_apex.py:183``), re-designed for one-process-per-MI355X:

* **Gradient arena**: every ``param.grad`` is a *view* into a flat per-bucket
  buffer, so a bucket is all-reduced in place (no flatten/unflatten copies) and
  ``zero_grad`` is one memset per bucket.  The arena also exists for world
  size 1 (single GPU) — only the communication is skipped.
* **Overlap**: per-parameter post-accumulate hooks count a bucket down; when it
  completes, the comm stream waits on a compute-stream event and the RCCL
  all-reduce (mean) is enqueued while backward continues.  An end-of-backward
  engine callback joins the comm stream back — no host synchronisation.
* **Bucket order** is re-planned from the observed ready order after the first
  backward (c10d's rebuild).
* ``delay_allreduce=True``: one flat bucket per dtype reduced after backward.
* Init: parameter-shape verification across ranks, then rank-0 broadcast of a
  flat parameter arena and the (flat) BatchNorm buffer arena (C-c); optional
  per-forward buffer broadcast (C-d, ``broadcast_buffers``).
* ``state_dict`` keys carry the ``module.`` prefix like the reference's files.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from .bucketing import MiB, BucketSpec, chunk_ranges, plan_buckets
from .comm import TorchComm, default_comm


class _Bucket:
    def __init__(self, spec: BucketSpec, params: List[nn.Parameter], device, dtype):
        self.spec = spec
        self.params = params
        self.flat = torch.zeros(spec.total, dtype=dtype, device=device)
        self.views = []
        for p, off, n in zip(params, spec.offsets, spec.numels):
            v = self.flat[off:off + n]
            if p.dim() == 4:
                k, c, r, s = p.shape
                v = v.view(k, r, s, c).permute(0, 3, 1, 2)      # logical OIHW over KRSC memory
            else:
                v = v.view(p.shape)
            self.views.append(v)
        self.pending = len(params)
        self.launched = False
        self.work = []


class GradArena:
    """Flat gradient buckets + (optional) reducer state machine."""

    def __init__(self, params: List[nn.Parameter], cap_bytes: int, first_cap_bytes: int,
                 order: Optional[List[int]] = None, single_bucket: bool = False):
        self.params = params
        n = len(params)
        order = list(range(n - 1, -1, -1)) if order is None else order
        dtypes = sorted({self.grad_dtype(p) for p in params}, key=str)
        keys = [dtypes.index(self.grad_dtype(p)) for p in params]
        if single_bucket:
            cap_bytes = first_cap_bytes = 1 << 62
        specs = plan_buckets([p.numel() for p in params],
                             [torch.empty((), dtype=self.grad_dtype(p)).element_size() for p in params],
                             keys, order, cap_bytes, first_cap_bytes)
        self.buckets: List[_Bucket] = []
        self.bucket_of: Dict[int, int] = {}
        for spec in specs:
            ps = [params[i] for i in spec.param_ids]
            b = _Bucket(spec, ps, ps[0].device, dtypes[spec.dtype_key])
            for pid in spec.param_ids:
                self.bucket_of[pid] = len(self.buckets)
            self.buckets.append(b)
        self.attach()

    @staticmethod
    def grad_dtype(p: torch.Tensor) -> torch.dtype:
        return p.dtype

    def attach(self) -> None:
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                p.grad = v

    def flat_buffers(self) -> List[torch.Tensor]:
        return [b.flat for b in self.buckets]

    def zero(self) -> None:
        for b in self.buckets:
            b.flat.zero_()


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, process_group=None,
                 bucket_cap_mb: float = 32.0, first_bucket_mb: float = 1.0,
                 delay_allreduce: bool = False, broadcast_buffers: bool = True,
                 comm: str = "auto", chunk_mb: float = 64.0, verify_shapes: bool = True):
        super().__init__()
        self.module = module
        self.delay_allreduce = delay_allreduce
        self.broadcast_buffers = broadcast_buffers
        self.cap_bytes = int(bucket_cap_mb * MiB)
        self.first_cap_bytes = int(first_bucket_mb * MiB)
        self.chunk_bytes = int(chunk_mb * MiB)
        self.require_backward_grad_sync = True
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.device = self.params[0].device
        self.on_gpu = self.device.type == "cuda"
        if dist.is_available() and dist.is_initialized():
            self.comm = default_comm(comm, process_group)
        else:
            self.comm = None
        self.world_size = self.comm.world_size if self.comm is not None else 1
        self.comm_stream = torch.cuda.Stream(self.device, priority=-1) if self.on_gpu else None

        if self.world_size > 1:
            if verify_shapes:
                self._verify_shapes()
            self._sync_module_states()
        self._buffer_arena()
        self.arena = GradArena(self.params, self.cap_bytes, self.first_cap_bytes,
                               single_bucket=delay_allreduce)
        self._ready_order: List[int] = []
        self._rebuilt = delay_allreduce or self.world_size == 1
        self._callback_queued = False
        self._index = {id(p): i for i, p in enumerate(self.params)}
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._make_hook(p))
        self.comm_bytes_per_step = sum(b.flat.numel() * b.flat.element_size()
                                       for b in self.arena.buckets)

    # ------------------------------------------------------------------ init-time sync
    def _verify_shapes(self) -> None:
        sig = torch.tensor([len(self.params)] + [p.numel() for p in self.params],
                           dtype=torch.int64, device=self.device)
        ref = sig.clone()
        self.comm.broadcast(ref, 0)
        if not torch.equal(sig.cpu(), ref.cpu()):
            raise RuntimeError("DDP: parameter shapes differ across ranks")

    def _sync_module_states(self) -> None:
        """Rank 0's parameters and buffers become everyone's (C-c), coalesced per dtype."""
        tensors = [p.data for p in self.module.parameters()] + \
                  [b.data for b in self.module.buffers()]
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dtype, group in by_dtype.items():
            flat = torch.cat([t.reshape(-1) if t.is_contiguous() else
                              self._dense_flat(t) for t in group])
            self.comm.broadcast(flat, 0)
            off = 0
            for t in group:
                n = t.numel()
                self._dense_assign(t, flat[off:off + n])
                off += n

    @staticmethod
    def _dense_flat(t: torch.Tensor) -> torch.Tensor:
        if t.dim() == 4:
            return t.permute(0, 2, 3, 1).reshape(-1)        # KRSC memory order
        return t.contiguous().reshape(-1)

    @staticmethod
    def _dense_assign(t: torch.Tensor, flat: torch.Tensor) -> None:
        if t.dim() == 4 and not t.is_contiguous():
            k, c, r, s = t.shape
            t.copy_(flat.view(k, r, s, c).permute(0, 3, 1, 2))
        else:
            t.copy_(flat.view(t.shape))

    def _buffer_arena(self) -> None:
        """Re-homes all float buffers (BN running stats) and all integer buffers
        into two flat tensors so the per-forward broadcast is two collectives."""
        self._buf_flat: List[torch.Tensor] = []
        groups: Dict[torch.dtype, list] = {}
        for mod in self.module.modules():
            for name, buf in mod._buffers.items():
                if buf is not None:
                    groups.setdefault(buf.dtype, []).append((mod, name, buf))
        for dtype, items in groups.items():
            total = sum(b.numel() for _, _, b in items)
            flat = torch.empty(total, dtype=dtype, device=items[0][2].device)
            off = 0
            for mod, name, buf in items:
                n = buf.numel()
                view = flat[off:off + n].view(buf.shape)
                view.copy_(buf)
                mod._buffers[name] = view
                off += n
            self._buf_flat.append(flat)

    # ------------------------------------------------------------------ forward
    def forward(self, *inputs, **kwargs):
        if self.world_size > 1 and self.broadcast_buffers and self.module.training \
                and self.require_backward_grad_sync:
            for flat in self._buf_flat:
                self.comm.broadcast(flat, 0)
        if not self._rebuilt and len(self._ready_order) == len(self.params):
            self._rebuild_buckets()
        return self.module(*inputs, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        old, self.require_backward_grad_sync = self.require_backward_grad_sync, False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    # ------------------------------------------------------------------ reducer
    def _rebuild_buckets(self) -> None:
        order = list(self._ready_order)
        old = self.arena
        self.arena = GradArena(self.params, self.cap_bytes, self.first_cap_bytes, order=order)
        # carry over nothing: grads were consumed by the optimizer step already
        del old
        self._rebuilt = True
        for opt in getattr(self, "_optimizers", []):
            opt.register_flat_grads(self.arena.flat_buffers())

    def bind_optimizer(self, optimizer) -> None:
        """Lets ``FusedSGD.zero_grad`` memset the flat buckets."""
        self._optimizers = getattr(self, "_optimizers", []) + [optimizer]
        if hasattr(optimizer, "register_flat_grads"):
            optimizer.register_flat_grads(self.arena.flat_buffers())

    def _make_hook(self, p: nn.Parameter):
        def hook(param):
            self._on_grad_ready(param)
        return hook

    def _on_grad_ready(self, p: nn.Parameter) -> None:
        pid = self._index[id(p)]
        bi = self.arena.bucket_of[pid]
        b = self.arena.buckets[bi]
        slot = b.spec.param_ids.index(pid)
        view = b.views[slot]
        if p.grad is not view:
            # someone set grads to None (torch's default zero_grad): adopt the fresh grad
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
        if not self.require_backward_grad_sync:
            return
        if not self._rebuilt:
            self._ready_order.append(pid)
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize_backward)
        b.pending -= 1
        if b.pending == 0 and not self.delay_allreduce:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        b.launched = True
        if self.world_size == 1:
            return
        nbytes = b.flat.numel() * b.flat.element_size()
        pieces = [b.flat]
        if nbytes > self.chunk_bytes:
            pieces = [b.flat[s:e] for s, e in
                      chunk_ranges(b.flat.numel(), b.flat.element_size(), self.chunk_bytes)]
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                for piece in pieces:
                    b.work.append(self.comm.all_reduce_mean(piece, async_op=True))
        else:
            for piece in pieces:
                b.work.append(self.comm.all_reduce_mean(piece, async_op=True))

    def _finalize_backward(self) -> None:
        self._callback_queued = False
        for b in self.arena.buckets:
            if not b.launched:                 # delay_allreduce, or parameters unused this step
                self._launch(b)
        for b in self.arena.buckets:
            for w in b.work:
                w.wait()                        # GPU: stream-level wait; gloo: host wait
            b.work = []
            b.pending = len(b.params)
            b.launched = False
        if self.on_gpu and self.world_size > 1:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if not self._rebuilt and len(self._ready_order) != len(self.params):
            self._ready_order = []             # partial graph: try again next iteration

    # ------------------------------------------------------------------ conveniences
    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero()

    def bucket_summary(self) -> List[dict]:
        return [{"bucket": i, "params": len(b.params),
                 "mbytes": b.flat.numel() * b.flat.element_size() / MiB}
                for i, b in enumerate(self.arena.buckets)]

"""hipGraph capture of the whole training step (``--hip-graph``).

One ResNet-50 step is ~700 kernel launches issued from Python; at batch 256 the GPU is the bottleneck,
but small models / small per-GPU batches are launch bound.  After a few eager warm-up steps (which build
every cache: weight copies, tap tables, the SGD tensor table, the gradient arena) the sequence

    zero_grad -> forward -> loss+metrics -> backward -> fused SGD step (-> loss-scaler update)

is captured ONCE into a HIP graph (``torch.cuda.CUDAGraph`` == hipGraph on ROCm; our kernels are launched
on torch's current stream, so stream capture records them like any ATen kernel) and replayed per step
with the batch copied into static input buffers.  Host-side values that change between steps (learning
rate, momentum, ...) live in a small device array that is refreshed BEFORE the replay, so schedules keep
working.  Any capture failure falls back to eager execution with a warning.

With DistributedDataParallel the gradient all-reduce is captured too: the reducer orders its communication stream
behind a "ready" event recorded on the capturing stream and the capturing stream behind the last bucket's "done"
event -- a capture fork / join -- so the collectives of the communication stream (the peer backend's pack /
barrier / reduce / gather kernels, whose barrier sequence numbers are counted on the device; RCCL's kernels, which
are capturable) become nodes of the same graph and still overlap with backward when it is replayed.  Capture starts
only after the buckets have been re-planned (step 2) and every rank agrees that its capture succeeded; otherwise all
ranks run eagerly.

The reference has no equivalent (``cudnn.benchmark`` autotuning is its only launch-side knob,
nd_imagenet.py:200, imagenet_ddp_apex.py:110) -- SURVEY.md section 7 "HIP streams and graphs".
"""
from __future__ import annotations

import warnings

import torch

from .. import amp
from ..ops import functional as F


class GraphedTrainStep:
    def __init__(self, model, optimizer, warmup: int = 3, enabled: bool = True):
        self.model, self.optimizer = model, optimizer
        self.warmup = max(2, warmup)
        self.enabled = enabled
        self.graph = None
        self.key = None
        self._seen_key, self._seen = None, 0
        self.replays = 0
        self.stream = None          # warm-up AND capture run on this side stream (see _on_side_stream)

    def _on_side_stream(self, images, target):
        """An eager warm-up step on the capture stream.  Autograd remembers the stream a node was created on; a
        parameter whose gradient arrives through AccumulateGrad (squeeze-excitation weights, layer scales) would
        otherwise make the engine synchronise the capturing stream with the stream of the warm-up steps -- a
        cross-stream wait that is illegal during capture (hipStreamEndCapture faulted on it)."""
        if self.stream is None:
            self.stream = torch.cuda.Stream(images.device)
        cur = torch.cuda.current_stream(images.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._step(images, target)
        cur.wait_stream(self.stream)
        return out

    # the eager step, also what gets captured
    def _step(self, images, target):
        output = self.model(images)
        loss, rec = F.loss_and_metrics(output, target)
        self.optimizer.zero_grad()
        with amp.scale_loss(loss, self.optimizer) as scaled:
            scaled.backward()
        self.optimizer.step()
        return loss, rec

    def _capture(self, images, target):
        base = getattr(images, "_dpa_s2d_base", None)
        if base is not None and images.stride() == base.stride() and images._base is base:
            # to_s2d() hands out a view of zero-padded rows (the stem reads the padding): keep that layout
            sbase = base.clone()
            off = images.storage_offset() - base.storage_offset()
            self.static_x = sbase.as_strided(images.shape, images.stride(), sbase.storage_offset() + off)
            self.static_x._dpa_s2d_base = sbase
        else:
            self.static_x = images.clone()
        self.static_t = target.clone()
        if getattr(images, "_dpa_nhwc", False):      # the prefetcher's "already normalised NHWC" tag
            self.static_x._dpa_nhwc = True
        if getattr(images, "_dpa_s2d", False):       # ... and its space-to-depth stem layout
            self.static_x._dpa_s2d = True
        sync = getattr(self.optimizer, "sync_hyper", None)
        if sync is not None:
            sync()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.stream is None:
            self.stream = torch.cuda.Stream(images.device)
        begin, end = getattr(self.model, "capture_begin", None), getattr(self.model, "capture_end", None)
        err = None
        if begin is not None:
            begin()
        try:
            # thread_local: threads that are not part of the step (a communication library's progress thread) may
            # keep calling into the runtime while the step is recorded
            mode = "thread_local" if self._world() > 1 else "global"
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                loss, rec = self._step(self.static_x, self.static_t)
        except Exception as exc:  # noqa: BLE001
            err = exc
        finally:
            if end is not None:
                end()
        if not self._all_ranks_ok(err is None, images.device):
            raise err if err is not None else RuntimeError("capture failed on another rank")
        self.graph, self.loss, self.rec = g, loss, rec
        self.key = self._key(images, target)

    def _world(self) -> int:
        return int(getattr(self.model, "world_size", 1)) if hasattr(self.model, "comm_summary") else 1

    def _all_ranks_ok(self, ok: bool, device) -> bool:
        """Collective: every rank replays a graph or none does (a mixed job would deadlock in its first all-reduce)."""
        if self._world() <= 1:
            return ok
        from ..parallel import process_group as pg

        flag = torch.tensor([0.0 if ok else 1.0], dtype=torch.float32, device=device)
        pg.all_reduce_sum(flag)
        return float(flag.item()) == 0.0

    @staticmethod
    def _key(images, target):
        return (tuple(images.shape), images.dtype, tuple(target.shape), target.dtype,
                bool(getattr(images, "_dpa_nhwc", False)), bool(getattr(images, "_dpa_s2d", False)))

    def __call__(self, images, target):
        if not self.enabled or not images.is_cuda:
            return self._step(images, target)
        key = self._key(images, target)
        if self.graph is None:
            if key != self._seen_key:
                self._seen_key, self._seen = key, 0
            self._seen += 1
            ready = getattr(self.model, "graph_ready", None)
            if self._seen <= self.warmup or (ready is not None and not ready()):
                return self._on_side_stream(images, target)
            try:
                self._capture(images, target)
            except Exception as exc:  # noqa: BLE001
                warnings.warn("hipGraph capture of the training step failed (%s: %s); running eagerly"
                              % (type(exc).__name__, exc))
                self.enabled = False
                self.graph = None
                torch.cuda.synchronize()
                return self._step(images, target)
        if key != self.key:                      # ragged last batch
            return self._step(images, target)
        self.static_x.copy_(images, non_blocking=True)
        self.static_t.copy_(target, non_blocking=True)
        sync = getattr(self.optimizer, "sync_hyper", None)
        if sync is not None:
            sync()
        self.graph.replay()
        done = getattr(self.optimizer, "after_graph_replay", None)
        if done is not None:
            done()
        self.replays += 1
        return self.loss, self.rec

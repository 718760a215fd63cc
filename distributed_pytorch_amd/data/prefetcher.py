"""StreamPrefetcher — async H2D + on-GPU normalise on a side HIP stream.

Same contract as the reference ``DataPrefetcher`` (``imagenet_ddp_apex.py:304-351``):
``next()`` returns the already-resident, already-normalised batch, makes the
compute stream wait on the copy stream, ``record_stream``s both tensors and
immediately starts the following batch.  Differences, by design:

* the uint8 batch is normalised by ONE fused HIP kernel (``u8_nhwc_to_norm``,
  N-K16) straight into the compute dtype and the 8-channel-padded NHWC layout the
  stem convolution consumes — not ``.float().sub_().div_()`` (3 passes);
* works on CPU too (no streams) so the loops are testable without a GPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..ops import functional as F


_copy_streams = {}


def copy_stream(device: torch.device) -> "torch.cuda.Stream":
    """The process's input copy stream for ``device`` (created on the first call, then shared).

    Call it right after ``torch.cuda.set_device`` -- BEFORE process groups, communicators or other streams exist.
    HIP multiplexes streams onto ``GPU_MAX_HW_QUEUES`` (4) hardware queues in creation order; a copy stream created
    after the communication streams lands on the compute stream's queue, and its 150 MB host-to-device copy then
    runs in order with the training step instead of beside it: 3 ms of idle GPU per step under DDP (measured with
    ``bench.py --force-ddp``: 10.75k -> 11.14k img/s, profiles/r1_notes.md)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _copy_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device)
        with torch.cuda.stream(s):
            torch.empty(1, device=device).zero_()     # touch it: the hardware queue is bound on first use
        _copy_streams[key] = s
        from .. import backend as _backend

        _backend.register_stream("copy")
    return s


class StreamPrefetcher:
    def __init__(self, loader, device: torch.device, dtype: torch.dtype = torch.float32,
                 pad_c: int = 0, normalize: bool = True, s2d: bool = False,
                 stream: Optional["torch.cuda.Stream"] = None):
        self.loader = iter(loader)
        self.s2d = s2d      # the model's stem takes the space-to-depth layout (model.stem_s2d): emit it directly
        self.device, self.dtype, self.pad_c, self.normalize = device, dtype, pad_c, normalize
        self.on_gpu = device.type == "cuda"
        self.stream: Optional[torch.cuda.Stream] = (stream or copy_stream(device)) if self.on_gpu else None
        self.next_input = None
        self.next_target = None
        self.preload()

    def _stage(self, images: torch.Tensor, target: torch.Tensor):
        images = images.to(self.device, non_blocking=True)
        target = target.to(self.device, non_blocking=True)
        if images.dtype == torch.uint8 and self.normalize:
            if self.s2d and F.s2d_ok(images):
                return F.to_s2d(images, self.dtype), target        # tagged _dpa_s2d / _dpa_nhwc
            layout = "nhwc" if images.shape[-1] in (1, 3) else "nchw"
            images = F.normalize_u8(images, self.dtype, self.pad_c, input_layout=layout)
            images._dpa_nhwc = True
        return images, target

    def preload(self) -> None:
        try:
            images, target = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = None
            return
        if self.on_gpu:
            with torch.cuda.stream(self.stream):
                self.next_input, self.next_target = self._stage(images, target)
        else:
            self.next_input, self.next_target = self._stage(images, target)

    def next(self):
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        images, target = self.next_input, self.next_target
        if self.on_gpu:
            if images is not None:
                images.record_stream(torch.cuda.current_stream(self.device))
            if target is not None:
                target.record_stream(torch.cuda.current_stream(self.device))
        self.preload()
        return images, target

    def __iter__(self):
        while True:
            images, target = self.next()
            if images is None:
                return
            yield images, target

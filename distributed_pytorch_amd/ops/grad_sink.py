"""Gradient sinks: where backward kernels put parameter gradients.

A ``GradSink`` ties a parameter to a persistent gradient buffer (a view into a flat DDP bucket /
single-GPU arena, same shape and memory layout as the parameter).  The HIP backward wrappers write
their result straight into ``sink.view`` -- overwrite on the first delivery of an iteration,
accumulate afterwards -- and return ``None`` to autograd for that input, so there is no
AccumulateGrad pass, no per-step memset and the optimizer's device table never changes.
``on_ready`` (the DDP reducer) is told as soon as the gradient is enqueued.

Replaces c10d's ``gradient_as_bucket_view`` + autograd hooks.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GradSink:
    __slots__ = ("param", "view", "index", "on_ready", "written", "zeroed", "arena")

    def __init__(self, param, view: torch.Tensor, index: int, on_ready: Optional[Callable] = None,
                 arena=None):
        self.param, self.view, self.index, self.on_ready = param, view, index, on_ready
        self.arena = arena
        self.written = False      # a gradient was delivered this iteration
        self.zeroed = True        # the view currently holds zeros (fresh / memset)

    def beta(self) -> float:
        """Accumulation factor for the next kernel writing into the view."""
        return 1.0 if self.written else 0.0

    def delivered(self) -> None:
        self.written = True
        self.zeroed = False
        p = self.param
        if p.grad is not self.view:
            p.grad = self.view
        if self.on_ready is not None:
            self.on_ready(self)


def sink_of(param) -> Optional[GradSink]:
    """The parameter's sink when kernels may write into it now (training backward on the GPU)."""
    s = getattr(param, "_dpa_sink", None)
    if s is None or not param.requires_grad or s.view.device != param.device \
            or s.view.dtype != torch.float32:
        return None
    return s

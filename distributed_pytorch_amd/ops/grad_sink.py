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
    __slots__ = ("param", "view", "index", "on_ready", "stamp", "zeroed", "arena", "uses")

    def __init__(self, param, view: torch.Tensor, index: int, on_ready: Optional[Callable] = None,
                 arena=None):
        self.param, self.view, self.index, self.on_ready = param, view, index, on_ready
        self.arena = arena
        self.stamp = -1           # arena epoch of the last delivery ("written this iteration")
        self.zeroed = True        # the view currently holds zeros (fresh / memset)
        self.uses = 0             # forward uses whose backward has not delivered yet (shared weights)

    @property
    def written(self) -> bool:
        return self.stamp == self.arena.epoch

    @written.setter
    def written(self, flag: bool) -> None:
        self.stamp = self.arena.epoch if flag else -1

    def beta(self) -> float:
        """Accumulation factor for the next kernel writing into the view."""
        return 1.0 if self.written else 0.0

    def delivered(self) -> None:
        """A kernel has just written (beta() = 0) or accumulated (1) this gradient into the view."""
        p = self.param
        g = p.grad
        if g is not None and g is not self.view:
            self.view.add_(g)     # a part that arrived through autograd earlier in this backward
        self.written = True
        self.zeroed = False
        if g is not self.view:
            p.grad = self.view
        self.uses -= 1
        if self.on_ready is not None and self.uses <= 0:   # the LAST use completes the gradient
            self.uses = 0
            self.on_ready(self)


def sink_of(param) -> Optional[GradSink]:
    """The parameter's sink when kernels may write into it now (training backward on the GPU)."""
    s = getattr(param, "_dpa_sink", None)
    if s is None or not param.requires_grad or s.view.device != param.device \
            or s.view.dtype != torch.float32:
        return None
    return s


def note_use(*params) -> None:
    """Forward-time bookkeeping: one more backward delivery is owed to each parameter's sink."""
    if not torch.is_grad_enabled():
        return
    for p in params:
        if p is not None:
            s = getattr(p, "_dpa_sink", None)
            if s is not None and p.requires_grad:
                s.uses += 1

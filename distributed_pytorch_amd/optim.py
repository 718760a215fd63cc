"""FusedSGD — SGD with momentum / weight decay as ONE multi-tensor HIP launch.

Semantics are those of ``torch.optim.SGD`` as used by the reference
(``nd_imagenet.py:174-176``; dampening 0, no nesterov by default):
``g += wd*p``; first step ``buf = g`` else ``buf = mu*buf + g``; ``p -= lr*buf``.
``state_dict()`` has torch's exact structure (``state[i]['momentum_buffer']``,
``param_groups[...]['params'] = [0..P-1]``) so optimizer state is interchangeable
with reference checkpoints.

GPU path (kernel ``sgd_momentum_multi_tensor``, N-K13/N-K14): one launch walks a
device-resident chunk table over all parameters; it reads the gradient (fp32 or
bf16/fp16) times ``1/loss_scale``, skips the whole update when the loss
scaler's ``found_inf`` flag is set, updates the fp32 master + momentum and writes
the low-precision compute copy used by the next forward.  ``lr`` / ``wd`` /
``momentum`` are read from a device buffer, so a per-iteration LR schedule
(``imagenet_ddp_apex.py:369``) costs no re-capture under hipGraph replay.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch.optim import Optimizer

from . import amp as _amp
from .parallel import process_group as _pg


class FusedSGD(Optimizer):
    fused_unscale = True

    def __init__(self, params, lr: float = 0.1, momentum: float = 0.0, dampening: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                        nesterov=nesterov, maximize=False, foreach=None, differentiable=False,
                        fused=None)
        super().__init__(params, defaults)
        self._scaler = None          # armed by amp.scale_loss for the next step
        self._amp_state = None
        self._plan = None            # device chunk table (GPU path)
        self._own_arena = None       # single-GPU gradient arena (a DDP wrapper brings its own)
        self._steps = 0

    # ------------------------------------------------------------------ amp hooks
    def set_amp(self, amp_state) -> None:
        self._amp_state = amp_state
        self._plan = None

    def arm_scaler(self, scaler) -> None:
        self._scaler = scaler

    def rebind(self) -> None:
        """Parameters were re-typed (amp O3): drop cached plans / state."""
        self._plan = None
        self.state.clear()

    # ------------------------------------------------------------------ gradient arenas
    def register_arena(self, arena) -> None:
        """Kept for API symmetry: arenas are discovered through the parameters' sinks."""
        self._plan = None

    def _arenas(self):
        """Gradient arenas behind this optimizer's parameters.  On the GPU every fp32 parameter
        gets one (DDP's buckets, or a private flat arena on a single GPU) so backward kernels write
        gradients in place and the device tensor table never changes."""
        params = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not params:
            return []
        missing = [p for p in params if getattr(p, "_dpa_sink", None) is None]
        if missing and all(p.is_cuda and p.dtype == torch.float32 for p in missing):
            from .parallel.distributed import GradArena

            self._own_arena = GradArena(missing)
            self._plan = None
        seen, out = set(), []
        for p in params:
            s = getattr(p, "_dpa_sink", None)
            if s is not None and s.arena is not None and id(s.arena) not in seen:
                seen.add(id(s.arena))
                out.append(s.arena)
        return out

    # ------------------------------------------------------------------ zero_grad
    def zero_grad(self, set_to_none: Optional[bool] = None) -> None:
        """With a gradient arena nothing is zeroed on the GPU: the first kernel that delivers a
        gradient overwrites its slot (N-K18); CPU arenas (gloo plumbing) get one memset per bucket."""
        arenas = self._arenas()
        if not arenas:
            super().zero_grad(set_to_none=True if set_to_none is None else set_to_none)
            return
        for arena in arenas:
            if arena.params[0].is_cuda and set_to_none is not False:
                arena.begin_step()
                for p in arena.params:
                    p.grad = None
            else:
                arena.zero()
                arena.attach()

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        scaler, self._scaler = self._scaler, None
        from .ops import hip_ops as _hip_ops

        _hip_ops.join_side_stream()
        if _pg.is_initialized():
            _pg.check_health()          # a timed-out SyncBN / metric all-reduce must not reach the weights
        for arena in self._arenas():
            arena.finish_step()
        any_cuda = any(p.is_cuda for g in self.param_groups for p in g["params"])
        if any_cuda:
            self._step_hip(scaler)
        else:
            self._step_reference(scaler)
        for arena in self._arenas():
            arena.end_step()
        if scaler is not None:
            scaler.update()
        self._steps += 1
        return loss

    def _momentum_buffer(self, p: torch.Tensor) -> torch.Tensor:
        st = self.state[p]
        buf = st.get("momentum_buffer")
        if buf is None:
            buf = torch.zeros_like(p, dtype=torch.float32 if p.dtype == torch.float32 else p.dtype,
                                   memory_format=torch.preserve_format)
            st["momentum_buffer"] = buf
        return buf

    def _step_reference(self, scaler) -> None:
        skip = scaler is not None and float(scaler.found_inf.item()) > 0
        if skip:
            return
        inv = None if scaler is None else scaler.inv_scale
        for group in self.param_groups:
            lr, mu, wd = group["lr"], group["momentum"], group["weight_decay"]
            damp, nesterov = group["dampening"], group["nesterov"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad.to(p.dtype)
                if inv is not None:
                    g = g * inv.to(g.dtype).squeeze(0)
                if wd != 0:
                    g = g.add(p, alpha=wd)
                if mu != 0:
                    first = "momentum_buffer" not in self.state[p] or \
                        self.state[p]["momentum_buffer"] is None
                    buf = self._momentum_buffer(p)
                    if first:
                        buf.copy_(g)
                    else:
                        buf.mul_(mu).add_(g, alpha=1 - damp)
                    g = g.add(buf, alpha=mu) if nesterov else buf
                p.add_(g.to(p.dtype), alpha=-lr)

    def _step_hip(self, scaler) -> None:
        from .ops import hip_ops

        if self._plan is None or not self._plan.matches(self):
            self._plan = hip_ops.SgdPlan(self)
        self._plan.launch(self, scaler)
        _amp.mark_copies_written(self._plan.params_with_copies)

    # ------------------------------------------------------------------ hipGraph support (utils/graph.py)
    def sync_hyper(self) -> None:
        if self._plan is not None:
            self._plan.sync_hyper(self)

    def after_graph_replay(self) -> None:
        """A replayed step ran the captured kernels only: advance the host-side bookkeeping."""
        from .ops import hip_ops

        self._steps += 1
        hip_ops.bump_generation()

    # ------------------------------------------------------------------ torch-compatible state
    def state_dict(self):
        sd = super().state_dict()
        if self._steps == 0:
            # torch.optim.SGD has empty state before its first step
            sd["state"] = {k: v for k, v in sd["state"].items()
                           if v.get("momentum_buffer") is not None and self._steps > 0}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._plan = None
        self._steps = max(self._steps, 1 if len(self.state) else 0)
        for p, st in self.state.items():
            buf = st.get("momentum_buffer")
            if buf is not None and p.dim() == 4 and not buf.is_contiguous(
                    memory_format=torch.channels_last):
                st["momentum_buffer"] = buf.contiguous(memory_format=torch.channels_last)

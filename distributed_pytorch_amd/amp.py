"""Native mixed precision: precision policy + dynamic loss scaling.

Replaces the NVIDIA Apex ``amp`` dependency of the reference
(``amp.initialize`` at ``imagenet_ddp_apex.py:169-172`` and ``amp.scale_loss`` at
``:380-381``) with a policy that is explicit instead of function-patching:

======  =====================================================================
level   meaning here (``dtype`` = ``--amp-dtype``, bf16 by default, fp16 optional)
======  =====================================================================
O0      pure fp32
O1      fp32 parameters; conv/linear run on low-precision *compute copies*
        that are re-cast lazily when a parameter changes; BN statistics fp32
O2      like O1, but the fused optimizer owns fp32 masters + momentum and
        writes the low-precision copies itself (one kernel, no re-cast pass)
O3      everything in ``dtype`` (BatchNorm too unless ``keep_batchnorm_fp32``)
======  =====================================================================

``loss_scale``: ``"dynamic"`` (default for fp16: start 2**16, halve + skip the
step on inf/nan, double after 2000 clean steps — Apex semantics) or a number
(default 1.0 for bf16, whose exponent range makes scaling unnecessary).  The
scaler state (scale, growth counter, found-inf flag) lives ON THE DEVICE; the
unscale + overflow check + conditional SGD update run without a host sync
(kernels ``multi_tensor_unscale_check`` / ``sgd_momentum_multi_tensor``).
"""
from __future__ import annotations

import contextlib
import weakref
from typing import Dict, Optional

import torch

_LP = {"bf16": torch.bfloat16, "fp16": torch.float16,
       torch.bfloat16: torch.bfloat16, torch.float16: torch.float16}


class _Entry:
    __slots__ = ("lp", "lp_t", "version", "generation", "ref")

    def __init__(self):
        self.lp = None          # low-precision copy, same logical shape / KRSC memory
        self.lp_t = None        # dgrad operand (conv: [C,R,S,K]; linear: [in,out]) built by hip_ops
        self.version = -1
        self.generation = -1
        self.ref = None


class AmpState:
    def __init__(self):
        self.reset()

    def reset(self):
        self.enabled = False
        self.opt_level = "O0"
        self.dtype = torch.float32
        self.keep_batchnorm_fp32 = True
        self.master_weights = False
        self.scaler: Optional["LossScaler"] = None
        self.entries: Dict[int, _Entry] = {}
        self.generation = 0     # bumped by fused optimizers that rewrite copies themselves


_state = AmpState()


def state() -> AmpState:
    return _state


def compute_dtype(default: torch.dtype = torch.float32) -> torch.dtype:
    return _state.dtype if _state.enabled else default


def entry_for(param: torch.Tensor) -> _Entry:
    e = _state.entries.get(id(param))
    if e is None or e.ref is None or e.ref() is not param:
        e = _Entry()
        e.ref = weakref.ref(param)
        _state.entries[id(param)] = e
    return e


def compute_copy(param: torch.Tensor, like: torch.Tensor) -> Optional[torch.Tensor]:
    """Low-precision compute copy of ``param`` for an op whose input is ``like``;
    ``None`` when the op should use ``param`` itself."""
    if param.dtype == like.dtype or not like.dtype.is_floating_point or param.is_cuda:
        return None       # GPU: ops/hip_ops.py owns the (cast / padded / transposed) compute copies
    e = entry_for(param)
    if e.lp is None or e.lp.dtype != like.dtype or e.lp.device != param.device \
            or e.version != param._version:
        with torch.no_grad():
            fmt = torch.channels_last if param.dim() == 4 else torch.contiguous_format
            src = param.detach()
            if e.lp is None or e.lp.dtype != like.dtype or e.lp.device != param.device:
                e.lp = torch.empty_like(src, dtype=like.dtype, memory_format=fmt)
            e.lp.copy_(src)
        e.lp_t = None
        e.version = param._version
    return e.lp


def mark_copies_written(params) -> None:
    """Called by FusedSGD after its kernel rewrote master AND low-precision copy:
    the cached copy is current, only derived operands (``lp_t``) are stale."""
    for p in params:
        e = _state.entries.get(id(p))
        if e is not None:
            e.version = p._version
            e.lp_t = None


class LossScaler:
    """Apex-compatible dynamic loss scaler whose state is three device scalars."""

    def __init__(self, device, init_scale: float = 2.0 ** 16, dynamic: bool = True,
                 growth_interval: int = 2000, growth_factor: float = 2.0,
                 backoff_factor: float = 0.5, max_scale: float = 2.0 ** 24, min_scale: float = 1.0):
        self.dynamic = dynamic
        self.growth_interval = growth_interval
        self.growth_factor, self.backoff_factor = growth_factor, backoff_factor
        self.max_scale, self.min_scale = max_scale, min_scale
        self.scale = torch.full((1,), float(init_scale), dtype=torch.float32, device=device)
        self.inv_scale = torch.full((1,), 1.0 / float(init_scale), dtype=torch.float32, device=device)
        self.good_steps = torch.zeros(1, dtype=torch.int32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=device)

    @property
    def is_identity(self) -> bool:
        return (not self.dynamic) and self._static_one

    _static_one = False

    def loss_scale(self) -> float:
        return float(self.scale.item())

    def check_overflow(self, grads) -> None:
        """Sets ``found_inf`` (device flag) if any grad has inf/nan.  No host sync."""
        grads = [g for g in grads if g is not None]
        if not grads:
            return
        if grads[0].is_cuda:
            from .ops import hip_ops

            hip_ops.multi_tensor_check_finite(grads, self.found_inf)
        else:
            bad = torch.zeros((), dtype=torch.float32)
            for g in grads:
                bad = torch.maximum(bad, (~torch.isfinite(g)).any().float())
            self.found_inf.copy_(torch.maximum(self.found_inf, bad.view(1)))

    def update(self) -> None:
        """Scale state machine, evaluated on the device (no ``.item()``)."""
        if not self.dynamic:
            self.found_inf.zero_()
            return
        if self.scale.is_cuda:
            from .ops import hip_ops

            hip_ops.scaler_update(self.scale, self.inv_scale, self.good_steps, self.found_inf,
                                  self.growth_interval, self.growth_factor, self.backoff_factor,
                                  self.max_scale, self.min_scale)
            return
        overflow = self.found_inf > 0
        good = torch.where(overflow, torch.zeros_like(self.good_steps), self.good_steps + 1)
        grow = good >= self.growth_interval
        new_scale = torch.where(overflow, self.scale * self.backoff_factor,
                                torch.where(grow, self.scale * self.growth_factor, self.scale))
        new_scale = new_scale.clamp(self.min_scale, self.max_scale)
        self.good_steps.copy_(torch.where(grow, torch.zeros_like(good), good))
        self.scale.copy_(new_scale)
        self.inv_scale.copy_(1.0 / new_scale)
        self.found_inf.zero_()

    def state_dict(self):
        return {"scale": float(self.scale.item()), "good_steps": int(self.good_steps.item()),
                "dynamic": self.dynamic}

    def load_state_dict(self, sd):
        self.scale.fill_(float(sd["scale"]))
        self.inv_scale.fill_(1.0 / float(sd["scale"]))
        self.good_steps.fill_(int(sd.get("good_steps", 0)))


def _parse_bool(v, default):
    if v is None:
        return default
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() in ("1", "true", "yes", "on")


def initialize(model, optimizer=None, opt_level: Optional[str] = "O1", keep_batchnorm_fp32=None,
               loss_scale=None, dtype="bf16", verbose: bool = False):
    """Configure precision for ``model``/``optimizer`` and return them
    (signature mirrors ``apex.amp.initialize``; strings accepted for the
    overrides exactly like the reference passes them from argparse)."""
    from .nn.modules import BatchNorm2d

    st = _state
    st.reset()
    level = (opt_level or "O0").upper()
    if level not in ("O0", "O1", "O2", "O3"):
        raise ValueError("unknown opt_level %r" % opt_level)
    lp = _LP[dtype]
    st.opt_level = level
    st.enabled = level != "O0"
    st.dtype = lp if st.enabled else torch.float32
    st.keep_batchnorm_fp32 = _parse_bool(keep_batchnorm_fp32, level != "O3")
    st.master_weights = level == "O2"
    device = next(model.parameters()).device

    if level == "O3":
        for mod in model.modules():
            if isinstance(mod, BatchNorm2d) and st.keep_batchnorm_fp32:
                continue
            for p in mod.parameters(recurse=False):
                p.data = p.data.to(lp)
        if optimizer is not None and hasattr(optimizer, "rebind"):
            optimizer.rebind()

    if loss_scale is None:
        loss_scale = "dynamic" if (st.enabled and lp == torch.float16 and level != "O3") else 1.0
    if isinstance(loss_scale, str) and loss_scale != "dynamic":
        loss_scale = float(loss_scale)
    if loss_scale == "dynamic":
        st.scaler = LossScaler(device, dynamic=True)
    else:
        st.scaler = LossScaler(device, init_scale=float(loss_scale), dynamic=False)
        st.scaler._static_one = float(loss_scale) == 1.0
    if optimizer is not None:
        optimizer._amp_scaler = st.scaler
        if hasattr(optimizer, "set_amp"):
            optimizer.set_amp(st)
    model._amp_dtype = st.dtype
    if verbose:
        print("amp: opt_level={} dtype={} keep_batchnorm_fp32={} loss_scale={}".format(
            level, st.dtype, st.keep_batchnorm_fp32, loss_scale))
    return (model, optimizer) if optimizer is not None else model


@contextlib.contextmanager
def scale_loss(loss: torch.Tensor, optimizer):
    """``with amp.scale_loss(loss, opt) as scaled: scaled.backward()``.

    On exit the gradients are checked for inf/nan and the optimizer is armed
    with ``1/scale``: a FusedSGD folds the unscale and the overflow-skip into
    its update kernel; any other optimizer gets its grads unscaled in place and
    its ``step`` skipped through a device-side predicate emulation.
    """
    scaler: Optional[LossScaler] = getattr(optimizer, "_amp_scaler", None) or _state.scaler
    if scaler is None or scaler.is_identity:
        yield loss
        return
    yield loss * scaler.scale.to(loss.dtype).squeeze(0)
    grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    scaler.check_overflow(grads)
    if getattr(optimizer, "fused_unscale", False):
        optimizer.arm_scaler(scaler)          # unscale + skip happen inside the SGD kernel
    else:
        inv = scaler.inv_scale
        for g in grads:
            g.mul_(inv.to(g.dtype).squeeze(0))
        _patch_step_once(optimizer, scaler)
    # the scale update itself runs after the step (needs found_inf intact for the skip)


def _patch_step_once(optimizer, scaler: LossScaler) -> None:
    """Generic optimizers: skip ``step`` on overflow (costs one host sync, like Apex)."""
    real_step = optimizer.step

    def guarded_step(*a, **kw):
        optimizer.step = real_step
        if float(scaler.found_inf.item()) > 0:
            scaler.update()
            return None
        out = real_step(*a, **kw)
        scaler.update()
        return out

    optimizer.step = guarded_step


def state_dict():
    return {"scaler": _state.scaler.state_dict() if _state.scaler is not None else None,
            "opt_level": _state.opt_level}


def load_state_dict(sd) -> None:
    if sd and sd.get("scaler") and _state.scaler is not None:
        _state.scaler.load_state_dict(sd["scaler"])

"""Checkpoint save / resume, load-compatible with the reference's files.

Writer: exactly the reference's dict — ``epoch`` (= epoch+1), ``arch``,
``state_dict``, ``best_acc1``, ``optimizer`` (``nd_imagenet.py:257-263``), fp32
OIHW tensors under torchvision names, ``module.`` prefix when the run is
distributed (the reference saves the DDP wrapper's state_dict,
``imagenet_ddp.py:219``).  Extra keys (``training_time`` on a desired-acc stop,
``imagenet_ddp.py:227-234``; ``amp`` scaler state; ``best_prec1`` alias for the
AMP script, ``imagenet_ddp_apex.py:275``) are additive.

Loader: accepts with/without ``module.`` / ``features.module.`` prefixes
(DataParallel on AlexNet/VGG ``features``, ``nd_imagenet.py:166``), ``best_acc1``
*or* ``best_prec1``, tensor or float, fp16/fp32 weights, optimizer state with or
without momentum buffers.  Files are read with ``weights_only=True``.

Fixes vs the reference: save is atomic (tmp + rename), only
one rank writes, and a barrier follows so no rank races ahead of the file.
"""
from __future__ import annotations

import os
import shutil
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch

CHECKPOINT_NAME = "checkpoint.pth.tar"
BEST_NAME = "model_best.pth.tar"


def strip_prefixes(state_dict: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[len("module."):]
        k = k.replace("features.module.", "features.")
        out[k] = v
    return out


def add_prefix(state_dict: Dict[str, torch.Tensor], prefix: str = "module.") -> "OrderedDict":
    return OrderedDict((prefix + k, v) for k, v in state_dict.items())


def model_state(model: torch.nn.Module, distributed: bool) -> "OrderedDict[str, torch.Tensor]":
    """fp32, OIHW-logical, CPU-portable state_dict with the reference's key names."""
    core = model.module if hasattr(model, "module") and isinstance(
        getattr(model, "module"), torch.nn.Module) else model
    sd = OrderedDict()
    for k, v in core.state_dict().items():
        t = v.detach()
        if t.is_floating_point() and t.dtype != torch.float32:
            t = t.float()
        sd[k] = t.contiguous().clone()      # plain OIHW in the file, whatever the compute layout
    return add_prefix(sd) if distributed else sd


def load_model_state(model: torch.nn.Module, state_dict: Dict[str, torch.Tensor],
                     strict: bool = True):
    core = model.module if hasattr(model, "module") and isinstance(
        getattr(model, "module"), torch.nn.Module) else model
    sd = strip_prefixes(state_dict)
    own = core.state_dict()
    fixed = OrderedDict()
    for k, v in sd.items():
        if k in own and v.is_floating_point() and v.dtype != own[k].dtype:
            v = v.to(own[k].dtype)
        fixed[k] = v
    return core.load_state_dict(fixed, strict=strict)


def _best_to_float(v) -> float:
    if isinstance(v, torch.Tensor):
        return float(v.detach().cpu().reshape(-1)[0].item()) if v.numel() else 0.0
    return float(v)


def save_checkpoint(state: Dict[str, Any], is_best: bool, out_dir: str = "",
                    filename: str = CHECKPOINT_NAME) -> str:
    """``torch.save`` + copy to ``model_best.pth.tar`` (reference
    ``nd_imagenet.py:356-359``; ``out_dir`` as in ``imagenet_ddp_apex.py:496-506``)."""
    from .parallel import process_group as _pg

    if _pg.is_initialized():
        _pg.check_health()      # never persist weights / running statistics that a failed collective poisoned
    path = os.path.join(out_dir, filename) if out_dir else filename
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(state, tmp)
    os.replace(tmp, path)
    if is_best:
        best = os.path.join(out_dir, BEST_NAME) if out_dir else BEST_NAME
        shutil.copyfile(path, best)
    return path


def build_state(epoch: int, arch: str, model, optimizer, best_acc1, distributed: bool,
                legacy_prec1_key: bool = False, extra: Optional[Dict[str, Any]] = None):
    best = best_acc1 if isinstance(best_acc1, torch.Tensor) else torch.tensor(float(best_acc1))
    state = {
        "epoch": epoch + 1,
        "arch": arch,
        "state_dict": model_state(model, distributed),
        "best_acc1": best.detach().cpu(),
        "optimizer": optimizer.state_dict(),
    }
    if legacy_prec1_key:
        state["best_prec1"] = _best_to_float(best)
    if extra:
        state.update(extra)
    return state


class CheckpointError(RuntimeError):
    pass


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as exc:  # truncated / corrupt / not a checkpoint
        raise CheckpointError("cannot read checkpoint %r: %s" % (path, exc)) from exc
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise CheckpointError("%r is not a training checkpoint (no 'state_dict')" % path)
    return ckpt


def resume(path: str, model, optimizer=None, map_location="cpu", verbose: bool = True):
    """-> (start_epoch, best_acc1 float).  Mirrors reference ``nd_imagenet.py:179-198``
    (and restores ``best_prec1`` properly, unlike ``imagenet_ddp_apex.py:191-196``)."""
    if not os.path.isfile(path):
        if verbose:
            print("=> no checkpoint found at '{}'".format(path))
        return None
    if verbose:
        print("=> loading checkpoint '{}'".format(path))
    ckpt = load_checkpoint(path, map_location)
    load_model_state(model, ckpt["state_dict"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        try:
            optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError) as exc:
            raise CheckpointError("optimizer state in %r does not match: %s" % (path, exc)) from exc
    best = ckpt.get("best_acc1", ckpt.get("best_prec1", 0.0))
    if ckpt.get("amp") is not None:
        from . import amp

        amp.load_state_dict(ckpt["amp"])
    if verbose:
        print("=> loaded checkpoint '{}' (epoch {})".format(path, ckpt.get("epoch", 0)))
    return int(ckpt.get("epoch", 0)), _best_to_float(best)

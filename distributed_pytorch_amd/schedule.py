"""Learning-rate schedules.

* ``step_lr``            — per-epoch ÷10 every 30 epochs (reference
  ``nd_imagenet.py:429-433``, ``imagenet_ddp.py:374-378``).
* ``warmup_step_lr``     — per-iteration schedule of the AMP script
  (``imagenet_ddp_apex.py:527-543``): ÷10 at epochs 30/60/80 and a 5-epoch
  linear warm-up.
* ``scale_lr_for_global_batch`` — linear scaling rule ``lr*global_batch/256``
  (``imagenet_ddp_apex.py:162``).

``apply_lr`` writes the value into every param group; optimizers that keep the
LR in a device scalar (``FusedSGD``) pick it up from there without a sync.
"""
from __future__ import annotations


def step_lr(base_lr: float, epoch: int, period: int = 30, gamma: float = 0.1) -> float:
    return base_lr * (gamma ** (epoch // period))


def warmup_step_lr(base_lr: float, epoch: int, step: int, len_epoch: int,
                   warmup_epochs: int = 5) -> float:
    """``step`` is 1-based inside the epoch, as in the reference loop."""
    factor = epoch // 30
    if epoch >= 80:
        factor += 1
    lr = base_lr * (0.1 ** factor)
    if epoch < warmup_epochs:
        lr = lr * float(1 + step + epoch * len_epoch) / (float(warmup_epochs) * len_epoch)
    return lr


def scale_lr_for_global_batch(lr: float, per_gpu_batch: int, world_size: int) -> float:
    return lr * float(per_gpu_batch * world_size) / 256.0


def apply_lr(optimizer, lr: float) -> float:
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def adjust_learning_rate(optimizer, epoch: int, args) -> float:
    """ND/DDP per-epoch schedule with the reference's call signature."""
    return apply_lr(optimizer, step_lr(args.lr, epoch))


def adjust_learning_rate_iter(optimizer, epoch: int, step: int, len_epoch: int, args) -> float:
    """AMP-script per-iteration schedule (``args.lr`` already batch-scaled)."""
    warm = getattr(args, "warmup_epochs", 5)
    return apply_lr(optimizer, warmup_step_lr(args.lr, epoch, step, len_epoch, warm))

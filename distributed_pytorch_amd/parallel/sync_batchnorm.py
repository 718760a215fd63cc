"""SyncBatchNorm: batch statistics over every rank of the process group.

Replaces ``apex.parallel.convert_syncbn_model`` (reference
``imagenet_ddp_apex.py:146-148``).  Forward: local per-channel moments ``[mean, M2, n]``
(M2 = sum of squared deviations from the LOCAL mean, never a raw sum of squares) -> gathered
from every rank (one small collective: an all-reduce of a zeroed ``[W, 2C+1]`` table in which
each rank fills its row) -> Chan / Welford parallel merge, as Apex ``welford_parallel`` does
-> normalise; ``running_var`` gets the unbiased global variance.  Backward: ONE all-reduce of
``[sum(dy), sum(dy*xhat)]`` (``2C`` floats), then the usual BN input gradient with
the global element count (collective inventory C-h / C-i).

This file holds the process-group plumbing and the CPU reference math; on the GPU
the local reductions and the normalise/backward passes are the HIP kernels
``syncbn_*`` (same kernels as plain BN, with the cross-rank reduction inserted
between their reduce and apply phases).
"""
from __future__ import annotations

from typing import Optional

import torch
from . import process_group as pg
from torch import nn

from ..nn.modules import BatchNorm2d, SyncBatchNorm


def _group(group):
    return None if group in (None, "world") else group


def _world(group) -> int:
    if not pg.is_initialized():
        return 1
    return pg.get_world_size(_group(group))


def all_reduce_sum_(t: torch.Tensor, group) -> torch.Tensor:
    if _world(group) > 1:
        pg.all_reduce_sum(t, group=_group(group))
    return t


def chan_merge(table: torch.Tensor, c: int):
    """[W, 2C+1] rows (mean_r, M2_r, n_r) -> (mean, biased var, n):  n = sum n_r,  mean = sum n_r mean_r / n,
    M2 = sum M2_r + sum n_r (mean_r - mean)^2  (Chan et al.; deviations from the MERGED mean, in double)."""
    t = table.double()
    n_r = t[:, 2 * c:2 * c + 1]
    count = n_r.sum()
    mean = (n_r * t[:, :c]).sum(0) / count
    m2 = t[:, c:2 * c].sum(0) + (n_r * (t[:, :c] - mean).square()).sum(0)
    return mean.float(), (m2 / count).float(), count.float()


class _SyncBNFunction(torch.autograd.Function):
    """Reference (pure torch) SyncBN over NHWC input; used for CPU tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, group):
        c = x.shape[-1]
        xf = x.reshape(-1, c).float()
        world = _world(group)
        rank = pg.get_rank(_group(group)) if world > 1 else 0
        lmean = xf.mean(0)
        table = torch.zeros(world, 2 * c + 1, dtype=torch.float32, device=x.device)
        table[rank, :c] = lmean
        table[rank, c:2 * c] = (xf - lmean).square().sum(0)
        table[rank, 2 * c] = float(xf.shape[0])
        all_reduce_sum_(table, group)
        mean, var, count = chan_merge(table, c)
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var * (count / (count - 1).clamp_min(1.0))
                running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
                if nbt is not None:
                    nbt.add_(1)
        xhat = (xf - mean) * invstd
        y = (xhat * weight.float() + bias.float()).to(x.dtype).reshape(x.shape)
        ctx.save_for_backward(xhat, weight, invstd, count)
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, invstd, count = ctx.saved_tensors
        c = dy.shape[-1]
        g = dy.reshape(-1, c).float()
        sums = torch.cat([g.sum(0), (g * xhat).sum(0)])
        dbias, dweight = sums[:c].clone(), sums[c:].clone()   # local: DDP averages params' grads
        all_reduce_sum_(sums, ctx.group)
        mean_dy, mean_dyx = sums[:c] / count, sums[c:] / count
        dx = (g - mean_dy - xhat * mean_dyx) * (weight.float() * invstd)
        return (dx.to(dy.dtype).reshape(dy.shape), dweight.to(weight.dtype),
                dbias.to(weight.dtype), None, None, None, None, None, None)


def sync_batch_norm_reference(x, weight, bias, running_mean, running_var, nbt, momentum, eps, group):
    return _SyncBNFunction.apply(x, weight, bias, running_mean, running_var, nbt, momentum, eps,
                                 group)


def convert_syncbn_model(module: nn.Module, process_group="world") -> nn.Module:
    """Recursively swaps every ``BatchNorm2d`` for ``SyncBatchNorm`` keeping
    parameters, buffers and names (Apex ``convert_syncbn_model`` contract)."""
    if isinstance(module, BatchNorm2d) and not isinstance(module, SyncBatchNorm):
        new = SyncBatchNorm(module.num_features, module.eps, module.momentum, process_group)
        new.weight, new.bias = module.weight, module.bias
        new._buffers.update(module._buffers)
        new.training = module.training
        return new
    for name, child in list(module.named_children()):
        module._modules[name] = convert_syncbn_model(child, process_group)
    return module

"""NHWC neural-network modules with torchvision-compatible parameter names.

Parameters keep torch's *logical* shapes (conv ``[K, C, R, S]``, linear
``[out, in]``) so ``state_dict()`` is interchangeable with checkpoints written by
the reference scripts, but conv weights use the
``channels_last`` memory format: their bytes are already the KRSC layout the HIP
implicit-GEMM kernels consume, so no per-step layout conversion exists.

Initialisation follows torchvision (kaiming-normal fan_out for ResNet convs,
default ``nn.Conv2d``/``nn.Linear`` init elsewhere) so that random-init
behaviour matches ``models.__dict__[arch]()`` in reference ``nd_imagenet.py:138``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from ..ops import functional as F
from .. import amp as _amp


class Conv2d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0,
                 bias: bool = True, groups: int = 1):
        super().__init__()
        k = F._pair(kernel_size)
        if in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride, self.padding = k, F._pair(stride), F._pair(padding)
        w = torch.empty(out_channels, in_channels // groups, k[0], k[1]).contiguous(
            memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.needs_input_grad = True  # the stem sets this False: no dgrad to the image
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels // self.groups * self.kernel_size[0] * self.kernel_size[1]
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: torch.Tensor, relu: bool = False, want_stats: bool = False, skip: bool = False):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, relu=relu,
                        w_lp=_amp.compute_copy(self.weight, x), want_stats=want_stats,
                        needs_input_grad=self.needs_input_grad, skip=skip, groups=self.groups)

    def extra_repr(self) -> str:
        return "{}, {}, kernel_size={}, stride={}, padding={}, bias={}{}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
            self.bias is not None, ", groups=%d" % self.groups if self.groups > 1 else "")


class BatchNorm2d(nn.Module):
    """BatchNorm over the channel (last) axis with optional fused residual-add and ReLU."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.process_group = None  # set by convert_syncbn_model

    def forward(self, x: torch.Tensor, relu: bool = False, residual: Optional[torch.Tensor] = None,
                stats=None, defer: bool = False) -> torch.Tensor:
        return F.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                            self.num_batches_tracked, self.training, self.momentum, self.eps,
                            relu=relu, residual=residual, stats=stats, group=self.process_group, defer=defer)

    def extra_repr(self) -> str:
        return "{}, eps={}, momentum={}{}".format(self.num_features, self.eps, self.momentum,
                                                  ", sync" if self.process_group is not None else "")


class SyncBatchNorm(BatchNorm2d):
    """BatchNorm2d whose batch statistics span every rank of ``process_group``
    (Apex ``convert_syncbn_model`` parity, reference ``imagenet_ddp_apex.py:146-148``)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1,
                 process_group="world"):
        super().__init__(num_features, eps, momentum)
        self.process_group = process_group


class ReLU(nn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        return F.relu(x)


class ReLU6(nn.Module):
    """Placeholder that keeps torchvision's module indices: the zoo applies ReLU6 inside the fused BN pass
    (``conv_bn_act(..., relu=6)``).  Called on its own it is a plain clamp."""

    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        return torch.clamp(x, 0.0, 6.0)


class _Act(nn.Module):
    kind = ""

    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        return F.activation(x, self.kind)


class Sigmoid(_Act):
    kind = "sigmoid"


class SiLU(_Act):
    kind = "silu"


class Hardsigmoid(_Act):
    kind = "hardsigmoid"


class Hardswish(_Act):
    kind = "hardswish"


class GELU(_Act):
    kind = "gelu"


class LayerNorm(nn.Module):
    """LayerNorm over the channel (last) axis of NHWC activations; stands for both ``nn.LayerNorm(C)`` and
    torchvision's ``LayerNorm2d(C)`` (same parameters: ``weight``, ``bias`` of shape ``[C]``)."""

    def __init__(self, normalized_shape: int, eps: float = 1e-5):
        super().__init__()
        self.normalized_shape, self.eps = (int(normalized_shape),), eps
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))

    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.eps)

    def extra_repr(self) -> str:
        return "{}, eps={}".format(self.normalized_shape, self.eps)


LayerNorm2d = LayerNorm


class StochasticDepth(nn.Module):
    """torchvision ``StochasticDepth(p, mode="row")`` (EfficientNet)."""

    def __init__(self, p: float, mode: str = "row"):
        super().__init__()
        if mode != "row":
            raise ValueError("only the per-sample ('row') mode is used by the zoo")
        self.p, self.mode = p, mode

    def forward(self, x):
        return F.stochastic_depth(x, self.p, self.training)


class SqueezeExcitation(nn.Module):
    """torchvision ``SqueezeExcitation``: ``x * gate(fc2(act(fc1(mean_hw(x)))))`` with 1x1-conv shaped parameters
    (``fc1.weight [S, C, 1, 1]`` ...) so checkpoints interchange.

    On the GPU the whole block is one autograd node (``hip_ops._SEBlockFn``): the activation-sized work -- the H*W
    mean, the gate multiply, the backward reduction of ``dy * x`` and ``dx = dy * s + d(mean) / HW`` in one pass -- runs
    in HIP kernels, and the excitation MLP (two ``[N, C] x [C, S]`` fp32 products, S = 4 .. a few hundred) is three
    launches of ``csrc/kernels/se_mlp.hip`` (forward, per-sample backward, parameter gradients)."""

    def __init__(self, input_channels: int, squeeze_channels: int, activation: str = "relu",
                 scale_activation: str = "sigmoid"):
        super().__init__()
        self.avgpool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(input_channels, squeeze_channels, 1)
        self.fc2 = Conv2d(squeeze_channels, input_channels, 1)
        self.activation = ReLU() if activation == "relu" else SiLU()
        self.scale_activation = Sigmoid() if scale_activation == "sigmoid" else Hardsigmoid()
        self._act, self._gate = activation, scale_activation

    def gate(self, x: torch.Tensor) -> torch.Tensor:
        n, c = x.shape[0], x.shape[-1]
        p = F.adaptive_avg_pool2d(x, (1, 1)).reshape(n, c).float()
        w1, w2 = self.fc1.weight.flatten(1).float(), self.fc2.weight.flatten(1).float()
        h = torch.addmm(self.fc1.bias.float(), p, w1.t())
        h = torch.relu(h) if self._act == "relu" else torch.nn.functional.silu(h)
        s = torch.addmm(self.fc2.bias.float(), h, w2.t())
        return torch.sigmoid(s) if self._gate == "sigmoid" else torch.nn.functional.hardsigmoid(s)

    def forward(self, x):
        if F._on_gpu(x) and F.se_fused():
            return F.se_block(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self._act, self._gate)
        return F.se_scale(x, self.gate(x))


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode: bool = False):
        super().__init__()
        self.kernel_size, self.stride, self.padding, self.ceil_mode = kernel_size, stride, padding, ceil_mode

    def forward(self, x):
        return F.max_pool2d(x, self.kernel_size, self.stride, self.padding, self.ceil_mode)


class AvgPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        return F.avg_pool2d(x, self.kernel_size, self.stride, self.padding)


class AdaptiveAvgPool2d(nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return F.adaptive_avg_pool2d(x, self.output_size)


class Dropout(nn.Module):
    def __init__(self, p: float = 0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F.dropout(x, self.p, self.training)


class Linear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, relu: bool = False, out_dtype=None):
        return F.linear(x, self.weight, self.bias, relu=relu,
                        w_lp=_amp.compute_copy(self.weight, x), out_dtype=out_dtype)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}".format(
            self.in_features, self.out_features, self.bias is not None)


class CrossEntropyLoss(nn.Module):
    """Mean softmax cross-entropy (reference ``nn.CrossEntropyLoss`` at ``nd_imagenet.py:172``)."""

    def forward(self, logits, target):
        return F.cross_entropy(logits, target)


class Flatten(nn.Module):
    """Flattens NHWC activations in **NCHW order** so Linear weights stay
    checkpoint-compatible with torchvision's ``torch.flatten(x, 1)``."""

    def forward(self, x):
        if x.dim() == 4:
            if x.shape[1] == 1 and x.shape[2] == 1:
                return x.reshape(x.shape[0], -1)
            return x.permute(0, 3, 1, 2).reshape(x.shape[0], -1)
        return x.reshape(x.shape[0], -1)


def conv_bn_relu_maxpool(x, conv: Conv2d, bn: BatchNorm2d, pool: "MaxPool2d"):
    """Stem: conv -> BN -> ReLU -> max-pool with the BN/ReLU/pool part as one fused pass (``F.bn_relu_max_pool2d``)."""
    fuse = F._on_gpu(x) and bn.training and bn.process_group is None
    out = conv(x, want_stats=fuse)
    stats = None
    if fuse:
        out, stats = out
    return F.bn_relu_max_pool2d(out, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                bn.training, bn.momentum, bn.eps, pool.kernel_size, pool.stride, pool.padding,
                                pool.ceil_mode, stats=stats, group=bn.process_group)


def conv_bn_cat_relu(a, conv3: Conv2d, bn3: BatchNorm2d, x, convd: Conv2d, bnd: BatchNorm2d):
    """Tail of a residual block with a downsample branch, ``relu(bn3(conv3(a)) + bnd(convd(x)))``, as ONE two-operand
    GEMM on the GPU in training mode (``hip_ops._ConvBnCatReluFn``); None when the fused path does not apply -- the
    caller then runs the two ``conv_bn_act`` calls."""
    if not (F._on_gpu(a) and bn3.training and bnd.training and bn3.process_group is None and bnd.process_group is None):
        return None
    for c in (conv3, convd):
        if c.groups != 1 or c.bias is not None or not c.needs_input_grad or tuple(c.kernel_size) != (1, 1):
            return None
    return F._hip().conv_bn_cat_relu(a, conv3.weight, conv3.stride, conv3.padding, bn3.weight, bn3.bias, bn3.running_mean,
                                     bn3.running_var, bn3.num_batches_tracked, bn3.momentum, bn3.eps,
                                     x, convd.weight, convd.stride, convd.padding, bnd.weight, bnd.bias,
                                     bnd.running_mean, bnd.running_var, bnd.num_batches_tracked, bnd.momentum, bnd.eps)


def conv_bn_act(x, conv: Conv2d, bn: BatchNorm2d, relu=True, residual=None, skip: bool = False, defer: bool = False,
                act: Optional[str] = None):
    """conv -> BN(+residual)(+ReLU; ``relu=6``: ReLU6).  ``act`` ("silu", "hardswish", ...): a smooth activation
    instead of ReLU, fused into the BN pass like ReLU is.  ``defer=True`` (downsample branches, ``relu=False``): the
    result may only be used as the ``residual`` of the block's last ``conv_bn_act`` (see ``F.batch_norm``).  On the GPU the conv epilogue emits the
    per-channel sum/sumsq partials that BN consumes, so the conv output is not
    re-read for statistics.  ``skip=True`` -> ``(out, x_alias)`` (see ``F.conv2d``)."""
    fuse = F._on_gpu(x) and bn.training and bn.process_group is None
    if fuse and residual is not None and relu is True and act is None and not skip and not defer \
            and conv.groups == 1 and conv.bias is None and conv.needs_input_grad:
        # bottleneck tail: statistics from the Gram matrix of x, BN + residual + ReLU in the convolution's epilogue
        y = F._hip().conv_bn_res_relu(x, conv.weight, conv.stride, conv.padding, bn.weight, bn.bias, bn.running_mean,
                                      bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps, residual)
        if y is not None:
            return y
    out = conv(x, want_stats=fuse, skip=skip)
    alias = None
    if skip:
        out, alias = out
    if act is not None:
        relu = act               # applied inside the BN pass (``F.batch_norm(relu="silu")``)
    if fuse:
        y, stats = out
        res = bn(y, relu=relu, residual=residual, stats=stats, defer=defer)
    else:
        res = bn(out, relu=relu, residual=residual, defer=defer)
    return (res, alias) if skip else res

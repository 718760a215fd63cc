"""RegNetX 400MF ... 32GF and RegNetY 400MF ... 128GF in NHWC (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Stage widths / depths come from the RegNet design-space quantisation (linear width ramp ``w_0 + w_a * i`` snapped to
powers of ``w_m``, multiples of 8, then made compatible with the group width) -- the same closed form torchvision
evaluates, so module names (``stem.0.weight``, ``trunk_output.block2.block2-1.f.b.0.weight``, ``fc.weight``) and
parameter counts equal its checkpoints: x_400mf 5,495,976; x_800mf 7,259,656; x_1_6gf 9,190,136; x_3_2gf 15,296,552;
x_8gf 39,572,648; x_16gf 54,278,536; x_32gf 107,811,560.

The 3x3 of every block is a grouped convolution (group width 16 ... 168 channels): one launch of the implicit-GEMM
kernel with the group on ``grid.y``.

The Y variants add a squeeze-excitation gate between the grouped 3x3 and the last 1x1 (``f.se.fc1.weight`` ...):
y_400mf 4,344,144; y_800mf 6,432,512; y_1_6gf 11,202,430; y_3_2gf 19,436,338; y_8gf 39,381,472; y_16gf 83,590,140;
y_32gf 145,046,770; y_128gf 644,812,894 parameters.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch import nn

from ..nn.modules import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Linear, ReLU, SqueezeExcitation, conv_bn_act
from .alexnet_vgg import _ImageNetCNN


def _make_divisible(v: float, divisor: int) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


def block_params(depth: int, w_0: int, w_a: float, w_m: float, group_width: int, bottleneck_multiplier: float = 1.0,
                 se_ratio=None):
    """-> per stage: (width_out, stride, depth, group_width, bottleneck_multiplier[, se_ratio])."""
    quant = 8
    if w_a < 0 or w_0 <= 0 or w_m <= 1 or w_0 % 8 != 0:
        raise ValueError("Invalid RegNet settings")
    widths_cont = torch.arange(depth) * w_a + w_0
    capacity = torch.round(torch.log(widths_cont / w_0) / math.log(w_m))
    block_widths = (torch.round(torch.divide(w_0 * torch.pow(w_m, capacity), quant)) * quant).int().tolist()
    num_stages = len(set(block_widths))
    split = [w != wp or r != rp for w, wp, r, rp in
             zip(block_widths + [0], [0] + block_widths, block_widths + [0], [0] + block_widths)]
    stage_widths = [w for w, t in zip(block_widths, split[:-1]) if t]
    edges = [d for d, t in enumerate(split) if t]
    stage_depths = [b - a for a, b in zip(edges[:-1], edges[1:])]
    mult = [bottleneck_multiplier] * num_stages
    gws = [group_width] * num_stages
    # make the bottleneck widths divisible by the group width
    bot = [int(w * b) for w, b in zip(stage_widths, mult)]
    gws = [min(g, wb) for g, wb in zip(gws, bot)]
    bot = [_make_divisible(wb, g) for wb, g in zip(bot, gws)]
    stage_widths = [int(wb / b) for wb, b in zip(bot, mult)]
    out = list(zip(stage_widths, [2] * num_stages, stage_depths, gws, mult))
    return out if se_ratio is None else [t + (se_ratio,) for t in out]


class _ConvBN(nn.Sequential):
    """torchvision's Conv2dNormActivation: index 0 conv, 1 norm (, 2 activation)."""

    def __init__(self, cin, cout, kernel_size, stride=1, groups=1, act=True):
        layers = [Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=False,
                         groups=groups), BatchNorm2d(cout)]
        if act:
            layers.append(ReLU(inplace=True))
        super().__init__(*layers)
        self.act = act

    def forward(self, x, residual=None, relu=None, defer=False):
        return conv_bn_act(x, self[0], self[1], relu=self.act if relu is None else relu, residual=residual,
                           defer=defer)


class BottleneckTransform(nn.Sequential):
    def __init__(self, width_in, width_out, stride, group_width, bottleneck_multiplier, se_ratio=None):
        w_b = int(round(width_out * bottleneck_multiplier))
        layers = OrderedDict()
        layers["a"] = _ConvBN(width_in, w_b, 1)
        layers["b"] = _ConvBN(w_b, w_b, 3, stride=stride, groups=w_b // group_width)
        if se_ratio:
            layers["se"] = SqueezeExcitation(w_b, int(round(se_ratio * width_in)), activation="relu")
        layers["c"] = _ConvBN(w_b, width_out, 1, act=False)
        super().__init__(layers)


class ResBottleneckBlock(nn.Module):
    def __init__(self, width_in, width_out, stride, group_width=1, bottleneck_multiplier=1.0, se_ratio=None):
        super().__init__()
        self.proj = None
        if width_in != width_out or stride != 1:
            self.proj = _ConvBN(width_in, width_out, 1, stride=stride, act=False)
        self.f = BottleneckTransform(width_in, width_out, stride, group_width, bottleneck_multiplier, se_ratio)
        self.has_se = bool(se_ratio)
        self.activation = ReLU(inplace=True)

    def forward(self, x):
        identity = self.proj(x, defer=True) if self.proj is not None else x      # applied inside f.c's BN pass
        h = self.f.b(self.f.a(x))
        if self.has_se:
            h = self.f.se(h)
        return self.f.c(h, residual=identity, relu=True)          # BN + identity + ReLU in one pass


class AnyStage(nn.Sequential):
    def __init__(self, width_in, width_out, stride, depth, group_width, bottleneck_multiplier, stage_index,
                 se_ratio=None):
        super().__init__()
        for i in range(depth):
            self.add_module("block%d-%d" % (stage_index, i),
                            ResBottleneckBlock(width_in if i == 0 else width_out, width_out, stride if i == 0 else 1,
                                               group_width, bottleneck_multiplier, se_ratio))


class RegNet(_ImageNetCNN):
    def __init__(self, params, num_classes: int = 1000, stem_width: int = 32):
        super().__init__()
        self.stem = _ConvBN(3, stem_width, 3, stride=2)
        self.stem[0].needs_input_grad = False
        width = stem_width
        blocks = []
        for i, (w_out, stride, depth, gw, mult, *se) in enumerate(params):
            blocks.append(("block%d" % (i + 1), AnyStage(width, w_out, stride, depth, gw, mult, i + 1,
                                                         se[0] if se else None)))
            width = w_out
        self.trunk_output = nn.Sequential(OrderedDict(blocks))
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(width, num_classes)
        for m in self.modules():
            if isinstance(m, Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, mean=0.0, std=math.sqrt(2.0 / fan_out))
            elif isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.trunk_output(self.stem(x))
        x = self.avgpool(x)
        return self.fc(x.reshape(x.shape[0], -1), out_dtype=torch.float32)


def regnet_x_400mf(**kw):
    return RegNet(block_params(depth=22, w_0=24, w_a=24.48, w_m=2.54, group_width=16), **kw)


def regnet_x_800mf(**kw):
    return RegNet(block_params(depth=16, w_0=56, w_a=35.73, w_m=2.28, group_width=16), **kw)


def regnet_x_1_6gf(**kw):
    return RegNet(block_params(depth=18, w_0=80, w_a=34.01, w_m=2.25, group_width=24), **kw)


def regnet_x_3_2gf(**kw):
    return RegNet(block_params(depth=25, w_0=88, w_a=26.31, w_m=2.25, group_width=48), **kw)


def regnet_x_8gf(**kw):
    return RegNet(block_params(depth=23, w_0=80, w_a=49.56, w_m=2.88, group_width=120), **kw)


def regnet_x_16gf(**kw):
    return RegNet(block_params(depth=22, w_0=216, w_a=55.59, w_m=2.1, group_width=128), **kw)


def regnet_x_32gf(**kw):
    return RegNet(block_params(depth=23, w_0=320, w_a=69.86, w_m=2.0, group_width=168), **kw)


def regnet_y_400mf(**kw):
    return RegNet(block_params(depth=16, w_0=48, w_a=27.89, w_m=2.09, group_width=8, se_ratio=0.25), **kw)


def regnet_y_800mf(**kw):
    return RegNet(block_params(depth=14, w_0=56, w_a=38.84, w_m=2.4, group_width=16, se_ratio=0.25), **kw)


def regnet_y_1_6gf(**kw):
    return RegNet(block_params(depth=27, w_0=48, w_a=20.71, w_m=2.65, group_width=24, se_ratio=0.25), **kw)


def regnet_y_3_2gf(**kw):
    return RegNet(block_params(depth=21, w_0=80, w_a=42.63, w_m=2.66, group_width=24, se_ratio=0.25), **kw)


def regnet_y_8gf(**kw):
    return RegNet(block_params(depth=17, w_0=192, w_a=76.82, w_m=2.19, group_width=56, se_ratio=0.25), **kw)


def regnet_y_16gf(**kw):
    return RegNet(block_params(depth=18, w_0=200, w_a=106.23, w_m=2.48, group_width=112, se_ratio=0.25), **kw)


def regnet_y_32gf(**kw):
    return RegNet(block_params(depth=20, w_0=232, w_a=115.89, w_m=2.53, group_width=232, se_ratio=0.25), **kw)


def regnet_y_128gf(**kw):
    return RegNet(block_params(depth=27, w_0=456, w_a=160.83, w_m=2.52, group_width=264, se_ratio=0.25), **kw)

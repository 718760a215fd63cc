"""EfficientNet B0 ... B7 and EfficientNetV2 S / M / L in NHWC (torchvision zoo members reachable through the
reference's ``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``features.1.0.block.0.0.weight``, ``features.2.0.block.2.fc1.weight``,
``classifier.1.weight``) so its checkpoints load.  MBConv = 1x1 expand -> depthwise k x k -> squeeze-excitation
(SiLU inside, sigmoid gate) -> 1x1 project, with per-sample stochastic depth on the residual branch; FusedMBConv (V2)
replaces expand + depthwise by one dense k x k convolution.  Depthwise convolutions are the grouped implicit-GEMM
kernel with groups == channels; BN is the fused (activation-free) normalise pass followed by the SiLU kernel; in eval
mode (no stochastic depth) the identity joins inside the last BN pass.  Parameter counts: b0 5,288,548; b1 7,794,184;
b2 9,109,994; b3 12,233,232; b4 19,341,616; b5 30,389,784; b6 43,040,704; b7 66,347,960; v2_s 21,458,488;
v2_m 54,139,356; v2_l 118,515,272.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ..nn.modules import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, Linear, SiLU, SqueezeExcitation,
                          StochasticDepth, conv_bn_act)
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN
from .mobilenet import _make_divisible


class _ConvBNSiLU(nn.Sequential):
    """torchvision's ``Conv2dNormActivation``: 0 conv, 1 norm (, 2 SiLU)."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, act=True, bn=None):
        layers = [Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=False,
                         groups=groups), (bn or BatchNorm2d)(cout)]
        if act:
            layers.append(SiLU(inplace=True))
        super().__init__(*layers)
        self.act = act

    def forward(self, x, residual=None):
        if self.act:
            return conv_bn_act(x, self[0], self[1], act="silu")
        return conv_bn_act(x, self[0], self[1], relu=False, residual=residual)


def _adjust_channels(channels: int, width_mult: float) -> int:
    return _make_divisible(channels * width_mult, 8)


class _Block(nn.Module):
    """Shared residual logic of MBConv / FusedMBConv: ``block`` ends in an activation-free conv + BN."""

    def _run(self, x):
        mods = list(self.block)
        h = x
        for m in mods[:-1]:
            h = m(h)
        last = mods[-1]
        if not self.use_res_connect:
            return last(h)
        if not isinstance(last, _ConvBNSiLU) or last.act:
            # FusedMBConv without expansion: the only layer carries the SiLU, so the identity joins afterwards
            return F.add(self.stochastic_depth(last(h)), x)
        if self.training and self.stochastic_depth.p > 0.0:
            return F.add(self.stochastic_depth(last(h)), x)
        return last(h, residual=x)                 # identity joins inside the BN pass

    def forward(self, x):
        return self._run(x)


class MBConv(_Block):
    def __init__(self, expand_ratio, kernel, stride, cin, cout, sd_prob, bn):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError("illegal stride value")
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        expanded = _adjust_channels(cin, expand_ratio)
        if expanded != cin:
            layers.append(_ConvBNSiLU(cin, expanded, 1, bn=bn))
        layers.append(_ConvBNSiLU(expanded, expanded, kernel, stride=stride, groups=expanded, bn=bn))
        layers.append(SqueezeExcitation(expanded, max(1, cin // 4), activation="silu"))
        layers.append(_ConvBNSiLU(expanded, cout, 1, act=False, bn=bn))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")
        self.out_channels = cout


class FusedMBConv(_Block):
    def __init__(self, expand_ratio, kernel, stride, cin, cout, sd_prob, bn):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError("illegal stride value")
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        expanded = _adjust_channels(cin, expand_ratio)
        if expanded != cin:
            layers.append(_ConvBNSiLU(cin, expanded, kernel, stride=stride, bn=bn))
            layers.append(_ConvBNSiLU(expanded, cout, 1, act=False, bn=bn))
        else:
            layers.append(_ConvBNSiLU(cin, cout, kernel, stride=stride, bn=bn))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")
        self.out_channels = cout


class EfficientNet(_ImageNetCNN):
    """``setting``: rows ``(block, expand_ratio, kernel, stride, cin, cout, num_layers)`` with channel counts and
    depths already scaled."""

    def __init__(self, setting, dropout: float, stochastic_depth_prob: float = 0.2, num_classes: int = 1000,
                 bn=None, last_channel=None):
        super().__init__()
        bn = bn or BatchNorm2d
        stem = _ConvBNSiLU(3, setting[0][4], 3, stride=2, bn=bn)
        stem[0].needs_input_grad = False
        layers = [stem]
        total = sum(row[6] for row in setting)
        done = 0
        for block, ratio, kernel, stride, cin, cout, n in setting:
            stage = []
            for i in range(n):
                sd = stochastic_depth_prob * float(done) / total
                stage.append(block(ratio, kernel, stride if i == 0 else 1, cin if i == 0 else cout, cout, sd, bn))
                done += 1
            layers.append(nn.Sequential(*stage))
        cin = setting[-1][5]
        cout = last_channel if last_channel is not None else 4 * cin
        layers.append(_ConvBNSiLU(cin, cout, 1, bn=bn))
        self.features = nn.Sequential(*layers)
        self.avgpool = AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(Dropout(dropout), Linear(cout, num_classes))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, Linear):
                r = 1.0 / math.sqrt(m.out_features)
                nn.init.uniform_(m.weight, -r, r)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.features(x)
        x = self.avgpool(x)
        x = self.classifier[0](x.reshape(x.shape[0], -1))
        return self.classifier[1](x, out_dtype=torch.float32)


_B0 = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
       (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]


def _scaled(width_mult: float, depth_mult: float):
    return [(MBConv, r, k, s, _adjust_channels(ci, width_mult), _adjust_channels(co, width_mult),
             int(math.ceil(n * depth_mult))) for r, k, s, ci, co, n in _B0]


def _bn_eps(eps: float, momentum: float = 0.1):
    return lambda c: BatchNorm2d(c, eps=eps, momentum=momentum)


def efficientnet_b0(**kw):
    return EfficientNet(_scaled(1.0, 1.0), 0.2, **kw)


def efficientnet_b1(**kw):
    return EfficientNet(_scaled(1.0, 1.1), 0.2, **kw)


def efficientnet_b2(**kw):
    return EfficientNet(_scaled(1.1, 1.2), 0.3, **kw)


def efficientnet_b3(**kw):
    return EfficientNet(_scaled(1.2, 1.4), 0.3, **kw)


def efficientnet_b4(**kw):
    return EfficientNet(_scaled(1.4, 1.8), 0.4, **kw)


def efficientnet_b5(**kw):
    return EfficientNet(_scaled(1.6, 2.2), 0.4, bn=_bn_eps(0.001, 0.01), **kw)


def efficientnet_b6(**kw):
    return EfficientNet(_scaled(1.8, 2.6), 0.5, bn=_bn_eps(0.001, 0.01), **kw)


def efficientnet_b7(**kw):
    return EfficientNet(_scaled(2.0, 3.1), 0.5, bn=_bn_eps(0.001, 0.01), **kw)


_V2_S = [(FusedMBConv, 1, 3, 1, 24, 24, 2), (FusedMBConv, 4, 3, 2, 24, 48, 4), (FusedMBConv, 4, 3, 2, 48, 64, 4),
         (MBConv, 4, 3, 2, 64, 128, 6), (MBConv, 6, 3, 1, 128, 160, 9), (MBConv, 6, 3, 2, 160, 256, 15)]
_V2_M = [(FusedMBConv, 1, 3, 1, 24, 24, 3), (FusedMBConv, 4, 3, 2, 24, 48, 5), (FusedMBConv, 4, 3, 2, 48, 80, 5),
         (MBConv, 4, 3, 2, 80, 160, 7), (MBConv, 6, 3, 1, 160, 176, 14), (MBConv, 6, 3, 2, 176, 304, 18),
         (MBConv, 6, 3, 1, 304, 512, 5)]
_V2_L = [(FusedMBConv, 1, 3, 1, 32, 32, 4), (FusedMBConv, 4, 3, 2, 32, 64, 7), (FusedMBConv, 4, 3, 2, 64, 96, 7),
         (MBConv, 4, 3, 2, 96, 192, 10), (MBConv, 6, 3, 1, 192, 224, 19), (MBConv, 6, 3, 2, 224, 384, 25),
         (MBConv, 6, 3, 1, 384, 640, 7)]


def efficientnet_v2_s(**kw):
    return EfficientNet(_V2_S, 0.2, bn=_bn_eps(1e-3), last_channel=1280, **kw)


def efficientnet_v2_m(**kw):
    return EfficientNet(_V2_M, 0.3, bn=_bn_eps(1e-3), last_channel=1280, **kw)


def efficientnet_v2_l(**kw):
    return EfficientNet(_V2_L, 0.4, bn=_bn_eps(1e-3), last_channel=1280, **kw)

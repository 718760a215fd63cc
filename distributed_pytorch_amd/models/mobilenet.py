"""Depthwise-separable families in NHWC: MobileNetV2, MobileNetV3 large / small and MNASNet 0.5 / 0.75 / 1.0 / 1.3 (torchvision zoo members
reachable through the reference's ``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``features.1.conv.0.0.weight``, ``layers.8.0.layers.3.weight``,
``classifier.1.weight``) so its checkpoints load.  A depthwise convolution is the grouped implicit-GEMM kernel
with groups == channels: 64 consecutive channels share one block-diagonal MFMA tile (``hip_ops.group_merge``),
one launch per pass.  BN + ReLU6 is the same fused normalise pass as BN + ReLU (activation code 6).  Parameter
counts: mobilenet_v2 3,504,872; mobilenet_v3_large 5,483,032; mobilenet_v3_small 2,542,856; mnasnet0_5 2,218,512; mnasnet0_75 3,170,208; mnasnet1_0 4,383,312;
mnasnet1_3 6,282,256.
"""
from __future__ import annotations

import torch
from torch import nn

from ..nn.modules import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, Hardswish, Linear, ReLU, ReLU6,
                          SqueezeExcitation, conv_bn_act)
from .alexnet_vgg import _ImageNetCNN, run_fused_sequential


def _make_divisible(v: float, divisor: int, min_value=None) -> int:
    min_value = divisor if min_value is None else min_value
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


class ConvBNReLU6(nn.Sequential):
    """torchvision's ``Conv2dNormActivation(..., activation_layer=ReLU6)``: indices 0 / 1 / 2."""

    def __init__(self, cin: int, cout: int, kernel_size: int = 3, stride: int = 1, groups: int = 1):
        super().__init__(Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=False,
                                groups=groups),
                         BatchNorm2d(cout), ReLU6(inplace=True))

    def forward(self, x):
        return conv_bn_act(x, self[0], self[1], relu=6)


class InvertedResidual(nn.Module):
    def __init__(self, inp: int, oup: int, stride: int, expand_ratio: int):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError("stride should be 1 or 2 instead of %d" % stride)
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU6(inp, hidden, kernel_size=1))
        layers += [ConvBNReLU6(hidden, hidden, stride=stride, groups=hidden),
                   Conv2d(hidden, oup, 1, bias=False), BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)
        self.out_channels = oup

    def forward(self, x):
        mods = list(self.conv)
        h = x
        for m in mods[:-2]:
            h = m(h)
        # linear bottleneck: BN without activation; the identity joins inside the same normalise pass
        return conv_bn_act(h, mods[-2], mods[-1], relu=False, residual=x if self.use_res_connect else None)


class MobileNetV2(_ImageNetCNN):
    def __init__(self, num_classes: int = 1000, width_mult: float = 1.0, round_nearest: int = 8,
                 dropout: float = 0.2):
        super().__init__()
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                   [6, 320, 1, 1]]
        cin = _make_divisible(32 * width_mult, round_nearest)
        self.last_channel = _make_divisible(1280 * max(1.0, width_mult), round_nearest)
        stem = ConvBNReLU6(3, cin, stride=2)
        stem[0].needs_input_grad = False
        features = [stem]
        for t, c, n, s in setting:
            cout = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                features.append(InvertedResidual(cin, cout, s if i == 0 else 1, t))
                cin = cout
        features.append(ConvBNReLU6(cin, self.last_channel, kernel_size=1))
        self.features = nn.Sequential(*features)
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.classifier = nn.Sequential(Dropout(dropout), Linear(self.last_channel, num_classes))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.features(x)
        x = self.avgpool(x)
        x = self.classifier[0](x.reshape(x.shape[0], -1))
        return self.classifier[1](x, out_dtype=torch.float32)


def mobilenet_v2(**kw):
    return MobileNetV2(**kw)



# ------------------------------------------------------------------------------------------ MobileNetV3
class _ConvBNAct(nn.Sequential):
    """torchvision's ``Conv2dNormActivation``: 0 conv, 1 norm (, 2 activation: ReLU or Hardswish)."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, act="relu", eps=1e-3, momentum=0.01):
        layers = [Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=False,
                         groups=groups), BatchNorm2d(cout, eps=eps, momentum=momentum)]
        if act == "relu":
            layers.append(ReLU(inplace=True))
        elif act == "hardswish":
            layers.append(Hardswish(inplace=True))
        elif act is not None:
            raise ValueError("unsupported activation %r" % (act,))
        super().__init__(*layers)
        self.act = act

    def forward(self, x, residual=None):
        if self.act == "hardswish":
            return conv_bn_act(x, self[0], self[1], act="hardswish")
        return conv_bn_act(x, self[0], self[1], relu=self.act == "relu", residual=residual)


class _V3InvertedResidual(nn.Module):
    def __init__(self, cin, kernel, expanded, cout, use_se, act, stride):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError("illegal stride value")
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        if expanded != cin:
            layers.append(_ConvBNAct(cin, expanded, 1, act=act))
        layers.append(_ConvBNAct(expanded, expanded, kernel, stride=stride, groups=expanded, act=act))
        if use_se:
            layers.append(SqueezeExcitation(expanded, _make_divisible(expanded // 4, 8), activation="relu",
                                            scale_activation="hardsigmoid"))
        layers.append(_ConvBNAct(expanded, cout, 1, act=None))
        self.block = nn.Sequential(*layers)
        self.out_channels = cout

    def forward(self, x):
        mods = list(self.block)
        h = x
        for m in mods[:-1]:
            h = m(h)
        return mods[-1](h, residual=x if self.use_res_connect else None)     # identity joins inside the BN pass


_V3_LARGE = [(16, 3, 16, 16, False, "relu", 1), (16, 3, 64, 24, False, "relu", 2), (24, 3, 72, 24, False, "relu", 1),
             (24, 5, 72, 40, True, "relu", 2), (40, 5, 120, 40, True, "relu", 1), (40, 5, 120, 40, True, "relu", 1),
             (40, 3, 240, 80, False, "hardswish", 2), (80, 3, 200, 80, False, "hardswish", 1),
             (80, 3, 184, 80, False, "hardswish", 1), (80, 3, 184, 80, False, "hardswish", 1),
             (80, 3, 480, 112, True, "hardswish", 1), (112, 3, 672, 112, True, "hardswish", 1),
             (112, 5, 672, 160, True, "hardswish", 2), (160, 5, 960, 160, True, "hardswish", 1),
             (160, 5, 960, 160, True, "hardswish", 1)]
_V3_SMALL = [(16, 3, 16, 16, True, "relu", 2), (16, 3, 72, 24, False, "relu", 2), (24, 3, 88, 24, False, "relu", 1),
             (24, 5, 96, 40, True, "hardswish", 2), (40, 5, 240, 40, True, "hardswish", 1),
             (40, 5, 240, 40, True, "hardswish", 1), (40, 5, 120, 48, True, "hardswish", 1),
             (48, 5, 144, 48, True, "hardswish", 1), (48, 5, 288, 96, True, "hardswish", 2),
             (96, 5, 576, 96, True, "hardswish", 1), (96, 5, 576, 96, True, "hardswish", 1)]


class MobileNetV3(_ImageNetCNN):
    def __init__(self, setting, last_channel: int, num_classes: int = 1000, dropout: float = 0.2):
        super().__init__()
        stem = _ConvBNAct(3, setting[0][0], 3, stride=2, act="hardswish")
        stem[0].needs_input_grad = False
        layers = [stem] + [_V3InvertedResidual(*cfg) for cfg in setting]
        cin = setting[-1][3]
        layers.append(_ConvBNAct(cin, 6 * cin, 1, act="hardswish"))
        self.features = nn.Sequential(*layers)
        self.avgpool = AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(Linear(6 * cin, last_channel), Hardswish(inplace=True), Dropout(dropout),
                                        Linear(last_channel, num_classes))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.features(x)
        x = self.avgpool(x)
        c = self.classifier
        x = c[2](c[1](c[0](x.reshape(x.shape[0], -1))))
        return c[3](x, out_dtype=torch.float32)


def mobilenet_v3_large(**kw):
    return MobileNetV3(_V3_LARGE, 1280, **kw)


def mobilenet_v3_small(**kw):
    return MobileNetV3(_V3_SMALL, 1024, **kw)


# ------------------------------------------------------------------------------------------ MNASNet
_MNAS_BN_MOMENTUM = 1 - 0.9997


class _MnasInvertedResidual(nn.Module):
    def __init__(self, cin: int, cout: int, kernel_size: int, stride: int, expansion: int):
        super().__init__()
        if stride not in (1, 2) or kernel_size not in (3, 5):
            raise ValueError("unsupported MNASNet block geometry")
        mid = cin * expansion
        self.apply_residual = cin == cout and stride == 1
        self.layers = nn.Sequential(
            Conv2d(cin, mid, 1, bias=False), BatchNorm2d(mid, momentum=_MNAS_BN_MOMENTUM), ReLU(inplace=True),
            Conv2d(mid, mid, kernel_size, padding=kernel_size // 2, stride=stride, groups=mid, bias=False),
            BatchNorm2d(mid, momentum=_MNAS_BN_MOMENTUM), ReLU(inplace=True),
            Conv2d(mid, cout, 1, bias=False), BatchNorm2d(cout, momentum=_MNAS_BN_MOMENTUM))

    def forward(self, x):
        L = self.layers
        h = conv_bn_act(x, L[0], L[1], relu=True)
        h = conv_bn_act(h, L[3], L[4], relu=True)
        return conv_bn_act(h, L[6], L[7], relu=False, residual=x if self.apply_residual else None)


def _mnas_stack(cin, cout, kernel_size, stride, expansion, repeats):
    first = _MnasInvertedResidual(cin, cout, kernel_size, stride, expansion)
    rest = [_MnasInvertedResidual(cout, cout, kernel_size, 1, expansion) for _ in range(1, repeats)]
    return nn.Sequential(first, *rest)


def _mnas_depths(alpha: float):
    return [_make_divisible(d * alpha, 8) for d in (32, 16, 24, 40, 80, 96, 192, 320)]


class MNASNet(_ImageNetCNN):
    def __init__(self, alpha: float, num_classes: int = 1000, dropout: float = 0.2):
        super().__init__()
        if alpha <= 0.0:
            raise ValueError("alpha should be greater than 0.0 instead of %r" % alpha)
        self.alpha, self.num_classes = alpha, num_classes
        d = _mnas_depths(alpha)
        bn = lambda c: BatchNorm2d(c, momentum=_MNAS_BN_MOMENTUM)     # noqa: E731
        first = Conv2d(3, d[0], 3, padding=1, stride=2, bias=False)
        first.needs_input_grad = False
        self.layers = nn.Sequential(
            first, bn(d[0]), ReLU(inplace=True),
            Conv2d(d[0], d[0], 3, padding=1, stride=1, groups=d[0], bias=False), bn(d[0]), ReLU(inplace=True),
            Conv2d(d[0], d[1], 1, bias=False), bn(d[1]),
            _mnas_stack(d[1], d[2], 3, 2, 3, 3), _mnas_stack(d[2], d[3], 5, 2, 3, 3),
            _mnas_stack(d[3], d[4], 5, 2, 6, 3), _mnas_stack(d[4], d[5], 3, 1, 6, 2),
            _mnas_stack(d[5], d[6], 5, 2, 6, 4), _mnas_stack(d[6], d[7], 3, 1, 6, 1),
            Conv2d(d[7], 1280, 1, bias=False), bn(1280), ReLU(inplace=True))
        self.avgpool = AdaptiveAvgPool2d((1, 1))      # x.mean([2, 3]) in torchvision: no parameters, no key
        self.classifier = nn.Sequential(Dropout(dropout), Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, Linear):
                nn.init.kaiming_uniform_(m.weight, mode="fan_out", nonlinearity="sigmoid")
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = run_fused_sequential(self.layers, x)
        x = self.avgpool(x)
        x = self.classifier[0](x.reshape(x.shape[0], -1))
        return self.classifier[1](x, out_dtype=torch.float32)


def mnasnet0_5(**kw):
    return MNASNet(0.5, **kw)


def mnasnet0_75(**kw):
    return MNASNet(0.75, **kw)


def mnasnet1_0(**kw):
    return MNASNet(1.0, **kw)


def mnasnet1_3(**kw):
    return MNASNet(1.3, **kw)

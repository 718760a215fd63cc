"""ResNet family (18/34/50/101/152, wide and ResNeXt variants) in NHWC with fused blocks.

Architecture and parameter names equal torchvision's ``resnet.py`` v1.5
(stride on the 3x3 of a bottleneck) so that ``state_dict`` keys —
``conv1.weight``, ``layer1.0.downsample.0.weight``, ``fc.weight`` … — load
checkpoints written by the reference (``models.__dict__[args.arch]()``,
reference ``nd_imagenet.py:133-138``).  Parameter counts: resnet18 11,689,512;
resnet50 25,557,032.

Each ``conv -> BN -> ReLU`` triple runs as: implicit-GEMM conv whose epilogue
emits per-channel sum/sumsq partials -> BN finalize -> one fused
normalise(+residual)(+ReLU) pass (``nn.conv_bn_act``).
"""
from __future__ import annotations

from typing import List, Optional, Type, Union

import torch
from torch import nn

from .. import amp
from ..nn.modules import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLU,
                          conv_bn_act, conv_bn_relu_maxpool, conv_bn_cat_relu)
from ..ops import functional as F

STEM_PAD_C = 8  # image channels 3 -> 8 on the GPU so every im2col chunk is one 16-byte vector


def conv3x3(cin, cout, stride=1, groups=1):
    return Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False, groups=groups)


def conv1x1(cin, cout, stride=1):
    return Conv2d(cin, cout, 1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, base_width=64, groups=1):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # `identity` aliases x: its gradient is added inside conv1's dgrad epilogue
        # bn1 + ReLU are deferred: conv2 applies them while it stages its input tiles (nothing is written in between)
        out, identity = conv_bn_act(x, self.conv1, self.bn1, relu=True, skip=True, defer=True)
        if self.downsample is not None:
            identity = conv_bn_act(identity, self.downsample[0], self.downsample[1], relu=False, defer=True)
        return conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, base_width=64, groups=1):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = conv3x3(width, width, stride, groups)     # ResNeXt: the cardinality lives here
        self.bn2 = BatchNorm2d(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = BatchNorm2d(planes * self.expansion)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # `identity` aliases x: its gradient is added inside conv1's dgrad epilogue
        # bn1 / bn2 + ReLU are deferred to the convolution that consumes them (applied on load in fprop and wgrad):
        # the two normalised activations of a bottleneck are never written to memory
        out, identity = conv_bn_act(x, self.conv1, self.bn1, relu=True, skip=True, defer=True)
        out = conv_bn_act(out, self.conv2, self.bn2, relu=True, defer=True)
        if self.downsample is not None:
            # both branches as the two operands of one GEMM (statistics from Gram matrices) where that applies
            y = conv_bn_cat_relu(out, self.conv3, self.bn3, identity, self.downsample[0], self.downsample[1])
            if y is not None:
                return y
            identity = conv_bn_act(identity, self.downsample[0], self.downsample[1], relu=False, defer=True)
        return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=identity)


class ResNet(nn.Module):
    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]], layers: List[int],
                 num_classes: int = 1000, zero_init_residual: bool = False,
                 groups: int = 1, width_per_group: int = 64):
        super().__init__()
        self.inplanes = 64
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.conv1.needs_input_grad = False
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.base_width, self.groups)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, base_width=self.base_width, groups=self.groups))
        return nn.Sequential(*layers)

    stem_s2d = True      # 7x7 stride-2 pad-3 stem: takes the space-to-depth image layout on the GPU (F.to_s2d)

    def prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        """Accepts torchvision-style float NCHW, or uint8 (NHWC/NCHW) straight from
        the prefetcher; returns normalised NHWC in the compute dtype."""
        dtype = amp.compute_dtype(torch.float32)
        pad = STEM_PAD_C if F._on_gpu(x) else 0
        if getattr(x, "_dpa_nhwc", False):
            return x
        if self.stem_s2d and F.s2d_ok(x):
            return F.to_s2d(x, dtype)
        if x.dtype == torch.uint8:
            layout = "nhwc" if x.shape[-1] in (1, 3) else "nchw"
            return F.normalize_u8(x, dtype, pad, input_layout=layout)
        if getattr(x, "_dpa_nhwc", False):
            return x
        return F.to_nhwc(x, dtype, pad)

    def forward(self, x):
        x = self.prepare_input(x)
        x = conv_bn_relu_maxpool(x, self.conv1, self.bn1, self.maxpool)     # fused stem
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)                       # [N,1,1,C]
        x = x.reshape(x.shape[0], -1)
        return self.fc(x, out_dtype=torch.float32)


def _resnet(block, layers, **kw):
    return ResNet(block, layers, **kw)


def resnet18(**kw):
    return _resnet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(**kw):
    return _resnet(BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(**kw):
    return _resnet(Bottleneck, [3, 4, 6, 3], **kw)


def resnet101(**kw):
    return _resnet(Bottleneck, [3, 4, 23, 3], **kw)


def resnet152(**kw):
    return _resnet(Bottleneck, [3, 8, 36, 3], **kw)


def wide_resnet50_2(**kw):
    return _resnet(Bottleneck, [3, 4, 6, 3], width_per_group=128, **kw)


def wide_resnet101_2(**kw):
    return _resnet(Bottleneck, [3, 4, 23, 3], width_per_group=128, **kw)


def resnext50_32x4d(**kw):
    return _resnet(Bottleneck, [3, 4, 6, 3], groups=32, width_per_group=4, **kw)


def resnext101_32x8d(**kw):
    return _resnet(Bottleneck, [3, 4, 23, 3], groups=32, width_per_group=8, **kw)


def resnext101_64x4d(**kw):
    return _resnet(Bottleneck, [3, 4, 23, 3], groups=64, width_per_group=4, **kw)

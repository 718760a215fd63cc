"""DenseNet-121 / 161 / 169 / 201 in NHWC (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Parameter names match torchvision (``features.denseblock1.denselayer1.norm1.weight`` ...).  DenseNet is
pre-activation (BN -> ReLU -> conv): BN+ReLU is one fused kernel pair (statistics + apply), the 1x1 bottleneck
convolution hands its BatchNorm statistics to ``norm2`` from its epilogue, and feature reuse is the NHWC
channel-concat kernel.  Parameter counts: 121: 7,978,856; 169: 14,149,480; 201: 20,013,928; 161: 28,681,000.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Tuple

import torch
from torch import nn

from ..nn.modules import (AdaptiveAvgPool2d, AvgPool2d, BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLU, conv_bn_act,
                          conv_bn_relu_maxpool)
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


class _DenseLayer(nn.Module):
    def __init__(self, num_input_features: int, growth_rate: int, bn_size: int):
        super().__init__()
        self.norm1 = BatchNorm2d(num_input_features)
        self.relu1 = ReLU(inplace=True)
        self.conv1 = Conv2d(num_input_features, bn_size * growth_rate, 1, bias=False)
        self.norm2 = BatchNorm2d(bn_size * growth_rate)
        self.relu2 = ReLU(inplace=True)
        self.conv2 = Conv2d(bn_size * growth_rate, growth_rate, 3, padding=1, bias=False)

    def forward(self, x):
        y = self.norm1(x, relu=True)
        y = conv_bn_act(y, self.conv1, self.norm2, relu=True)     # conv1 -> norm2 -> relu2
        return self.conv2(y)


class _DenseBlock(nn.ModuleDict):
    def __init__(self, num_layers: int, num_input_features: int, bn_size: int, growth_rate: int):
        super().__init__()
        for i in range(num_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size)

    def forward(self, x):
        for layer in self.values():
            x = F.cat_channels([x, layer(x)])
        return x


class _Transition(nn.Sequential):
    def __init__(self, num_input_features: int, num_output_features: int):
        super().__init__(OrderedDict([
            ("norm", BatchNorm2d(num_input_features)),
            ("relu", ReLU(inplace=True)),
            ("conv", Conv2d(num_input_features, num_output_features, 1, bias=False)),
            ("pool", AvgPool2d(2, 2)),
        ]))

    def forward(self, x):
        return self.pool(self.conv(self.norm(x, relu=True)))


class DenseNet(_ImageNetCNN):
    stem_s2d = True
    def __init__(self, growth_rate: int = 32, block_config: Tuple[int, ...] = (6, 12, 24, 16),
                 num_init_features: int = 64, bn_size: int = 4, num_classes: int = 1000):
        super().__init__()
        conv0 = Conv2d(3, num_init_features, 7, stride=2, padding=3, bias=False)
        conv0.needs_input_grad = False
        self.features = nn.Sequential(OrderedDict([
            ("conv0", conv0),
            ("norm0", BatchNorm2d(num_init_features)),
            ("relu0", ReLU(inplace=True)),
            ("pool0", MaxPool2d(3, 2, 1)),
        ]))
        n = num_init_features
        for i, num_layers in enumerate(block_config):
            self.features.add_module("denseblock%d" % (i + 1), _DenseBlock(num_layers, n, bn_size, growth_rate))
            n += num_layers * growth_rate
            if i != len(block_config) - 1:
                self.features.add_module("transition%d" % (i + 1), _Transition(n, n // 2))
                n //= 2
        self.features.add_module("norm5", BatchNorm2d(n))
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.classifier = Linear(n, num_classes)
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.prepare_input(x)
        f = self.features
        x = conv_bn_relu_maxpool(x, f.conv0, f.norm0, f.pool0)          # fused stem
        for name, m in f.named_children():
            if name.startswith("denseblock") or name.startswith("transition"):
                x = m(x)
        x = f.norm5(x, relu=True)
        x = self.avgpool(x)
        x = torch.flatten(x, 1)
        return self.classifier(x, out_dtype=torch.float32)


def densenet121(**kw):
    return DenseNet(32, (6, 12, 24, 16), 64, **kw)


def densenet161(**kw):
    return DenseNet(48, (6, 12, 36, 24), 96, **kw)


def densenet169(**kw):
    return DenseNet(32, (6, 12, 32, 32), 64, **kw)


def densenet201(**kw):
    return DenseNet(32, (6, 12, 48, 32), 64, **kw)

"""SqueezeNet 1.0 / 1.1 in NHWC (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``features.3.squeeze.weight``, ``classifier.1.bias``) so its
checkpoints load.  Every convolution is conv+bias+ReLU in one kernel launch; the two expand
branches of a fire module are joined by the NHWC channel-concat kernel.  Parameter counts:
squeezenet1_0 1,248,424; squeezenet1_1 1,235,496.
"""
from __future__ import annotations

import torch
from torch import nn

from ..nn.modules import AdaptiveAvgPool2d, Conv2d, Dropout, MaxPool2d, ReLU
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


class Fire(nn.Module):
    def __init__(self, inplanes: int, squeeze_planes: int, expand1x1_planes: int, expand3x3_planes: int):
        super().__init__()
        self.squeeze = Conv2d(inplanes, squeeze_planes, 1)
        self.squeeze_activation = ReLU(inplace=True)
        self.expand1x1 = Conv2d(squeeze_planes, expand1x1_planes, 1)
        self.expand1x1_activation = ReLU(inplace=True)
        self.expand3x3 = Conv2d(squeeze_planes, expand3x3_planes, 3, padding=1)
        self.expand3x3_activation = ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze(x, relu=True)
        return F.cat_channels([self.expand1x1(x, relu=True), self.expand3x3(x, relu=True)])


class SqueezeNet(_ImageNetCNN):
    def __init__(self, version: str = "1_0", num_classes: int = 1000, dropout: float = 0.5):
        super().__init__()
        self.num_classes = num_classes
        if version == "1_0":
            first = Conv2d(3, 96, 7, stride=2)
            self.features = nn.Sequential(
                first, ReLU(inplace=True), MaxPool2d(3, 2, ceil_mode=True),
                Fire(96, 16, 64, 64), Fire(128, 16, 64, 64), Fire(128, 32, 128, 128),
                MaxPool2d(3, 2, ceil_mode=True),
                Fire(256, 32, 128, 128), Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256),
                MaxPool2d(3, 2, ceil_mode=True),
                Fire(512, 64, 256, 256),
            )
        elif version == "1_1":
            first = Conv2d(3, 64, 3, stride=2)
            self.features = nn.Sequential(
                first, ReLU(inplace=True), MaxPool2d(3, 2, ceil_mode=True),
                Fire(64, 16, 64, 64), Fire(128, 16, 64, 64),
                MaxPool2d(3, 2, ceil_mode=True),
                Fire(128, 32, 128, 128), Fire(256, 32, 128, 128),
                MaxPool2d(3, 2, ceil_mode=True),
                Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256), Fire(512, 64, 256, 256),
            )
        else:
            raise ValueError("unsupported SqueezeNet version %r" % version)
        first.needs_input_grad = False
        final_conv = Conv2d(512, num_classes, 1)
        self.classifier = nn.Sequential(Dropout(dropout), final_conv, ReLU(inplace=True), AdaptiveAvgPool2d((1, 1)))
        for m in self.modules():
            if isinstance(m, Conv2d):
                if m is final_conv:
                    nn.init.normal_(m.weight, mean=0.0, std=0.01)
                else:
                    nn.init.kaiming_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.prepare_input(x)
        mods = list(self.features)
        x = mods[0](x, relu=True)                 # conv + bias + ReLU fused; mods[1] is the ReLU placeholder
        for m in mods[2:]:
            x = m(x)
        x = self.classifier[0](x)
        x = self.classifier[1](x, relu=True)
        x = self.classifier[3](x)
        return torch.flatten(x, 1).float()


def squeezenet1_0(**kw):
    return SqueezeNet("1_0", **kw)


def squeezenet1_1(**kw):
    return SqueezeNet("1_1", **kw)

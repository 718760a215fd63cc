"""Inception v3 in NHWC -- a torchvision zoo member reachable through the reference's ``models.__dict__[arch]()``
(``nd_imagenet.py:22-24,133-138``).  Native input extent 299 x 299 (any extent >= 75 runs).

Module names match torchvision (``Conv2d_1a_3x3.conv.weight``, ``Mixed_6b.branch7x7_2.bn.running_mean``,
``AuxLogits.conv1.conv.weight``, ``fc.weight``) so its checkpoints load.  Every BasicConv2d is the fused
conv -> BN(eps 1e-3) -> ReLU triple on the implicit-GEMM kernels -- the factorised 1x7 / 7x1 / 1x3 / 3x1 filters are
ordinary tap tables there (a 1x3 is one filter row of the filter-row kernel); branches are joined by the NHWC
channel-concat kernel; the 3x3 stride-1 pad-1 branch pools run on the fixed-window average-pool kernel.

``aux_logits`` defaults to False for the same reason as GoogLeNet (the reference's ``criterion(output, target)`` cannot
consume torchvision's training-mode namedtuple); ``aux_logits=True`` builds ``AuxLogits`` and returns
``InceptionOutputs(logits, aux_logits)`` while training.  Parameter counts: 23,834,568 without, 27,161,264 with the
auxiliary classifier.
"""
from __future__ import annotations

from collections import namedtuple

import torch
from torch import nn

from ..nn.modules import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, Linear, MaxPool2d
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN
from .googlenet import BasicConv2d

InceptionOutputs = namedtuple("InceptionOutputs", ["logits", "aux_logits"])


def _branch_pool(x):
    return F.avg_pool2d(x, 3, 1, 1)


class InceptionA(nn.Module):
    def __init__(self, cin: int, pool_features: int):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        return F.cat_channels([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                               self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                               self.branch_pool(_branch_pool(x))])


class InceptionB(nn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return F.cat_channels([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                               F.max_pool2d(x, 3, 2)])


class InceptionC(nn.Module):
    def __init__(self, cin: int, channels_7x7: int):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        d = self.branch7x7dbl_1(x)
        for m in (self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4, self.branch7x7dbl_5):
            d = m(d)
        return F.cat_channels([self.branch1x1(x), b7, d, self.branch_pool(_branch_pool(x))])


class InceptionD(nn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b7 = self.branch7x7x3_1(x)
        for m in (self.branch7x7x3_2, self.branch7x7x3_3, self.branch7x7x3_4):
            b7 = m(b7)
        return F.cat_channels([self.branch3x3_2(self.branch3x3_1(x)), b7, F.max_pool2d(x, 3, 2)])


class InceptionE(nn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        d = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        return F.cat_channels([self.branch1x1(x), self.branch3x3_2a(b3), self.branch3x3_2b(b3),
                               self.branch3x3dbl_3a(d), self.branch3x3dbl_3b(d), self.branch_pool(_branch_pool(x))])


class InceptionAux(nn.Module):
    def __init__(self, cin: int, num_classes: int):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(128, 768, kernel_size=5)
        self.conv1.stddev = 0.01
        self.fc = Linear(768, num_classes)
        self.fc.stddev = 0.001
        self.pool = AdaptiveAvgPool2d((1, 1))

    def forward(self, x):
        x = self.conv1(self.conv0(F.avg_pool2d(x, 5, 3)))          # 17 x 17 -> 5 x 5 -> 1 x 1
        x = self.pool(x)
        return self.fc(x.reshape(x.shape[0], -1), out_dtype=torch.float32)


class Inception3(_ImageNetCNN):
    def __init__(self, num_classes: int = 1000, aux_logits: bool = False, dropout: float = 0.5):
        super().__init__()
        self.aux_logits = aux_logits
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_1a_3x3.conv.needs_input_grad = False
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.maxpool1 = MaxPool2d(3, stride=2)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.maxpool2 = MaxPool2d(3, stride=2)
        self.Mixed_5b = InceptionA(192, pool_features=32)
        self.Mixed_5c = InceptionA(256, pool_features=64)
        self.Mixed_5d = InceptionA(288, pool_features=64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, channels_7x7=128)
        self.Mixed_6c = InceptionC(768, channels_7x7=160)
        self.Mixed_6d = InceptionC(768, channels_7x7=160)
        self.Mixed_6e = InceptionC(768, channels_7x7=192)
        self.AuxLogits = InceptionAux(768, num_classes) if aux_logits else None
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.dropout = Dropout(dropout)
        self.fc = Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, BasicConv2d) and hasattr(m, "stddev"):
                nn.init.trunc_normal_(m.conv.weight, mean=0.0, std=m.stddev, a=-2, b=2)
            elif isinstance(m, (Conv2d, Linear)):
                nn.init.trunc_normal_(m.weight, mean=0.0, std=float(getattr(m, "stddev", 0.1)), a=-2, b=2)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.AuxLogits is not None:            # the loop above reached conv1.conv after the BasicConv2d branch
            nn.init.trunc_normal_(self.AuxLogits.conv1.conv.weight, mean=0.0, std=0.01, a=-2, b=2)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.maxpool1(self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x))))
        x = self.maxpool2(self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x)))
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        aux = self.AuxLogits(x) if self.AuxLogits is not None and self.training else None
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = self.avgpool(x)
        x = self.dropout(x.reshape(x.shape[0], -1))
        x = self.fc(x, out_dtype=torch.float32)
        if self.training and self.aux_logits:
            return InceptionOutputs(x, aux)
        return x


def inception_v3(**kw):
    return Inception3(**kw)

"""GoogLeNet (Inception v1, BN variant) in NHWC -- a torchvision zoo member reachable through the reference's
``models.__dict__[arch]()`` (``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``conv1.conv.weight``, ``inception3a.branch2.1.bn.running_mean``, ``fc.weight``) so its
checkpoints load; like torchvision the "5x5" branch is a 3x3 convolution.  Every BasicConv2d is the fused
conv -> BN(eps 1e-3) -> ReLU triple; the four branches of an Inception block are joined by the NHWC channel-concat
kernel; the pooling branch is the ceil-mode max-pool kernel at stride 1.

``aux_logits`` defaults to False here: with the auxiliary heads torchvision returns a namedtuple in training mode, which
the reference's ``criterion(output, target)`` (``nd_imagenet.py:292``) cannot consume.  ``aux_logits=True`` builds them
(``aux1`` / ``aux2``) and returns ``GoogLeNetOutputs(logits, aux_logits2, aux_logits1)`` while training.
Parameter counts: 6,624,904 without, 13,004,888 with the auxiliary classifiers.
"""
from __future__ import annotations

from collections import namedtuple

import torch
from torch import nn

from ..nn.modules import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, Linear, MaxPool2d, conv_bn_act,
                          conv_bn_relu_maxpool)
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN

GoogLeNetOutputs = namedtuple("GoogLeNetOutputs", ["logits", "aux_logits2", "aux_logits1"])


class BasicConv2d(nn.Module):
    def __init__(self, cin: int, cout: int, **kw):
        super().__init__()
        self.conv = Conv2d(cin, cout, bias=False, **kw)
        self.bn = BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return conv_bn_act(x, self.conv, self.bn, relu=True)


class Inception(nn.Module):
    def __init__(self, cin, ch1x1, ch3x3red, ch3x3, ch5x5red, ch5x5, pool_proj):
        super().__init__()
        self.branch1 = BasicConv2d(cin, ch1x1, kernel_size=1)
        self.branch2 = nn.Sequential(BasicConv2d(cin, ch3x3red, kernel_size=1),
                                     BasicConv2d(ch3x3red, ch3x3, kernel_size=3, padding=1))
        self.branch3 = nn.Sequential(BasicConv2d(cin, ch5x5red, kernel_size=1),
                                     BasicConv2d(ch5x5red, ch5x5, kernel_size=3, padding=1))
        self.branch4 = nn.Sequential(MaxPool2d(3, stride=1, padding=1, ceil_mode=True),
                                     BasicConv2d(cin, pool_proj, kernel_size=1))

    def forward(self, x):
        return F.cat_channels([self.branch1(x), self.branch2(x), self.branch3(x), self.branch4(x)])


class InceptionAux(nn.Module):
    def __init__(self, cin: int, num_classes: int, dropout: float = 0.7):
        super().__init__()
        self.conv = BasicConv2d(cin, 128, kernel_size=1)
        self.fc1 = Linear(2048, 1024)
        self.fc2 = Linear(1024, num_classes)
        self.dropout = Dropout(dropout)
        self.pool = AdaptiveAvgPool2d((4, 4))

    def forward(self, x):
        x = self.conv(self.pool(x))                                   # [N, 4, 4, 128]
        # torchvision flattens NCHW: fc1's input index is c*16 + h*4 + w -- keep that order so its weights load
        x = x.permute(0, 3, 1, 2).reshape(x.shape[0], -1)
        x = self.dropout(self.fc1(x, relu=True))
        return self.fc2(x, out_dtype=torch.float32)


class GoogLeNet(_ImageNetCNN):
    stem_s2d = True
    def __init__(self, num_classes: int = 1000, aux_logits: bool = False, dropout: float = 0.2,
                 dropout_aux: float = 0.7):
        super().__init__()
        self.aux_logits = aux_logits
        self.conv1 = BasicConv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.conv1.conv.needs_input_grad = False
        self.maxpool1 = MaxPool2d(3, stride=2, ceil_mode=True)
        self.conv2 = BasicConv2d(64, 64, kernel_size=1)
        self.conv3 = BasicConv2d(64, 192, kernel_size=3, padding=1)
        self.maxpool2 = MaxPool2d(3, stride=2, ceil_mode=True)
        self.inception3a = Inception(192, 64, 96, 128, 16, 32, 32)
        self.inception3b = Inception(256, 128, 128, 192, 32, 96, 64)
        self.maxpool3 = MaxPool2d(3, stride=2, ceil_mode=True)
        self.inception4a = Inception(480, 192, 96, 208, 16, 48, 64)
        self.inception4b = Inception(512, 160, 112, 224, 24, 64, 64)
        self.inception4c = Inception(512, 128, 128, 256, 24, 64, 64)
        self.inception4d = Inception(512, 112, 144, 288, 32, 64, 64)
        self.inception4e = Inception(528, 256, 160, 320, 32, 128, 128)
        self.maxpool4 = MaxPool2d(2, stride=2, ceil_mode=True)
        self.inception5a = Inception(832, 256, 160, 320, 32, 128, 128)
        self.inception5b = Inception(832, 384, 192, 384, 48, 128, 128)
        if aux_logits:
            self.aux1 = InceptionAux(512, num_classes, dropout_aux)
            self.aux2 = InceptionAux(528, num_classes, dropout_aux)
        else:
            self.aux1 = self.aux2 = None
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.dropout = Dropout(dropout)
        self.fc = Linear(1024, num_classes)
        for m in self.modules():
            if isinstance(m, (Conv2d, Linear)):
                nn.init.trunc_normal_(m.weight, mean=0.0, std=0.01, a=-2, b=2)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.prepare_input(x)
        x = conv_bn_relu_maxpool(x, self.conv1.conv, self.conv1.bn, self.maxpool1)      # fused stem
        x = self.maxpool2(self.conv3(self.conv2(x)))
        x = self.maxpool3(self.inception3b(self.inception3a(x)))
        x = self.inception4a(x)
        aux1 = self.aux1(x) if self.aux1 is not None and self.training else None
        x = self.inception4d(self.inception4c(self.inception4b(x)))
        aux2 = self.aux2(x) if self.aux2 is not None and self.training else None
        x = self.maxpool4(self.inception4e(x))
        x = self.inception5b(self.inception5a(x))
        x = self.avgpool(x)
        x = self.dropout(x.reshape(x.shape[0], -1))
        x = self.fc(x, out_dtype=torch.float32)
        if self.training and self.aux_logits:
            return GoogLeNetOutputs(x, aux2, aux1)
        return x


def googlenet(**kw):
    return GoogLeNet(**kw)

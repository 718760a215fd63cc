"""AlexNet and VGG-11/13/16/19 (+BN variants) in NHWC.

Module indices inside ``features`` / ``classifier`` match torchvision so keys
such as ``features.3.weight`` / ``classifier.6.bias`` load reference
checkpoints (the reference special-cases these families at
``nd_imagenet.py:165-167``; README.md:32-36 recommends ``--lr 0.01``).
Conv+bias+ReLU and Linear+bias+ReLU run as single fused kernels: the
``ReLU`` modules stay in the ``Sequential`` only to keep the indices.
Parameter counts: alexnet 61,100,840; vgg16 138,357,544.
"""
from __future__ import annotations

from typing import List, Union

import torch
from torch import nn

from .. import amp
from ..nn.modules import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, Flatten, Linear,
                          MaxPool2d, ReLU, ReLU6, conv_bn_act)
from ..ops import functional as F
from .resnet import STEM_PAD_C


def run_fused_sequential(seq: nn.Sequential, x, final_dtype=None):
    """Executes a torchvision-shaped Sequential, fusing ``Conv/Linear (+BN) + ReLU``."""
    mods = list(seq)
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        nxt2 = mods[i + 2] if i + 2 < n else None
        if isinstance(m, Conv2d) and isinstance(nxt, BatchNorm2d):
            relu = 6 if isinstance(nxt2, ReLU6) else isinstance(nxt2, ReLU)
            x = conv_bn_act(x, m, nxt, relu=relu)
            i += 3 if relu else 2
        elif isinstance(m, Conv2d) and isinstance(nxt, ReLU):
            x = m(x, relu=True)
            i += 2
        elif isinstance(m, Linear):
            relu = isinstance(nxt, ReLU)
            last = (i + (2 if relu else 1)) >= n
            x = m(x, relu=relu, out_dtype=final_dtype if last else None)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x


class _ImageNetCNN(nn.Module):
    stem_s2d = False     # True for 7x7 stride-2 pad-3 stems: the space-to-depth image layout (F.to_s2d)

    def prepare_input(self, x):
        dtype = amp.compute_dtype(torch.float32)
        pad = STEM_PAD_C if F._on_gpu(x) else 0
        if getattr(x, "_dpa_nhwc", False):      # already normalised (StreamPrefetcher)
            return x
        if self.stem_s2d and F.s2d_ok(x):
            return F.to_s2d(x, dtype)
        if x.dtype == torch.uint8:
            layout = "nhwc" if x.shape[-1] in (1, 3) else "nchw"
            return F.normalize_u8(x, dtype, pad, input_layout=layout)
        if getattr(x, "_dpa_nhwc", False):      # already normalised NHWC (StreamPrefetcher)
            return x
        return F.to_nhwc(x, dtype, pad)

    def forward(self, x):
        x = self.prepare_input(x)
        x = run_fused_sequential(self.features, x)
        x = self.avgpool(x)
        x = self.flatten(x)
        return run_fused_sequential(self.classifier, x, final_dtype=torch.float32)


class AlexNet(_ImageNetCNN):
    def __init__(self, num_classes: int = 1000, dropout: float = 0.5):
        super().__init__()
        first = Conv2d(3, 64, 11, stride=4, padding=2)
        first.needs_input_grad = False
        self.features = nn.Sequential(
            first, ReLU(inplace=True), MaxPool2d(3, 2),
            Conv2d(64, 192, 5, padding=2), ReLU(inplace=True), MaxPool2d(3, 2),
            Conv2d(192, 384, 3, padding=1), ReLU(inplace=True),
            Conv2d(384, 256, 3, padding=1), ReLU(inplace=True),
            Conv2d(256, 256, 3, padding=1), ReLU(inplace=True), MaxPool2d(3, 2),
        )
        self.avgpool = AdaptiveAvgPool2d((6, 6))
        self.flatten = Flatten()
        self.classifier = nn.Sequential(
            Dropout(dropout), Linear(256 * 6 * 6, 4096), ReLU(inplace=True),
            Dropout(dropout), Linear(4096, 4096), ReLU(inplace=True),
            Linear(4096, num_classes),
        )


VGG_CFGS = {
    "A": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "B": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "D": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "E": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
          512, 512, 512, 512, "M"],
}


def _vgg_features(cfg: List[Union[str, int]], batch_norm: bool) -> nn.Sequential:
    layers: List[nn.Module] = []
    cin = 3
    for v in cfg:
        if v == "M":
            layers.append(MaxPool2d(2, 2))
            continue
        conv = Conv2d(cin, int(v), 3, padding=1)
        if cin == 3:
            conv.needs_input_grad = False
        layers.append(conv)
        if batch_norm:
            layers.append(BatchNorm2d(int(v)))
        layers.append(ReLU(inplace=True))
        cin = int(v)
    return nn.Sequential(*layers)


class VGG(_ImageNetCNN):
    def __init__(self, cfg: str, batch_norm: bool = False, num_classes: int = 1000,
                 dropout: float = 0.5, init_weights: bool = True):
        super().__init__()
        self.features = _vgg_features(VGG_CFGS[cfg], batch_norm)
        self.avgpool = AdaptiveAvgPool2d((7, 7))
        self.flatten = Flatten()
        self.classifier = nn.Sequential(
            Linear(512 * 7 * 7, 4096), ReLU(inplace=True), Dropout(dropout),
            Linear(4096, 4096), ReLU(inplace=True), Dropout(dropout),
            Linear(4096, num_classes),
        )
        if init_weights:
            for m in self.modules():
                if isinstance(m, Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
                elif isinstance(m, Linear):
                    nn.init.normal_(m.weight, 0, 0.01)
                    nn.init.constant_(m.bias, 0)


def alexnet(**kw):
    return AlexNet(**kw)


def vgg11(**kw):
    return VGG("A", False, **kw)


def vgg11_bn(**kw):
    return VGG("A", True, **kw)


def vgg13(**kw):
    return VGG("B", False, **kw)


def vgg13_bn(**kw):
    return VGG("B", True, **kw)


def vgg16(**kw):
    return VGG("D", False, **kw)


def vgg16_bn(**kw):
    return VGG("D", True, **kw)


def vgg19(**kw):
    return VGG("E", False, **kw)


def vgg19_bn(**kw):
    return VGG("E", True, **kw)

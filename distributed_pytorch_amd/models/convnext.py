"""ConvNeXt tiny / small / base / large in NHWC (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``features.0.0.weight``, ``features.1.0.block.0.weight`` (7x7 depthwise),
``features.1.0.block.2.weight`` (LayerNorm), ``features.1.0.block.3.weight`` / ``block.5.weight`` (the two pointwise
Linear layers), ``features.1.0.layer_scale`` ``[C, 1, 1]``, ``classifier.2.weight``) so its checkpoints load.

ConvNeXt is written channels-last: torchvision permutes NCHW -> NHWC in front of every LayerNorm / Linear pair and back
again; here activations already are NHWC, so the permutes are index-keeping no-ops, ``LayerNorm2d`` and ``LayerNorm``
are the same row-wise HIP kernel, and the pointwise MLP is the MFMA linear kernel over ``N*H*W`` rows.  The depthwise
7x7 is the grouped implicit-GEMM kernel; GELU (erf form), the layer-scale multiply and row-mode stochastic depth are
elementwise HIP kernels.  Parameter counts: tiny 28,589,128; small 50,223,688; base 88,591,464; large 197,767,336.
"""
from __future__ import annotations

import torch
from torch import nn

from ..nn.modules import AdaptiveAvgPool2d, Conv2d, GELU, LayerNorm, Linear, StochasticDepth
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


class Permute(nn.Module):
    """Index placeholder for torchvision's ``Permute`` (activations never leave NHWC here)."""

    def __init__(self, dims):
        super().__init__()
        self.dims = list(dims)

    def forward(self, x):
        return x


class CNBlock(nn.Module):
    def __init__(self, dim: int, layer_scale: float, stochastic_depth_prob: float):
        super().__init__()
        self.block = nn.Sequential(
            Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True),
            Permute([0, 2, 3, 1]),
            LayerNorm(dim, eps=1e-6),
            Linear(dim, 4 * dim, bias=True),
            GELU(),
            Linear(4 * dim, dim, bias=True),
            Permute([0, 3, 1, 2]),
        )
        self.layer_scale = nn.Parameter(torch.ones(dim, 1, 1) * layer_scale)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")

    def forward(self, x):
        b = self.block
        h = b[5](b[4](b[3](b[2](b[0](x)))))
        n, c = h.shape[0], h.shape[-1]
        h = F.se_scale(h, self.layer_scale.reshape(1, c).float().expand(n, c))      # per-channel layer scale
        return F.add(self.stochastic_depth(h), x)


class ConvNeXt(_ImageNetCNN):
    def __init__(self, setting, stochastic_depth_prob: float = 0.0, layer_scale: float = 1e-6,
                 num_classes: int = 1000):
        super().__init__()
        first = Conv2d(3, setting[0][0], kernel_size=4, stride=4, padding=0, bias=True)
        first.needs_input_grad = False
        layers = [nn.Sequential(first, LayerNorm(setting[0][0], eps=1e-6))]
        total = sum(n for _, _, n in setting)
        done = 0
        for cin, cout, n in setting:
            stage = []
            for _ in range(n):
                stage.append(CNBlock(cin, layer_scale, stochastic_depth_prob * done / (total - 1.0)))
                done += 1
            layers.append(nn.Sequential(*stage))
            if cout is not None:
                layers.append(nn.Sequential(LayerNorm(cin, eps=1e-6), Conv2d(cin, cout, kernel_size=2, stride=2)))
        self.features = nn.Sequential(*layers)
        self.avgpool = AdaptiveAvgPool2d(1)
        last = setting[-1][1] if setting[-1][1] is not None else setting[-1][0]
        self.classifier = nn.Sequential(LayerNorm(last, eps=1e-6), nn.Flatten(1), Linear(last, num_classes))
        for m in self.modules():
            if isinstance(m, (Conv2d, Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.features(x)
        x = self.classifier[0](self.avgpool(x))
        return self.classifier[2](x.reshape(x.shape[0], -1), out_dtype=torch.float32)


def convnext_tiny(**kw):
    return ConvNeXt([(96, 192, 3), (192, 384, 3), (384, 768, 9), (768, None, 3)],
                    stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.1), **kw)


def convnext_small(**kw):
    return ConvNeXt([(96, 192, 3), (192, 384, 3), (384, 768, 27), (768, None, 3)],
                    stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.4), **kw)


def convnext_base(**kw):
    return ConvNeXt([(128, 256, 3), (256, 512, 3), (512, 1024, 27), (1024, None, 3)],
                    stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.5), **kw)


def convnext_large(**kw):
    return ConvNeXt([(192, 384, 3), (384, 768, 3), (768, 1536, 27), (1536, None, 3)],
                    stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.5), **kw)

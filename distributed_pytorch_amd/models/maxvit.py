"""MaxViT-T (torchvision zoo member reachable through the reference's ``models.__dict__[arch]()``,
``nd_imagenet.py:22-24,133-138``).  30,919,624 parameters.

Module names equal torchvision's ``MaxVit`` (``stem.0.0.weight``, ``blocks.0.layers.0.layers.MBconv.layers.conv_a.0.weight``,
``blocks.0.layers.0.layers.window_attention.attn_layer.1.to_qkv.weight``,
``blocks.3.layers.1.layers.grid_attention.mlp_layer.3.bias``, ``classifier.5.weight``) so its checkpoints load.

A MaxViT layer is MBConv (1x1 -> depthwise 3x3 -> squeeze-excitation -> 1x1, BatchNorm + GELU fused into the normalise
pass) followed by block attention over 7 x 7 windows and grid attention over a 7 x 7 strided grid.  In NHWC both
partitions are one gather kernel each way (``F.window_partition`` window / grid mode), the attention core is the fused
kernel of ``csrc/kernels/attention.hip`` with the learnable relative position bias added in registers (its gradient is
reduced over all windows inside the backward kernel).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Tuple

import torch
from torch import nn

from ..nn.modules import (GELU, AdaptiveAvgPool2d, AvgPool2d, BatchNorm2d, Conv2d, Dropout, Flatten, LayerNorm, Linear,
                          SqueezeExcitation, StochasticDepth, conv_bn_act)
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


def _conv_out(size: Tuple[int, int], k: int, s: int, p: int) -> Tuple[int, int]:
    return (size[0] - k + 2 * p) // s + 1, (size[1] - k + 2 * p) // s + 1


class _BN(BatchNorm2d):
    def __init__(self, c: int):
        super().__init__(c, eps=1e-3, momentum=0.01)


class _ConvNormAct(nn.Sequential):
    """torchvision ``Conv2dNormActivation(norm=BatchNorm2d(1e-3, 0.01), activation=GELU)``: indices 0 / 1 / 2."""

    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, groups: int = 1):
        super().__init__(Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False, groups=groups),
                         _BN(cout), GELU())

    def forward(self, x):
        return conv_bn_act(x, self[0], self[1], relu=False, act="gelu")


class MBConv(nn.Module):
    def __init__(self, cin: int, cout: int, expansion_ratio: float, squeeze_ratio: float, stride: int, p_sd: float):
        super().__init__()
        should_proj = stride != 1 or cin != cout
        if should_proj:
            proj: List[nn.Module] = [Conv2d(cin, cout, 1, stride=1, bias=True)]
            if stride == 2:
                proj = [AvgPool2d(3, stride=stride, padding=1)] + proj
            self.proj = nn.Sequential(*proj)
        else:
            self.proj = nn.Identity()
        mid, sqz = int(cout * expansion_ratio), int(cout * squeeze_ratio)
        self.stochastic_depth = StochasticDepth(p_sd, "row") if p_sd else nn.Identity()
        self.layers = nn.Sequential(OrderedDict(
            pre_norm=_BN(cin),
            conv_a=_ConvNormAct(cin, mid, 1),
            conv_b=_ConvNormAct(mid, mid, 3, stride=stride, groups=mid),
            squeeze_excitation=SqueezeExcitation(mid, sqz, activation="silu"),
            conv_c=Conv2d(mid, cout, 1, bias=True)))

    def forward(self, x):
        res = self.proj(x)
        return F.add(res, self.stochastic_depth(self.layers(x)))


def _relative_position_index(h: int, w: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += h - 1
    rel[:, :, 1] += w - 1
    rel[:, :, 0] *= 2 * w - 1
    return rel.sum(-1)


class RelativePositionalMultiHeadAttention(nn.Module):
    def __init__(self, feat_dim: int, head_dim: int, max_seq_len: int):
        super().__init__()
        if feat_dim % head_dim:
            raise ValueError("feat_dim: %d must be divisible by head_dim: %d" % (feat_dim, head_dim))
        self.n_heads, self.head_dim, self.max_seq_len = feat_dim // head_dim, head_dim, max_seq_len
        self.size = int(math.sqrt(max_seq_len))
        self.to_qkv = Linear(feat_dim, self.n_heads * head_dim * 3)
        self.scale_factor = feat_dim ** -0.5              # torchvision scales by the FEATURE dim, not the head dim
        self.merge = Linear(head_dim * self.n_heads, feat_dim)
        self.relative_position_bias_table = nn.Parameter(
            torch.empty(((2 * self.size - 1) * (2 * self.size - 1), self.n_heads), dtype=torch.float32))
        self.register_buffer("relative_position_index", _relative_position_index(self.size, self.size))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def _bias(self) -> torch.Tensor:
        n = self.max_seq_len
        return self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1)

    def forward(self, x):
        """x: [windows, tokens, C]"""
        a = F.attention(self.to_qkv(x), self.n_heads, self.scale_factor, self._bias())
        return self.merge(a)


class PartitionAttentionLayer(nn.Module):
    def __init__(self, in_channels: int, head_dim: int, partition_size: int, partition_type: str,
                 grid_size: Tuple[int, int], mlp_ratio: int, attention_dropout: float, mlp_dropout: float, p_sd: float):
        super().__init__()
        if partition_type not in ("grid", "window"):
            raise ValueError("partition_type must be either 'grid' or 'window'")
        self.partition_type, self.partition_size, self.grid_size = partition_type, partition_size, grid_size
        self.attn_layer = nn.Sequential(
            LayerNorm(in_channels, eps=1e-5),
            RelativePositionalMultiHeadAttention(in_channels, head_dim, partition_size ** 2),
            Dropout(attention_dropout))
        self.mlp_layer = nn.Sequential(
            LayerNorm(in_channels, eps=1e-5), Linear(in_channels, in_channels * mlp_ratio), GELU(),
            Linear(in_channels * mlp_ratio, in_channels), Dropout(mlp_dropout))
        self.stochastic_dropout = StochasticDepth(p_sd, "row")

    def _sd(self, y, B):
        """Stochastic depth drops whole IMAGES (torchvision keeps the batch axis in front of the windows)."""
        if not (self.training and self.stochastic_dropout.p):
            return y
        shp = y.shape
        return self.stochastic_dropout(y.reshape(B, shp[0] // B, shp[1], shp[2])).reshape(shp)

    def forward(self, x):
        B, H, W, Cc = x.shape
        p = self.partition_size
        if H % p or W % p:
            raise ValueError("the feature map must be divisible by the partition size")
        geom = F.window_geometry(B, H, W, p, p, grid=self.partition_type == "grid")
        t = F.window_partition(x, geom)                                   # [B * windows, p * p, C]
        a = self.attn_layer[1](self.attn_layer[0](t))
        if self.attn_layer[2].p:
            a = self.attn_layer[2](a)
        t = F.add(t, self._sd(a, B))
        m = self.mlp_layer
        y = m[3](F.activation(m[1](m[0](t)), "gelu"))
        if m[4].p:
            y = m[4](y)
        t = F.add(t, self._sd(y, B))
        return F.window_reverse(t, geom)


class MaxVitLayer(nn.Module):
    def __init__(self, cin, cout, squeeze_ratio, expansion_ratio, stride, head_dim, mlp_ratio, mlp_dropout,
                 attention_dropout, p_sd, partition_size, grid_size):
        super().__init__()
        self.layers = nn.Sequential(OrderedDict(
            MBconv=MBConv(cin, cout, expansion_ratio, squeeze_ratio, stride, p_sd),
            window_attention=PartitionAttentionLayer(cout, head_dim, partition_size, "window", grid_size, mlp_ratio,
                                                     attention_dropout, mlp_dropout, p_sd),
            grid_attention=PartitionAttentionLayer(cout, head_dim, partition_size, "grid", grid_size, mlp_ratio,
                                                   attention_dropout, mlp_dropout, p_sd)))

    def forward(self, x):
        return self.layers(x)


class MaxVitBlock(nn.Module):
    def __init__(self, cin, cout, squeeze_ratio, expansion_ratio, head_dim, mlp_ratio, mlp_dropout, attention_dropout,
                 partition_size, input_grid_size, n_layers, p_stochastic: List[float]):
        super().__init__()
        if len(p_stochastic) != n_layers:
            raise ValueError("p_stochastic must have length n_layers=%d" % n_layers)
        self.grid_size = _conv_out(input_grid_size, 3, 2, 1)
        self.layers = nn.ModuleList(
            MaxVitLayer(cin if i == 0 else cout, cout, squeeze_ratio, expansion_ratio, 2 if i == 0 else 1, head_dim,
                        mlp_ratio, mlp_dropout, attention_dropout, p, partition_size, self.grid_size)
            for i, p in enumerate(p_stochastic))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class MaxVit(_ImageNetCNN):
    def __init__(self, input_size: Tuple[int, int] = (224, 224), stem_channels: int = 64, partition_size: int = 7,
                 block_channels: List[int] = (64, 128, 256, 512), block_layers: List[int] = (2, 2, 5, 2),
                 head_dim: int = 32, stochastic_depth_prob: float = 0.2, squeeze_ratio: float = 0.25,
                 expansion_ratio: float = 4, mlp_ratio: int = 4, mlp_dropout: float = 0.0, attention_dropout: float = 0.0,
                 num_classes: int = 1000):
        super().__init__()
        size = tuple(input_size)
        for _ in range(len(block_channels) + 1):
            size = _conv_out(size, 3, 2, 1)
        # every attention stage must tile into partition_size windows (torchvision's _make_block_input_shapes check)
        chk = _conv_out(tuple(input_size), 3, 2, 1)
        for _ in block_channels:
            chk = _conv_out(chk, 3, 2, 1)
            if chk[0] % partition_size or chk[1] % partition_size:
                raise ValueError("Input size %s is not divisible by partition size %d at some stage"
                                 % (tuple(input_size), partition_size))
        first = Conv2d(3, stem_channels, 3, stride=2, padding=1, bias=False)
        first.needs_input_grad = False
        self.stem = nn.Sequential(
            nn.Sequential(first, _BN(stem_channels), GELU()),
            nn.Sequential(Conv2d(stem_channels, stem_channels, 3, stride=1, padding=1, bias=True)))
        grid = _conv_out(tuple(input_size), 3, 2, 1)
        self.partition_size = partition_size
        self.blocks = nn.ModuleList()
        cins = [stem_channels] + list(block_channels[:-1])
        p_sd = [stochastic_depth_prob * i / max(sum(block_layers) - 1, 1) for i in range(sum(block_layers))]
        # torchvision: np.linspace(0, stochastic_depth_prob, sum(block_layers))
        idx = 0
        for cin, cout, n in zip(cins, block_channels, block_layers):
            blk = MaxVitBlock(cin, cout, squeeze_ratio, expansion_ratio, head_dim, mlp_ratio, mlp_dropout,
                              attention_dropout, partition_size, grid, n, p_sd[idx:idx + n])
            self.blocks.append(blk)
            grid = blk.grid_size
            idx += n
        c = block_channels[-1]
        self.classifier = nn.Sequential(AdaptiveAvgPool2d(1), Flatten(), LayerNorm(c, eps=1e-5), Linear(c, c), nn.Tanh(),
                                        Linear(c, num_classes, bias=False))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, Linear):
                nn.init.normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        s0, s1 = self.stem[0], self.stem[1]
        x = conv_bn_act(x, s0[0], s0[1], relu=False, act="gelu")
        x = s1[0](x)
        for blk in self.blocks:
            x = blk(x)
        c = self.classifier
        x = c[1](c[0](x))
        x = c[3](c[2](x))
        x = torch.tanh(x)                      # [N, 512]: parameter-sized
        return c[5](x, out_dtype=torch.float32)


def maxvit_t(**kw):
    return MaxVit(**kw)

"""Vision Transformer B/16, B/32, L/16, L/32, H/14 (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Parameter and module names equal torchvision's ``VisionTransformer`` (``conv_proj.weight``, ``class_token``,
``encoder.pos_embedding``, ``encoder.layers.encoder_layer_0.self_attention.in_proj_weight``,
``encoder.layers.encoder_layer_0.mlp.3.bias``, ``encoder.ln.weight``, ``heads.head.weight``) so its checkpoints load.
Parameter counts at 224 x 224: vit_b_16 86,567,656; vit_b_32 88,224,232; vit_l_16 304,326,632; vit_l_32 306,535,400;
vit_h_14 632,045,800.

Data flow on the GPU: the patch embedding is the implicit-GEMM convolution (16 x 16 stride 16 -> ``[N, 14, 14, hidden]``,
which IS the token-major ``[N, 196, hidden]`` layout in NHWC, no permute), every Linear is the MFMA GEMM, LayerNorm the
row kernel of ``layernorm.hip``, and the attention core one fused kernel per direction on the packed ``[3][H][D]``
projection (``F.attention``, ``csrc/kernels/attention.hip``): no head transposes, no ``[S, S]`` score tensor in memory.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch import nn

from ..nn.modules import GELU, Conv2d, Dropout, LayerNorm, Linear
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


class _OutProj(Linear):
    """``nn.MultiheadAttention.out_proj`` (a Linear whose bias starts at zero)."""

    def reset_parameters(self) -> None:
        super().reset_parameters()
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class MultiheadSelfAttention(nn.Module):
    """``nn.MultiheadAttention(embed_dim, num_heads, batch_first=True)`` restricted to self-attention: parameters
    ``in_proj_weight [3E, E]``, ``in_proj_bias [3E]``, ``out_proj.{weight, bias}``."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        if dropout:
            raise ValueError("attention dropout is not used by the zoo")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, x):
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)
        a = F.attention(qkv, self.num_heads, self.head_dim ** -0.5)
        return self.out_proj(a)


class MLPBlock(nn.Sequential):
    """torchvision ``MLPBlock``: indices 0 Linear, 1 GELU, 2 Dropout, 3 Linear, 4 Dropout."""

    def __init__(self, in_dim: int, mlp_dim: int, dropout: float):
        super().__init__(Linear(in_dim, mlp_dim), GELU(), Dropout(dropout), Linear(mlp_dim, in_dim), Dropout(dropout))
        for m in self:
            if isinstance(m, Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x):
        x = F.activation(self[0](x), "gelu")
        if self[2].p:
            x = self[2](x)
        x = self[3](x)
        return self[4](x) if self[4].p else x


class EncoderBlock(nn.Module):
    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float, attention_dropout: float):
        super().__init__()
        self.num_heads = num_heads
        self.ln_1 = LayerNorm(hidden_dim, eps=1e-6)
        self.self_attention = MultiheadSelfAttention(hidden_dim, num_heads, dropout=attention_dropout)
        self.dropout = Dropout(dropout)
        self.ln_2 = LayerNorm(hidden_dim, eps=1e-6)
        self.mlp = MLPBlock(hidden_dim, mlp_dim, dropout)

    def forward(self, x):
        y = self.self_attention(self.ln_1(x))
        if self.dropout.p:
            y = self.dropout(y)
        x = F.add(x, y)
        return F.add(x, self.mlp(self.ln_2(x)))


class Encoder(nn.Module):
    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float,
                 attention_dropout: float):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = Dropout(dropout)
        self.layers = nn.Sequential(OrderedDict(
            ("encoder_layer_%d" % i, EncoderBlock(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout))
            for i in range(num_layers)))
        self.ln = LayerNorm(hidden_dim, eps=1e-6)

    def forward(self, x):
        x = x + self.pos_embedding.to(x.dtype)
        if self.dropout.p:
            x = self.dropout(x)
        return self.ln(self.layers(x))


class VisionTransformer(_ImageNetCNN):
    def __init__(self, image_size: int, patch_size: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int,
                 dropout: float = 0.0, attention_dropout: float = 0.0, num_classes: int = 1000):
        super().__init__()
        if image_size % patch_size:
            raise ValueError("Input shape indivisible by patch size!")
        self.image_size, self.patch_size, self.hidden_dim = image_size, patch_size, hidden_dim
        self.conv_proj = Conv2d(3, hidden_dim, patch_size, stride=patch_size)
        self.conv_proj.needs_input_grad = False
        seq_length = (image_size // patch_size) ** 2 + 1
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        self.encoder = Encoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout)
        self.seq_length = seq_length
        self.heads = nn.Sequential(OrderedDict(head=Linear(hidden_dim, num_classes)))
        fan_in = 3 * patch_size * patch_size
        nn.init.trunc_normal_(self.conv_proj.weight, std=math.sqrt(1 / fan_in))
        nn.init.zeros_(self.conv_proj.bias)
        nn.init.zeros_(self.heads.head.weight)
        nn.init.zeros_(self.heads.head.bias)

    def forward(self, x):
        x = self.prepare_input(x)
        if x.shape[1] != self.image_size or x.shape[2] != self.image_size:
            raise ValueError("Wrong image size: expected %d x %d" % (self.image_size, self.image_size))
        x = self.conv_proj(x)                                   # [N, h, w, hidden] == token-major [N, h*w, hidden]
        n = x.shape[0]
        x = x.reshape(n, -1, self.hidden_dim)
        x = torch.cat([self.class_token.to(x.dtype).expand(n, -1, -1), x], dim=1)
        x = self.encoder(x)
        return self.heads.head(x[:, 0].contiguous(), out_dtype=torch.float32)


def _vit(patch_size, num_layers, num_heads, hidden_dim, mlp_dim, **kw):
    kw.setdefault("image_size", 224)
    return VisionTransformer(patch_size=patch_size, num_layers=num_layers, num_heads=num_heads, hidden_dim=hidden_dim,
                             mlp_dim=mlp_dim, **kw)


def vit_b_16(**kw):
    return _vit(16, 12, 12, 768, 3072, **kw)


def vit_b_32(**kw):
    return _vit(32, 12, 12, 768, 3072, **kw)


def vit_l_16(**kw):
    return _vit(16, 24, 16, 1024, 4096, **kw)


def vit_l_32(**kw):
    return _vit(32, 24, 16, 1024, 4096, **kw)


def vit_h_14(**kw):
    return _vit(14, 32, 16, 1280, 5120, **kw)

"""Swin Transformer T / S / B and Swin Transformer V2 T / S / B (torchvision zoo members reachable through the
reference's ``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names equal torchvision's ``SwinTransformer`` (``features.0.0.weight``, ``features.1.0.attn.qkv.weight``,
``features.1.0.attn.relative_position_bias_table``, ``features.2.reduction.weight``, ``norm.weight``, ``head.weight``;
V2 adds ``attn.logit_scale``, ``attn.cpb_mlp.{0,2}.weight``) so its checkpoints load.  Parameter counts: swin_t
28,288,354; swin_s 49,606,258; swin_b 87,768,224; swin_v2_t 28,351,570; swin_v2_s 49,737,442; swin_v2_b 87,930,848.

The framework is NHWC, which is Swin's native ``[B, H, W, C]`` layout.  Per block on the GPU:

* cyclic shift + zero padding + window partition are ONE gather kernel (``F.window_partition``), its inverse another;
* ``qkv`` / ``proj`` / MLP are the MFMA GEMM, LayerNorm the row kernel;
* the attention core is the fused kernel of ``csrc/kernels/attention.hip`` on the packed projection, with the relative
  position bias (learnable, gradient reduced over all windows inside the kernel) and the shifted-window mask added to
  the logits in registers;
* V2's cosine attention is a pre-pass on the packed projection (``F.qk_normalize``: q <- logit_scale * q / |q|,
  k <- k / |k|) followed by the same kernel with scale 1.

Only parameter-sized tensors (bias tables, the continuous-position-bias MLP, the mask) go through plain torch ops.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
from torch import nn

from ..nn.modules import GELU, Conv2d, Dropout, LayerNorm, Linear, ReLU, StochasticDepth
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN

_MERGE_ORDER = (0, 2, 1, 3)      # our 2x2 window order (i, j) -> torchvision's concat order x0, x1, x2, x3 = (j, i)


def _trunc_linear(m: Linear) -> Linear:
    nn.init.trunc_normal_(m.weight, std=0.02)
    if m.bias is not None:
        nn.init.zeros_(m.bias)
    return m


class PatchMerging(nn.Module):
    """2 x 2 neighbourhood -> channels, LayerNorm(4C), Linear(4C -> 2C).  The neighbourhood gather is the window-partition
    kernel with 2 x 2 windows; its token order differs from torchvision's concat order by a fixed block permutation,
    which is applied to the (tiny) norm / reduction parameters instead of the activation."""

    v2 = False

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.reduction = _trunc_linear(Linear(4 * dim, 2 * dim, bias=False))
        self.norm = LayerNorm(2 * dim if self.v2 else 4 * dim, eps=1e-5)

    def _perm(self, device) -> torch.Tensor:
        return (torch.tensor(_MERGE_ORDER, device=device).reshape(4, 1) * self.dim
                + torch.arange(self.dim, device=device).reshape(1, -1)).reshape(-1)

    def forward(self, x):
        B, H, W, Cc = x.shape
        geom = F.window_geometry(B, H, W, 2, 2)
        t = F.window_partition(x, geom)                              # zero pads odd extents, like torchvision
        t = t.reshape(B, geom[5], geom[6], 4 * Cc)
        idx = self._perm(x.device)
        w = self.reduction.weight.index_select(1, idx)
        if self.v2:
            return self.norm(F.linear(t, w, None))
        t = F.layer_norm(t, self.norm.weight.index_select(0, idx), self.norm.bias.index_select(0, idx), self.norm.eps)
        return F.linear(t, w, None)


class PatchMergingV2(PatchMerging):
    v2 = True


_mask_cache: Dict[tuple, torch.Tensor] = {}


def _shift_mask(Hp: int, Wp: int, ws: List[int], shift: List[int], device) -> torch.Tensor:
    """torchvision's shifted-window mask, ``[nW, N, N]`` of 0 / -100."""
    key = (Hp, Wp, tuple(ws), tuple(shift), str(device))
    m = _mask_cache.get(key)
    if m is None:
        a = torch.zeros((Hp, Wp))
        hs = ((0, -ws[0]), (-ws[0], -shift[0]), (-shift[0], None))
        wsl = ((0, -ws[1]), (-ws[1], -shift[1]), (-shift[1], None))
        count = 0
        for h in hs:
            for w in wsl:
                a[h[0]:h[1], w[0]:w[1]] = count
                count += 1
        a = a.view(Hp // ws[0], ws[0], Wp // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(-1, ws[0] * ws[1])
        d = a.unsqueeze(1) - a.unsqueeze(2)
        m = torch.where(d != 0, torch.tensor(-100.0), torch.tensor(0.0)).to(device)
        _mask_cache[key] = m
    return m


class ShiftedWindowAttention(nn.Module):
    v2 = False

    def __init__(self, dim: int, window_size: List[int], shift_size: List[int], num_heads: int, qkv_bias: bool = True,
                 proj_bias: bool = True):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads, self.dim = list(window_size), list(shift_size), num_heads, dim
        self.qkv = _trunc_linear(Linear(dim, dim * 3, bias=qkv_bias))
        self.proj = _trunc_linear(Linear(dim, dim, bias=proj_bias))
        self._define_bias()
        ws = self.window_size
        coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 0] *= 2 * ws[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1).flatten())

    def _define_bias(self):
        ws = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1), self.num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def _bias(self) -> torch.Tensor:
        N = self.window_size[0] * self.window_size[1]
        return self.relative_position_bias_table[self.relative_position_index].view(N, N, -1).permute(2, 0, 1)

    def forward(self, x):
        B, H, W, Cc = x.shape
        ws = self.window_size
        Hp, Wp = -(-H // ws[0]) * ws[0], -(-W // ws[1]) * ws[1]
        shift = [0 if ws[0] >= Hp else self.shift_size[0], 0 if ws[1] >= Wp else self.shift_size[1]]
        geom = F.window_geometry(B, H, W, ws[0], ws[1], shift[0], shift[1])
        win = F.window_partition(x, geom)
        qkv_bias = self.qkv.bias
        head_dim = Cc // self.num_heads
        if self.v2 and qkv_bias is not None:
            keep = torch.ones_like(qkv_bias)
            keep[Cc:2 * Cc] = 0.0                                      # the key projection has no bias in V2
            qkv_bias = qkv_bias * keep
        qkv = F.linear(win, self.qkv.weight, qkv_bias)
        scale = head_dim ** -0.5
        if self.v2:
            ls = torch.clamp(self.logit_scale, max=math.log(100.0)).exp().reshape(-1)
            qkv = F.qk_normalize(qkv, ls, self.num_heads)
            scale = 1.0
        mask = _shift_mask(Hp, Wp, ws, shift, x.device) if sum(shift) > 0 else None
        a = F.attention(qkv, self.num_heads, scale, self._bias(), mask)
        return F.window_reverse(self.proj(a), geom)


class ShiftedWindowAttentionV2(ShiftedWindowAttention):
    v2 = True

    def __init__(self, dim, window_size, shift_size, num_heads, qkv_bias=True, proj_bias=True):
        super().__init__(dim, window_size, shift_size, num_heads, qkv_bias, proj_bias)
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(Linear(2, 512, bias=True), ReLU(inplace=True), Linear(512, num_heads, bias=False))
        for m in self.cpb_mlp:
            if isinstance(m, Linear):
                _trunc_linear(m)
        if qkv_bias:
            with torch.no_grad():
                self.qkv.bias[dim:2 * dim].zero_()

    def _define_bias(self):
        ws = self.window_size
        ch = torch.arange(-(ws[0] - 1), ws[0], dtype=torch.float32)
        cw = torch.arange(-(ws[1] - 1), ws[1], dtype=torch.float32)
        t = torch.stack(torch.meshgrid([ch, cw], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
        t[:, :, :, 0] /= ws[0] - 1
        t[:, :, :, 1] /= ws[1] - 1
        t *= 8
        t = torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / 3.0
        self.register_buffer("relative_coords_table", t)

    def _bias(self) -> torch.Tensor:
        N = self.window_size[0] * self.window_size[1]
        # a 2 -> 512 -> heads MLP over (2w - 1)^2 = 225 table rows: parameter-sized, plain torch ops on any device
        t = self.relative_coords_table.float()
        h = torch.relu(torch.nn.functional.linear(t, self.cpb_mlp[0].weight.float(), self.cpb_mlp[0].bias.float()))
        tab = torch.nn.functional.linear(h, self.cpb_mlp[2].weight.float()).view(-1, self.num_heads)
        bias = tab[self.relative_position_index].view(N, N, -1).permute(2, 0, 1)
        return 16 * torch.sigmoid(bias)


class _MLP(nn.Sequential):
    """torchvision ``MLP(dim, [hidden, dim], activation_layer=GELU)``: indices 0 Linear, 1 GELU, 2 Dropout, 3 Linear,
    4 Dropout."""

    def __init__(self, dim: int, hidden: int, dropout: float):
        super().__init__(Linear(dim, hidden), GELU(), Dropout(dropout), Linear(hidden, dim), Dropout(dropout))
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


class SwinTransformerBlock(nn.Module):
    attn_layer = ShiftedWindowAttention
    v2 = False

    def __init__(self, dim: int, num_heads: int, window_size: List[int], shift_size: List[int], mlp_ratio: float = 4.0,
                 dropout: float = 0.0, stochastic_depth_prob: float = 0.0):
        super().__init__()
        self.norm1 = LayerNorm(dim, eps=1e-5)
        self.attn = self.attn_layer(dim, window_size, shift_size, num_heads)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")
        self.norm2 = LayerNorm(dim, eps=1e-5)
        self.mlp = _MLP(dim, int(dim * mlp_ratio), dropout)

    def forward(self, x):
        if self.v2:
            x = F.add(x, self.stochastic_depth(self.norm1(self.attn(x))))
            return F.add(x, self.stochastic_depth(self.norm2(self.mlp(x))))
        x = F.add(x, self.stochastic_depth(self.attn(self.norm1(x))))
        return F.add(x, self.stochastic_depth(self.mlp(self.norm2(x))))


class SwinTransformerBlockV2(SwinTransformerBlock):
    attn_layer = ShiftedWindowAttentionV2
    v2 = True


class _PatchEmbed(nn.Sequential):
    """features.0: Conv2d(3, C, 4, 4), Permute (a no-op in NHWC, kept for torchvision's indices), LayerNorm(C)."""

    def __init__(self, embed_dim: int, patch: List[int]):
        conv = Conv2d(3, embed_dim, (patch[0], patch[1]), stride=(patch[0], patch[1]))
        conv.needs_input_grad = False
        super().__init__(conv, nn.Identity(), LayerNorm(embed_dim, eps=1e-5))


class SwinTransformer(_ImageNetCNN):
    def __init__(self, patch_size: List[int], embed_dim: int, depths: List[int], num_heads: List[int],
                 window_size: List[int], mlp_ratio: float = 4.0, dropout: float = 0.0,
                 stochastic_depth_prob: float = 0.1, num_classes: int = 1000, v2: bool = False):
        super().__init__()
        block = SwinTransformerBlockV2 if v2 else SwinTransformerBlock
        merge = PatchMergingV2 if v2 else PatchMerging
        layers: List[nn.Module] = [_PatchEmbed(embed_dim, patch_size)]
        total = sum(depths)
        bid = 0
        for i_stage, depth in enumerate(depths):
            dim = embed_dim * 2 ** i_stage
            stage = []
            for i_layer in range(depth):
                sd = stochastic_depth_prob * float(bid) / (total - 1)
                stage.append(block(dim, num_heads[i_stage], window_size,
                                   [0 if i_layer % 2 == 0 else w // 2 for w in window_size], mlp_ratio, dropout, sd))
                bid += 1
            layers.append(nn.Sequential(*stage))
            if i_stage < len(depths) - 1:
                layers.append(merge(dim))
        self.features = nn.Sequential(*layers)
        nf = embed_dim * 2 ** (len(depths) - 1)
        self.norm = LayerNorm(nf, eps=1e-5)
        self.head = _trunc_linear(Linear(nf, num_classes))

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.features(x)
        x = self.norm(x)
        x = F.adaptive_avg_pool2d(x, (1, 1))
        return self.head(x.reshape(x.shape[0], -1), out_dtype=torch.float32)


def _swin(embed_dim, depths, num_heads, window, sd, v2=False, **kw):
    kw.setdefault("stochastic_depth_prob", sd)
    return SwinTransformer([4, 4], embed_dim, depths, num_heads, window, v2=v2, **kw)


def swin_t(**kw):
    return _swin(96, [2, 2, 6, 2], [3, 6, 12, 24], [7, 7], 0.2, **kw)


def swin_s(**kw):
    return _swin(96, [2, 2, 18, 2], [3, 6, 12, 24], [7, 7], 0.3, **kw)


def swin_b(**kw):
    return _swin(128, [2, 2, 18, 2], [4, 8, 16, 32], [7, 7], 0.5, **kw)


def swin_v2_t(**kw):
    return _swin(96, [2, 2, 6, 2], [3, 6, 12, 24], [8, 8], 0.2, v2=True, **kw)


def swin_v2_s(**kw):
    return _swin(96, [2, 2, 18, 2], [3, 6, 12, 24], [8, 8], 0.3, v2=True, **kw)


def swin_v2_b(**kw):
    return _swin(128, [2, 2, 18, 2], [4, 8, 16, 32], [8, 8], 0.5, v2=True, **kw)

"""Functional op layer.  Activations are **NHWC** tensors ``[N, H, W, C]``.

Dispatch rule (see ``backend.py``): CPU tensors -> pure PyTorch reference math
(the oracle for every test and the BASELINE "config #1" plumbing backend);
GPU tensors -> hand-written HIP kernels in ``csrc/kernels`` through
``ops/hip_ops.py``.  The op set is the implicit native surface of the reference
scripts: what cuDNN / cuBLAS / ATen / Apex ran for
``model(images)``, ``criterion``, ``loss.backward()`` and ``optimizer.step()``
(reference ``nd_imagenet.py:291-303``, ``imagenet_ddp_apex.py:339-383``).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as TF


def _on_gpu(t: torch.Tensor) -> bool:
    """GPU tensors ALWAYS run the HIP kernels; the PyTorch ops below are the CPU reference backend.  (The stock
    PyTorch-ROCm comparison point lives outside the framework: ``bench/torch_baseline.py``.)"""
    return t.is_cuda


def _hip():
    from . import hip_ops

    return hip_ops


def _pair(v) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


# ----------------------------------------------------------------------------- conv
def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           stride=1, padding=0, *, relu: bool = False, w_lp: Optional[torch.Tensor] = None,
           want_stats: bool = False, needs_input_grad: bool = True, skip: bool = False, groups: int = 1):
    """NHWC convolution.  ``weight`` is logical ``[K, C/groups, R, S]`` (memory KRSC).

    ``w_lp``: optional low-precision compute copy of ``weight`` (AMP O1/O2);
    gradients still flow to ``weight``.  ``want_stats`` additionally returns the
    per-channel ``(sum, sumsq)`` of the output computed in the conv epilogue
    (feeds the fused BatchNorm).  ``needs_input_grad=False`` skips dgrad (stem).
    ``skip=True`` returns ``(result, x_alias)``: feed ``x_alias`` to the residual branch and its
    gradient is added inside this convolution's dgrad epilogue instead of a separate accumulate pass.
    Contract: use ``x_alias`` ONCE -- as the ``residual`` of the block's last BatchNorm, or as the input of the
    block's downsample branch.  (A residual BN + ReLU returns the alias's gradient unmasked together with its ReLU
    bit mask, which only this convolution's epilogue knows how to apply.)
    ``groups > 1``: grouped / depthwise convolution, every pass one launch with the group on ``grid.y``.
    """
    stride, padding = _pair(stride), _pair(padding)
    if getattr(x, "_dpa_s2d", False):       # space-to-depth image (to_s2d): the 7x7 stride-2 stem as a 4x4 convolution
        if groups != 1 or skip or not _on_gpu(x):
            raise ValueError("space-to-depth input is only valid for a dense GPU stem convolution")
        return _hip().stem_s2d_conv(x, weight, bias, stride, padding, relu, want_stats)
    if groups > 1:
        if _on_gpu(x):
            out = _hip().group_conv2d(x, weight, bias, stride, padding, groups, relu, want_stats, needs_input_grad)
        else:
            w = weight if w_lp is None else _StraightThroughCast.apply(weight, x.dtype)
            if w.dtype != x.dtype:
                w = w.to(x.dtype)
            b = None if bias is None else bias.to(x.dtype)
            y = TF.conv2d(x.permute(0, 3, 1, 2), w, b, stride, padding, 1, groups).permute(0, 2, 3, 1)
            y = (TF.relu(y) if relu else y).contiguous()
            out = (y, None) if want_stats else y
        return (out, x) if skip else out
    if _on_gpu(x):
        return _hip().conv2d(x, weight, bias, stride, padding, relu, w_lp, want_stats,
                             needs_input_grad, skip)
    w = weight if w_lp is None else _StraightThroughCast.apply(weight, x.dtype)
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
    b = None if bias is None else bias.to(x.dtype)
    y = TF.conv2d(x.permute(0, 3, 1, 2), w, b, stride, padding).permute(0, 2, 3, 1)
    if relu:
        y = TF.relu(y)
    y = y.contiguous()
    out = (y, None) if want_stats else y
    return (out, x) if skip else out


class _StraightThroughCast(torch.autograd.Function):
    """fp32 master -> low precision for compute; gradient returns in fp32."""

    @staticmethod
    def forward(ctx, w, dtype):
        return w.to(dtype)

    @staticmethod
    def backward(ctx, g):
        return g.float(), None


# ----------------------------------------------------------------------------- batch norm
def batch_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
               running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
               num_batches_tracked: Optional[torch.Tensor], training: bool,
               momentum: float = 0.1, eps: float = 1e-5, *, relu: bool = False,
               residual: Optional[torch.Tensor] = None, stats=None, group=None, defer: bool = False) -> torch.Tensor:
    """Fused ``BN [+ residual] [+ ReLU]`` over NHWC (N-K4/K5/K7; SyncBN when
    ``group`` is a process group: N-K6).  Statistics are always fp32.
    ``running_var`` receives the unbiased variance, as in torch / cuDNN.
    ``relu=6`` applies ReLU6 (MobileNet) inside the same pass; ``relu="silu"`` / ``"hardswish"`` / ... a smooth
    activation (EfficientNet, MobileNetV3): the GPU training path applies it inside the normalise pass and folds its
    slope into the BN backward kernels, so no separate activation pass runs in either direction.
    ``defer=True`` (activation-free BN of a downsample branch): on the GPU in training mode nothing is written --
    the result aliases ``x`` and MUST only be used as the ``residual`` of another ``batch_norm``, which applies this
    BN's scale / shift on the fly (one activation-sized write and read less).  Ignored elsewhere.
    """
    if _on_gpu(x):
        return _hip().batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked,
                                 training, momentum, eps, relu, residual, stats, group, defer)
    if group is not None and training:
        from ..parallel.sync_batchnorm import sync_batch_norm_reference

        y = sync_batch_norm_reference(x, weight, bias, running_mean, running_var,
                                      num_batches_tracked, momentum, eps, group)
    else:
        xc = x.permute(0, 3, 1, 2)
        if training and num_batches_tracked is not None:
            num_batches_tracked.add_(1)
        y = TF.batch_norm(xc.float(), running_mean, running_var, weight.float(), bias.float(),
                          training, momentum, eps).to(x.dtype).permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual
    if isinstance(relu, str):
        y = _CPU_ACTS[relu](y)
    elif relu:
        y = TF.relu6(y) if (relu is not True and relu == 6) else TF.relu(y)
    return y.contiguous()


def bn_relu_max_pool2d(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps,
                       kernel_size, stride=None, padding=0, ceil_mode: bool = False, *, stats=None, group=None):
    """``max_pool2d(relu(batch_norm(x)))`` -- the stem of ResNet / DenseNet / GoogLeNet.  On the GPU in training mode
    (no SyncBN) this is the fused stem path: the normalised full-resolution tensor is never materialised."""
    k, s, p = _pair(kernel_size), _pair(stride if stride is not None else kernel_size), _pair(padding)
    if _on_gpu(x) and training and group is None and _hip().can_fuse_bn_relu_maxpool(x):
        return _hip().bn_relu_maxpool(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps,
                                      stats, k, s, p, ceil_mode)
    y = batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps,
                   relu=True, stats=stats, group=group)
    return max_pool2d(y, k, s, p, ceil_mode)


# ----------------------------------------------------------------------------- pooling
def max_pool2d(x: torch.Tensor, kernel_size, stride=None, padding=0, ceil_mode: bool = False) -> torch.Tensor:
    k, s, p = _pair(kernel_size), _pair(stride if stride is not None else kernel_size), _pair(padding)
    if _on_gpu(x):
        return _hip().max_pool2d(x, k, s, p, ceil_mode)
    return TF.max_pool2d(x.permute(0, 3, 1, 2), k, s, p, ceil_mode=ceil_mode).permute(0, 2, 3, 1).contiguous()


def avg_pool2d(x: torch.Tensor, kernel_size, stride=None, padding=0) -> torch.Tensor:
    """Average pooling (floor mode, padding counted in the divisor like torch's default).  Non-overlapping windows
    that tile the input (DenseNet transitions: 2x2 stride 2) run on the adaptive average-pool kernel; everything
    else (Inception's 3x3 stride-1 pad-1 branch pool, its 5x5 stride-3 auxiliary pool) on the fixed-window kernel."""
    k, s, p = _pair(kernel_size), _pair(stride if stride is not None else kernel_size), _pair(padding)
    if k == s and p == (0, 0) and x.shape[1] % k[0] == 0 and x.shape[2] % k[1] == 0:
        return adaptive_avg_pool2d(x, (x.shape[1] // k[0], x.shape[2] // k[1]))
    if _on_gpu(x):
        return _hip().avg_pool2d_window(x, k, s, p)
    return TF.avg_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).contiguous()


def cat_channels(xs) -> torch.Tensor:
    """Concatenate NHWC tensors along the channel axis."""
    xs = list(xs)
    if len(xs) == 1:
        return xs[0]
    if _on_gpu(xs[0]):
        return _hip().cat_channels(xs)
    return torch.cat(xs, dim=-1)


def attention(qkv: torch.Tensor, num_heads: int, scale, rel_bias: Optional[torch.Tensor] = None,
              mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Multi-head self-attention on the packed projection: ``qkv`` is ``[B, S, 3 * H * D]`` laid out ``[3][H][D]`` per
    token (what ``nn.MultiheadAttention``'s in-projection, Swin's and MaxViT's ``qkv`` Linear produce), the result
    ``[B, S, H * D]`` feeds the output projection.

        softmax(scale * q k^T + rel_bias[h] + mask[b % nW]) v

    ``scale``: a float or a per-head tensor ``[H]`` (not differentiated here).  ``rel_bias``: optional learnable
    ``[H, S, S]`` (Swin / MaxViT relative position bias; gets a gradient).  ``mask``: optional constant ``[nW, S, S]``
    (Swin's shifted-window mask), batch entry ``b`` uses window ``b % nW``.  On the GPU this is one fused MFMA kernel
    per direction (``csrc/kernels/attention.hip``): the scores never reach memory."""
    if _on_gpu(qkv):
        return _hip().attention(qkv, num_heads, scale, rel_bias, mask)
    B, S, W3 = qkv.shape
    H = num_heads
    D = W3 // (3 * H)
    q, k, v = qkv.reshape(B, S, 3, H, D).permute(2, 0, 3, 1, 4).float().unbind(0)      # [B, H, S, D]
    sc = scale.float().reshape(1, H, 1, 1) if isinstance(scale, torch.Tensor) else float(scale)
    att = (q @ k.transpose(-1, -2)) * sc
    if rel_bias is not None:
        att = att + rel_bias.float().reshape(1, H, S, S)
    if mask is not None:
        nW = mask.shape[0]
        att = (att.reshape(B // nW, nW, H, S, S) + mask.float().reshape(1, nW, 1, S, S)).reshape(B, H, S, S)
    out = att.softmax(-1) @ v
    return out.transpose(1, 2).reshape(B, S, H * D).to(qkv.dtype)


def window_geometry(B: int, H: int, W: int, wh: int, ww: int, sh: int = 0, sw: int = 0, grid: bool = False):
    """Geometry tuple of ``window_partition``: ``(B, H, W, wh, ww, nh, nw, sh, sw, grid)``.  Window mode: windows of
    ``wh x ww`` tokens over the image zero-padded to a multiple of the window, after a cyclic shift by ``(sh, sw)``
    (Swin).  Grid mode (MaxViT): ``wh x ww`` is the GRID size, token ``(i, j)`` of window ``(gh, gw)`` is pixel
    ``(i * nh + gh, j * nw + gw)``; the image must divide evenly."""
    if grid:
        if H % wh or W % ww:
            raise ValueError("grid partition needs H, W divisible by the grid size")
        return (B, H, W, wh, ww, H // wh, W // ww, 0, 0, True)
    return (B, H, W, wh, ww, -(-H // wh), -(-W // ww), sh, sw, False)


def window_partition(x: torch.Tensor, geom) -> torch.Tensor:
    """``[B, H, W, C] -> [B * nh * nw, wh * ww, C]`` (shift + pad + partition in one gather on the GPU)."""
    if _on_gpu(x):
        return _hip().window_partition(x, geom, False)
    B, H, W, wh, ww, nh, nw, sh, sw, grid = geom
    C = x.shape[-1]
    if grid:
        return x.reshape(B, wh, nh, ww, nw, C).permute(0, 2, 4, 1, 3, 5).reshape(B * nh * nw, wh * ww, C)
    x = TF.pad(x, (0, 0, 0, nw * ww - W, 0, nh * wh - H))
    if sh or sw:
        x = torch.roll(x, shifts=(-sh, -sw), dims=(1, 2))
    return x.reshape(B, nh, wh, nw, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nh * nw, wh * ww, C)


def window_reverse(win: torch.Tensor, geom) -> torch.Tensor:
    """Inverse of ``window_partition`` (the zero padding is dropped)."""
    if _on_gpu(win):
        return _hip().window_partition(win, geom, True)
    B, H, W, wh, ww, nh, nw, sh, sw, grid = geom
    C = win.shape[-1]
    if grid:
        return win.reshape(B, nh, nw, wh, ww, C).permute(0, 3, 1, 4, 2, 5).reshape(B, H, W, C)
    x = win.reshape(B, nh, nw, wh, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nh * wh, nw * ww, C)
    if sh or sw:
        x = torch.roll(x, shifts=(sh, sw), dims=(1, 2))
    return x[:, :H, :W, :].contiguous()


def qk_normalize(qkv: torch.Tensor, mult: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Swin V2 cosine attention pre-pass on the packed ``[.., 3 * H * D]`` projection: ``q <- mult[h] * q / |q|``,
    ``k <- k / |k|`` (``F.normalize`` semantics, eps 1e-12), ``v`` unchanged.  Differentiable in ``qkv`` and ``mult``."""
    if _on_gpu(qkv):
        return _hip().qk_normalize(qkv, mult, num_heads)
    shp = qkv.shape
    H = num_heads
    D = shp[-1] // (3 * H)
    x = qkv.reshape(-1, 3, H, D).float()
    q = TF.normalize(x[:, 0], dim=-1) * mult.float().reshape(1, H, 1)
    k = TF.normalize(x[:, 1], dim=-1)
    return torch.stack([q, k, x[:, 2]], dim=1).reshape(shp).to(qkv.dtype)


def shuffle_pair(a: torch.Tensor, b: torch.Tensor, bf: int):
    """ShuffleNetV2's ``channel_shuffle(cat(a, b), groups=2)`` followed by the ``chunk(2)`` of the next block, as one op:
    returns the two halves.  ``a`` / ``b`` hold ``bf`` real channels (+ zero padding up to their width on the GPU)."""
    if _on_gpu(a):
        return _hip().shuffle_pair(a, b, bf)
    out = torch.stack([a[..., :bf], b[..., :bf]], dim=-1).flatten(-2)        # out[2i] = a[i], out[2i+1] = b[i]
    lo, ro = out[..., :bf], out[..., bf:]
    pad = a.shape[-1] - bf
    if pad:
        lo, ro = TF.pad(lo, (0, pad)), TF.pad(ro, (0, pad))
    return lo.contiguous(), ro.contiguous()


def adaptive_avg_pool2d(x: torch.Tensor, output_size) -> torch.Tensor:
    o = _pair(output_size)
    if _on_gpu(x):
        return _hip().adaptive_avg_pool2d(x, o)
    return TF.adaptive_avg_pool2d(x.permute(0, 3, 1, 2), o).permute(0, 2, 3, 1).contiguous()


def relu(x: torch.Tensor) -> torch.Tensor:
    if _on_gpu(x):
        return _hip().relu(x)
    return TF.relu(x)


def add_relu(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Residual ``relu(x + y)`` (N-K17) when it is not fused into BatchNorm."""
    if _on_gpu(x):
        return _hip().add_relu(x, y)
    return TF.relu(x + y)


def add(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Residual ``x + y`` without activation (EfficientNet / MobileNetV3 under stochastic depth)."""
    if _on_gpu(x):
        return _hip().add(x, y)
    return x + y


_CPU_ACTS = {"sigmoid": torch.sigmoid, "silu": TF.silu, "hardsigmoid": TF.hardsigmoid, "hardswish": TF.hardswish,
             "gelu": TF.gelu}


def activation(x: torch.Tensor, kind: str) -> torch.Tensor:
    """``sigmoid`` / ``silu`` / ``hardsigmoid`` / ``hardswish`` / ``gelu`` (erf form): EfficientNet, MobileNetV3,
    squeeze-excitation, ConvNeXt."""
    if kind not in _CPU_ACTS:
        raise ValueError("unknown activation %r" % (kind,))
    if _on_gpu(x):
        return _hip().activation(x, kind)
    return _CPU_ACTS[kind](x)


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """LayerNorm over the last (channel) axis -- in NHWC torchvision's ``LayerNorm2d`` and ``nn.LayerNorm`` coincide."""
    if _on_gpu(x):
        return _hip().layer_norm(x, weight, bias, eps)
    return TF.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).to(x.dtype)


def se_scale(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """``x[n, h, w, c] * s[n, c]``: the squeeze-excitation gate.  ``s`` is fp32 ``[N, C]``."""
    if _on_gpu(x):
        return _hip().se_scale(x, s)
    return x * s.to(x.dtype).view(s.shape[0], 1, 1, s.shape[1])


def se_fused() -> bool:
    return _hip()._SE_FUSED


def se_block(x, w1, b1, w2, b2, act: str, gate: str) -> torch.Tensor:
    """GPU only: the whole squeeze-excitation block as one autograd node (see ``hip_ops._SEBlockFn``); ``w1`` / ``w2``
    are the 1x1-conv shaped parameters of ``nn.SqueezeExcitation``."""
    return _hip().se_block(x, w1, b1, w2, b2, act, gate)


def stochastic_depth(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """torchvision ``StochasticDepth(p, "row")``: drops the whole residual branch of a sample with probability
    ``p`` and rescales the survivors by ``1 / (1 - p)``."""
    if not training or p == 0.0:
        return x
    keep = 1.0 - p
    noise = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device).bernoulli_(keep)
    if keep > 0.0:
        noise.div_(keep)
    if _on_gpu(x):
        return _hip().se_scale(x, noise.expand(x.shape[0], x.shape[-1]))
    return x * noise.to(x.dtype).view(-1, 1, 1, 1)


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    if not training or p == 0.0:
        return x
    if _on_gpu(x):
        return _hip().dropout(x, p)
    return TF.dropout(x, p, True)


# ----------------------------------------------------------------------------- linear
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           relu: bool = False, w_lp: Optional[torch.Tensor] = None,
           out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``y = x @ weight.T + bias`` (N-K10), optional fused ReLU epilogue."""
    if _on_gpu(x):
        return _hip().linear(x, weight, bias, relu, w_lp, out_dtype)
    w = weight if w_lp is None else _StraightThroughCast.apply(weight, x.dtype)
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
    y = TF.linear(x, w, None if bias is None else bias.to(x.dtype))
    if relu:
        y = TF.relu(y)
    if out_dtype is not None:
        y = y.to(out_dtype)
    return y


# ----------------------------------------------------------------------------- loss / metrics
def cross_entropy(logits: torch.Tensor, target: torch.Tensor, loss_scale=None) -> torch.Tensor:
    """Mean softmax cross-entropy (N-K11).  fp32 math regardless of input dtype."""
    if _on_gpu(logits):
        return _hip().cross_entropy(logits, target)
    return TF.cross_entropy(logits.float(), target)


def topk_correct(output: torch.Tensor, target: torch.Tensor, topk: Sequence[int] = (1, 5)) -> torch.Tensor:
    """Number of rows whose target is within the top-k, float32 ``[len(topk)]`` (N-K12)."""
    if _on_gpu(output):
        return _hip().topk_correct(output, target, tuple(topk))
    z = output.float()
    zt = z.gather(1, target.view(-1, 1))
    rank = (z > zt).sum(dim=1)
    return torch.stack([(rank < k).sum().float() for k in topk])


def loss_and_metrics(logits: torch.Tensor, target: torch.Tensor):
    """-> (mean loss with grad, float32[4] record = loss*n, correct@1, correct@5, n)."""
    if _on_gpu(logits):
        return _hip().loss_and_metrics(logits, target)
    loss = cross_entropy(logits, target)
    with torch.no_grad():
        n = float(target.numel())
        k5 = min(5, logits.shape[1])
        c = topk_correct(logits, target, (1, k5))
        rec = torch.stack([loss.detach().float() * n, c[0], c[1],
                           torch.tensor(n, device=logits.device)])
    return loss, rec


# ----------------------------------------------------------------------------- input pipeline
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_u8(x: torch.Tensor, dtype: torch.dtype = torch.float32, pad_c: int = 0,
                 mean=IMAGENET_MEAN, std=IMAGENET_STD, input_layout: str = "nhwc") -> torch.Tensor:
    """uint8 image batch -> normalised NHWC ``dtype`` (N-K16).

    ``(x - 255*mean_c) / (255*std_c)``, identical to the reference prefetcher
    (``imagenet_ddp_apex.py:311-340``).  ``pad_c`` zero-pads channels 3 -> pad_c
    so the stem convolution sees a vector-width-aligned C.
    """
    if _on_gpu(x):
        return _hip().normalize_u8(x, dtype, pad_c, mean, std, input_layout)
    if input_layout == "nchw":
        x = x.permute(0, 2, 3, 1)
    m = torch.tensor(mean, dtype=torch.float32, device=x.device) * 255.0
    s = torch.tensor(std, dtype=torch.float32, device=x.device) * 255.0
    y = ((x.float() - m) / s).to(dtype)
    if pad_c and pad_c > y.shape[-1]:
        y = TF.pad(y, (0, pad_c - y.shape[-1]))
    return y.contiguous()


def s2d_ok(x: torch.Tensor) -> bool:
    """True when ``to_s2d`` applies: a 3-channel image batch with even extents on the GPU."""
    return _on_gpu(x) and _hip().s2d_ok(x)


def to_s2d(x: torch.Tensor, dtype: torch.dtype, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """Image batch (uint8 NHWC / NCHW: normalised like ``normalize_u8``; float NCHW: converted) -> the space-to-depth
    layout ``[N, H/2, W/2, 16]`` consumed by 7x7 stride-2 stems (ResNet, DenseNet, GoogLeNet).  GPU only."""
    return _hip().to_s2d(x, dtype, mean, std)


def to_nhwc(x: torch.Tensor, dtype: Optional[torch.dtype] = None, pad_c: int = 0) -> torch.Tensor:
    """Float NCHW (torchvision convention) -> NHWC compute layout."""
    if _on_gpu(x):
        return _hip().nchw_to_nhwc(x, dtype or x.dtype, pad_c)
    y = x.permute(0, 2, 3, 1)
    if dtype is not None:
        y = y.to(dtype)
    if pad_c and pad_c > y.shape[-1]:
        y = TF.pad(y, (0, pad_c - y.shape[-1]))
    return y.contiguous()

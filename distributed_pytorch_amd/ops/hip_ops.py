"""GPU implementations of the functional ops: autograd wrappers over the hand-written
gfx950 kernels in ``csrc/`` (no ATen / MIOpen / rocBLAS compute on this path).

Every wrapper allocates its outputs with ``torch.empty`` (caching allocator, so a
step can be captured in a hipGraph), launches on torch's *current* stream and never
synchronises with the host.  Tensors are NHWC; conv weights are logical
``[K, C, R, S]`` whose memory is KRSC.

Reference parity: these ops are what cuDNN / cuBLAS / ATen / Apex executed for the
reference's ``model(images)``, ``criterion``, ``loss.backward()`` and
``optimizer.step()`` (``nd_imagenet.py:291-303``, ``imagenet_ddp_apex.py:372-383``).
"""
from __future__ import annotations

import math
import functools
import collections
import os
import types
from typing import Dict, Optional, Tuple

import torch

from .. import backend
from . import grad_sink

_C = None


_BN_REVERSE = int(os.environ.get("DPA_BN_REVERSE", "0"))


def C():
    global _C
    if _C is None:
        _C = backend.require()
        _C.bn_set_reverse(_BN_REVERSE)
        _C.igemm_set_row3(int(os.environ.get("DPA_IGEMM_ROW3", "-1")))   # filter-row kernel: -1 auto, 0 off
        if "DPA_IGEMM_TAIL_MIN" in os.environ:          # masked K-tile tails: min channels per tap; 0 off; -n: n, no filter-row tails
            _C.igemm_set_tail_min(int(os.environ["DPA_IGEMM_TAIL_MIN"]))
        _C.igemm_set_nt_store(int(os.environ.get("DPA_IGEMM_NT", "0")))  # 0: by size, 1: always, -1: never
        _C.igemm_set_lean(int(os.environ.get("DPA_IGEMM_LEAN", "-1")))   # lean 1x1 GEMM: -1 auto, 0 off
        # convolutions on the 256 x 256 multi-phase pipeline (csrc/kernels/conv8.hip): -1 auto, 0 off, 1 / 2 forced
        _C.wgrad_row3_set_mode(int(os.environ.get("DPA_WGRAD_ROW3", "-1")))     # filter-row weight gradient: -1 auto, 0 off
        _C.wgrad8_set_mode(int(os.environ.get("DPA_WGRAD8", "-1")))             # 256 x 256 dense 1x1 weight gradient
        _C.conv8_set_mode(int(os.environ.get("DPA_CONV8", "-1")))
        _C.conv8_set_min_k(int(os.environ.get("DPA_CONV8_MIN_K", "512")))
        _C.conv8_set_tail_split(int(os.environ.get("DPA_CONV8_TAIL", "4")))
        if "DPA_BN_GRID" in os.environ:          # workgroups of the BN streaming passes (default 1280, see bn.hip)
            _C.bn_set_tuning(int(os.environ["DPA_BN_GRID"]), 1)
        if "DPA_DW_SLIDING" in os.environ:       # 0: window-per-pixel depthwise kernels only
            _C.dw_set_sliding(int(os.environ["DPA_DW_SLIDING"]))
        if "DPA_STREAM_GRID" in os.environ:      # workgroup cap of the elementwise / pooling / depthwise kernels
            _C.set_stream_grid(int(os.environ["DPA_STREAM_GRID"]))
        if "DPA_NT_MASK" in os.environ:          # 1 BN apply, 2 BN backward apply, 4 BN backward reduce, 8 igemm stores
            _C.set_nt_mask(int(os.environ["DPA_NT_MASK"]))
        if "DPA_NT_MIN_MB" in os.environ:        # streaming threshold (default 32 MB); negative: never
            _C.set_nt_min_bytes(int(float(os.environ["DPA_NT_MIN_MB"]) * (1 << 20)))
    return _C


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _dt(t) -> int:
    try:
        return _DT[t if isinstance(t, torch.dtype) else t.dtype]
    except KeyError:
        raise TypeError("unsupported dtype %r for HIP kernels" % (t,))


def _vec(dtype: torch.dtype) -> int:
    return 4 if dtype == torch.float32 else 8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------
# weight compute copies (cast / channel padding / dgrad transposes), cached per parameter
# ------------------------------------------------------------------------------------------
_generation = 0          # bumped by optimizers that rewrite parameters through raw pointers


def bump_generation() -> None:
    global _generation
    _generation += 1
    _pending_bnred.clear()          # never carry a fused BN-backward slab across optimizer steps
    _premasked.clear()
    _pending_mask.clear()
    _pending_compact.clear()
    _pending_fold.clear()
    _pending_colsum.clear()


class _WCache:
    __slots__ = ("wc", "wc_key", "wc_stamp", "wt", "wt_key", "wt_stamp")

    def __init__(self):
        self.wc = self.wt = None
        self.wc_key = self.wt_key = None
        self.wc_stamp = self.wt_stamp = None


def _cache(p: torch.Tensor) -> _WCache:
    c = getattr(p, "_dpa_wcache", None)
    if c is None:
        c = _WCache()
        p._dpa_wcache = c
    return c


def _stamp(p: torch.Tensor):
    return (p._version, _generation, p.data_ptr())


def _krsc_view(w: torch.Tensor) -> torch.Tensor:
    """[K,C,R,S] logical -> [K,R,S,C] tensor that is contiguous in memory (no copy when the
    parameter already is channels_last)."""
    v = w.detach().permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def conv_weight(w: torch.Tensor, dtype: torch.dtype, cpad: int) -> torch.Tensor:
    """Compute copy [K, R, S, cpad] in ``dtype`` (the parameter itself when nothing changes)."""
    K, Cw, R, S = w.shape
    c = _cache(w)
    key = (dtype, cpad)
    if c.wc is not None and c.wc_key == key and c.wc_stamp == _stamp(w):
        return c.wc
    src = _krsc_view(w)
    if cpad == Cw and dtype == w.dtype:
        wc = src
    else:
        if c.wc is None or c.wc_key != key:
            c.wc = torch.zeros((K, R, S, cpad), dtype=dtype, device=w.device)
        wc = c.wc
        if cpad == Cw:
            C().cast(src.data_ptr(), _dt(src), wc.data_ptr(), _dt(dtype), src.numel(), _stream())
        else:
            # channel padding (stem only): [K*R*S] "images" of one pixel with C channels -> cpad
            C().nchw_to_nhwc(src.data_ptr(), wc.data_ptr(), K * R * S, Cw, 1, cpad, _dt(src), _dt(dtype),
                             _stream())
    c.wc, c.wc_key, c.wc_stamp = wc, key, _stamp(w)
    return wc


def conv_weight_t(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """dgrad operand [C, R, S, K] in ``dtype``."""
    K, Cw, R, S = w.shape
    c = _cache(w)
    if c.wt is not None and c.wt_key == dtype and c.wt_stamp == _stamp(w):
        return c.wt
    src = _krsc_view(w)
    if c.wt is None or c.wt_key != dtype:
        c.wt = torch.empty((Cw, R, S, K), dtype=dtype, device=w.device)
    C().transpose_ktc(src.data_ptr(), c.wt.data_ptr(), K, R * S, Cw, _dt(src), _dt(dtype), _stream())
    c.wt_key, c.wt_stamp = dtype, _stamp(w)
    return c.wt


def linear_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    c = _cache(w)
    key = (dtype, 0)
    if c.wc is not None and c.wc_key == key and c.wc_stamp == _stamp(w):
        return c.wc
    src = w.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    if dtype == w.dtype:
        wc = src
    else:
        if c.wc is None or c.wc_key != key:
            c.wc = torch.empty(w.shape, dtype=dtype, device=w.device)
        wc = c.wc
        C().cast(src.data_ptr(), _dt(src), wc.data_ptr(), _dt(dtype), src.numel(), _stream())
    c.wc, c.wc_key, c.wc_stamp = wc, key, _stamp(w)
    return wc


def linear_weight_t(w: torch.Tensor, dtype: torch.dtype, outp: int) -> torch.Tensor:
    """[in, outp] (zero padded columns) for the input-gradient GEMM."""
    out_f, in_f = w.shape
    c = _cache(w)
    key = (dtype, outp)
    if c.wt is not None and c.wt_key == key and c.wt_stamp == _stamp(w):
        return c.wt
    src = w.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    if c.wt is None or c.wt_key != key:
        c.wt = torch.zeros((in_f, outp), dtype=dtype, device=w.device)
    if outp == out_f:
        C().transpose_ktc(src.data_ptr(), c.wt.data_ptr(), out_f, 1, in_f, _dt(src), _dt(dtype), _stream())
    else:
        tmp = torch.empty((in_f, out_f), dtype=dtype, device=w.device)
        C().transpose_ktc(src.data_ptr(), tmp.data_ptr(), out_f, 1, in_f, _dt(src), _dt(dtype), _stream())
        # widen rows: "images" of in_f pixels... reuse the channel-padding kernel (C=out_f -> outp)
        C().nchw_to_nhwc(tmp.data_ptr(), c.wt.data_ptr(), in_f, out_f, 1, outp, _dt(dtype), _dt(dtype), _stream())
    c.wt_key, c.wt_stamp = key, _stamp(w)
    return c.wt


# ------------------------------------------------------------------------------------------
# tap tables
# ------------------------------------------------------------------------------------------
_taps: Dict[tuple, torch.Tensor] = {}


def _tap_table(device, entries) -> torch.Tensor:
    key = (device.index, tuple(entries))
    t = _taps.get(key)
    if t is None:
        t = torch.tensor([v for e in entries for v in e], dtype=torch.int32, device=device)
        _taps[key] = t
        if tuple(entries) == ((0, 0, 0),):
            _dense1_tables.add(t.data_ptr())
    return t


_dense1_tables = set()      # tap tables holding the single tap (0, 0, 0): 1x1 convolutions without padding, Linear


def _fprop_taps(R, S, ph, pw):
    return tuple((r - ph, s - pw, r * S + s) for r in range(R) for s in range(S))


def _dgrad_classes(R, S, sh, sw, ph, pw):
    """Per output-parity class (py, px): taps (dy, dx, wtap) over the dY grid."""
    out = []
    for py in range(sh):
        for px in range(sw):
            taps = []
            for r in range(R):
                if (py + ph - r) % sh:
                    continue
                for s in range(S):
                    if (px + pw - s) % sw:
                        continue
                    taps.append(((py + ph - r) // sh, (px + pw - s) // sw, r * S + s))
            out.append((py, px, tuple(taps)))
    return out


# ------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------
def _conv_out(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def _igemm(A, B, Cout_t, bias, stats, taps, M, N, Cred, ntaps, lda, ldb, ldc, IH, IW, SY, SX, OH, OW,
           scatter=None, relu=False, dtype=None, out_dtype=None, accum=None, bnred=None, groups=None, row3=False,
           accum_mask=None, axf=None, rstride=None, xmask=None, cat=None, relu_mask=None):
    """``cat``: (A2, lda2, Cred2[, B2, ldb2]) -- the reduction continues over a second gathered operand against the next
    Cred2 columns of B, or against B2 (lean GEMM, dense 1x1; needs a bias).
    ``xmask``: a ReLU output with the layout of the result; result elements where it is <= 0 are stored as 0.
    ``groups``: (count, a_gstride, b_gstride, c_gstride, stats_ld) -- one launch, grid.y = group.
    ``rstride``: (OHc, OWc, sy, sx) -- ``accum`` is the COMPACT gradient [NB, OHc, OWc, N] of a stride-s 1x1 branch
    (added at the output pixels whose coordinates are multiples of s; lean GEMM only).
    ``row3``: the tap table is sorted into filter rows (dy, dx ascending, three taps per row): stride-1 "same"
    convolutions may use the filter-row kernel (one staged activation tile per three taps)."""
    flags = (C().IGEMM_RELU if relu else 0) | (C().IGEMM_ROW3 if row3 else 0)
    if ntaps == 1 and taps.data_ptr() in _dense1_tables:
        flags |= C().IGEMM_DENSE1       # eligible for the lean GEMM (csrc/kernels/gemm_lean.hip)
    a_scale = a_shift = 0
    if axf is not None:                 # coef[4][Cred] of the BatchNorm + ReLU applied while A is staged
        flags |= C().IGEMM_AXF
        a_scale, a_shift = axf[2].data_ptr(), axf[3].data_ptr()
    gargs = (1, 0, 0, 0, 0) if groups is None else groups
    bn_x = bn_z = bn_coef = bn_relu = 0
    if bnred is not None:               # (bn input, bn output or None, coef[4][N], relu mode)
        flags |= C().IGEMM_BN_BWD
        bn_x, bn_z, bn_coef, bn_relu = bnred[0].data_ptr(), _p(bnred[1]), bnred[2].data_ptr(), bnred[3]
        if len(bnred) > 4 and bnred[4]:     # "bn fold": masked store, sum dz_m only, the BN input is not read
            flags |= C().IGEMM_BN_FOLD
    if xmask is not None:
        flags |= C().IGEMM_XMASK
        bn_x = xmask.data_ptr()
    if relu_mask is not None:           # C = relu(acc + bias + accum), ReLU bit mask written to `relu_mask` (lean GEMM)
        flags |= C().IGEMM_RELU_MASK
        bn_z = relu_mask.data_ptr()
    OHF = OWF = OSY = OSX = OY0 = OX0 = 0
    if scatter is not None:
        flags |= C().IGEMM_SCATTER
        OHF, OWF, OSY, OSX, OY0, OX0 = scatter
    if rstride is not None:
        flags |= C().IGEMM_RSTRIDED
        OHF, OWF, OSY, OSX = rstride
    C().igemm(A.data_ptr(), B.data_ptr(), Cout_t.data_ptr(), _p(accum), _p(bias), _p(stats), taps.data_ptr(), M, N, Cred,
              ntaps, lda, ldb, ldc, IH, IW, SY, SX, OH, OW, OHF, OWF, OSY, OSX, OY0, OX0, flags,
              _dt(dtype), _dt(out_dtype), _stream(), bn_x, bn_z, bn_coef, bn_relu, *gargs, _p(accum_mask),
              a_scale, a_shift, *((cat[0].data_ptr(), cat[1], cat[2]) if cat is not None else (0, 0, 0)),
              *((cat[3].data_ptr(), cat[4]) if cat is not None and len(cat) > 3 else (0, 0)))


def _stats_slab(nparts: int, C_: int, device) -> torch.Tensor:
    """Forward-statistics slab [nparts][2][C] = Chan partials (sum, M2 about the part's mean) with the parts' row
    counts stored right behind it (``kernels.h: bn_stats_floats``).  The view keeps the tail alive."""
    flat = torch.empty(nparts * 2 * C_ + nparts, dtype=torch.float32, device=device)
    return flat[:nparts * 2 * C_].view(nparts, 2, C_)


def _colsum(dy2d: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of [M, C] (bias gradients) through the BN column reducer."""
    M, Cc = dy2d.shape
    nblk = C().bn_partial_rows(M, Cc)
    part = _stats_slab(nblk, Cc, dy2d.device)
    C().bn_stats(dy2d.data_ptr(), M, Cc, _dt(dy2d), part.data_ptr(), nblk, _stream())
    s1 = torch.empty(Cc, dtype=torch.float32, device=dy2d.device)
    s2 = torch.empty(Cc, dtype=torch.float32, device=dy2d.device)
    C().bn_bwd_finalize(part.data_ptr(), nblk, Cc, s2.data_ptr(), s1.data_ptr(), _stream())
    return s1


def _bias_grad(bias_p, dy2d: torch.Tensor, keep=None):
    """Gradient of a bias = column sums of the [M, C >= keep] output gradient, handed to the parameter's sink (returns
    None) or to autograd.  When the sink has not been written this iteration the fold of the partial sums writes
    straight into its view: two launches instead of three (+ two temporaries).  A one-launch variant (in-kernel ordered
    fold behind a ticket) measured slower: wide enough for the bandwidth it leaves a long serial fold to one workgroup
    (56 us), narrow enough for the fold it pays an L2 write-back per workgroup (profiles/r3_notes.md)."""
    M, Cc = dy2d.shape
    keep = Cc if keep is None else keep
    sink = grad_sink.sink_of(bias_p)
    if sink is not None and keep == Cc and sink.beta() == 0.0 and sink.view.is_contiguous() and sink.view.numel() == Cc \
            and dy2d.is_contiguous():
        nblk = C().bn_partial_rows(M, Cc)
        part = _stats_slab(nblk, Cc, dy2d.device)
        st = _stream()
        C().bn_stats(dy2d.data_ptr(), M, Cc, _dt(dy2d), part.data_ptr(), nblk, st)
        s2 = torch.empty(Cc, dtype=torch.float32, device=dy2d.device)
        C().bn_bwd_finalize(part.data_ptr(), nblk, Cc, s2.data_ptr(), sink.view.data_ptr(), st)
        sink.delivered()
        return None
    return _deliver(bias_p, _colsum(dy2d if dy2d.is_contiguous() else dy2d.contiguous())[:keep].to(bias_p.dtype))


def _flat_same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    if a.shape != b.shape:
        return False
    sa = [st for st, n in zip(a.stride(), a.shape) if n > 1]
    sb = [st for st, n in zip(b.stride(), b.shape) if n > 1]
    return sa == sb


def _deliver(param, g):
    """Hand a finished parameter gradient to its sink (returns None) or to autograd (returns g)."""
    s = grad_sink.sink_of(param)
    if s is None or g is None:
        return g
    if not _flat_same_layout(g, s.view):
        g = torch.empty_like(s.view).copy_(g)
    C().axpby(g.data_ptr(), _dt(g), s.view.data_ptr(), g.numel(), 1.0, s.beta(), _stream())
    s.delivered()
    return None


# ------------------------------------------------------------------------------------------
# BatchNorm backward folded into the producing 1x1 convolution ("bn fold", csrc/kernels/bnfold.hip)
# ------------------------------------------------------------------------------------------
_BN_FOLD = os.environ.get("DPA_BN_FOLD", "1") == "1"
# expansion ratio (output / input channels) from which the fold is taken: the passes it removes stream the WIDE tensor,
# the GEMMs it adds run on the narrow one
_BN_FOLD_MIN_RATIO = float(os.environ.get("DPA_BN_FOLD_MIN_RATIO", "2"))
# size (MiB of the BatchNorm input) from which it is taken: measured on ResNet-50 at batch 1024, the 7x7 blocks (205 MiB)
# lose 0.1 ms to the extra small launches, everything larger gains (profiles/r3/bn_fold.txt)
_BN_FOLD_MIN_MB = float(os.environ.get("DPA_BN_FOLD_MIN_MB", "300"))
# ... and from half that size where the convolution's input has at most 256 channels: the fold's own cost is launches plus
# work on C x C matrices, so at batch 256 the 28x28 128 -> 512 blocks (205 MiB, C = 128) gain 3 % of the step while at batch
# 1024 the 7x7 512 -> 2048 ones (also 205 MiB, C = 512) lose (profiles/r3_notes.md)
_BN_FOLD_NARROW_C = 256
# data_ptr of the NaN token a folded BatchNorm returned as "the gradient of its input" -> (token, dy (ReLU-masked), partial
# slab with sum dy_m per part, BN coef [4][N], gamma, beta parameter, count, generation).  Consumed by _ConvFn.backward of
# the 1x1 convolution that produced the BatchNorm's input.
_pending_fold: Dict[int, tuple] = {}
# data_ptr of a MASKED block-output gradient handed on to the residual branch -> (partial slab whose first row holds its
# column sums per 128-row part, version, generation)
_pending_colsum: Dict[int, tuple] = {}
bn_fold_counts = {"folded": 0}


def _bn_fold_eligible(x, weight, bias, stride, padding, relu, want_stats, needs_input_grad, skip) -> bool:
    if not (_BN_FOLD and want_stats and bias is None and not relu and not skip and needs_input_grad):
        return False
    if not (torch.is_grad_enabled() and x.is_cuda and x.requires_grad and weight.requires_grad):
        return False
    K, Cw, R, S = weight.shape
    if (R, S) != (1, 1) or tuple(padding) != (0, 0) or tuple(stride) not in ((1, 1), (2, 2)):
        return False
    if x.dtype not in (torch.bfloat16, torch.float16) or weight.dtype != torch.float32 or x.dim() != 4:
        return False
    Cx = x.shape[-1]
    if Cx != Cw or Cx % 64 or K % 64 or K < _BN_FOLD_MIN_RATIO * Cx:
        return False
    min_mb = _BN_FOLD_MIN_MB * (0.5 if Cx <= _BN_FOLD_NARROW_C else 1.0)
    if x.numel() // Cx // (stride[0] * stride[1]) * K * x.element_size() < min_mb * 2 ** 20:
        return False
    return grad_sink.sink_of(weight) is not None


def _wgrad_raw(dy2d_src, x_src, M, N, Cc, IH, IW, gram: bool = False, out_dtype=torch.float32, ldy=None, ldx=None) -> torch.Tensor:
    """[N][Cc] = dy^T x over M pixel rows (dy: [M][ldy >= N], x: [M][ldx >= Cc]) through the weight-gradient kernels (fp32
    accumulation, result in ``out_dtype``).  ``gram``: the same launch also produces x^T x; the result is [N + Cc][Cc]
    (x is staged once for both)."""
    ldy = N if ldy is None else ldy
    ldx = Cc if ldx is None else ldx
    rows = N + (Cc if gram else 0)
    st = _stream()
    # deep layers: the 256 x 256 LDS-DMA pipeline (csrc/kernels/wgrad8.hip); 0 splits = not covered
    splits = 0
    if (dy2d_src.data_ptr() | x_src.data_ptr()) % 16 == 0:
        splits, pps = C().wgrad8_splits(M, N, Cc if gram else 0, Cc, _dt(dy2d_src))
    if splits > 0:
        slab = torch.empty((splits, rows, Cc), dtype=torch.float32, device=dy2d_src.device)
        C().wgrad8(dy2d_src.data_ptr(), x_src.data_ptr(), slab.data_ptr(), M, N, Cc, ldy, ldx, splits, pps,
                   _dt(dy2d_src), st, x_src.data_ptr() if gram else 0, Cc if gram else 0, ldx if gram else 0)
    else:
        splits, pps = C().wgrad_choose_splits(M, N, Cc, _dt(dy2d_src))
        slab = torch.empty((splits, rows, Cc), dtype=torch.float32, device=dy2d_src.device)
        taps = _tap_table(dy2d_src.device, ((0, 0, 0),))
        C().wgrad(dy2d_src.data_ptr(), x_src.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, N, Cc, 1, ldy, ldx, IH, IW,
                  1, 1, IH, IW, splits, pps, 1, _dt(dy2d_src), st, 1, 0, 0, 0, 0,
                  x_src.data_ptr() if gram else 0, Cc if gram else 0, ldx if gram else 0)
    out = torch.empty((rows, Cc), dtype=out_dtype, device=dy2d_src.device)
    C().slab_reduce(slab.data_ptr(), splits, rows * Cc, out.data_ptr(), _dt(out_dtype), 0.0, st)
    return out


def _grad_into(param, fill):
    """Gradient of `param` produced by `fill(view, beta)` (fp32, the parameter's memory layout): straight into the sink
    view when there is one, else into a temporary that joins ``param.grad`` like autograd's accumulation would."""
    s = grad_sink.sink_of(param)
    if s is not None and _flat_same_layout(s.view, param):
        fill(s.view, s.beta())
        s.delivered()
        return
    g = torch.empty_like(param, dtype=torch.float32)
    fill(g, 0.0)
    g = g.to(param.dtype)
    param.grad = g if param.grad is None else param.grad + g


def _subsample(x: torch.Tensor, s: int) -> torch.Tensor:
    """x[:, ::s, ::s, :] as a dense tensor (the pixels a stride-s 1x1 convolution reads)."""
    if s == 1:
        return x
    NB, IH, IW, Cc = x.shape
    if not x.is_contiguous() or (Cc * x.element_size()) % 16:
        return x[:, ::s, ::s, :].contiguous()
    y = torch.empty((NB, (IH + s - 1) // s, (IW + s - 1) // s, Cc), dtype=x.dtype, device=x.device)
    C().subsample(x.data_ptr(), y.data_ptr(), NB, IH, IW, Cc * x.element_size(), s, _stream())
    return y


def _conv_bn_fold_backward(ctx, fent, pre=None):
    """Backward of (1x1 convolution -> training BatchNorm [+ residual + ReLU]) without the BatchNorm-backward apply pass:
    see csrc/kernels/bnfold.hip for the algebra.  `fent` comes from _BNFn.backward, `ctx` is the convolution's.
    ``pre`` = (gram, colsum(a), W gram) kept from a forward pass that took the BatchNorm statistics from them
    (_ConvBnResReluFn); ``part`` may then be None (no producer reduced sum dy_m: it is reduced here)."""
    _, dy, part, coef, gamma, beta_p, count, _ = fent
    a, weight, _ = ctx.saved_tensors
    full_shape = tuple(a.shape)
    (sh, sw) = ctx.geom[0]
    if (sh, sw) != (1, 1):
        # strided 1x1 branch: everything below is a dense problem on the pixels the convolution really read
        a = pre[3] if (pre is not None and len(pre) > 3) else (_subsample(a, sh) if sh == sw else a[:, ::sh, ::sw, :].contiguous())
    NB, IH, IW, Cx = a.shape
    N = weight.shape[0]
    M = NB * IH * IW
    dev, dt, st = a.device, a.dtype, _stream()
    bn_fold_counts["folded"] += 1
    # ---- reductions over the pixels: raw = dy_m^T a, gram = a^T a, cs = colsum(a), dbeta = colsum(dy_m) (producer's slab)
    if pre is None:
        rg = _wgrad_raw(dy, a, M, N, Cx, IH, IW, gram=True)     # one launch: a is staged once for both products
        raw, gram = rg[:N], rg[N:]
        cs = _colsum(a.view(M, Cx))
    else:
        raw = _wgrad_raw(dy, a, M, N, Cx, IH, IW)
        gram, cs = pre[0], pre[1]
    if pre is not None and len(pre) > 4 and pre[4] is not None:
        dbeta_sum = pre[4]              # (a sibling BatchNorm on the same masked gradient already reduced it)
    elif part is None:
        dbeta_sum = _colsum(dy.view(M, N))
    else:
        dbeta_sum = torch.empty(N, dtype=torch.float32, device=dev)
        scratch = torch.empty(N, dtype=torch.float32, device=dev)
        partf = _fold(part, N)              # (thousands of 128-row parts: folded by many workgroups first)
        C().bn_bwd_finalize(partf.data_ptr(), partf.shape[0], N, scratch.data_ptr(), dbeta_sum.data_ptr(), st)
    if hasattr(ctx, "dbeta_out"):
        ctx.dbeta_out = dbeta_sum
    # ---- BatchNorm coefficients (A, B, K), dgamma / dbeta, and the scaled weight operands
    wlp = conv_weight(weight, dt, Cx)                                 # [N][Cx] as the forward pass used it
    abk = torch.empty((3, N), dtype=torch.float32, device=dev)
    wt_scaled = torch.empty((Cx, N), dtype=dt, device=dev)            # (A o W)^T: the data gradient's first weight operand
    bw = torch.empty((N, Cx + 8), dtype=dt, device=dev)               # (B o W | K | 0), output-channel major
    g32 = _f32(gamma.detach())
    box = {}

    def fill_gamma(view, beta):
        box["g"] = (view, beta)

    def fill_beta(view, beta):
        box["b"] = (view, beta)
        gv, gb = box["g"]
        C().bnfold_coeffs(raw.data_ptr(), wlp.data_ptr(), Cx, dbeta_sum.data_ptr(), coef[0].data_ptr(),
                          coef[1].data_ptr(), g32.data_ptr(), float(count), N, Cx, gv.data_ptr(), float(gb),
                          view.data_ptr(), float(beta), abk.data_ptr(), wt_scaled.data_ptr(), N, bw.data_ptr(), _dt(a), st)

    # (gamma's view is only noted; the kernel that fills both runs when beta's is known)
    sg, sb = grad_sink.sink_of(gamma), grad_sink.sink_of(beta_p)
    if sg is not None and sb is not None:
        fill_gamma(sg.view, sg.beta())
        fill_beta(sb.view, sb.beta())
        sg.delivered()
        sb.delivered()
    else:
        tg = torch.empty(N, dtype=torch.float32, device=dev)
        tb = torch.empty(N, dtype=torch.float32, device=dev)
        fill_gamma(tg, 0.0)
        fill_beta(tb, 0.0)
        for prm, t in ((gamma, tg), (beta_p, tb)):
            r = _deliver(prm, t.to(prm.dtype))
            if r is not None:
                prm.grad = r if prm.grad is None else prm.grad + r
    one = _tap_table(dev, ((0, 0, 0),))
    dx = None
    if ctx.needs_input_grad[0] and ctx.geom[3]:
        # ---- G = W^T diag(B) W and bvec = K^T W: a reduction over the N output channels with both operands stored
        #      channel-major -- the shape of a weight gradient with N "pixels" (split across workgroups; as an ordinary
        #      GEMM it was one or two workgroups walking all of K: 69 us per call)
        gk = _wgrad_raw(wlp, bw, N, Cx, Cx + 8, 1, N, out_dtype=dt, ldy=Cx, ldx=Cx + 8)        # [Cx][Cx + 8]
        bvec = gk[:, Cx].float().contiguous()
        # ---- da = dy_m (A o W) + a G + bvec in one launch (two gathered operands back to back), with the usual fused reduction for the BN before
        src, ctx.bnsrc = ctx.bnsrc, None
        bnred = slab = None
        if src is not None and src.uses == 1 and src.mode != 1:
            bnred = (src.x, src.mask if src.mode == 3 else None, src.coef, src.mode, bool(src.fold))
            rows = C().igemm_stats_rows(M, Cx, N, 1, _dt(dy), _dt(dy))
            slab = torch.empty((rows, 2, Cx), dtype=torch.float32, device=dev)
        dx = torch.empty((NB, IH, IW, Cx), dtype=dt, device=dev)
        _igemm(dy, wt_scaled, dx, bvec, slab, one, M, Cx, N, 1, N, N, Cx, IH, IW, 1, 1, IH, IW, dtype=dt, out_dtype=dt,
               bnred=bnred, cat=(a, Cx, Cx, gk, Cx + 8))
        if slab is not None:
            _pending_bnred[dx.data_ptr()] = (slab, bnred[0].data_ptr(), dx._version, bool(bnred[4]))
        if (sh, sw) != (1, 1):
            if ctx.compact:
                # the owner of the alias this branch read adds a COMPACT gradient in its own epilogue (see _ConvFn.backward)
                fake = torch.full((1,), float("nan"), dtype=dt, device=dev)
                dxc, dx = dx, fake.expand(full_shape)
                _pending_compact[dx.data_ptr()] = (fake, dxc, (sh, sw), _generation, full_shape)
            else:
                dx = _expand_compact((None, dx, (sh, sw), _generation, full_shape))
    # ---- weight gradient: A o raw + B o (W gram) + K (x) cs, W gram in fp32 (the terms cancel)
    if ctx.needs_input_grad[1]:
        if pre is not None:
            wg = pre[2]
        else:
            wg = torch.empty((N, Cx), dtype=torch.float32, device=dev)
            w32 = weight.detach()
            _igemm(w32, gram, wg, None, None, one, N, Cx, Cx, 1, Cx, Cx, Cx, 1, N, 1, 1, 1, N, dtype=torch.float32,
                   out_dtype=torch.float32)

        def fill_w(view, beta):
            C().bnfold_wgrad(raw.data_ptr(), wg.data_ptr(), abk.data_ptr(), cs.data_ptr(), view.data_ptr(), N, Cx,
                             float(beta), st)

        _grad_into(weight, fill_w)
    return dx, None, None, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------
# Tail of a bottleneck in ONE GEMM: out = relu(BN(a W^T) + identity) with the BatchNorm statistics taken from the Gram
# matrix of a (csrc/kernels/bnfold.hip: bnfold_fwd) -- the pre-BatchNorm tensor is never written or read.  Backward is
# the "bn fold" above with the Gram matrix, colsum(a) and W gram kept from forward.
# ------------------------------------------------------------------------------------------
_BN_FWD_FOLD = os.environ.get("DPA_BN_FWD_FOLD", "1") == "1"
# colsum(a) from the pass that materialises a (bn_apply_colsum) instead of a reduction pass of its own
_COLSUM_IN_APPLY = os.environ.get("DPA_COLSUM_IN_APPLY", "1") == "1"
bn_fwd_fold_counts = {"fused": 0}


class _ConvBnResReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, weight, gamma, beta, residual, running_mean, running_var, nbt, momentum, eps, bnsrc, cs_pre=None):
        NB, IH, IW, Cx = a.shape
        N = weight.shape[0]
        M = NB * IH * IW
        dev, dt, st = a.device, a.dtype, _stream()
        one = _tap_table(dev, ((0, 0, 0),))
        # ---- statistics of z = a W^T from a^T a and colsum(a)
        gram = _wgrad_raw(a, a, M, Cx, Cx, IH, IW)
        cs = cs_pre if cs_pre is not None else _colsum(a.view(M, Cx))
        w32 = weight.detach()               # [N, Cx, 1, 1]: N rows of Cx in either memory format
        wg = torch.empty((N, Cx), dtype=torch.float32, device=dev)
        _igemm(w32, gram, wg, None, None, one, N, Cx, Cx, 1, Cx, Cx, Cx, 1, N, 1, 1, 1, N, dtype=torch.float32,
               out_dtype=torch.float32)
        coef = torch.empty((4, N), dtype=torch.float32, device=dev)       # mean, invstd, scale, shift
        wsc = torch.empty((N, Cx), dtype=dt, device=dev)                   # scale o W
        g32, b32 = _f32(gamma.detach()), _f32(beta.detach())
        C().bnfold_fwd(wg.data_ptr(), w32.data_ptr(), cs.data_ptr(), float(M), g32.data_ptr(), b32.data_ptr(), float(eps),
                       float(momentum), _p(running_mean), _p(running_var), _p(nbt), coef.data_ptr(), wsc.data_ptr(), N, Cx,
                       _dt(a), st)
        # ---- out = relu(a (scale o W)^T + shift + identity), ReLU bit mask on the side
        out = torch.empty((NB, IH, IW, N), dtype=dt, device=dev)
        mask = torch.empty(M * N // _vec(dt), dtype=torch.uint8, device=dev)
        _igemm(a, wsc, out, coef[3], None, one, M, N, Cx, 1, Cx, Cx, N, IH, IW, 1, 1, IH, IW, dtype=dt, out_dtype=dt,
               accum=residual, relu_mask=mask)
        bn_fwd_fold_counts["fused"] += 1
        ctx.bnsrc = None
        if bnsrc is not None:
            bnsrc.uses += 1
            if _FUSE_BN_BWD and bnsrc.x.shape == a.shape and bnsrc.x.dtype == a.dtype \
                    and _FUSE_BN_BWD_MIN <= a.numel() * a.element_size() <= _FUSE_BN_BWD_MAX:
                ctx.bnsrc = bnsrc
        ctx.save_for_backward(a, weight, gamma, coef, mask, gram, cs, wg)
        ctx.beta_ref = beta
        ctx.count = float(M)
        ctx.out_ptr = out.data_ptr()
        ctx.set_materialize_grads(False)
        # (a detached alias, not `out` itself: the object ends up as an attribute OF out, and a reference cycle would keep
        # every block output alive until the cyclic collector runs)
        _ConvBnResReluFn.last = _BnSrc(out.detach(), coef, 3, mask, True)
        return out

    @staticmethod
    def backward(ctx, dy):
        a, weight, gamma, coef, mask, gram, cs, wg = ctx.saved_tensors
        if dy is None:
            return (None,) * 12
        part = None
        fused = _pending_bnred.pop(dy.data_ptr(), None)
        if fused is not None and len(fused) > 3 and fused[3] and fused[1] == ctx.out_ptr and fused[2] == dy._version \
                and dy.is_contiguous():
            part = fused[0]         # the consumer's data gradient stored dy THROUGH the mask and reduced sum dy_m
        else:
            ent = _pending_mask.pop(dy.data_ptr(), None)
            dy = dy if dy.is_contiguous() else dy.contiguous()
            if ent is not None and ent[1] == dy._version and ent[2] == _generation \
                    and ent[0].numel() * _vec(dy.dtype) == dy.numel():
                dy = apply_bit_mask(dy, ent[0])     # (a lazily masked gradient of some other branch)
            dy = apply_bit_mask(dy, mask)
        fent = (None, dy, part, coef, gamma, ctx.beta_ref, ctx.count, _generation)
        fctx = types.SimpleNamespace(saved_tensors=(a, weight, None), geom=((1, 1), (0, 0), False, True, False),
                                     bnsrc=ctx.bnsrc, compact=False,
                                     needs_input_grad=(ctx.needs_input_grad[0], ctx.needs_input_grad[1]))
        ctx.bnsrc = None
        da = _conv_bn_fold_backward(fctx, fent, pre=(gram, cs, wg))[0]
        dres = dy if ctx.needs_input_grad[4] else None
        return da, None, None, None, dres, None, None, None, None, None, None, None


_ConvBnResReluFn.last = None


def conv_bn_res_relu(x, weight, stride, padding, gamma, beta, running_mean, running_var, nbt, momentum, eps, residual):
    """``relu(batch_norm(conv1x1(x)) + residual)`` in training mode as one GEMM, or None when the launch is not one the
    fused path takes (the caller then runs conv -> batch_norm)."""
    if not (_BN_FWD_FOLD and residual is not None and momentum is not None and running_mean is not None):
        return None
    if getattr(x, "_dpa_deferred_coef", None) is not None and not getattr(x, "_dpa_deferred_relu", False):
        return None
    if getattr(residual, "_dpa_deferred_coef", None) is not None:
        return None
    if tuple(stride) != (1, 1) or not _bn_fold_eligible(x, weight, None, stride, padding, False, True, True, False):
        return None
    NB, IH, IW, Cx = x.shape
    N = weight.shape[0]
    if tuple(residual.shape) != (NB, IH, IW, N) or residual.dtype != x.dtype or not residual.is_contiguous():
        return None
    if not gamma.requires_grad or not beta.requires_grad or N % 128:
        return None
    grad_sink.note_use(weight, None)
    grad_sink.note_use(gamma, beta)
    if getattr(x, "_dpa_deferred_coef", None) is not None:
        x = materialize(x, want_colsum=True)    # the GEMMs of both directions run on the normalised activation itself
    x = x if x.is_contiguous() else x.contiguous()
    out = _ConvBnResReluFn.apply(x, weight, gamma, beta, residual, running_mean, running_var, nbt, float(momentum),
                                 float(eps), getattr(x, "_dpa_bn", None), getattr(x, "_dpa_colsum", None))
    if _ConvBnResReluFn.last is not None and torch.is_grad_enabled():
        out._dpa_bn = _ConvBnResReluFn.last     # the consuming convolution stores its data gradient through the mask
    _ConvBnResReluFn.last = None
    return out


class _ConvBnCatReluFn(torch.autograd.Function):
    """Tail of a bottleneck WITH a downsample branch in one GEMM over two operands:
    ``out = relu(BN3(a W3^T) + BNd(x_s Wd^T))`` with both BatchNorms' statistics from Gram matrices (bnfold_fwd), the two
    scaled weight matrices as the two B operands of the lean GEMM and shift3 + shiftd as its bias.  Neither pre-BatchNorm
    tensor nor the downsample branch's output is ever written.  Backward: the "bn fold" twice on the same masked gradient."""

    @staticmethod
    def forward(ctx, a, w3, g3, b3, x, wd, gd, bd, stride, rm3, rv3, nbt3, mom3, eps3, rmd, rvd, nbtd, momd, epsd,
                bnsrc, compact_ok, cs_pre=None):
        NB, IH, IW, C3 = a.shape
        N = w3.shape[0]
        M = NB * IH * IW
        dev, dt, st = a.device, a.dtype, _stream()
        one = _tap_table(dev, ((0, 0, 0),))
        xs = _subsample(x, stride)
        Cd = xs.shape[-1]

        def stats(inp, Cc, w, g, b, rm, rv, nbt, mom, eps, cs=None):
            gram = _wgrad_raw(inp, inp, M, Cc, Cc, IH, IW)
            cs = cs if cs is not None else _colsum(inp.view(M, Cc))
            w32 = w.detach()
            wg = torch.empty((N, Cc), dtype=torch.float32, device=dev)
            _igemm(w32, gram, wg, None, None, one, N, Cc, Cc, 1, Cc, Cc, Cc, 1, N, 1, 1, 1, N, dtype=torch.float32,
                   out_dtype=torch.float32)
            coef = torch.empty((4, N), dtype=torch.float32, device=dev)
            wsc = torch.empty((N, Cc), dtype=dt, device=dev)
            g32, b32 = _f32(g.detach()), _f32(b.detach())
            C().bnfold_fwd(wg.data_ptr(), w32.data_ptr(), cs.data_ptr(), float(M), g32.data_ptr(), b32.data_ptr(),
                           float(eps), float(mom), _p(rm), _p(rv), _p(nbt), coef.data_ptr(), wsc.data_ptr(), N, Cc,
                           _dt(a), st)
            return gram, cs, wg, coef, wsc

        gram3, cs3, wg3, coef3, wsc3 = stats(a, C3, w3, g3, b3, rm3, rv3, nbt3, mom3, eps3, cs_pre)
        gramd, csd, wgd, coefd, wscd = stats(xs, Cd, wd, gd, bd, rmd, rvd, nbtd, momd, epsd)
        shift = coef3[3] + coefd[3]
        out = torch.empty((NB, IH, IW, N), dtype=dt, device=dev)
        mask = torch.empty(M * N // _vec(dt), dtype=torch.uint8, device=dev)
        _igemm(a, wsc3, out, shift, None, one, M, N, C3, 1, C3, C3, N, IH, IW, 1, 1, IH, IW, dtype=dt, out_dtype=dt,
               cat=(xs, Cd, Cd, wscd, Cd), relu_mask=mask)
        bn_fwd_fold_counts["fused"] += 1
        ctx.bnsrc = None
        if bnsrc is not None:
            bnsrc.uses += 1
            if _FUSE_BN_BWD and bnsrc.x.shape == a.shape and bnsrc.x.dtype == a.dtype \
                    and _FUSE_BN_BWD_MIN <= a.numel() * a.element_size() <= _FUSE_BN_BWD_MAX:
                ctx.bnsrc = bnsrc
        ctx.save_for_backward(a, w3, g3, coef3, gram3, cs3, wg3, x, wd, gd, coefd, gramd, csd, wgd,
                              xs if stride != 1 else None, mask)
        ctx.b3_ref, ctx.bd_ref = b3, bd
        ctx.count = float(M)
        ctx.stride = stride
        ctx.compact = bool(compact_ok and stride in (2, 4) and N % 32 == 0)
        ctx.out_ptr = out.data_ptr()
        ctx.set_materialize_grads(False)
        _ConvBnResReluFn.last = _BnSrc(out.detach(), coef3, 3, mask, True)      # (detached: see _ConvBnResReluFn)
        return out

    @staticmethod
    def backward(ctx, dy):
        a, w3, g3, coef3, gram3, cs3, wg3, x, wd, gd, coefd, gramd, csd, wgd, xs, mask = ctx.saved_tensors
        if dy is None:
            return (None,) * 22
        part = None
        fused = _pending_bnred.pop(dy.data_ptr(), None)
        if fused is not None and len(fused) > 3 and fused[3] and fused[1] == ctx.out_ptr and fused[2] == dy._version \
                and dy.is_contiguous():
            part = fused[0]
        else:
            ent = _pending_mask.pop(dy.data_ptr(), None)
            dy = dy if dy.is_contiguous() else dy.contiguous()
            if ent is not None and ent[1] == dy._version and ent[2] == _generation \
                    and ent[0].numel() * _vec(dy.dtype) == dy.numel():
                dy = apply_bit_mask(dy, ent[0])
            dy = apply_bit_mask(dy, mask)
        s_ = ctx.stride
        fctx3 = types.SimpleNamespace(saved_tensors=(a, w3, None), geom=((1, 1), (0, 0), False, True, False),
                                      bnsrc=ctx.bnsrc, compact=False, dbeta_out=None,
                                      needs_input_grad=(ctx.needs_input_grad[0], ctx.needs_input_grad[1]))
        ctx.bnsrc = None
        da = _conv_bn_fold_backward(fctx3, (None, dy, part, coef3, g3, ctx.b3_ref, ctx.count, _generation),
                                    pre=(gram3, cs3, wg3))[0]
        fctxd = types.SimpleNamespace(saved_tensors=(x, wd, None), geom=((s_, s_), (0, 0), False, True, False),
                                      bnsrc=None, compact=ctx.compact,
                                      needs_input_grad=(ctx.needs_input_grad[4], ctx.needs_input_grad[5]))
        dx = _conv_bn_fold_backward(fctxd, (None, dy, part, coefd, gd, ctx.bd_ref, ctx.count, _generation),
                                    pre=(gramd, csd, wgd, xs, fctx3.dbeta_out))[0]
        return (da, None, None, None, dx) + (None,) * 17


def conv_bn_cat_relu(a, w3, stride3, padding3, g3, b3, rm3, rv3, nbt3, mom3, eps3,
                     x, wd, strided, paddingd, gd, bd, rmd, rvd, nbtd, momd, epsd):
    """``relu(batch_norm(conv1x1(a)) + batch_norm(conv1x1_stride_s(x)))`` (training mode) as one two-operand GEMM, or None
    when this is not a launch the fused path takes."""
    if not _BN_FWD_FOLD or mom3 is None or momd is None or rm3 is None or rmd is None:
        return None
    if getattr(a, "_dpa_deferred_coef", None) is not None and not getattr(a, "_dpa_deferred_relu", False):
        return None
    if getattr(x, "_dpa_deferred_coef", None) is not None:
        return None
    sd = tuple(strided)
    if tuple(stride3) != (1, 1) or sd[0] != sd[1] or sd[0] not in (1, 2) or tuple(paddingd) != (0, 0):
        return None
    if not _bn_fold_eligible(a, w3, None, stride3, padding3, False, True, True, False) \
            or not _bn_fold_eligible(x, wd, None, sd, paddingd, False, True, True, False):
        return None
    NB, IH, IW, C3 = a.shape
    N = w3.shape[0]
    if wd.shape[0] != N or N % 128 or x.dtype != a.dtype or not x.is_contiguous():
        return None
    if x.shape[0] != NB or (x.shape[1] + sd[0] - 1) // sd[0] != IH or (x.shape[2] + sd[0] - 1) // sd[0] != IW:
        return None
    if not all(t.requires_grad for t in (g3, b3, gd, bd)):
        return None
    grad_sink.note_use(w3, None)
    grad_sink.note_use(wd, None)
    grad_sink.note_use(g3, b3)
    grad_sink.note_use(gd, bd)
    if getattr(a, "_dpa_deferred_coef", None) is not None:
        a = materialize(a, want_colsum=True)
    a = a if a.is_contiguous() else a.contiguous()
    out = _ConvBnCatReluFn.apply(a, w3, g3, b3, x, wd, gd, bd, sd[0], rm3, rv3, nbt3, float(mom3), float(eps3),
                                 rmd, rvd, nbtd, float(momd), float(epsd), getattr(a, "_dpa_bn", None),
                                 bool(getattr(x, "_dpa_compact_ok", False)), getattr(a, "_dpa_colsum", None))
    if _ConvBnResReluFn.last is not None and torch.is_grad_enabled():
        out._dpa_bn = _ConvBnResReluFn.last
    _ConvBnResReluFn.last = None
    return out


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu, want_stats, needs_input_grad, skip, bnsrc=None,
                axf=None, relu_in=False):
        # axf: x is the RAW input of a training-mode BatchNorm + ReLU (a "deferred" BN output) and `axf` that BN's
        # coefficient table: forward and the weight gradient apply relu(scale * x + shift) while they stage the
        # operand, the normalised activation is never written (profiles/r2_notes.md)
        x_in = x
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cx = x.shape
        K, Cw, R, S = weight.shape
        if Cx < Cw or Cx % _vec(x.dtype):
            raise ValueError("conv2d: input channels %d incompatible with weight %s" % (Cx, tuple(weight.shape)))
        sh, sw = stride
        ph, pw = padding
        OH, OW = _conv_out(IH, R, sh, ph), _conv_out(IW, S, sw, pw)
        if OH <= 0 or OW <= 0:
            raise ValueError("conv2d: input %s too small for kernel %s" % (tuple(x.shape), (R, S)))
        wc = conv_weight(weight, x.dtype, Cx)
        y = torch.empty((NB, OH, OW, K), dtype=x.dtype, device=x.device)
        M = NB * OH * OW
        stats = None
        if want_stats:
            tiles = C().igemm_stats_rows(M, K, Cx, R * S, _dt(x), _dt(x))
            stats = _stats_slab(tiles, K, x.device)
        taps = _tap_table(x.device, _fprop_taps(R, S, ph, pw))
        bias_f = None
        if bias is not None:
            bias_f = bias.detach() if bias.dtype == torch.float32 else bias.detach().float()
        _igemm(x, wc, y, bias_f, stats, taps, M, K, Cx, R * S, Cx, R * S * Cx, K, IH, IW, sh, sw, OH, OW,
               relu=relu, dtype=x.dtype, out_dtype=x.dtype, row3=(S == 3 and sw == 1 and pw == 1), axf=axf)
        ctx.axf = axf
        # this convolution is the strided 1x1 branch of a block whose first convolution (the owner of the alias it
        # reads) can add a COMPACT gradient: backward then returns [NB, OH, OW, Cx] instead of scattering it into a
        # zero-filled full-size tensor (one fill + one scatter write + one full read less per downsample branch)
        ctx.compact = bool(getattr(x_in, "_dpa_compact_ok", False) and (R, S, ph, pw) == (1, 1, 0, 0) and sh == sw
                           and sh in (2, 4) and x is x_in and Cx == Cw and K % 32 == 0 and not relu)
        ctx.geom = (stride, padding, relu, needs_input_grad, bias is not None)
        ctx.bias_ref = bias
        # x is the output of a convolution + ReLU (VGG): the 3x3 stride-1 data gradient is stored through that ReLU's
        # mask and the producer skips its relu_bwd pass (_premasked)
        ctx.xmask = bool(relu_in and _DGRAD_XMASK and x is x_in and axf is None and x.dtype != torch.float32
                         and (sh, sw) == (1, 1) and S == 3 and pw == 1 and Cx == Cw and K % 32 == 0)
        # x is the output of a training-mode BatchNorm: the data-gradient epilogue can produce that BN's
        # backward partial sums (stride 1: one launch covers every input pixel exactly once)
        ctx.bnsrc = None
        if bnsrc is not None:
            bnsrc.uses += 1
            strided_ok = _FUSE_BN_BWD_STRIDED and (sh, sw) == (2, 2) and (R, S) == (3, 3) and (ph, pw) == (1, 1) \
                and IH % 2 == 0 and IW % 2 == 0 and bnsrc.mode != 1 and not bnsrc.fold
            if _FUSE_BN_BWD and x is x_in and ((sh, sw) == (1, 1) or strided_ok) and Cx == Cw \
                    and x.dtype != torch.float32 \
                    and bnsrc.x.shape == x.shape and bnsrc.x.dtype == x.dtype \
                    and _FUSE_BN_BWD_MIN <= x.numel() * x.element_size() <= _FUSE_BN_BWD_MAX:
                ctx.bnsrc = bnsrc
        ctx.save_for_backward(x, weight, y if relu else None)
        # no zero tensors for the outputs nobody differentiates (the statistics slab is up to 51 MB per layer)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        # skip=True also returns an alias of the input for the residual branch: its gradient comes back into
        # THIS node and is added inside the dgrad epilogue (no separate accumulate kernel, one rounding)
        alias = x.view(x.shape) if skip else None
        return y, stats, alias

    @staticmethod
    def backward(ctx, dy, dstats=None, dalias=None):
        amask = None
        if dalias is not None:              # gradient of the residual branch, possibly still carrying its ReLU mask
            ent = _pending_mask.pop(dalias.data_ptr(), None)
            if ent is not None and ent[1] == dalias._version and ent[2] == _generation \
                    and ent[0].numel() * _vec(dalias.dtype) == dalias.numel():
                amask = ent[0]
        compact = None
        if dalias is not None:
            ent = _pending_compact.pop(dalias.data_ptr(), None)
            if ent is not None and ent[3] == _generation and ent[0].data_ptr() == dalias.data_ptr() \
                    and tuple(dalias.shape) == ent[4]:
                compact = ent
        if dy is None:                      # only the alias was used downstream
            if compact is not None:
                dalias = _expand_compact(compact)
            if amask is not None:
                dalias = apply_bit_mask(dalias, amask)
            return dalias, None, None, None, None, None, None, None, None, None, None, None
        fent = _pending_fold.pop(dy.data_ptr(), None)
        if fent is not None:
            if fent[-1] != _generation or tuple(fent[1].shape) != tuple(dy.shape):
                raise RuntimeError("conv2d backward: stale BatchNorm-fold token")
            return _conv_bn_fold_backward(ctx, fent)
        x, weight, y = ctx.saved_tensors
        axf = ctx.axf
        (sh, sw), (ph, pw), relu, needs_input_grad, has_bias = ctx.geom
        NB, IH, IW, Cx = x.shape
        K, Cw, R, S = weight.shape
        OH, OW = dy.shape[1], dy.shape[2]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        M = NB * OH * OW
        st = _stream()
        if relu:
            ent = _premasked.pop(dy.data_ptr(), None)
            if ent is None or ent != (dy._version, _generation, y.data_ptr()):
                dym = torch.empty_like(dy)
                C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), st)
                dy = dym
        dx = dw = db = None
        wgrad_done = False
        _sk = grad_sink.sink_of(weight)
        if ctx.needs_input_grad[1] and _WGRAD_STREAM and dy.is_cuda and (_sk is None or _sk.on_ready is None):
            # weight gradient on the side stream, enqueued BEFORE the data gradient so the two run concurrently
            # (the dgrad -> BN chain is full of launches that leave most of the chip idle)
            side = side_stream(dy.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dw = conv_wgrad(dy, x, weight, (sh, sw), (ph, pw), axf=axf)
            # keep the operands alive until the side stream is past this launch.  NOT record_stream(): that defers
            # the blocks' reuse to an event poll at every later allocation, the main stream (far ahead of the side
            # stream) then misses its cache for every activation-sized tensor and falls back to hipMalloc -- measured
            # 380 instead of 84 ms per step at batch 1024
            ev = torch.cuda.Event()
            ev.record(side)
            _side_keep.append((ev, dy, x))
            while _side_keep and _side_keep[0][0].query():
                _side_keep.popleft()
            if dw is not None:              # handed back to autograd on the main stream
                torch.cuda.current_stream().wait_stream(side)
            wgrad_done = True
        src, ctx.bnsrc = ctx.bnsrc, None           # drop the reference: it pins the BN's input until the graph dies
        if ctx.needs_input_grad[0] and needs_input_grad:
            bnred = None
            if src is not None and src.uses == 1:                  # sole consumer of that BN's output
                bnred = (src.x, x if src.mode == 1 else (src.mask if src.mode == 3 else None), src.coef, src.mode,
                         bool(src.fold))
            if ctx.compact and dalias is None:
                # compact gradient: a plain GEMM over the output grid; autograd receives a NaN placeholder of the
                # input's shape (zero strides, no memory) that only the alias' owner knows how to replace -- anything
                # else that touched it would turn the loss into NaN instead of silently training on garbage
                dxc = torch.empty((NB, OH, OW, Cx), dtype=dy.dtype, device=dy.device)
                _igemm(dy, conv_weight_t(weight, dy.dtype), dxc, None, None, _tap_table(dy.device, ((0, 0, 0),)),
                       NB * OH * OW, Cx, K, 1, K, K, Cx, OH, OW, 1, 1, OH, OW, dtype=dy.dtype, out_dtype=dy.dtype)
                fake = torch.full((1,), float("nan"), dtype=dy.dtype, device=dy.device)
                dx = fake.expand(NB, IH, IW, Cx)
                _pending_compact[dx.data_ptr()] = (fake, dxc, (sh, sw), _generation, (NB, IH, IW, Cx))
            elif compact is not None:
                if amask is not None or (bnred is not None and bnred[3] == 1):
                    dalias, compact = _expand_compact(compact), None     # combinations the lean epilogue does not take
                    dx = conv_dgrad(dy, weight, (NB, IH, IW, Cx), (sh, sw), (ph, pw), accum=dalias, bnred=bnred,
                                    accum_mask=amask)
                else:
                    dx = conv_dgrad(dy, weight, (NB, IH, IW, Cx), (sh, sw), (ph, pw), accum=compact[1], bnred=bnred,
                                    accum_stride=compact[2])
            else:
                dx = conv_dgrad(dy, weight, (NB, IH, IW, Cx), (sh, sw), (ph, pw), accum=dalias, bnred=bnred,
                                accum_mask=amask, xmask=x if (ctx.xmask and bnred is None) else None)
        elif dalias is not None:
            if compact is not None:
                dalias = _expand_compact(compact)
            dx = apply_bit_mask(dalias, amask) if amask is not None else dalias
        if ctx.needs_input_grad[1] and not wgrad_done:
            dw = conv_wgrad(dy, x, weight, (sh, sw), (ph, pw), axf=axf)
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            db = _bias_grad(bias_p, dy.view(M, K))
        return dx, dw, db, None, None, None, None, None, None, None, None, None


# Fused BN-backward reduction in the data-gradient epilogue (DPA_FUSE_BN_BWD=0 disables).  First measured a wash
# (27.85 vs 27.65 ms/step at batch 256: the epilogue paid what the reduce kernel took); with streaming stores, the
# bit-mask ReLU and batch 1024 it is worth +0.6 % (profiles/r1_notes.md), so it is on.
# Side stream for weight gradients (single-process runs; see _ConvFn.backward).  join_side_stream() orders the
# current stream after everything enqueued there -- called before the optimizer reads the gradients.
_WGRAD_STREAM = os.environ.get("DPA_WGRAD_STREAM", "0") == "1"
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def side_stream(device) -> "torch.cuda.Stream":
    s = _side_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device)
        _side_streams[device.index] = s
        backend.register_stream("wgrad-side")
    return s


_side_keep = collections.deque()     # (event on the side stream, tensors its launch reads)


def join_side_stream() -> None:
    for s in _side_streams.values():
        torch.cuda.current_stream(s.device).wait_stream(s)
    _side_keep.clear()               # everything the main stream does from here on is ordered behind the side stream


_FUSE_BN_BWD = os.environ.get("DPA_FUSE_BN_BWD", "1") == "1"
# size window (MiB of the BN input) inside which the fusion is taken -- tuning hook
# the same fusion in the GROUPED data gradient (ResNeXt / RegNet 3x3): implemented and tested, off by default -- a
# group's reduction is only 9 * 64 channels deep, too short to hide the epilogue's extra operand loads: resnext50_32x4d
# 8,765 -> 8,675 img/s, resnext101_32x8d 3,019 -> 2,994 img/s with it on (profiles/r2/group_bnb_ab.txt)
_FUSE_BN_BWD_GROUPED = os.environ.get("DPA_FUSE_BN_BWD_GROUPED", "0") == "1"
# ... also in the parity-class launches of a STRIDED data gradient (3x3 stride 2: four launches, one shared slab).
# Implemented and tested, off: the twelve launches of ResNet-50 lose more (general-kernel epilogue) than the three
# stand-alone reductions cost -- 65.5 vs 65.2 ms/step (profiles/r3_notes.md)
_FUSE_BN_BWD_STRIDED = os.environ.get("DPA_FUSE_BN_BWD_STRIDED", "0") == "1"
_FUSE_BN_BWD_MIN = int(float(os.environ.get("DPA_FUSE_BN_BWD_MIN_MB", "0")) * 2 ** 20)
_FUSE_BN_BWD_MAX = int(float(os.environ.get("DPA_FUSE_BN_BWD_MAX_MB", "1e9")) * 2 ** 20)
# data_ptr of a freshly produced data gradient -> (partial slab, data_ptr of the BN input it was reduced
# against, tensor version).  BatchNorm backward uses the slab only if the gradient it receives is that very
# tensor, unmodified: autograd sums gradients of a multiply-used tensor either out of place (new data_ptr)
# or in place (version bump), and both must fall back to the full reduction.
_pending_bnred: Dict[int, tuple] = {}
# data_ptr of a gradient handed to a residual branch WITHOUT its ReLU mask applied -> (bit mask, version, generation).
# Only produced for residuals tagged `_dpa_alias` (whose sole consumer is _ConvFn.backward, which applies the mask
# inside the data-gradient epilogue): saves writing the masked copy (one activation-sized tensor per block).
_pending_mask: Dict[int, tuple] = {}
# data_ptr of the NaN placeholder a strided 1x1 branch returned as its input gradient -> (placeholder, compact gradient
# [NB, OH, OW, C], (sy, sx), generation, full shape).  Consumed by the convolution that owns the alias the branch read.
_pending_compact: Dict[int, tuple] = {}
# data_ptr of a max-pool's input gradient that ALREADY carries the ReLU mask of the convolution that produced the pool's
# input -> (version, generation, data_ptr of that convolution's output).  relu(x) followed by max-pool: a window's
# arg-max is inactive exactly when the pooled value is 0, so the mask can be applied to the (4x or more smaller) pooled
# gradient instead of the full-size one; the convolution's backward then skips its own relu_bwd pass (VGG, AlexNet).
# (What autograd carries between the two nodes is then the gradient of the PRE-activation, not of the ReLU output: only a
# hook / retain_grad() on that intermediate tensor can tell; any other consumer of it makes autograd hand the producer a
# SUM, a new tensor, which matches no entry and takes the ordinary relu_bwd pass -- masking twice is idempotent.)
_premasked: Dict[int, tuple] = {}
_POOL_PREMASK = os.environ.get("DPA_POOL_PREMASK", "1") == "1"
_DGRAD_XMASK = os.environ.get("DPA_DGRAD_XMASK", "1") == "1"     # the same for conv + ReLU -> 3x3 conv (dgrad epilogue)
_COMPACT_DRES = os.environ.get("DPA_COMPACT_DRES", "1") == "1"


def _expand_compact(ent) -> torch.Tensor:
    """Cold paths: the full-size gradient a compact entry stands for."""
    _, dxc, (sy, sx), _, shape = ent
    full = torch.zeros(shape, dtype=dxc.dtype, device=dxc.device)
    full[:, ::sy, ::sx] = dxc
    return full
_LAZY_DRES = os.environ.get("DPA_LAZY_DRES", "1") == "1"


class _LazyTok:
    """One per residual BatchNorm + ReLU call that may hand back an UNMASKED residual gradient.  A tensor hook on the
    residual sees the gradient autograd is about to deliver to the residual's producer: when the residual had a
    second consumer that is a SUM of several gradients (a new tensor, or an in-place accumulation), the pending
    mask can no longer be applied by the producer, so the hook repairs the sum here (cold path)."""
    __slots__ = ("ent",)

    def __init__(self):
        self.ent = None         # (dy as handed back, bit mask, dy._version)


def _lazy_res_hook(tok: "_LazyTok", grad: torch.Tensor):
    ent, tok.ent = tok.ent, None
    if ent is None:
        return None
    dy, mask, ver = ent
    if grad.data_ptr() == dy.data_ptr() and grad._version == ver and grad.shape == dy.shape:
        return None             # the only consumer: the producer pops _pending_mask and masks in its own kernel
    _pending_mask.pop(dy.data_ptr(), None)
    # grad = (unmasked dy) + others  ->  (masked dy) + others
    return grad - (dy - apply_bit_mask(dy, mask))


bn_bwd_counts = {"fused": 0, "reduced": 0}      # how BatchNorm backward obtained its partial sums (tests, tuning)


class _BnSrc:
    """Attached (as ``_dpa_bn``) to the output of a training-mode BatchNorm: what a consuming convolution needs
    to reduce that BN's backward sums inside its data-gradient epilogue."""
    __slots__ = ("x", "coef", "mode", "uses", "mask", "fold")

    def __init__(self, x, coef, mode, mask=None, fold=False):
        self.x, self.coef, self.mode, self.uses, self.mask = x, coef, mode, 0, mask
        self.fold = fold        # the BN's backward is folded into its producing 1x1 convolution (see _conv_bn_fold_backward)


def apply_bit_mask(t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """t * mask for a bit mask written by the BN apply pass (one byte per 16-byte chunk) -- cold paths only."""
    vec = _vec(t.dtype)
    bits = (mask.view(-1, 1) >> torch.arange(vec, device=t.device, dtype=torch.uint8)) & 1
    return (t.contiguous().view(-1, vec) * bits.to(t.dtype)).view(t.shape)


def conv_dgrad(dy, weight, xshape, stride, padding, accum=None, bnred=None, accum_mask=None, accum_stride=None,
               xmask=None, wt_override=None):
    NB, IH, IW, Cx = xshape
    K, Cw, R, S = weight.shape
    sh, sw = stride
    ph, pw = padding
    OH, OW = dy.shape[1], dy.shape[2]
    wt = conv_weight_t(weight, dy.dtype) if wt_override is None else wt_override          # [Cw, R, S, K]
    if Cx != Cw:
        raise ValueError("dgrad into a channel-padded input is not supported (stem has no input grad)")
    classes = _dgrad_classes(R, S, sh, sw, ph, pw)
    need_zero = any(len(t) == 0 for _, _, t in classes)
    rstride = None
    if accum_stride is not None:         # compact gradient of a strided 1x1 branch (see _ConvFn.backward)
        if (R, S, sh, sw, ph, pw) != (1, 1, 1, 1, 0, 0) or accum is None or accum_mask is not None:
            raise ValueError("compact strided accumulate: 1x1 stride-1 data gradients only")
        rstride = (accum.shape[1], accum.shape[2], accum_stride[0], accum_stride[1])
    if accum is not None:
        accum = accum if accum.is_contiguous() else accum.contiguous()
    if accum_mask is not None and (need_zero or accum is None):
        accum = apply_bit_mask(accum, accum_mask) if accum is not None else None
        accum_mask = None
    if need_zero and accum is not None:
        dx = accum.clone()                  # uncovered positions keep the other branch's gradient
        accum = dx                          # covered ones accumulate in place (same thread reads then writes)
    else:
        dx = (torch.zeros if need_zero else torch.empty)((NB, IH, IW, Cx), dtype=dy.dtype, device=dy.device)
    row3 = S == 3 and (sh, sw) == (1, 1) and pw == 1 and OW == IW
    # fused BN-backward sums: one launch that covers every input pixel once, or -- strided convolutions -- parity classes
    # that do so together (every class non-empty): each class leaves its rows in one shared slab
    fuse_bnb = bnred is not None and K % 32 == 0 and not need_zero and \
        (len(classes) == 1 or (_FUSE_BN_BWD_STRIDED and all(len(t) > 0 for _, _, t in classes)))
    slab_all, slab_row = None, 0
    if fuse_bnb and len(classes) > 1:
        total = 0
        for py, px, taps in classes:
            SH, SW = (IH - py + sh - 1) // sh, (IW - px + sw - 1) // sw
            if SH <= 0 or SW <= 0:
                fuse_bnb = False
                break
            total += C().igemm_stats_rows(NB * SH * SW, Cx, K, len(taps), _dt(dy), _dt(dy))
        if fuse_bnb:
            slab_all = torch.empty((total, 2, Cx), dtype=torch.float32, device=dy.device)
    for py, px, taps in classes:
        if not taps:
            continue
        SH, SW = (IH - py + sh - 1) // sh, (IW - px + sw - 1) // sw   # sub-grid extent
        if SH <= 0 or SW <= 0:
            continue
        if row3:
            taps = tuple(sorted(taps))          # filter rows: dy, then dx ascending (-1, 0, 1)
        tt = _tap_table(dy.device, taps)
        scatter = None if (sh == 1 and sw == 1) else (IH, IW, sh, sw, py, px)
        slab = None
        if fuse_bnb:
            rows = C().igemm_stats_rows(NB * SH * SW, Cx, K, len(taps), _dt(dy), _dt(dy))
            if slab_all is not None:
                slab = slab_all[slab_row:slab_row + rows]
                slab_row += rows
            else:
                slab = torch.empty((rows, 2, Cx), dtype=torch.float32, device=dy.device)
        # dx is the gradient of a ReLU output (xmask): store it through that ReLU's mask (filter-row launches)
        xm = xmask if (xmask is not None and slab is None and row3 and scatter is None and len(classes) == 1
                       and not need_zero and rstride is None and Cx % _vec(dy.dtype) == 0
                       and tuple(xmask.shape) == (NB, IH, IW, Cx) and xmask.is_contiguous()) else None
        _igemm(dy, wt, dx, None, slab, tt, NB * SH * SW, Cx, K, len(taps), K, R * S * K, Cx, OH, OW, 1, 1,
               SH, SW, scatter=scatter, dtype=dy.dtype, out_dtype=dy.dtype, accum=accum,
               bnred=bnred if slab is not None else None, row3=row3, accum_mask=accum_mask, rstride=rstride, xmask=xm)
        if slab is not None:
            _pending_bnred[dx.data_ptr()] = (slab_all if slab_all is not None else slab, bnred[0].data_ptr(), dx._version,
                                             len(bnred) > 4 and bool(bnred[4]))
        if xm is not None:
            _premasked[dx.data_ptr()] = (dx._version, _generation, xm.data_ptr())
    return dx


def conv_wgrad(dy, x, weight, stride, padding, axf=None):
    NB, IH, IW, Cx = x.shape
    K, Cw, R, S = weight.shape
    sh, sw = stride
    ph, pw = padding
    OH, OW = dy.shape[1], dy.shape[2]
    M = NB * OH * OW
    Ktot = R * S * Cx
    st = _stream()
    splits = 0
    if S == 3 and (sh, sw) == (1, 1) and pw == 1 and (OH, OW) == (IH, IW) and axf is None and x.is_contiguous():
        # filter-row kernel: one staged activation row tile per filter ROW instead of per tap (csrc/kernels/wgrad_row3.hip)
        splits, rps = C().wgrad_row3_splits(NB, IH, IW, K, Cx, R, _dt(dy))
    dense8 = gather8 = 0
    if axf is None and x.is_contiguous() and dy.is_contiguous() and splits == 0 \
            and (x.data_ptr() | dy.data_ptr()) % 16 == 0:
        # wide layers: the 256 x 256 LDS-DMA pipeline (csrc/kernels/wgrad8.hip), dense or with gathered taps
        if R * S == 1 and (sh, sw) == (1, 1) and ph == 0 and pw == 0:
            dense8, pps8 = C().wgrad8_splits(M, K, 0, Cx, _dt(dy))
        else:
            gather8, pps8 = C().wgrad8_splits(M, K, 0, Cx, _dt(dy), R * S, True)
    if splits > 0:
        slab = torch.empty((splits, K, Ktot), dtype=torch.float32, device=dy.device)
        C().wgrad_row3(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), NB, IH, IW, K, Cx, R, ph, K, Cx, splits, rps,
                       _dt(dy), st)
    elif dense8 > 0:
        splits = dense8
        slab = torch.empty((splits, K, Ktot), dtype=torch.float32, device=dy.device)
        C().wgrad8(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), M, K, Cx, K, Cx, splits, pps8, _dt(dy), st)
    elif gather8 > 0:
        splits = gather8
        slab = torch.empty((splits, K, Ktot), dtype=torch.float32, device=dy.device)
        taps = _tap_table(dy.device, _fprop_taps(R, S, ph, pw))
        C().wgrad(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, K, Cx, R * S, K, Cx, IH, IW,
                  sh, sw, OH, OW, splits, pps8, 0, _dt(dy), st, use8=1)
    else:
        splits, pps = C().wgrad_choose_splits(M, K, Ktot, _dt(dy))
        slab = torch.empty((splits, K, Ktot), dtype=torch.float32, device=dy.device)
        taps = _tap_table(dy.device, _fprop_taps(R, S, ph, pw))
        C().wgrad(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, K, Cx, R * S, K, Cx, IH, IW,
                  sh, sw, OH, OW, splits, pps, 1 if (R * S == 1 and ph == 0 and pw == 0) else 0, _dt(dy), st,
                  1, 0, 0, axf[2].data_ptr() if axf is not None else 0, axf[3].data_ptr() if axf is not None else 0)
    gdtype = weight.dtype
    sink = grad_sink.sink_of(weight)
    if Cx == Cw and sink is not None and _flat_same_layout(sink.view, weight) \
            and (weight.is_contiguous(memory_format=torch.channels_last) or R * S == 1):
        # the slab fold writes the gradient straight into the parameter's bucket view
        C().slab_reduce(slab.data_ptr(), splits, K * Ktot, sink.view.data_ptr(), _dt(gdtype), sink.beta(), st)
        sink.delivered()
        return None
    if Cx == Cw:
        g = torch.empty((K, R, S, Cw), dtype=gdtype, device=dy.device)
        C().slab_reduce(slab.data_ptr(), splits, K * Ktot, g.data_ptr(), _dt(gdtype), 0.0, st)
    else:
        full = torch.empty((K, R, S, Cx), dtype=torch.float32, device=dy.device)
        C().slab_reduce(slab.data_ptr(), splits, K * Ktot, full.data_ptr(), 0, 0.0, st)
        g = torch.empty((K, R, S, Cw), dtype=gdtype, device=dy.device)
        # drop the padded channels: "NCHW -> NHWC" over K*R*S one-pixel images keeps the first Cw of Cx
        _slice_channels(full, g, Cw)
    return _deliver(weight, g.permute(0, 3, 1, 2))       # logical [K, C, R, S] with channels_last strides


def _slice_channels(src: torch.Tensor, dst: torch.Tensor, keep: int) -> None:
    """dst[..., :keep] = src[..., :keep] for tiny tensors (stem weight gradient)."""
    n = src.numel() // src.shape[-1]
    # [n][Cx] -> transpose to [Cx][n] -> take first `keep` rows -> transpose back
    Cx = src.shape[-1]
    tmp = torch.empty((Cx, n), dtype=torch.float32, device=src.device)
    C().transpose_ktc(src.data_ptr(), tmp.data_ptr(), n, 1, Cx, 0, 0, _stream())
    C().transpose_ktc(tmp.data_ptr(), dst.data_ptr(), keep, 1, n, 0, _dt(dst), _stream())


_AXF = os.environ.get("DPA_BN_APPLY_ON_LOAD", "1") == "1"
# Apply-on-load re-does the BatchNorm arithmetic every time an activation element is staged: once per output-channel
# tile and filter tap (row) in forward, once per output-channel tile and tap in the weight gradient.  Beyond this many
# stagings per element the stand-alone bn_apply pass (one read + one write) is cheaper (profiles/r2_notes.md).
_AXF_MAX_RED = float(os.environ.get("DPA_AXF_MAX_RED", "8"))


def _axf_redundancy(K: int, R: int, S: int, stride) -> float:
    ntile = (K + 127) // 128
    row3 = S == 3 and stride[1] == 1
    return ntile * (R * S / (3.0 if row3 else 1.0)) + ntile * R * S


class _MaterializeFn(torch.autograd.Function):
    """relu(scale * x + shift) of a deferred BatchNorm + ReLU for a consumer that cannot apply it on load.  The
    BatchNorm's own backward recomputes the ReLU mask from x, so the gradient passes through unchanged."""

    last_colsum = None

    @staticmethod
    def forward(ctx, x, coef, want_colsum=False):
        y = torch.empty_like(x)
        Cc = x.shape[-1]
        M = x.numel() // Cc
        _MaterializeFn.last_colsum = None
        nparts = C().bn_apply_colsum_parts(M, Cc, _dt(x)) if (want_colsum and x.is_contiguous()) else 0
        if nparts > 0:
            # the same pass also leaves the column sums of what it writes (the forward fold's colsum(a))
            part = torch.empty((nparts, 2, Cc), dtype=torch.float32, device=x.device)
            C().bn_apply_colsum(x.data_ptr(), y.data_ptr(), M, Cc, _dt(x), coef[2].data_ptr(), coef[3].data_ptr(),
                                part.data_ptr(), nparts, _stream())
            s1 = torch.empty(Cc, dtype=torch.float32, device=x.device)
            s2 = torch.empty(Cc, dtype=torch.float32, device=x.device)
            C().bn_bwd_finalize(part.data_ptr(), nparts, Cc, s2.data_ptr(), s1.data_ptr(), _stream())
            _MaterializeFn.last_colsum = s1
            return y
        C().bn_apply(x.data_ptr(), 0, y.data_ptr(), M, Cc, _dt(x), coef[2].data_ptr(), coef[3].data_ptr(),
                     1, _stream(), 0, 0, 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


def materialize(x: torch.Tensor, want_colsum: bool = False) -> torch.Tensor:
    """The normalised activation behind a deferred BatchNorm + ReLU output (identity for ordinary tensors).
    ``want_colsum``: the pass also reduces the column sums of its result (``y._dpa_colsum``, fp32 [C])."""
    coef = getattr(x, "_dpa_deferred_coef", None)
    if coef is None:
        return x
    if not getattr(x, "_dpa_deferred_relu", False):
        raise ValueError("a deferred activation-free BatchNorm output can only be the `residual` of another BatchNorm")
    y = _MaterializeFn.apply(x, coef, bool(want_colsum and _COLSUM_IN_APPLY))
    if _MaterializeFn.last_colsum is not None:
        y._dpa_colsum = _MaterializeFn.last_colsum
        _MaterializeFn.last_colsum = None
    bn = getattr(x, "_dpa_bn", None)
    if bn is not None:
        y._dpa_bn = bn
    return y


def _is_relu_out(x) -> bool:
    """x is still exactly what a convolution + ReLU of this module wrote: tagged, and not modified in place since
    (a stale tag would mask the consumer's gradient with values the ReLU never produced)."""
    tag = getattr(x, "_dpa_relu_out", None)
    return tag is not None and tag is not False and x._version == tag


def conv2d(x, weight, bias, stride, padding, relu, w_lp, want_stats, needs_input_grad, skip=False):
    grad_sink.note_use(weight, bias)
    axf = getattr(x, "_dpa_deferred_coef", None)
    if axf is not None:
        K, Cw = weight.shape[0], weight.shape[1]
        NB, IH, IW, Cx = x.shape
        ok = getattr(x, "_dpa_deferred_relu", False) and not skip and x.is_contiguous() and Cx == Cw \
            and _axf_redundancy(K, weight.shape[2], weight.shape[3], stride) <= _AXF_MAX_RED \
            and C().igemm_axf_ok(_conv_out(IH, weight.shape[2], stride[0], padding[0])
                                 * _conv_out(IW, weight.shape[3], stride[1], padding[1]), IH, IW, Cx, K, Cx,
                                 weight.shape[2] * weight.shape[3] * Cx, _dt(x), _dt(x), 1)
        if not ok:
            x, axf = materialize(x), None
    fold_ok = _bn_fold_eligible(x, weight, bias, stride, padding, relu, want_stats, needs_input_grad, skip)
    if fold_ok and axf is not None:
        x, axf = materialize(x), None       # the folded backward runs GEMMs on the normalised activation itself
    y, stats, alias = _ConvFn.apply(x, weight, bias, tuple(stride), tuple(padding), bool(relu), bool(want_stats),
                                    bool(needs_input_grad), bool(skip), getattr(x, "_dpa_bn", None), axf,
                                    _is_relu_out(x))
    if fold_ok:
        y._dpa_fold_ok = True
    if alias is not None:
        # its gradient returns to _ConvFn.backward (and nowhere else): a BatchNorm that takes it as the residual may
        # hand back the UNMASKED gradient plus its ReLU bit mask instead of writing a masked copy
        alias._dpa_alias = True
        K_, Cw_, R_, S_ = weight.shape
        # the owner can add a compact strided gradient in its data-gradient epilogue (lean GEMM, 16-bit, dense 1x1)
        alias._dpa_compact_ok = bool(_COMPACT_DRES and (R_, S_) == (1, 1) and tuple(stride) == (1, 1)
                                     and tuple(padding) == (0, 0) and x.dtype in (torch.bfloat16, torch.float16)
                                     and x.shape[-1] == Cw_ and Cw_ % 64 == 0 and K_ % 32 == 0 and x.is_contiguous()
                                     and needs_input_grad)
    if relu:
        # a max-pool / 3x3 conv that consumes it may hand back an already masked gradient (_premasked); the tag holds
        # the tensor's version so an in-place edit between producer and consumer (y.sub_(1)) voids it
        y._dpa_relu_out = y._version
    out = (y, stats) if want_stats else y
    return (out, alias) if skip else out


# ------------------------------------------------------------------------------------------
# grouped convolution (ResNeXt cardinality, depthwise): ONE launch per pass, grid.y = group
# ------------------------------------------------------------------------------------------
_GROUP_ALIGN = int(os.environ.get("DPA_GROUP_ALIGN", "0"))


def group_merge(groups: int, Cg: int, Kg: int, vec: int) -> int:
    """How many consecutive groups one MFMA tile column block covers.

    The matrix cores want >= 64 output channels per tile and 16-byte channel vectors; a group with 4 (ResNeXt
    32x4d) or 1 (depthwise) channels offers neither.  ``m`` consecutive groups are therefore computed as one
    block-diagonal group of ``m*Cg -> m*Kg`` channels (zeros off the diagonal).  A tile computes 64 output
    channels whether they are used or not, so MFMA work is ``~ roundup(m*Kg, 64) * Cg`` per input channel group:
    merging is free up to 64 output channels and only then starts to cost.  Hence: the LARGEST divisor of
    ``groups`` with ``m*Kg <= 64`` whose slices are 16-byte aligned (fewest, fullest tiles); if there is none
    (wide groups, or alignment needs more), the smallest aligned divisor."""
    ok = [m for m in range(1, groups + 1) if groups % m == 0 and (m * Cg) % vec == 0 and (m * Kg) % vec == 0]
    if not ok:
        raise ValueError("grouped conv2d: %d groups of %d -> %d channels cannot be laid out in 16-byte vectors"
                         % (groups, Cg, Kg))
    fit = [m for m in ok if m * Kg <= 64]
    best = max(fit) if fit else min(ok)
    if _GROUP_ALIGN > 0 and vec == 8 and ((best * Cg) % 32 or (best * Kg) % 32):
        # 64-byte K tiles (the LDS-DMA "aligned" kernels) need 32-channel slices in BOTH directions (forward reduces
        # over m*Cg, the data gradient over m*Kg): RegNet group widths 24 / 56 / 112 miss that and fall to the
        # gather kernels.  Trade block-diagonal zeros for the fast path, up to _GROUP_ALIGN merged channels.
        # Measured a loss (regnet_y_8gf, 56-wide groups merged by 4: 4,074 -> 3,607 img/s), so off by default.
        al = [m for m in ok if (m * Cg) % 32 == 0 and (m * Kg) % 32 == 0 and max(m * Cg, m * Kg) <= _GROUP_ALIGN]
        if al:
            best = min(al)
    return best


def _gw_src(w: torch.Tensor, Gs: int, m: int) -> torch.Tensor:
    K, Cg, R, S = w.shape
    return w.detach().view(Gs, m, K // (Gs * m), Cg, R, S)


def group_weight(w: torch.Tensor, dtype: torch.dtype, groups: int, m: int) -> torch.Tensor:
    """fprop operand [K, R, S, m*Cg]: row k holds its group's filter at channel offset (group % m) * Cg."""
    if m == 1:
        return conv_weight(w, dtype, w.shape[1])
    K, Cg, R, S = w.shape
    c = _cache(w)
    key = (dtype, "g", m)
    if c.wc is not None and c.wc_key == key and c.wc_stamp == _stamp(w):
        return c.wc
    Gs, Kg = groups // m, K // groups
    if c.wc is None or c.wc_key != key:
        c.wc = torch.zeros((Gs, m, Kg, R, S, m, Cg), dtype=dtype, device=w.device)
    # only the diagonal blocks are ever written; the zeros around them persist across refreshes
    torch.diagonal(c.wc, dim1=1, dim2=5).copy_(_gw_src(w, Gs, m).permute(0, 2, 4, 5, 3, 1))
    c.wc_key, c.wc_stamp = key, _stamp(w)
    return c.wc


def group_weight_t(w: torch.Tensor, dtype: torch.dtype, groups: int, m: int) -> torch.Tensor:
    """dgrad operand [C, R, S, m*Kg] (input channel major, the merged group's output channels contiguous)."""
    K, Cg, R, S = w.shape
    c = _cache(w)
    key = (dtype, "g", m)
    if c.wt is not None and c.wt_key == key and c.wt_stamp == _stamp(w):
        return c.wt
    Gs, Kg = groups // m, K // groups
    if c.wt is None or c.wt_key != key:
        c.wt = torch.zeros((Gs, m, Cg, R, S, m, Kg), dtype=dtype, device=w.device)
    torch.diagonal(c.wt, dim1=1, dim2=5).copy_(_gw_src(w, Gs, m).permute(0, 3, 4, 5, 2, 1))
    c.wt_key, c.wt_stamp = key, _stamp(w)
    return c.wt


class _GroupConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, groups, relu, want_stats, needs_input_grad, bnsrc=None):
        x_in = x
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cx = x.shape
        K, Cg, R, S = weight.shape
        if Cx != Cg * groups or K % groups:
            raise ValueError("conv2d: input channels %d / weight %s / groups %d do not match"
                             % (Cx, tuple(weight.shape), groups))
        Kg = K // groups
        m = group_merge(groups, Cg, Kg, _vec(x.dtype))
        Gs, Cin, Kout = groups // m, m * Cg, m * Kg
        sh, sw = stride
        ph, pw = padding
        OH, OW = _conv_out(IH, R, sh, ph), _conv_out(IW, S, sw, pw)
        if OH <= 0 or OW <= 0:
            raise ValueError("conv2d: input %s too small for kernel %s" % (tuple(x.shape), (R, S)))
        wc = group_weight(weight, x.dtype, groups, m)
        y = torch.empty((NB, OH, OW, K), dtype=x.dtype, device=x.device)
        M = NB * OH * OW
        stats = None
        if want_stats:
            tiles = C().igemm_stats_rows(M, Kout, Cin, R * S, _dt(x), _dt(x))
            stats = _stats_slab(tiles, K, x.device)
        taps = _tap_table(x.device, _fprop_taps(R, S, ph, pw))
        bias_f = None
        if bias is not None:
            bias_f = bias.detach() if bias.dtype == torch.float32 else bias.detach().float()
        _igemm(x, wc, y, bias_f, stats, taps, M, Kout, Cin, R * S, Cx, R * S * Cin, K, IH, IW, sh, sw, OH, OW,
               relu=relu, dtype=x.dtype, out_dtype=x.dtype, groups=(Gs, Cin, Kout * R * S * Cin, Kout, K),
               row3=(S == 3 and sw == 1 and pw == 1))
        ctx.geom = (stride, padding, groups, m, relu, needs_input_grad, bias is not None)
        ctx.bias_ref = bias
        # x is the output of a training-mode BatchNorm (ResNeXt / RegNet bottlenecks): the stride-1 data gradient
        # reduces that BN's backward sums in its epilogue, one group slice per launch row (see _ConvFn.forward)
        ctx.bnsrc = None
        if bnsrc is not None:
            bnsrc.uses += 1
            if _FUSE_BN_BWD and _FUSE_BN_BWD_GROUPED and x is x_in and (sh, sw) == (1, 1) \
                    and x.dtype != torch.float32 and Kout % 32 == 0 \
                    and Cin % _vec(x.dtype) == 0 and bnsrc.x.shape == x.shape and bnsrc.x.dtype == x.dtype \
                    and _FUSE_BN_BWD_MIN <= x.numel() * x.element_size() <= _FUSE_BN_BWD_MAX:
                ctx.bnsrc = bnsrc
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, dstats=None):
        src, ctx.bnsrc = ctx.bnsrc, None
        if dy is None:
            return (None,) * 10
        x, weight, y = ctx.saved_tensors
        (sh, sw), (ph, pw), groups, m, relu, needs_input_grad, has_bias = ctx.geom
        NB, IH, IW, Cx = x.shape
        K, Cg, R, S = weight.shape
        Kg = K // groups
        Gs, Cin, Kout = groups // m, m * Cg, m * Kg
        OH, OW = dy.shape[1], dy.shape[2]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        M = NB * OH * OW
        st = _stream()
        if relu:
            dym = torch.empty_like(dy)
            C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), st)
            dy = dym
        dx = dw = db = None
        if ctx.needs_input_grad[0] and needs_input_grad:
            wt = group_weight_t(weight, dy.dtype, groups, m)
            classes = _dgrad_classes(R, S, sh, sw, ph, pw)
            need_zero = any(len(t) == 0 for _, _, t in classes)
            dx = (torch.zeros if need_zero else torch.empty)((NB, IH, IW, Cx), dtype=dy.dtype, device=dy.device)
            row3 = S == 3 and (sh, sw) == (1, 1) and pw == 1 and OW == IW       # filter-row kernel (see conv_dgrad)
            for py, px, taps in classes:
                SH, SW = (IH - py + sh - 1) // sh, (IW - px + sw - 1) // sw
                if not taps or SH <= 0 or SW <= 0:
                    continue
                if row3:
                    taps = tuple(sorted(taps))
                scatter = None if (sh == 1 and sw == 1) else (IH, IW, sh, sw, py, px)
                slab = bnred = None
                if src is not None and src.uses == 1 and scatter is None and len(classes) == 1:
                    bnred = (src.x, x if src.mode == 1 else (src.mask if src.mode == 3 else None), src.coef, src.mode)
                    rows = C().igemm_stats_rows(NB * SH * SW, Cin, Kout, len(taps), _dt(dy), _dt(dy))
                    slab = torch.empty((rows, 2, Cx), dtype=torch.float32, device=dy.device)
                _igemm(dy, wt, dx, None, slab, _tap_table(dy.device, taps), NB * SH * SW, Cin, Kout, len(taps), K,
                       R * S * Kout, Cx, OH, OW, 1, 1, SH, SW, scatter=scatter, dtype=dy.dtype, out_dtype=dy.dtype,
                       groups=(Gs, Kout, Cin * R * S * Kout, Cin, Cx if slab is not None else 0), bnred=bnred,
                       row3=row3)
                if slab is not None:
                    _pending_bnred[dx.data_ptr()] = (slab, src.x.data_ptr(), dx._version)
        if ctx.needs_input_grad[1]:
            Ktot = R * S * Cin
            splits, pps = C().wgrad_choose_splits(M, Kout, Ktot, _dt(dy), Gs)
            slab = torch.empty((splits, K, Ktot), dtype=torch.float32, device=dy.device)
            taps = _tap_table(dy.device, _fprop_taps(R, S, ph, pw))
            C().wgrad(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, Kout, Cin, R * S, K, Cx, IH, IW,
                      sh, sw, OH, OW, splits, pps, 1 if (R * S == 1 and ph == 0 and pw == 0) else 0, _dt(dy), st,
                      Gs, Kout, Cin)
            sink = grad_sink.sink_of(weight)
            if m == 1 and sink is not None and _flat_same_layout(sink.view, weight) \
                    and (weight.is_contiguous(memory_format=torch.channels_last) or R * S == 1):
                C().slab_reduce(slab.data_ptr(), splits, K * Ktot, sink.view.data_ptr(), _dt(weight.dtype),
                                sink.beta(), st)
                sink.delivered()
            else:
                full = torch.empty((Gs, m, Kg, R, S, m, Cg), dtype=torch.float32, device=dy.device)
                C().slab_reduce(slab.data_ptr(), splits, K * Ktot, full.data_ptr(), 0, 0.0, st)
                # the diagonal blocks are the gradient; the rest multiplied structural zeros
                g = torch.diagonal(full, dim1=1, dim2=5).permute(0, 5, 1, 4, 2, 3).reshape(K, Cg, R, S)
                dw = _deliver(weight, g.to(weight.dtype))
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            db = _bias_grad(bias_p, dy.view(M, K))
        return dx, dw, db, None, None, None, None, None, None, None


_DW_DIRECT = os.environ.get("DPA_DW_DIRECT", "1") == "1"


def _dw_filter(weight):
    """The depthwise filter as the kernels read it: memory [C][KH][KW] (fp32 master or compute dtype)."""
    K, _, R, S = weight.shape
    w = weight.detach()
    if w.stride(0) != R * S or w.stride(2) != S or w.stride(3) != 1:
        w = w.reshape(K, R * S).contiguous()
    return w


class _DepthwiseConvFn(torch.autograd.Function):
    """groups == channels: direct bandwidth kernels (``csrc/kernels/depthwise.hip``) instead of block-diagonal MFMA
    tiles.  The filter is read from the parameter itself; the weight gradient is folded from one partial row per
    workgroup straight into the parameter's gradient sink."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu, want_stats, needs_input_grad):
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cx = x.shape
        K, _, R, S = weight.shape
        sh, sw = stride
        ph, pw = padding
        OH, OW = _conv_out(IH, R, sh, ph), _conv_out(IW, S, sw, pw)
        if OH <= 0 or OW <= 0:
            raise ValueError("conv2d: input %s too small for kernel %s" % (tuple(x.shape), (R, S)))
        w = _dw_filter(weight)
        y = torch.empty((NB, OH, OW, K), dtype=x.dtype, device=x.device)
        stats = None
        if want_stats:
            stats = _stats_slab(C().dw_conv_rows(NB * OH * OW, K, _dt(x)), K, x.device)
        bias_f = None if bias is None else _f32(bias.detach()).contiguous()
        C().dw_conv(x.data_ptr(), w.data_ptr(), _dt(w), _p(bias_f), y.data_ptr(), _p(stats), 0, NB, IH, IW, K, OH, OW,
                    R, S, sh, sw, ph, pw, 1 if relu else 0, _dt(x), _stream())
        ctx.geom = (stride, padding, relu, needs_input_grad, bias is not None)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, dstats=None):
        if dy is None:
            return (None,) * 8
        x, weight, y = ctx.saved_tensors
        (sh, sw), (ph, pw), relu, needs_input_grad, has_bias = ctx.geom
        NB, IH, IW, Cx = x.shape
        K, _, R, S = weight.shape
        OH, OW = dy.shape[1], dy.shape[2]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        st = _stream()
        if relu:
            dym = torch.empty_like(dy)
            C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), st)
            dy = dym
        dx = dw = db = None
        if ctx.needs_input_grad[0] and needs_input_grad:
            w = _dw_filter(weight)
            dx = torch.empty((NB, IH, IW, Cx), dtype=dy.dtype, device=dy.device)
            C().dw_conv(dy.data_ptr(), w.data_ptr(), _dt(w), 0, dx.data_ptr(), 0, 1, NB, IH, IW, K, OH, OW, R, S, sh, sw,
                        ph, pw, 0, _dt(dy), st)
        if ctx.needs_input_grad[1]:
            rows = C().dw_wgrad_rows(NB * OH * OW, K, R, _dt(dy))
            partial = torch.empty((rows, K, R * S), dtype=torch.float32, device=dy.device)
            C().dw_wgrad(dy.data_ptr(), x.data_ptr(), partial.data_ptr(), NB, IH, IW, K, OH, OW, R, S, sh, sw, ph, pw,
                         _dt(dy), st)
            sink = grad_sink.sink_of(weight)
            if sink is not None and _flat_same_layout(sink.view, weight) and weight.stride(0) == R * S \
                    and weight.stride(2) == S and weight.stride(3) == 1:
                C().slab_reduce(partial.data_ptr(), rows, K * R * S, sink.view.data_ptr(), _dt(weight.dtype),
                                sink.beta(), st)
                sink.delivered()
            else:
                full = torch.empty((K, 1, R, S), dtype=torch.float32, device=dy.device)
                C().slab_reduce(partial.data_ptr(), rows, K * R * S, full.data_ptr(), 0, 0.0, st)
                dw = _deliver(weight, full.to(weight.dtype))
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            db = _bias_grad(bias_p, dy.view(NB * OH * OW, K))
        return dx, dw, db, None, None, None, None, None


def group_conv2d(x, weight, bias, stride, padding, groups, relu, want_stats, needs_input_grad=True):
    grad_sink.note_use(weight, bias)
    x = materialize(x)
    K, Cg, R, S = weight.shape
    if _DW_DIRECT and groups == K and Cg == 1 and x.shape[-1] == K and R <= 7 and S <= 7 \
            and weight.dtype in (torch.float32, x.dtype):
        y, stats = _DepthwiseConvFn.apply(x, weight, bias, tuple(stride), tuple(padding), bool(relu), bool(want_stats),
                                          bool(needs_input_grad))
        return (y, stats) if want_stats else y
    y, stats = _GroupConvFn.apply(x, weight, bias, tuple(stride), tuple(padding), int(groups), bool(relu),
                                  bool(want_stats), bool(needs_input_grad), getattr(x, "_dpa_bn", None))
    return (y, stats) if want_stats else y


# ------------------------------------------------------------------------------------------
# batch norm (+ residual, + ReLU), SyncBN
# ------------------------------------------------------------------------------------------
def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _fold(part: torch.Tensor, C_: int, limit: int = 64) -> torch.Tensor:
    n = part.shape[0]
    if n <= limit:
        return part
    nout = 32
    out = torch.empty((nout, 2, C_), dtype=torch.float32, device=part.device)
    C().bn_fold_partials(part.data_ptr(), n, C_, out.data_ptr(), nout, _stream())
    return out


_bn_tickets: Dict[int, torch.Tensor] = {}


def _bn_scratch(nparts: int, C_: int, device):
    """(tmp slab or None, ticket counters) for the one-launch fold + finalize kernels.  The counters are zeroed
    once; every launch leaves them zeroed.  BatchNorm kernels of one device run on one stream."""
    t = _bn_tickets.get(device.index)
    if t is None:
        t = torch.zeros(256, dtype=torch.int32, device=device)      # one per 64-channel group: C <= 16384
        _bn_tickets[device.index] = t
    n = C().bn_fold_chunks(nparts)
    tmp = torch.empty((n, 2, C_), dtype=torch.float64, device=device) if n > 1 else None
    return tmp, t


_FORCE_SYNC_PATH = False     # tests: take the SyncBN code path (all-reduce of a 1-rank group) on one GPU


def _group_size(group) -> int:
    from ..parallel import process_group as pg

    if _FORCE_SYNC_PATH and group is not None:
        return 2
    if group is None or not pg.is_initialized():
        return 1
    return pg.get_world_size(None if group == "world" else group)


def _gather_world(group):
    """(ranks, this rank) of the group the statistics are merged over (1, 0 without a process group)."""
    from ..parallel import process_group as pg

    if group is None or not pg.is_initialized():
        return 1, 0
    g = None if group == "world" else group
    return pg.get_world_size(g), pg.get_rank(g)


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    from ..parallel import process_group as pg

    if pg.is_initialized():
        pg.all_reduce_sum(t, group=None if group == "world" else group)


class _BNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, nbt, momentum, eps, relu, stats,
                group, defer=False, lazy_tok=None):
        fold_ok = bool(getattr(x, "_dpa_fold_ok", False)) and x.is_contiguous()
        ctx.fold = False
        x = x if x.is_contiguous() else x.contiguous()
        ctx.lazy_tok = lazy_tok
        Cc = x.shape[-1]
        M = x.numel() // Cc
        st = _stream()
        dev = x.device
        # relu >= 16: a smooth activation (ACT_CODES << 4) applied inside the normalise pass; backward multiplies dy by
        # its slope at the recomputed pre-activation inside the BN reduce / apply kernels (no activation pass at all)
        act = relu >> 4
        if act:
            relu = 0
            if residual is not None:
                raise ValueError("batch_norm: a smooth activation cannot be combined with a residual")
        if stats is None:
            nblk = C().bn_partial_rows(M, Cc)
            stats = _stats_slab(nblk, Cc, dev)
            C().bn_stats(x.data_ptr(), M, Cc, _dt(x), stats.data_ptr(), nblk, st)
        count = float(M)
        count_dev = None
        world = _group_size(group)
        g32, b32 = _f32(weight.detach()), _f32(bias.detach())
        coef = torch.empty((4, Cc), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        if world > 1:
            # SyncBN: local moments (mean, M2, n) -> all ranks -> Chan / Welford merge (Apex `welford_parallel`,
            # imagenet_ddp_apex.py:146-148).  The gather is an all-reduce of a zeroed [W][2C + 1] table in which
            # every rank fills its own row (x + 0 is exact), so it rides the small-message peer path.
            W, r = _gather_world(group)
            table = torch.zeros((W, 2 * Cc + 1), dtype=torch.float32, device=dev)
            row = table[r]
            tmp, tickets = _bn_scratch(stats.shape[0], Cc, dev)
            C().bn_fold_moments(stats.data_ptr(), stats.shape[0], Cc, _p(tmp), tickets.data_ptr(), count,
                                row[:Cc].data_ptr(), row[Cc:2 * Cc].data_ptr(), row[2 * Cc:].data_ptr(), st)
            _all_reduce_sum(table, group)
            count_dev = torch.empty(1, dtype=torch.float32, device=dev)
            C().bn_sync_merge(table.data_ptr(), W, Cc, g32.data_ptr(), b32.data_ptr(), eps, momentum,
                              _p(running_mean), _p(running_var), _p(nbt), coef[0].data_ptr(), coef[1].data_ptr(),
                              coef[2].data_ptr(), coef[3].data_ptr(), count_dev.data_ptr(), st)
        else:
            # one launch folds the per-tile partial slab and writes mean / invstd / scale / shift + running stats
            tmp, tickets = _bn_scratch(stats.shape[0], Cc, dev)
            C().bn_fold_finalize(stats.data_ptr(), stats.shape[0], Cc, _p(tmp), tickets.data_ptr(), count,
                                 g32.data_ptr(), b32.data_ptr(), eps, momentum, _p(running_mean), _p(running_var),
                                 _p(nbt), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                 coef[3].data_ptr(), st)
        if defer and relu == 1 and not act and residual is None:
            # Deferred BN + ReLU: nothing is written either.  The consumer -- a dense convolution -- applies
            # relu(scale * x + shift) while it stages its operand tiles, in forward and in its weight gradient
            # (igemm kIgemmAxf / wgrad x_scale); backward is the ordinary one with the mask recomputed from x.
            ctx.save_for_backward(x, None, weight, coef, count_dev)
            ctx.bias_ref = bias
            ctx.lazy_res = False
            ctx.res_coef = None
            ctx.cfg = (2, False, group, count)
            _BNFn.last = _BnSrc(x, coef, 2, None)
            _BNFn.deferred = coef
            _BNFn.deferred_relu = True
            return x.view(x.shape)
        if defer and not relu and not act and residual is None and world == 1:
            # Deferred: nothing is written.  The result is an alias of the raw input carrying this BN's (scale,
            # shift); the BatchNorm that takes it as `residual` applies them on the fly.  Backward is the normal one.
            ctx.save_for_backward(x, None, weight, coef, count_dev)
            ctx.bias_ref = bias
            ctx.lazy_res = False
            ctx.cfg = (0, False, group, count)
            ctx.fold = fold_ok          # (its gradient is the block's masked output gradient: see _pending_colsum)
            _BNFn.last = None
            _BNFn.deferred = coef
            _BNFn.deferred_relu = False
            return x.view(x.shape)
        res_coef = getattr(residual, "_dpa_deferred_coef", None) if residual is not None else None
        if residual is not None and not residual.is_contiguous():
            if res_coef is not None:
                raise ValueError("a deferred BatchNorm output must reach its consumer untouched")
            residual = residual.contiguous()
        y = torch.empty_like(x)
        clip = 4 if relu == 6 else 0            # ReLU6: bit 2 of the kernels' activation code
        # ReLU mask in backward: recomputed from x (scale*x + shift > 0) when there is no residual, so the
        # output is neither saved by this op nor read again; with a residual the apply pass writes one BIT per
        # element (1/16 of the output's bytes) and backward reads that instead of the output tensor
        relu_mode = 0 if not relu else (1 if residual is not None else 2)
        mask = None
        if relu_mode == 1 and not clip and _BN_BITMASK:
            relu_mode = 3
            mask = torch.empty(x.numel() // _vec(x.dtype), dtype=torch.uint8, device=dev)
        C().bn_apply(x.data_ptr(), _p(residual), y.data_ptr(), M, Cc, _dt(x), coef[2].data_ptr(),
                     coef[3].data_ptr(), (1 if relu else 0) | clip | (act << 4), st, _p(mask),
                     res_coef[2].data_ptr() if res_coef is not None else 0,
                     res_coef[3].data_ptr() if res_coef is not None else 0)
        ctx.res_coef = res_coef                 # keeps the deferred BN's coefficients alive until backward is done
        ctx.save_for_backward(x, mask if relu_mode == 3 else (y if relu_mode == 1 else None), weight, coef, count_dev)
        ctx.bias_ref = bias
        # residual = the block input handed out by the first convolution (`skip`): its gradient goes back to that
        # convolution only, so backward may return dy itself + the bit mask instead of a masked copy
        # ... or the output of an activation-free BatchNorm (the downsample branch), whose backward takes the bit
        # mask as its own "ReLU" mask
        ctx.lazy_res = bool(relu_mode == 3 and _LAZY_DRES and lazy_tok is not None
                            and residual.shape == x.shape and residual.dtype == x.dtype
                            and (getattr(residual, "_dpa_alias", False) or getattr(residual, "_dpa_bn_plain", False)))
        ctx.cfg = (relu_mode | clip | (act << 4), residual is not None, group, count)
        # "bn fold": this BN's input comes from a 1x1 convolution whose backward can absorb the BN backward
        ctx.fold = bool(fold_ok and relu_mode == 3 and not clip and not act and world == 1)
        # ReLU6 without a residual: the consumer recomputes 0 < scale * x + shift < 6 (mode 2 | 4)
        _BNFn.last = None if (act or (clip and relu_mode != 2)) else _BnSrc(x, coef, relu_mode | clip, mask, ctx.fold)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, coef, count_dev = ctx.saved_tensors
        relu, has_res, group, count = ctx.cfg
        ent = _pending_mask.pop(dy.data_ptr(), None)
        if ent is not None and ent[1] == dy._version and ent[2] == _generation \
                and ent[0].numel() * _vec(dy.dtype) == dy.numel():
            # dy is the unmasked gradient of a residual branch: its ReLU bit mask plays the role of this BN's own mask
            if relu == 0 and dy.is_contiguous():
                relu, y = 3, ent[0]
            else:
                dy = apply_bit_mask(dy, ent[0])
        shared = _pending_colsum.pop(dy.data_ptr(), None)
        if shared is not None and ctx.fold and relu == 0 and not has_res and dy.is_contiguous() \
                and shared[1] == dy._version and shared[2] == _generation and ctx.needs_input_grad[0]:
            token = torch.full((1,), float("nan"), dtype=dy.dtype, device=x.device).expand(x.shape)
            _pending_fold[token.data_ptr()] = (token, dy, shared[0], coef, weight, ctx.bias_ref, count, _generation)
            return token, None, None, None, None, None, None, None, None, None, None, None, None, None
        dy = dy if dy.is_contiguous() else dy.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        st = _stream()
        dev = x.device
        fused = _pending_bnred.pop(dy.data_ptr(), None)
        if fused is not None and len(fused) > 3 and fused[3]:
            # the producer stored dy THROUGH the ReLU mask and reduced sum dy_m only ("bn fold")
            if ctx.fold and fused[1] == x.data_ptr() and fused[2] == dy._version and ctx.needs_input_grad[0]:
                token = torch.full((1,), float("nan"), dtype=dy.dtype, device=dev).expand(x.shape)
                _pending_fold[token.data_ptr()] = (token, dy, fused[0], coef, weight, ctx.bias_ref, count, _generation)
                dres = dy if (has_res and ctx.needs_input_grad[3]) else None
                if dres is not None:
                    # the residual branch receives the very same masked tensor: an activation-free BatchNorm there (the
                    # downsample branch) can fold too, with the same column sums
                    _pending_colsum[dy.data_ptr()] = (fused[0], dy._version, _generation)
                return token, None, None, dres, None, None, None, None, None, None, None, None, None, None
            fused = None            # (a masked gradient is still a valid input of the ordinary path: masking is idempotent)
        if fused is not None and fused[1] == x.data_ptr() and fused[2] == dy._version:
            part = fused[0]         # the producing data-gradient kernel already reduced (sum dy_m, sum dy_m*xhat)
            bn_bwd_counts["fused"] += 1
        else:
            bn_bwd_counts["reduced"] += 1
            nblk = C().bn_partial_rows(M, Cc)
            part = torch.empty((nblk, 2, Cc), dtype=torch.float32, device=dev)
            C().bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), _p(y), M, Cc, _dt(x), coef[0].data_ptr(),
                              coef[1].data_ptr(), relu, coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(),
                              nblk, st)
        # local [dgamma, dbeta]: written straight into the parameters' gradient views when possible
        bias_p = ctx.bias_ref
        sw_, sb_ = grad_sink.sink_of(weight), grad_sink.sink_of(bias_p)
        direct = sw_ is not None and sb_ is not None and not sw_.written and not sb_.written \
            and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        if direct:
            dg_t, db_t = sw_.view, sb_.view
        else:
            sums = torch.empty((2, Cc), dtype=torch.float32, device=dev)
            dg_t, db_t = sums[0], sums[1]
        g32 = _f32(weight.detach())
        wk = torch.empty((5, Cc), dtype=torch.float32, device=dev)
        sync = _group_size(group) > 1
        if not sync:
            # one launch: fold the partial slab, dgamma / dbeta and the apply coefficients
            tmp, tickets = _bn_scratch(part.shape[0], Cc, dev)
            C().bn_bwd_fold_finalize(part.data_ptr(), part.shape[0], Cc, _p(tmp), tickets.data_ptr(), count,
                                     g32.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                     coef[3].data_ptr(), dg_t.data_ptr(), db_t.data_ptr(), wk.data_ptr(), st)
        else:
            part = _fold(part, Cc)
            C().bn_bwd_finalize_fused(part.data_ptr(), part.shape[0], Cc, dg_t.data_ptr(), db_t.data_ptr(),
                                      coef[0].data_ptr(), coef[1].data_ptr(), g32.data_ptr(), count,
                                      0, coef[2].data_ptr(), coef[3].data_ptr(), st)
        if sync:
            gs = torch.stack([dg_t, db_t])
            _all_reduce_sum(gs, group)
            C().bn_bwd_coeffs(gs[0].data_ptr(), gs[1].data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                              g32.data_ptr(), count, _p(count_dev), Cc, wk.data_ptr(), coef[2].data_ptr(),
                              coef[3].data_ptr(), st)
        dx = torch.empty_like(x)
        lazy = has_res and ctx.lazy_res and ctx.needs_input_grad[3]
        dres = torch.empty_like(x) if (has_res and not lazy) else None
        C().bn_bwd_apply(dy.data_ptr(), x.data_ptr(), _p(y), dx.data_ptr(), _p(dres), M, Cc, _dt(x),
                         wk.data_ptr(), relu, st)
        if lazy:                 # y is the bit mask here (relu mode 3); the consumer applies it in its dgrad epilogue
            _pending_mask[dy.data_ptr()] = (y, dy._version, _generation)
            ctx.lazy_tok.ent = (dy, y, dy._version)       # holding dy also keeps autograd from accumulating into it
            dres = dy
        if direct:
            sw_.delivered()
            sb_.delivered()
            dgamma = dbeta = None
        else:
            dgamma = _deliver(weight, dg_t.to(weight.dtype)) if ctx.needs_input_grad[1] else None
            dbeta = _deliver(bias_p, db_t.to(bias_p.dtype)) if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None


_FUSE_STEM = os.environ.get("DPA_FUSE_STEM", "1") == "1"
_BN_BITMASK = os.environ.get("DPA_BN_BITMASK", "1") == "1"   # residual BN+ReLU: backward reads a bit mask, not the output


def can_fuse_bn_relu_maxpool(x: torch.Tensor) -> bool:
    """The fused stem kernels keep one channel chunk per thread: C / vector width must divide 256."""
    if not _FUSE_STEM or x.dim() != 4:
        return False
    cc, rem = divmod(x.shape[-1], _vec(x.dtype))
    return rem == 0 and cc > 0 and 256 % cc == 0 and x.numel() // x.shape[-1] < (1 << 31)


class _BNReluMaxPoolFn(torch.autograd.Function):
    """Training-mode BN + ReLU + max-pool of a stem: the normalised full-resolution tensor is never written.

    Forward: fold the conv epilogue's statistics, then ONE pass reads the conv output and writes the pooled tensor
    and the arg-max codes.  Backward: the pooled gradient is gathered through the codes inside the BatchNorm
    reduce pass and again inside the apply pass (dx = A*g + B*x + K) -- no max-pool backward tensor either.
    Per ResNet-50 step at batch 1024 this removes ~7.5 GB of HBM traffic (profiles/r1_notes.md)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, stats, k, s, p, ceil_mode):
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cc = x.shape
        M = NB * IH * IW
        st = _stream()
        dev = x.device
        if stats is None:
            nblk = C().bn_partial_rows(M, Cc)
            stats = _stats_slab(nblk, Cc, dev)
            C().bn_stats(x.data_ptr(), M, Cc, _dt(x), stats.data_ptr(), nblk, st)
        g32, b32 = _f32(weight.detach()), _f32(bias.detach())
        coef = torch.empty((4, Cc), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        tmp, tickets = _bn_scratch(stats.shape[0], Cc, dev)
        C().bn_fold_finalize(stats.data_ptr(), stats.shape[0], Cc, _p(tmp), tickets.data_ptr(), float(M),
                             g32.data_ptr(), b32.data_ptr(), eps, momentum, _p(running_mean), _p(running_var),
                             _p(nbt), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                             coef[3].data_ptr(), st)
        OH, OW = _pool_out(IH, k[0], s[0], p[0], ceil_mode), _pool_out(IW, k[1], s[1], p[1], ceil_mode)
        y = torch.empty((NB, OH, OW, Cc), dtype=x.dtype, device=dev)
        idx = torch.empty((NB, OH, OW, Cc), dtype=torch.uint8, device=dev)
        geom = (NB, IH, IW, Cc, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1])
        C().bn_relu_maxpool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                *geom, _dt(x), st)
        ctx.save_for_backward(x, idx, weight, coef)
        ctx.bias_ref = bias
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx, weight, coef = ctx.saved_tensors
        geom = ctx.geom
        NB, IH, IW, Cc = geom[:4]
        M = NB * IH * IW
        dy = dy if dy.is_contiguous() else dy.contiguous()
        st = _stream()
        dev = x.device
        nblk = C().bn_relu_maxpool_partial_rows(M * (Cc // _vec(x.dtype)))
        part = torch.empty((nblk, 2, Cc), dtype=torch.float32, device=dev)
        C().bn_relu_maxpool_bwd_reduce(idx.data_ptr(), dy.data_ptr(), x.data_ptr(), coef.data_ptr(), part.data_ptr(),
                                       nblk, *geom, _dt(x), st)
        bias_p = ctx.bias_ref
        sw_, sb_ = grad_sink.sink_of(weight), grad_sink.sink_of(bias_p)
        direct = sw_ is not None and sb_ is not None and not sw_.written and not sb_.written \
            and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        if direct:
            dg_t, db_t = sw_.view, sb_.view
        else:
            sums = torch.empty((2, Cc), dtype=torch.float32, device=dev)
            dg_t, db_t = sums[0], sums[1]
        g32 = _f32(weight.detach())
        wk = torch.empty((5, Cc), dtype=torch.float32, device=dev)
        tmp, tickets = _bn_scratch(part.shape[0], Cc, dev)
        C().bn_bwd_fold_finalize(part.data_ptr(), part.shape[0], Cc, _p(tmp), tickets.data_ptr(), float(M),
                                 g32.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                 coef[3].data_ptr(), dg_t.data_ptr(), db_t.data_ptr(), wk.data_ptr(), st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            C().bn_relu_maxpool_bwd_apply(idx.data_ptr(), dy.data_ptr(), x.data_ptr(), coef.data_ptr(), wk.data_ptr(),
                                          dx.data_ptr(), *geom, _dt(x), st)
        if direct:
            sw_.delivered()
            sb_.delivered()
            dgamma = dbeta = None
        else:
            dgamma = _deliver(weight, dg_t.to(weight.dtype)) if ctx.needs_input_grad[1] else None
            dbeta = _deliver(bias_p, db_t.to(bias_p.dtype)) if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def bn_relu_maxpool(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, stats, k, s, p,
                    ceil_mode):
    grad_sink.note_use(weight, bias)
    return _BNReluMaxPoolFn.apply(x, weight, bias, running_mean, running_var, num_batches_tracked,
                                  float(momentum if momentum is not None else 0.1), float(eps), stats, tuple(k),
                                  tuple(s), tuple(p), bool(ceil_mode))


class _BNEvalFn(torch.autograd.Function):
    """Inference-mode BN (running statistics): y = x*scale + shift (+res)(relu)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, eps, relu):
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        st = _stream()
        coef = torch.empty((3, Cc), dtype=torch.float32, device=x.device)  # scale, shift, zero
        coef[2].zero_()
        C().bn_eval_coeffs(_f32(weight.detach()).data_ptr(), _f32(bias.detach()).data_ptr(),
                           _f32(running_mean).data_ptr(), _f32(running_var).data_ptr(), eps, Cc,
                           coef[0].data_ptr(), coef[1].data_ptr(), st)
        if residual is not None and not residual.is_contiguous():
            residual = residual.contiguous()
        y = torch.empty_like(x)
        C().bn_apply(x.data_ptr(), _p(residual), y.data_ptr(), M, Cc, _dt(x), coef[0].data_ptr(),
                     coef[1].data_ptr(), (1 if relu else 0) | (4 if relu == 6 else 0), st)
        # fine-tuning with frozen statistics still trains gamma / beta (torch does): their gradients need x and the
        # running moments in backward
        affine = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        ctx.save_for_backward(y if relu else None, coef, x if affine else None,
                              running_mean if affine else None, running_var if affine else None)
        ctx.cfg = (relu, residual is not None, float(eps))
        ctx.params = (weight, bias) if affine else None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, coef, x, rm, rv = ctx.saved_tensors
        relu, has_res, eps = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        Cc = dy.shape[-1]
        M = dy.numel() // Cc
        st = _stream()
        if relu:
            dym = torch.empty_like(dy)
            C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), st)
            dy = dym * (y < 6) if relu == 6 else dym     # eval-mode backward (fine-tuning frozen BN) is not a hot path
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            C().bn_apply(dy.data_ptr(), 0, dx.data_ptr(), M, Cc, _dt(dy), coef[0].data_ptr(), coef[2].data_ptr(), 0, st)
        dgamma = dbeta = None
        if ctx.params is not None and x is not None:
            weight, bias = ctx.params
            # dgamma = sum dy * (x - running_mean) * invstd, dbeta = sum dy: the training-mode reducer with the running
            # moments as its (mean, invstd)
            mom = torch.stack([_f32(rm), torch.rsqrt(_f32(rv) + eps)]).contiguous()
            nblk = C().bn_partial_rows(M, Cc)
            part = torch.empty((nblk, 2, Cc), dtype=torch.float32, device=dy.device)
            C().bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), 0, M, Cc, _dt(x), mom[0].data_ptr(), mom[1].data_ptr(), 0,
                              coef[0].data_ptr(), coef[1].data_ptr(), part.data_ptr(), nblk, st)
            dg = torch.empty(Cc, dtype=torch.float32, device=dy.device)
            db = torch.empty(Cc, dtype=torch.float32, device=dy.device)
            C().bn_bwd_finalize(part.data_ptr(), nblk, Cc, dg.data_ptr(), db.data_ptr(), st)
            dgamma = _deliver(weight, dg.to(weight.dtype)) if ctx.needs_input_grad[1] else None
            dbeta = _deliver(bias, db.to(bias.dtype)) if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta, (dy if has_res else None), None, None, None, None


def _act_code(relu) -> int:
    """False / True / 6 -> 0 (none), 1 (ReLU), 6 (ReLU6); "silu", "hardswish", ... -> ACT_CODES << 4."""
    if isinstance(relu, str):
        return ACT_CODES[relu] << 4
    return 6 if (relu is not True and relu == 6) else (1 if relu else 0)


_DEFER_BN = os.environ.get("DPA_DEFER_BN", "1") == "1"


def batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps,
               relu, residual, stats, group, defer=False):
    if getattr(x, "_dpa_deferred_coef", None) is not None:
        raise ValueError("a deferred BatchNorm output can only be used as the `residual` of another BatchNorm")
    if training or running_mean is None:
        grad_sink.note_use(weight, bias)
        _BNFn.last = None
        _BNFn.deferred = None
        tok = None
        if _LAZY_DRES and residual is not None and residual.requires_grad and torch.is_grad_enabled() \
                and (getattr(residual, "_dpa_alias", False) or getattr(residual, "_dpa_bn_plain", False)):
            # the lazy residual gradient is only right when this BN is the residual's sole consumer; nothing in the
            # forward pass can know that, so a hook on the residual checks what autograd finally delivers
            tok = _LazyTok()
            residual.register_hook(functools.partial(_lazy_res_hook, tok))
        y = _BNFn.apply(x, weight, bias, residual, running_mean if training else None,
                        running_var if training else None, num_batches_tracked if training else None,
                        float(momentum if momentum is not None else 0.1), float(eps), _act_code(relu), stats, group,
                        bool(defer and training and torch.is_grad_enabled()
                             and (_AXF if relu else _DEFER_BN)), tok)
        if _BNFn.deferred is not None:
            y._dpa_deferred_coef = _BNFn.deferred
            y._dpa_deferred_relu = _BNFn.deferred_relu
            _BNFn.deferred = None
        if _BNFn.last is not None and torch.is_grad_enabled():
            y._dpa_bn = _BNFn.last      # lets the consuming convolution fuse this BN's backward reduction
            _BNFn.last = None
        if not relu and residual is None:
            y._dpa_bn_plain = True      # activation-free BN output: as a residual its gradient may arrive unmasked + bit mask
        return y
    if torch.is_grad_enabled():
        grad_sink.note_use(weight, bias)
    if isinstance(relu, str):          # eval mode is not a hot path: BN from running statistics, then the activation kernel
        y = _BNEvalFn.apply(x, weight, bias, residual, running_mean, running_var, float(eps), 0)
        return activation(y, relu)
    return _BNEvalFn.apply(x, weight, bias, residual, running_mean, running_var, float(eps), _act_code(relu))


# ------------------------------------------------------------------------------------------
# pooling / elementwise
# ------------------------------------------------------------------------------------------
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p, ceil_mode=False, relu_in=False):
        ctx.relu_src = x.data_ptr() if (relu_in and x.is_contiguous()) else 0
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cc = x.shape
        OH, OW = _pool_out(IH, k[0], s[0], p[0], ceil_mode), _pool_out(IW, k[1], s[1], p[1], ceil_mode)
        y = torch.empty((NB, OH, OW, Cc), dtype=x.dtype, device=x.device)
        if k[0] * k[1] > 250:
            raise ValueError("max_pool2d: window too large")
        idx = torch.empty((NB, OH, OW, Cc), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        C().maxpool_fwd(x.data_ptr(), y.data_ptr(), _p(idx), NB, IH, IW, Cc, OH, OW, k[0], k[1], s[0], s[1],
                        p[0], p[1], _dt(x), _stream())
        if ctx.relu_src and idx is not None:
            ctx.save_for_backward(idx, y)           # the pooled values: where they are 0 the whole window was inactive
        else:
            ctx.relu_src = 0
            ctx.save_for_backward(idx)
        ctx.cfg = (k, s, p, OH, OW, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        idx = ctx.saved_tensors[0]
        k, s, p, OH, OW, xshape = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        NB, IH, IW, Cc = xshape
        if ctx.relu_src:
            y = ctx.saved_tensors[1]
            dym = torch.empty_like(dy)
            C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), _stream())
            dy = dym
        dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
        C().maxpool_bwd(idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), NB, IH, IW, Cc, OH, OW, k[0], k[1], s[0], s[1],
                        p[0], p[1], _dt(dy), _stream())
        if ctx.relu_src:
            _premasked[dx.data_ptr()] = (dx._version, _generation, ctx.relu_src)
        return dx, None, None, None, None, None


def max_pool2d(x, k, s, p, ceil_mode=False):
    relu_in = _POOL_PREMASK and _is_relu_out(x) and torch.is_grad_enabled() and x.requires_grad
    return _MaxPoolFn.apply(x, tuple(k), tuple(s), tuple(p), bool(ceil_mode), bool(relu_in))


def _pool_out(i, k, s, p, ceil_mode):
    """torch's pooling output extent; with ceil_mode the last window may hang over the right / bottom edge but
    must start inside the (left-padded) input."""
    if not ceil_mode:
        return (i + 2 * p - k) // s + 1
    o = -(-(i + 2 * p - k) // s) + 1
    if (o - 1) * s >= i + p:
        o -= 1
    return o


class _CatFn(torch.autograd.Function):
    """Channel concat of NHWC tensors (SqueezeNet fire modules, DenseNet feature reuse): one strided copy kernel
    per source; backward splits the gradient with the same kernel."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [t if t.is_contiguous() else t.contiguous() for t in xs]
        lead = xs[0].shape[:-1]
        rows = xs[0].numel() // xs[0].shape[-1]
        cs = [t.shape[-1] for t in xs]
        es = xs[0].element_size()
        if any(t.shape[:-1] != lead or t.dtype != xs[0].dtype for t in xs) or any((c * es) % 16 for c in cs):
            raise ValueError("cat_channels: inputs must share shape/dtype and have 16-byte multiples of channels")
        tot = sum(cs)
        out = torch.empty(tuple(lead) + (tot,), dtype=xs[0].dtype, device=xs[0].device)
        off = 0
        for t, c in zip(xs, cs):
            C().copy_rows(t.data_ptr(), c * es, out.data_ptr() + off * es, tot * es, rows, c * es, _stream())
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy if dy.is_contiguous() else dy.contiguous()
        cs, es = ctx.cs, dy.element_size()
        tot = sum(cs)
        rows = dy.numel() // tot
        outs, off = [], 0
        for i, c in enumerate(cs):
            if ctx.needs_input_grad[i]:
                g = torch.empty(dy.shape[:-1] + (c,), dtype=dy.dtype, device=dy.device)
                C().copy_rows(dy.data_ptr() + off * es, tot * es, g.data_ptr(), c * es, rows, c * es, _stream())
                outs.append(g)
            else:
                outs.append(None)
            off += c
        return tuple(outs)


def cat_channels(xs):
    return _CatFn.apply(*xs)


# ------------------------------------------------------------------------------------------
# fused multi-head self-attention (csrc/kernels/attention.hip)
# ------------------------------------------------------------------------------------------
_scale_vecs: Dict[tuple, torch.Tensor] = {}


def _scale_vec(scale, H: int, device) -> torch.Tensor:
    if isinstance(scale, torch.Tensor):
        return scale.detach().float().reshape(H).contiguous()
    key = (float(scale), H, device.index)
    t = _scale_vecs.get(key)
    if t is None:
        t = _scale_vecs[key] = torch.full((H,), float(scale), dtype=torch.float32, device=device)
    return t


class _AttnFn(torch.autograd.Function):
    """qkv [B, S, 3*H*D] -> [B, S, H*D].  fp32 activations run with fp16 MFMA operands and fp32 accumulation
    (the kernel has no fp32 MFMA path; 11-bit operands, like TF32 elsewhere) -- bf16 / fp16 run natively."""

    @staticmethod
    def forward(ctx, qkv, H, scale, rel_bias, mask):
        B, S, W3 = qkv.shape
        D = W3 // (3 * H)
        in_dtype = qkv.dtype
        q16 = qkv if qkv.is_contiguous() else qkv.contiguous()
        if in_dtype == torch.float32:
            q16 = q16.to(torch.float16)
        dev = qkv.device
        bias, nbias = None, 1
        if rel_bias is not None:
            bias = rel_bias.detach().float().reshape(1, H, S, S)
            if mask is not None:
                bias = bias + mask.detach().float().reshape(-1, 1, S, S)
            bias = bias.contiguous()
            nbias = bias.shape[0]
        elif mask is not None:
            bias = mask.detach().float().reshape(-1, 1, S, S).expand(-1, H, S, S).contiguous()
            nbias = bias.shape[0]
        if bias is not None and B % nbias:
            raise ValueError("attention: batch %d is not a multiple of the %d mask windows" % (B, nbias))
        sc = _scale_vec(scale, H, dev)
        out = torch.empty((B, S, H * D), dtype=q16.dtype, device=dev)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
        C().attn_fwd(q16.data_ptr(), out.data_ptr(), lse.data_ptr(), _p(bias), nbias, sc.data_ptr(), B, H, S, D,
                     _dt(q16), _stream())
        ctx.save_for_backward(q16, out, lse, bias, sc)
        ctx.cfg = (H, D, nbias, in_dtype, rel_bias is not None and rel_bias.requires_grad,
                   None if rel_bias is None else (rel_bias.shape, rel_bias.dtype))
        return out if in_dtype == out.dtype else out.to(in_dtype)

    @staticmethod
    def backward(ctx, dout):
        q16, out, lse, bias, sc = ctx.saved_tensors
        H, D, nbias, in_dtype, want_dbias, rb = ctx.cfg
        B, S, _ = q16.shape
        dev = q16.device
        dout = dout if dout.is_contiguous() else dout.contiguous()
        if dout.dtype != q16.dtype:
            dout = dout.to(q16.dtype)
        dqkv = torch.empty_like(q16)
        delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
        slab, nsplit = None, 1
        if want_dbias:
            if S > 64:
                raise ValueError("attention: a learnable bias is supported up to 64 tokens per window")
            nsplit = C().attn_dbias_splits(B, H)
            slab = torch.empty((nsplit, H, S, S), dtype=torch.float32, device=dev)
        C().attn_bwd(q16.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), _p(bias), nbias,
                     sc.data_ptr(), dqkv.data_ptr(), _p(slab), nsplit, B, H, S, D, _dt(q16), _stream())
        dbias = None
        if want_dbias:
            dbias = torch.empty((H, S, S), dtype=torch.float32, device=dev)
            C().slab_reduce(slab.data_ptr(), nsplit, H * S * S, dbias.data_ptr(), 0, 0.0, _stream())
            dbias = dbias.reshape(rb[0]).to(rb[1])
        if in_dtype != dqkv.dtype:
            dqkv = dqkv.to(in_dtype)
        return dqkv, None, None, dbias, None


def attention(qkv, num_heads, scale, rel_bias=None, mask=None):
    return _AttnFn.apply(qkv, int(num_heads), scale, rel_bias, mask)


class _WindowFn(torch.autograd.Function):
    """Window / grid partition of an NHWC token map and its inverse (``attention.hip: window_partition_kernel``);
    each direction is the other's gradient."""

    @staticmethod
    def forward(ctx, x, geom, reverse):
        B, H, W, wh, ww, nh, nw, sh, sw, grid = geom
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[-1]
        if reverse:
            out = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
        else:
            out = torch.empty((B * nh * nw, wh * ww, Cc), dtype=x.dtype, device=x.device)
        C().window_partition(x.data_ptr(), out.data_ptr(), 1 if reverse else 0, B, H, W, Cc, wh, ww, nh, nw, sh, sw,
                             1 if grid else 0, _dt(x), _stream())
        ctx.cfg = (geom, reverse)
        return out

    @staticmethod
    def backward(ctx, dy):
        geom, reverse = ctx.cfg
        return _WindowFn.apply(dy, geom, not reverse), None, None


def window_partition(x, geom, reverse):
    return _WindowFn.apply(x, tuple(geom), bool(reverse))


class _QKNormFn(torch.autograd.Function):
    """Swin V2 cosine attention pre-pass on the packed projection: q <- mult[h] * q / |q|, k <- k / |k|."""

    @staticmethod
    def forward(ctx, qkv, mult, H):
        in_dtype = qkv.dtype
        x = qkv if qkv.is_contiguous() else qkv.contiguous()
        if in_dtype == torch.float32:
            x = x.to(torch.float16)
        tokens = x.numel() // x.shape[-1]
        D = x.shape[-1] // (3 * H)
        m = mult.detach().float().reshape(H).contiguous()
        out = torch.empty_like(x)
        inv = torch.empty((tokens, 2, H), dtype=torch.float32, device=x.device)
        C().qk_normalize(x.data_ptr(), 0, out.data_ptr(), m.data_ptr(), inv.data_ptr(), 0, tokens, H, D, 0, _dt(x),
                         _stream())
        ctx.save_for_backward(x, m, inv)
        ctx.cfg = (H, D, in_dtype, mult.shape, mult.dtype)
        return out if in_dtype == out.dtype else out.to(in_dtype)

    @staticmethod
    def backward(ctx, dy):
        x, m, inv = ctx.saved_tensors
        H, D, in_dtype, mshape, mdtype = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        tokens = x.numel() // x.shape[-1]
        dx = torch.empty_like(x)
        rows4 = _round_up(tokens, 4)
        part = torch.empty((rows4, H), dtype=torch.float32, device=x.device)
        if rows4 != tokens:
            part[tokens:].zero_()
        C().qk_normalize(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), m.data_ptr(), inv.data_ptr(), part.data_ptr(), tokens,
                         H, D, 1, _dt(x), _stream())
        # d mult[h] = sum over tokens: the column reducer wants 16-byte rows, so four tokens form one row of 4 * H columns
        dm = _colsum(part.view(rows4 // 4, 4 * H)).view(4, H).sum(0).reshape(mshape).to(mdtype)
        return (dx if in_dtype == dx.dtype else dx.to(in_dtype)), dm, None


def qk_normalize(qkv, mult, num_heads):
    return _QKNormFn.apply(qkv, mult, int(num_heads))


class _ShufflePairFn(torch.autograd.Function):
    """(L', R') = split(channel_shuffle(cat(a, b), 2)) on padded halves (``elementwise.hip: shuffle_pair_kernel``)."""

    @staticmethod
    def forward(ctx, a, b, bf):
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        if a.shape != b.shape or a.dtype != b.dtype:
            raise ValueError("shuffle_pair: the two halves must have the same shape and dtype")
        P = a.shape[-1]
        o0, o1 = torch.empty_like(a), torch.empty_like(a)
        C().shuffle_pair(a.data_ptr(), b.data_ptr(), o0.data_ptr(), o1.data_ptr(), a.numel() // P, bf, P, 0, _dt(a),
                         _stream())
        ctx.bf = bf
        return o0, o1

    @staticmethod
    def backward(ctx, d0, d1):
        if d0 is None:
            d0 = torch.zeros_like(d1)
        if d1 is None:
            d1 = torch.zeros_like(d0)
        d0 = d0 if d0.is_contiguous() else d0.contiguous()
        d1 = d1 if d1.is_contiguous() else d1.contiguous()
        P = d0.shape[-1]
        da, db = torch.empty_like(d0), torch.empty_like(d0)
        C().shuffle_pair(d0.data_ptr(), d1.data_ptr(), da.data_ptr(), db.data_ptr(), d0.numel() // P, ctx.bf, P, 1,
                         _dt(d0), _stream())
        return da, db, None


def shuffle_pair(a, b, bf):
    return _ShufflePairFn.apply(a, b, int(bf))


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, o):
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cc = x.shape
        y = torch.empty((NB, o[0], o[1], Cc), dtype=x.dtype, device=x.device)
        C().avgpool_fwd(x.data_ptr(), y.data_ptr(), NB, IH, IW, Cc, o[0], o[1], _dt(x), _stream())
        ctx.cfg = (x.shape, o)
        return y

    @staticmethod
    def backward(ctx, dy):
        (NB, IH, IW, Cc), o = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty((NB, IH, IW, Cc), dtype=dy.dtype, device=dy.device)
        C().avgpool_bwd(dy.data_ptr(), dx.data_ptr(), NB, IH, IW, Cc, o[0], o[1], _dt(dy), _stream())
        return dx, None


class _AvgPoolWinFn(torch.autograd.Function):
    """Fixed-window average pooling, overlapping windows and zero padding allowed (count_include_pad=True)."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x if x.is_contiguous() else x.contiguous()
        NB, IH, IW, Cc = x.shape
        OH, OW = (IH + 2 * p[0] - k[0]) // s[0] + 1, (IW + 2 * p[1] - k[1]) // s[1] + 1
        y = torch.empty((NB, OH, OW, Cc), dtype=x.dtype, device=x.device)
        C().avgpool_win(x.data_ptr(), y.data_ptr(), 0, NB, IH, IW, Cc, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1],
                        _dt(x), _stream())
        ctx.cfg = (x.shape, (OH, OW), k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (NB, IH, IW, Cc), (OH, OW), k, s, p = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty((NB, IH, IW, Cc), dtype=dy.dtype, device=dy.device)
        C().avgpool_win(dy.data_ptr(), dx.data_ptr(), 1, NB, IH, IW, Cc, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1],
                        _dt(dy), _stream())
        return dx, None, None, None


def avg_pool2d_window(x, k, s, p):
    return _AvgPoolWinFn.apply(x, tuple(k), tuple(s), tuple(p))


def adaptive_avg_pool2d(x, o):
    if x.shape[1] == o[0] and x.shape[2] == o[1]:
        return x                      # identity at 224x224 for AlexNet (6x6) / VGG (7x7)
    return _AvgPoolFn.apply(x, tuple(o))


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(x)
        C().relu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _dt(x), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dy)
        C().relu_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), _dt(dy), _stream())
        return dx


def relu(x):
    return _ReluFn.apply(x)


class _AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        y = torch.empty_like(a)
        C().add_relu_fwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _dt(a), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dy)
        C().relu_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), _dt(dy), _stream())
        return dx, dx


def add_relu(a, b):
    return _AddReluFn.apply(a, b)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        y = torch.empty_like(a)
        C().add_fwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _dt(a), _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("add: operands must have the same shape and dtype")
    return _AddFn.apply(a, b)


ACT_CODES = {"sigmoid": 1, "silu": 2, "hardsigmoid": 3, "hardswish": 4, "gelu": 5}


class _ActFn(torch.autograd.Function):
    """Smooth activations of EfficientNet / MobileNetV3.  The input is saved and the slope recomputed in backward."""

    @staticmethod
    def forward(ctx, x, code):
        x = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(x)
        C().act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), code, _dt(x), _stream())
        ctx.save_for_backward(x)
        ctx.code = code
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dy)
        C().act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), ctx.code, _dt(dy), _stream())
        return dx, None


def activation(x, kind: str):
    return _ActFn.apply(x, ACT_CODES[kind])


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last axis (ConvNeXt).  fp32 statistics, one read + one write forward; backward writes dx and
    folds dgamma / dbeta from one partial row per workgroup (fixed order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        if weight.shape != (Cc,) or bias.shape != (Cc,):
            raise ValueError("layer_norm: weight / bias must be [C]")
        if Cc % _vec(x.dtype):
            raise ValueError("layer_norm: C must be a multiple of %d" % _vec(x.dtype))
        w, b = _f32(weight.detach()).contiguous(), _f32(bias.detach()).contiguous()
        y = torch.empty_like(x)
        stat = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        C().ln_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stat[0].data_ptr(), stat[1].data_ptr(), rows,
                   Cc, float(eps), _dt(x), _stream())
        ctx.save_for_backward(x, w, stat)
        ctx.refs = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stat = ctx.saved_tensors
        weight, bias = ctx.refs
        dy = dy if dy.is_contiguous() else dy.contiguous()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        dx = torch.empty_like(x)
        nparts = C().ln_partial_rows(rows, Cc, _dt(x))
        partial = torch.empty((nparts, 2, Cc), dtype=torch.float32, device=x.device)
        dwb = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        C().ln_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), stat[0].data_ptr(), stat[1].data_ptr(), dx.data_ptr(),
                   partial.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), 0.0, 0.0, rows, Cc, _dt(x), _stream())
        dw = _deliver(weight, dwb[0].to(weight.dtype)) if ctx.needs_input_grad[1] else None
        db = _deliver(bias, dwb[1].to(bias.dtype)) if ctx.needs_input_grad[2] else None
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None


def layer_norm(x, weight, bias, eps):
    grad_sink.note_use(weight, bias)
    return _LayerNormFn.apply(x, weight, bias, float(eps))


class _SEScaleFn(torch.autograd.Function):
    """``y[n, h, w, c] = x[n, h, w, c] * s[n, c]`` -- the squeeze-excitation gate (and stochastic depth, where ``s``
    needs no gradient).  Backward: one pass writes ``dx = dy * s`` and reduces ``ds = sum_hw dy * x``."""

    @staticmethod
    def forward(ctx, x, s):
        x = x if x.is_contiguous() else x.contiguous()
        NB, H, W, Cc = x.shape
        if s.shape != (NB, Cc):
            raise ValueError("se_scale: scale must be [N, C]")
        if Cc % _vec(x.dtype):
            raise ValueError("se_scale: C must be a multiple of %d" % _vec(x.dtype))
        s = _f32(s.detach()).contiguous()
        y = torch.empty_like(x)
        C().se_scale_fwd(x.data_ptr(), s.data_ptr(), y.data_ptr(), NB, H * W, Cc, _dt(x), _stream())
        ctx.save_for_backward(x if ctx.needs_input_grad[1] else None, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        NB, H, W, Cc = dy.shape
        dx = torch.empty_like(dy)
        if x is None:                               # the scale is a constant (stochastic depth)
            C().se_scale_fwd(dy.data_ptr(), s.data_ptr(), dx.data_ptr(), NB, H * W, Cc, _dt(dy), _stream())
            return dx, None
        ds = torch.empty((NB, Cc), dtype=torch.float32, device=dy.device)
        C().se_scale_bwd(dy.data_ptr(), x.data_ptr(), s.data_ptr(), dx.data_ptr(), ds.data_ptr(), NB, H * W, Cc,
                         _dt(dy), _stream())
        return dx, ds


_SE_FUSED = os.environ.get("DPA_SE_FUSED", "1") == "1"


class _SEBlockFn(torch.autograd.Function):
    """Whole squeeze-excitation block ``x * gate(W2 act(W1 mean_hw(x) + b1) + b2)`` as ONE autograd node.

    As separate nodes autograd hands ``x`` two gradients -- ``dy * s`` from the gate multiply and the broadcast
    ``d(mean) / HW`` from the pooling -- writes both and adds them (4 activation-sized reads, 3 writes).  Here backward
    first reduces ``ds = sum_hw dy * x`` (2 reads), runs the excitation MLP's backward on ``[N, C]`` tensors, then
    writes ``dx = dy * s + d(mean) / HW`` in one pass (1 read, 1 write).  The MLP is three launches in total
    (``csrc/kernels/se_mlp.hip``: forward, per-sample backward, parameter gradients summed over the batch in a fixed
    order); its parameter gradients are delivered to the gradient sinks like every kernel's."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, gate):
        x = x if x.is_contiguous() else x.contiguous()
        NB, H, W, Cc = x.shape
        S = w1.shape[0]
        st = _stream()
        dev = x.device
        pooled = torch.empty((NB, 1, 1, Cc), dtype=x.dtype, device=dev)
        C().avgpool_fwd(x.data_ptr(), pooled.data_ptr(), NB, H, W, Cc, 1, 1, _dt(x), st)
        p = pooled.view(NB, Cc).float()
        W1 = w1.detach().reshape(S, Cc).float().contiguous()
        W2 = w2.detach().reshape(Cc, S).float().contiguous()
        B1, B2 = b1.detach().float().contiguous(), b2.detach().float().contiguous()
        buf = torch.empty((2, NB, S), dtype=torch.float32, device=dev)        # z1, h
        z2 = torch.empty((NB, Cc), dtype=torch.float32, device=dev)
        s = torch.empty((NB, Cc), dtype=torch.float32, device=dev)
        ia, ig = (0 if act == "relu" else 1), (0 if gate == "sigmoid" else 1)
        C().se_mlp_fwd(p.data_ptr(), W1.data_ptr(), B1.data_ptr(), W2.data_ptr(), B2.data_ptr(), buf[0].data_ptr(),
                       buf[1].data_ptr(), z2.data_ptr(), s.data_ptr(), NB, Cc, S, ia, ig, st)
        y = torch.empty_like(x)
        C().se_scale_fwd(x.data_ptr(), s.data_ptr(), y.data_ptr(), NB, H * W, Cc, _dt(x), st)
        ctx.save_for_backward(x, s, p, buf, z2, w1, w2)
        ctx.refs = (b1, b2)
        ctx.cfg = (ia, ig)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s, p, buf, z2, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.refs
        ia, ig = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        NB, H, W, Cc = x.shape
        S = w1.shape[0]
        st = _stream()
        dev = x.device
        ds = torch.empty((NB, Cc), dtype=torch.float32, device=dev)
        C().se_scale_bwd(dy.data_ptr(), x.data_ptr(), s.data_ptr(), 0, ds.data_ptr(), NB, H * W, Cc, _dt(dy), st)
        W1 = w1.detach().reshape(S, Cc).float().contiguous()
        W2t = w2.detach().reshape(Cc, S).float().t().contiguous()
        dz2 = torch.empty((NB, Cc), dtype=torch.float32, device=dev)
        dz1 = torch.empty((NB, S), dtype=torch.float32, device=dev)
        dp = torch.empty((NB, Cc), dtype=torch.float32, device=dev)
        want_w = any(ctx.needs_input_grad[1:5])
        gw = torch.empty(2 * S * Cc + S + Cc, dtype=torch.float32, device=dev) if want_w else None
        o1, o2, o3 = S * Cc, 2 * S * Cc, 2 * S * Cc + S
        C().se_mlp_bwd(ds.data_ptr(), s.data_ptr(), z2.data_ptr(), buf[0].data_ptr(), p.data_ptr(), buf[1].data_ptr(),
                       W1.data_ptr(), W2t.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), dp.data_ptr(),
                       gw.data_ptr() if want_w else 0, gw[o2:].data_ptr() if want_w else 0,
                       gw[o1:].data_ptr() if want_w else 0, gw[o3:].data_ptr() if want_w else 0,
                       NB, Cc, S, ia, ig, 1.0 / (H * W), st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            C().se_scale_fwd(dy.data_ptr(), s.data_ptr(), dx.data_ptr(), NB, H * W, Cc, _dt(dy), st, dp.data_ptr())
        dw1 = _deliver(w1, gw[:o1].reshape(w1.shape).to(w1.dtype)) if ctx.needs_input_grad[1] else None
        db1 = _deliver(b1, gw[o2:o3].to(b1.dtype)) if ctx.needs_input_grad[2] else None
        dw2 = _deliver(w2, gw[o1:o2].reshape(w2.shape).to(w2.dtype)) if ctx.needs_input_grad[3] else None
        db2 = _deliver(b2, gw[o3:].to(b2.dtype)) if ctx.needs_input_grad[4] else None
        return dx, dw1, db1, dw2, db2, None, None


def se_block(x, w1, b1, w2, b2, act, gate):
    if x.shape[-1] % _vec(x.dtype):
        raise ValueError("se_block: C must be a multiple of %d" % _vec(x.dtype))
    grad_sink.note_use(w1, b1, w2, b2)
    return _SEBlockFn.apply(x, w1, b1, w2, b2, act, gate)


def se_scale(x, s):
    return _SEScaleFn.apply(x, s)


_dropout_counters: Dict[int, torch.Tensor] = {}


def _dropout_counter(device) -> torch.Tensor:
    """Device-resident RNG stream position (advanced by the kernel launch itself): masks stay fresh when the
    step is replayed from a hipGraph, where host-side offsets would be frozen at capture time."""
    t = _dropout_counters.get(device.index)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _dropout_counters[device.index] = t
    return t


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        C().dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), p, seed, 0,
                        _dropout_counter(x.device).data_ptr(), _dt(x), _stream())
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dy)
        C().dropout_bwd(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), ctx.p, _dt(dy), _stream())
        return dx, None


def dropout(x, p):
    return _DropoutFn.apply(x, float(p))


# ------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------
_ONE_TAP = ((0, 0, 0),)


def _padded_rows(t: torch.Tensor, n: int, dtype: torch.dtype) -> torch.Tensor:
    """[M, n] tensor -> contiguous [M, round_up(n, 8)] buffer in ``dtype`` with zero pad columns."""
    M = t.shape[0]
    ld = _round_up(n, 8)
    st = _stream()
    if ld == n:
        src = t if t.is_contiguous() else t.contiguous()
        if src.dtype == dtype:
            return src
        out = torch.empty((M, n), dtype=dtype, device=t.device)
        C().cast(src.data_ptr(), _dt(src), out.data_ptr(), _dt(dtype), src.numel(), st)
        return out
    src = t.contiguous()
    out = torch.empty((M, ld), dtype=dtype, device=t.device)
    C().nchw_to_nhwc(src.data_ptr(), out.data_ptr(), M, n, 1, ld, _dt(src), _dt(dtype), st)
    return out


_GEMM8 = int(os.environ.get("DPA_GEMM8", "1"))      # 256x256 multi-phase GEMM: 1 by shape, 0 never, 2 whenever legal


def _gemm8_wins(M: int, N: int, K: int, dtype, out_dtype) -> bool:
    """The 256x256 multi-phase kernel (csrc/kernels/gemm8.hip) against the lean 128-row GEMM, measured on one MI355X
    (bench/experiments/gemm8_check.py -> profiles/r2/gemm8_vs_lean.txt): it needs a long K loop to amortise its pipeline fill
    (1.1 PFLOP/s at 4096^3, 870 vs 708 TFLOP/s at 25216 x 3072 x 768) and a tile count that fills whole rounds of
    the 256 CUs (25216 x 768: 296 tiles = 2 rounds at 58 % -> the lean kernel wins there)."""
    if _GEMM8 == 0 or dtype == torch.float32 or out_dtype != dtype or K % 64 or N % 8:
        return False
    if _GEMM8 >= 2:
        return True
    tiles = -(-M // 256) * -(-N // 256)
    rounds = -(-tiles // 256)
    return K >= 768 and tiles >= 256 and tiles / (rounds * 256.0) >= 0.85


def _plain_gemm(A, B, Cbuf, bias_f, M, N, K, lda, ldb, ldc, relu, dtype, out_dtype):
    """Cbuf[M, N] = A[M, K] @ B[N, K]^T (+ bias) (+ ReLU): the multi-phase kernel where it wins, else the lean GEMM."""
    if _gemm8_wins(M, N, K, dtype, out_dtype) and lda % 8 == 0 and ldb % 8 == 0 and ldc % 8 == 0:
        C().gemm8(A.data_ptr(), B.data_ptr(), Cbuf.data_ptr(), _p(bias_f), M, N, K, lda, ldb, ldc, int(bool(relu)),
                  _dt(dtype), _stream())
        return
    taps = _tap_table(A.device, _ONE_TAP)
    _igemm(A, B, Cbuf, bias_f, None, taps, M, N, K, 1, lda, ldb, ldc, 1, 1, 1, 1, 1, 1, relu=relu, dtype=dtype,
           out_dtype=out_dtype)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, out_dtype):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, in_f = x2.shape
        out_f = weight.shape[0]
        if in_f % _vec(x2.dtype):
            raise ValueError("linear: in_features must be a multiple of %d" % _vec(x2.dtype))
        wc = linear_weight(weight, x2.dtype)
        odt = out_dtype or x2.dtype
        ldc = _round_up(out_f, 8)
        buf = torch.empty((M, ldc), dtype=odt, device=x.device)
        bias_f = None if bias is None else _f32(bias.detach())
        _plain_gemm(x2, wc, buf, bias_f, M, out_f, in_f, in_f, in_f, ldc, relu, x2.dtype, odt)
        y = buf[:, :out_f] if ldc != out_f else buf
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.bias_ref = bias
        ctx.cfg = (relu, bias is not None, lead)
        return y.reshape(*lead, out_f) if len(lead) != 1 else y

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        relu, has_bias, lead = ctx.cfg
        M, in_f = x2.shape
        out_f = weight.shape[0]
        st = _stream()
        cdt = x2.dtype
        dy2 = dy.reshape(M, out_f)
        if relu:
            dyc, yc = dy2.contiguous(), y.contiguous()
            dym = torch.empty_like(dyc)
            C().relu_bwd(yc.data_ptr(), dyc.data_ptr(), dym.data_ptr(), dyc.numel(), _dt(dyc), st)
            dy2 = dym
        dyp = _padded_rows(dy2, out_f, cdt)          # [M, outp], zero pad columns
        outp = dyp.shape[1]
        dx = dw = db = None
        taps = _tap_table(dy.device, _ONE_TAP)
        if ctx.needs_input_grad[0]:
            wt = linear_weight_t(weight, cdt, outp)   # [in, outp]
            dx = torch.empty((M, in_f), dtype=cdt, device=dy.device)
            _plain_gemm(dyp, wt, dx, None, M, in_f, outp, outp, outp, in_f, False, cdt, cdt)
            dx = dx.reshape(*lead, in_f)
        if ctx.needs_input_grad[1]:
            splits = 0
            if x2.is_contiguous() and dyp.is_contiguous() and (x2.data_ptr() | dyp.data_ptr()) % 16 == 0:
                splits, pps = C().wgrad8_splits(M, outp, 0, in_f, _dt(cdt))      # wide layers: csrc/kernels/wgrad8.hip
            if splits > 0:
                slab = torch.empty((splits, outp, in_f), dtype=torch.float32, device=dy.device)
                C().wgrad8(dyp.data_ptr(), x2.data_ptr(), slab.data_ptr(), M, outp, in_f, outp, in_f, splits, pps,
                           _dt(cdt), st)
            else:
                splits, pps = C().wgrad_choose_splits(M, outp, in_f, _dt(cdt))
                slab = torch.empty((splits, outp, in_f), dtype=torch.float32, device=dy.device)
                C().wgrad(dyp.data_ptr(), x2.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, outp, in_f, 1,
                          outp, in_f, 1, 1, 1, 1, 1, 1, splits, pps, 1, _dt(cdt), st)
            sink = grad_sink.sink_of(weight)
            if sink is not None and sink.view.is_contiguous() and outp == out_f:
                # the slab fold writes the gradient straight into the parameter's bucket view
                C().slab_reduce(slab.data_ptr(), splits, outp * in_f, sink.view.data_ptr(), _dt(weight.dtype),
                                sink.beta(), st)
                sink.delivered()
            else:
                full = torch.empty((outp, in_f), dtype=weight.dtype, device=dy.device)
                C().slab_reduce(slab.data_ptr(), splits, outp * in_f, full.data_ptr(), _dt(weight.dtype), 0.0, st)
                dw = _deliver(weight, full[:out_f])
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            db = _bias_grad(bias_p, dyp, out_f)
        return dx, dw, db, None, None


def linear(x, weight, bias, relu, w_lp, out_dtype):
    grad_sink.note_use(weight, bias)
    return _LinearFn.apply(x, weight, bias, bool(relu), out_dtype)


# ------------------------------------------------------------------------------------------
# loss / metrics
# ------------------------------------------------------------------------------------------
class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, metrics):
        N, Cc = logits.shape
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        ldl = logits.stride(0)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        rowbuf = torch.empty((3, N), dtype=torch.float32, device=logits.device)
        need_grad = logits.requires_grad
        dl = torch.empty((N, ldl), dtype=logits.dtype, device=logits.device) if need_grad else None
        tgt = target if target.dtype == torch.int64 and target.is_contiguous() else target.long().contiguous()
        C().xent_fwd_bwd(logits.data_ptr(), tgt.data_ptr(), N, Cc, ldl, _dt(logits), loss.data_ptr(), _p(dl),
                         rowbuf.data_ptr(), _p(metrics), _stream())
        ctx.dl = dl
        ctx.ncls = Cc
        if metrics is not None:
            ctx.mark_dirty(metrics)
            return loss, metrics
        return loss

    @staticmethod
    def backward(ctx, gout, *unused):
        dl = ctx.dl
        if dl is None:
            return None, None, None
        ctx.dl = None
        g = gout if gout.dtype == torch.float32 else gout.float()
        g = g.contiguous()
        C().xent_bwd_scale(dl.data_ptr(), dl.numel(), _dt(dl), g.data_ptr(), _stream())
        out = dl[:, :ctx.ncls] if dl.shape[1] != ctx.ncls else dl
        return out, None, None


def cross_entropy(logits, target):
    return _XentFn.apply(logits, target, None)


def loss_and_metrics(logits, target):
    """-> (mean loss, float32[4] = loss*n, correct@1, correct@5, n), one fused launch pair."""
    rec = torch.zeros(4, dtype=torch.float32, device=logits.device)
    loss, rec = _XentFn.apply(logits, target, rec)
    return loss, rec.detach()


def topk_correct(output, target, topk):
    rec = torch.zeros(4, dtype=torch.float32, device=output.device)
    with torch.no_grad():
        _XentFn.apply(output.detach(), target, rec)
    picks = []
    for k in topk:
        if k == 1:
            picks.append(rec[1])
        elif k == 5 or k == min(5, output.shape[1]):
            picks.append(rec[2])
        else:
            z = output.float()
            zt = z.gather(1, target.view(-1, 1))
            picks.append(((z > zt).sum(1) < k).sum().float())
    return torch.stack(picks)


# ------------------------------------------------------------------------------------------
# input pipeline
# ------------------------------------------------------------------------------------------
_norm_consts: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def normalize_u8(x, dtype, pad_c, mean, std, input_layout):
    key = (x.device.index, tuple(mean), tuple(std))
    if key not in _norm_consts:
        m = torch.tensor([v * 255.0 for v in mean], dtype=torch.float32, device=x.device)
        s = torch.tensor([1.0 / (v * 255.0) for v in std], dtype=torch.float32, device=x.device)
        _norm_consts[key] = (m, s)
    m, s = _norm_consts[key]
    x = x if x.is_contiguous() else x.contiguous()
    nchw = input_layout == "nchw"
    if nchw:
        NB, cin, H, W = x.shape
    else:
        NB, H, W, cin = x.shape
    cout = max(pad_c or 0, _round_up(cin, _vec(dtype)))
    cout = _round_up(cout, _vec(dtype))
    y = torch.empty((NB, H, W, cout), dtype=dtype, device=x.device)
    C().normalize_u8(x.data_ptr(), y.data_ptr(), NB * H * W, cin, cout, H * W, 1 if nchw else 0, m.data_ptr(),
                     s.data_ptr(), _dt(dtype), _stream())
    return y


# ------------------------------------------------------------------------------------------
# space-to-depth stem: 7x7 stride-2 pad-3 convolution of a 3-channel image == 4x4 stride-1 convolution (taps
# -2..1) of the [N, H/2, W/2, 16] space-to-depth image.  K = 256 instead of 7*7*8 = 392, input tensor half the size.
# ------------------------------------------------------------------------------------------
_STEM_S2D = os.environ.get("DPA_STEM_S2D", "1") == "1"
_S2D_TAPS = tuple((ty - 2, tx - 2, ty * 4 + tx) for ty in range(4) for tx in range(4))
# With the image rows padded by two zero pixels on the left and two on the right (to_s2d does that), the four
# horizontal taps of an output pixel are 64 contiguous, always in-range channels: the stem becomes a 4-tap x
# 64-channel problem (128-byte aligned rows) and runs on the aligned implicit-GEMM / weight-gradient kernels instead
# of the 32-byte generic gather (fprop 1.65 -> ~0.9 ms, wgrad 1.35 -> ~0.8 ms at batch 1024).
_S2D_PADL, _S2D_PADR = 2, 2
_S2D_ROW_TAPS = tuple((ty - 2, 0, ty) for ty in range(4))
_s2d_idx: Dict[int, torch.Tensor] = {}


def s2d_ok(x) -> bool:
    """Image batch (uint8 NHWC/NCHW or float NCHW) with 3 channels and even extents."""
    if not _STEM_S2D or x.dim() != 4:
        return False
    if x.dtype == torch.uint8 and x.shape[-1] == 3:
        H, W = x.shape[1], x.shape[2]
    elif x.shape[1] == 3:
        H, W = x.shape[2], x.shape[3]
    else:
        return False
    return H % 2 == 0 and W % 2 == 0 and H >= 2 and W >= 2


def to_s2d(x, dtype, mean, std):
    x = x if x.is_contiguous() else x.contiguous()
    nhwc = x.dtype == torch.uint8 and x.shape[-1] == 3
    if nhwc:
        NB, H, W, _ = x.shape
    else:
        NB, _, H, W = x.shape
    base = torch.empty((NB, H // 2, W // 2 + _S2D_PADL + _S2D_PADR, 16), dtype=dtype, device=x.device)
    y = base[:, :, _S2D_PADL:_S2D_PADL + W // 2]     # [N, H/2, W/2, 16] view of the zero-padded rows
    if x.dtype == torch.uint8:
        key = (x.device.index, tuple(mean), tuple(std))
        if key not in _norm_consts:
            m = torch.tensor([v * 255.0 for v in mean], dtype=torch.float32, device=x.device)
            s = torch.tensor([1.0 / (v * 255.0) for v in std], dtype=torch.float32, device=x.device)
            _norm_consts[key] = (m, s)
        m, s = _norm_consts[key]
        C().to_s2d(x.data_ptr(), base.data_ptr(), NB, H // 2, W // 2, 0 if nhwc else 1, -1, m.data_ptr(), s.data_ptr(),
                   _dt(dtype), _stream(), _S2D_PADL, _S2D_PADR)
    else:
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        C().to_s2d(x.data_ptr(), base.data_ptr(), NB, H // 2, W // 2, 1, _dt(x), 0, 0, _dt(dtype), _stream(),
                   _S2D_PADL, _S2D_PADR)
    y._dpa_s2d_base = base
    y._dpa_s2d = True
    y._dpa_nhwc = True
    return y


def _s2d_index(device) -> torch.Tensor:
    """Position of w[k, c, r, s] (flattened (r, s, c)) inside the [16 taps][16 channels] row of the s2d filter."""
    t = _s2d_idx.get(device.index)
    if t is None:
        idx = [(((r + 1) // 2) * 4 + (s + 1) // 2) * 16 + (((r + 1) % 2) * 2 + (s + 1) % 2) * 3 + c
               for r in range(7) for s in range(7) for c in range(3)]
        t = torch.tensor(idx, dtype=torch.int64, device=device)
        _s2d_idx[device.index] = t
    return t


def stem_weight_s2d(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    K = w.shape[0]
    c = _cache(w)
    key = (dtype, "s2d")
    if c.wc is not None and c.wc_key == key and c.wc_stamp == _stamp(w):
        return c.wc
    if c.wc is None or c.wc_key != key:
        c.wc = torch.zeros((K, 256), dtype=dtype, device=w.device)
    c.wc[:, _s2d_index(w.device)] = w.detach().permute(0, 2, 3, 1).reshape(K, 147).to(dtype)
    c.wc_key, c.wc_stamp = key, _stamp(w)
    return c.wc


class _StemS2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, want_stats):
        # `x` from to_s2d() is a view of zero-padded rows (`_dpa_s2d_base`): 4 taps x 64 contiguous channels; any
        # other [N, H/2, W/2, 16] tensor takes the 16 taps x 16 channels form
        _tap_table(x.device, _S2D_TAPS)          # both tables exist after the first (eager) call: a table made
        _tap_table(x.device, _S2D_ROW_TAPS)      # inside a hipGraph capture would be an illegal host-to-device copy
        base = getattr(x, "_dpa_s2d_base", None)
        padded = base is not None and x.data_ptr() == base.data_ptr() + _S2D_PADL * 16 * x.element_size() \
            and x.stride() == base.stride()
        if not padded:
            x = x if x.is_contiguous() else x.contiguous()
        NB, H2, W2, Cx = x.shape
        K = weight.shape[0]
        if Cx != 16 or tuple(weight.shape[1:]) != (3, 7, 7):
            raise ValueError("space-to-depth stem needs a [N, H/2, W/2, 16] input and a [K, 3, 7, 7] filter")
        w2 = stem_weight_s2d(weight, x.dtype)
        M = NB * H2 * W2
        y = torch.empty((NB, H2, W2, K), dtype=x.dtype, device=x.device)
        src, cred, ntaps, iw = (base, 64, 4, W2 + _S2D_PADL + _S2D_PADR) if padded else (x, 16, 16, W2)
        stats = None
        if want_stats:
            tiles = C().igemm_stats_rows(M, K, cred, ntaps, _dt(x), _dt(x))
            stats = _stats_slab(tiles, K, x.device)
        taps = _tap_table(x.device, _S2D_ROW_TAPS if padded else _S2D_TAPS)
        bias_f = None
        if bias is not None:
            bias_f = bias.detach() if bias.dtype == torch.float32 else bias.detach().float()
        _igemm(src, w2, y, bias_f, stats, taps, M, K, cred, ntaps, 16, 256, K, H2, iw, 1, 1, H2, W2, relu=relu,
               dtype=x.dtype, out_dtype=x.dtype)
        ctx.padded = padded
        ctx.src = src if padded else None
        ctx.cfg = (relu, bias is not None)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, dstats=None):
        if dy is None:
            return None, None, None, None, None
        x, weight, y = ctx.saved_tensors
        relu, has_bias = ctx.cfg
        NB, H2, W2, _ = x.shape
        K = weight.shape[0]
        M = NB * H2 * W2
        dy = dy if dy.is_contiguous() else dy.contiguous()
        st = _stream()
        if relu:
            dym = torch.empty_like(dy)
            C().relu_bwd(y.data_ptr(), dy.data_ptr(), dym.data_ptr(), dy.numel(), _dt(dy), st)
            dy = dym
        dw = db = None
        if ctx.needs_input_grad[1]:
            splits, pps = C().wgrad_choose_splits(M, K, 256, _dt(dy))
            slab = torch.empty((splits, K, 256), dtype=torch.float32, device=dy.device)
            if ctx.padded:
                taps = _tap_table(dy.device, _S2D_ROW_TAPS)
                C().wgrad(dy.data_ptr(), ctx.src.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, K, 64, 4, K, 16, H2,
                          W2 + _S2D_PADL + _S2D_PADR, 1, 1, H2, W2, splits, pps, 0, _dt(dy), st)
            else:
                taps = _tap_table(dy.device, _S2D_TAPS)
                C().wgrad(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), taps.data_ptr(), M, K, 16, 16, K, 16, H2, W2,
                          1, 1, H2, W2, splits, pps, 0, _dt(dy), st)
            full = torch.empty((K, 256), dtype=torch.float32, device=dy.device)
            C().slab_reduce(slab.data_ptr(), splits, K * 256, full.data_ptr(), 0, 0.0, st)
            g = full[:, _s2d_index(dy.device)].view(K, 7, 7, 3).permute(0, 3, 1, 2)
            dw = _deliver(weight, g.to(weight.dtype))
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            db = _bias_grad(bias_p, dy.view(M, K))
        return None, dw, db, None, None


def stem_s2d_conv(x, weight, bias, stride, padding, relu, want_stats):
    if tuple(stride) != (2, 2) or tuple(padding) != (3, 3):
        raise ValueError("space-to-depth input reached a convolution that is not a 7x7 stride-2 pad-3 stem")
    grad_sink.note_use(weight, bias)
    y, stats = _StemS2DFn.apply(x, weight, bias, bool(relu), bool(want_stats))
    return (y, stats) if want_stats else y


def nchw_to_nhwc(x, dtype, pad_c):
    x = x if x.is_contiguous() else x.contiguous()
    NB, Cc, H, W = x.shape
    cout = _round_up(max(pad_c or 0, Cc), _vec(dtype))
    y = torch.empty((NB, H, W, cout), dtype=dtype, device=x.device)
    C().nchw_to_nhwc(x.data_ptr(), y.data_ptr(), NB, Cc, H * W, cout, _dt(x), _dt(dtype), _stream())
    return y


# ------------------------------------------------------------------------------------------
# optimizer / AMP helpers
# ------------------------------------------------------------------------------------------
def multi_tensor_check_finite(grads, found_inf: torch.Tensor) -> None:
    st = _stream()
    for g in grads:
        gg = g if g.is_contiguous() or _dense(g) else g.contiguous()
        C().check_finite(gg.data_ptr(), gg.numel(), _dt(gg), found_inf.data_ptr(), st)


def scaler_update(scale, inv_scale, good_steps, found_inf, growth_interval, growth_factor, backoff_factor,
                  max_scale, min_scale) -> None:
    C().scaler_update(scale.data_ptr(), inv_scale.data_ptr(), good_steps.data_ptr(), found_inf.data_ptr(),
                      int(growth_interval), float(growth_factor), float(backoff_factor), float(max_scale),
                      float(min_scale), _stream())


def _dense(t: torch.Tensor) -> bool:
    """True when the tensor covers its memory densely (any permutation of a contiguous block)."""
    if t.is_contiguous():
        return True
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return True
    return False


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    if a.shape != b.shape:
        return False
    sa = [s for s, n in zip(a.stride(), a.shape) if n > 1]
    sb = [s for s, n in zip(b.stride(), b.shape) if n > 1]
    return sa == sb


def _lp_copy_of(p: torch.Tensor):
    cache = getattr(p, "_dpa_wcache", None)
    if cache is not None and cache.wc is not None and cache.wc.dtype != torch.float32 \
            and cache.wc.numel() == p.numel():
        return cache.wc
    return None


class SgdPlan:
    """Device-resident tensor tables (one per param group) for the fused multi-tensor SGD launch."""

    def __init__(self, opt):
        import struct

        chunk = C().sgd_chunk_elems()
        self.groups = []
        self.params_with_copies = []
        for group in opt.param_groups:
            entries, keep, plist, nchunks = [], [], [], 0
            g_dtype = lp_dtype = None
            device = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32:
                    raise TypeError("FusedSGD HIP path needs fp32 parameters, got %s" % p.dtype)
                if not _dense(p):
                    raise ValueError("FusedSGD: parameter is not densely laid out")
                g = p.grad
                if not _same_layout(p, g):
                    g = torch.empty_like(p).copy_(g)
                    p.grad = g
                device = p.device
                buf = opt._momentum_buffer(p) if group["momentum"] != 0 else None
                if buf is not None and not _same_layout(p, buf):
                    buf = torch.empty_like(p).copy_(buf)
                    opt.state[p]["momentum_buffer"] = buf
                if g_dtype is None:
                    g_dtype = _dt(g)
                elif g_dtype != _dt(g):
                    raise TypeError("FusedSGD: mixed gradient dtypes in one group")
                lp = _lp_copy_of(p)
                if lp is not None:
                    if lp_dtype is None:
                        lp_dtype = _dt(lp)
                    elif lp_dtype != _dt(lp):
                        lp = None
                n = p.numel()
                entries.append((p.data_ptr(), g.data_ptr(), _p(buf), _p(lp), n, nchunks))
                keep.extend([p, g, buf, lp])
                plist.append((p, lp))
                nchunks += (n + chunk - 1) // chunk
            if not entries:
                continue
            raw = b"".join(struct.pack("<QQQQqq", *e) for e in entries)
            assert len(raw) == len(entries) * C().sgd_table_entry_bytes()
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
            self.groups.append(dict(group=group, table=table, n=len(entries), chunks=nchunks, g_dtype=g_dtype,
                                    lp_dtype=lp_dtype if lp_dtype is not None else 0,
                                    hyper=torch.zeros(6, dtype=torch.float32, device=device), hyper_host=None,
                                    plist=plist, keep=keep))
            self.params_with_copies.extend(p for p, lp in plist if lp is not None)
        self.sig = self.signature(opt)

    @staticmethod
    def signature(opt):
        return tuple((p.data_ptr(), p.grad.data_ptr(), _p(_lp_copy_of(p)))
                     for g in opt.param_groups for p in g["params"] if p.grad is not None)

    def matches(self, opt) -> bool:
        return self.sig == self.signature(opt)

    def sync_hyper(self, opt) -> None:
        """Refresh the device-side hyper-parameter array (lr, momentum, wd, dampening, nesterov, first-step)
        when the host values changed.  Separate from launch() so a captured step never records the copy."""
        for gt in self.groups:
            group = gt["group"]
            first = 1.0 if opt._steps == 0 else 0.0
            host = (float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]),
                    float(group["dampening"]), 1.0 if group["nesterov"] else 0.0, first)
            if gt["hyper_host"] != host:
                gt["hyper"].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)
                gt["hyper_host"] = host

    def launch(self, opt, scaler) -> None:
        st = _stream()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper(opt)
        for gt in self.groups:
            inv = found = 0
            if scaler is not None and not scaler.is_identity:
                inv, found = scaler.inv_scale.data_ptr(), scaler.found_inf.data_ptr()
            C().sgd_multi_tensor(gt["table"].data_ptr(), gt["n"], gt["chunks"], gt["g_dtype"], gt["lp_dtype"],
                                 gt["hyper"].data_ptr(), inv, found, st)
        # parameters changed behind torch's back: derived copies (transposes, padded casts) are stale,
        # the low-precision copies the kernel wrote are current
        bump_generation()
        for gt in self.groups:
            for p, lp in gt["plist"]:
                if lp is not None:
                    p._dpa_wcache.wc_stamp = _stamp(p)

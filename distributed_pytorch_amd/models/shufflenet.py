"""ShuffleNetV2 x0.5 / x1.0 / x1.5 / x2.0 in NHWC (torchvision zoo members reachable through the reference's
``models.__dict__[arch]()``, ``nd_imagenet.py:22-24,133-138``).

Module names match torchvision (``conv1.0.weight``, ``stage2.0.branch1.0.weight``, ``stage3.4.branch2.5.weight``,
``conv5.1.running_mean``, ``fc.weight``) so its checkpoints load.  Parameter counts: x0_5 1,366,792; x1_0 2,278,604;
x1_5 3,503,624; x2_0 7,393,996.

Two things are different from every other family of the zoo:

* **The channel shuffle.**  ``channel_shuffle(cat(x1, branch2(x2)), 2)`` is always followed by the next block's
  ``chunk(2)``, so a stage's activation lives as a PAIR of half-width tensors and shuffle + split is one kernel
  (``F.shuffle_pair``, ``csrc/kernels/elementwise.hip``): no concatenated tensor, no strided halves.
* **58 / 116 / 122 / 244-channel halves** (x1.0, x2.0) do not fill the 16-byte channel vectors every kernel of this
  framework moves.  On the GPU each half is stored padded to the vector width with ZERO channels (58 -> 64,
  122 -> 128) and the tiny parameter tensors are scattered into the padded layout by differentiable torch ops
  (``_scatter``) right before they are handed to the HIP convolution / BatchNorm kernels: zero filters produce zero
  channels, the parameters (and their checkpoints) keep torchvision's shapes, and their gradients return through the
  scatter's autograd node.  x0.5 / x1.5 (24 / 48 / 96, 88 / 176 / 352) need no padding and take the ordinary fused path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from ..nn.modules import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLU, conv_bn_act
from ..ops import functional as F
from .alexnet_vgg import _ImageNetCNN


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Layout:
    """Where the logical channels of a tensor sit physically: segments (the halves of a pair, concatenated), each
    padded to ``vec`` channels."""
    _cache: Dict[tuple, "_Layout"] = {}

    def __init__(self, segs: Tuple[int, ...], vec: int):
        pos: List[int] = []
        off = 0
        for n in segs:
            pos.extend(range(off, off + n))
            off += _round_up(n, vec)
        self.pos, self.width, self.n = pos, off, sum(segs)
        self.identity = self.width == self.n
        self.single = len(segs) == 1              # the padding is one tail: scatter == pad, gather == a leading slice
        self._idx: Dict[torch.device, torch.Tensor] = {}

    @classmethod
    def of(cls, segs, vec: int) -> "_Layout":
        key = (tuple(segs), vec)
        lay = cls._cache.get(key)
        if lay is None:
            lay = cls._cache[key] = cls(tuple(segs), vec)
        return lay

    def index(self, device) -> torch.Tensor:
        t = self._idx.get(device)
        if t is None:
            t = self._idx[device] = torch.tensor(self.pos, dtype=torch.long, device=device)
        return t


def _scatter(p: torch.Tensor, dim: int, lay: _Layout, fill: float = 0.0) -> torch.Tensor:
    """``p`` with its ``dim`` spread over the padded layout (differentiable; ``fill`` in the padding)."""
    if lay.identity:
        return p
    if lay.single:          # one launch forward, a view backward (these models are host-launch bound)
        return torch.nn.functional.pad(p, [0, 0] * (p.dim() - 1 - dim) + [0, lay.width - lay.n], value=fill)
    shape = list(p.shape)
    shape[dim] = lay.width
    return p.new_full(shape, fill).index_copy(dim, lay.index(p.device), p)


def _padded_running_stats(bn: BatchNorm2d, lay: _Layout):
    """Running statistics in the padded layout.  Single-segment layouts keep them PERSISTENTLY padded: the module's
    ``running_mean`` / ``running_var`` buffers become leading views of the padded tensors the kernels update in place
    (checkpoints still see torchvision's shapes), which removes eight small launches per BatchNorm and step.  Returns
    (rm, rv, copy_back)."""
    if bn.running_mean is None:
        return None, None, False
    if not lay.single:
        with torch.no_grad():
            return _scatter(bn.running_mean, 0, lay), _scatter(bn.running_var, 0, lay, 1.0), True
    pad = getattr(bn, "_dpa_pad_stats", None)
    if pad is None or pad[0].shape[0] != lay.width or pad[0].data_ptr() != bn.running_mean.data_ptr() \
            or pad[1].data_ptr() != bn.running_var.data_ptr():
        bn._dpa_pad_stats = None
        if bn.running_mean._base is not None or bn.running_var._base is not None:
            # the buffers are views of somebody else's storage -- DistributedDataParallel re-homes every buffer into
            # one flat arena that it broadcasts per step: leave them there and copy back after the kernel
            with torch.no_grad():
                return _scatter(bn.running_mean, 0, lay), _scatter(bn.running_var, 0, lay, 1.0), True
        with torch.no_grad():           # first call, or the module was moved / reloaded since
            rm, rv = _scatter(bn.running_mean, 0, lay), _scatter(bn.running_var, 0, lay, 1.0)
        bn.running_mean, bn.running_var = rm[:lay.n], rv[:lay.n]
        pad = bn._dpa_pad_stats = (rm, rv)
    return pad[0], pad[1], False


def _cba(x: torch.Tensor, conv: Conv2d, bn: BatchNorm2d, relu: bool, lin: _Layout, lout: _Layout,
         depthwise: bool = False) -> torch.Tensor:
    """conv -> BN (-> ReLU) on padded layouts.  Aligned layouts take the zoo's ordinary fused path."""
    if lin.identity and lout.identity:
        return conv_bn_act(x, conv, bn, relu=relu)
    w = _scatter(conv.weight, 0, lout)
    groups = conv.groups
    if depthwise:
        groups = lout.width                      # one zero filter per padding channel
    else:
        w = _scatter(w, 1, lin)
    fuse = F._on_gpu(x) and bn.training and bn.process_group is None
    y = F.conv2d(x, w, None, conv.stride, conv.padding, want_stats=fuse, needs_input_grad=conv.needs_input_grad,
                 groups=groups)
    stats = None
    if fuse:
        y, stats = y
    rm, rv, copy_back = _padded_running_stats(bn, lout)
    out = F.batch_norm(y, _scatter(bn.weight, 0, lout, 1.0), _scatter(bn.bias, 0, lout), rm, rv,
                       bn.num_batches_tracked, bn.training, bn.momentum, bn.eps, relu=relu, stats=stats,
                       group=bn.process_group)
    if bn.training and copy_back:
        with torch.no_grad():
            if lout.single:
                bn.running_mean.copy_(rm[:lout.n])
                bn.running_var.copy_(rv[:lout.n])
            else:
                idx = lout.index(rm.device)
                bn.running_mean.copy_(rm.index_select(0, idx))
                bn.running_var.copy_(rv.index_select(0, idx))
    return out


class InvertedResidual(nn.Module):
    def __init__(self, inp: int, oup: int, stride: int):
        super().__init__()
        if not 1 <= stride <= 3:
            raise ValueError("illegal stride value")
        self.stride = stride
        bf = oup // 2
        if stride == 1 and inp != bf << 1:
            raise ValueError("Invalid combination of stride %d, inp %d and oup %d" % (stride, inp, oup))
        self.inp, self.bf = inp, bf
        if stride > 1:
            self.branch1 = nn.Sequential(
                Conv2d(inp, inp, 3, stride=stride, padding=1, bias=False, groups=inp), BatchNorm2d(inp),
                Conv2d(inp, bf, 1, bias=False), BatchNorm2d(bf), ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(
            Conv2d(inp if stride > 1 else bf, bf, 1, bias=False), BatchNorm2d(bf), ReLU(inplace=True),
            Conv2d(bf, bf, 3, stride=stride, padding=1, bias=False, groups=bf), BatchNorm2d(bf),
            Conv2d(bf, bf, 1, bias=False), BatchNorm2d(bf), ReLU(inplace=True))

    def _branch2(self, x, lin: _Layout, vec: int):
        b, half = self.branch2, _Layout.of((self.bf,), vec)
        h = _cba(x, b[0], b[1], True, lin, half)
        h = _cba(h, b[3], b[4], False, half, half, depthwise=True)
        return _cba(h, b[5], b[6], True, half, half)

    def forward(self, x, lin: Optional[_Layout], vec: int):
        """stride 1: ``x`` is the pair (left, right) of the previous block; stride 2: one tensor in layout ``lin``.
        Returns the pair after the channel shuffle."""
        half = _Layout.of((self.bf,), vec)
        if self.stride == 1:
            left, right = x
            return F.shuffle_pair(left, self._branch2(right, half, vec), self.bf)
        b = self.branch1
        h = _cba(x, b[0], b[1], False, lin, lin, depthwise=True)
        b1 = _cba(h, b[2], b[3], True, lin, half)
        return F.shuffle_pair(b1, self._branch2(x, lin, vec), self.bf)


class ShuffleNetV2(_ImageNetCNN):
    def __init__(self, stages_repeats: List[int], stages_out_channels: List[int], num_classes: int = 1000):
        super().__init__()
        if len(stages_repeats) != 3 or len(stages_out_channels) != 5:
            raise ValueError("expected 3 stage repeats and 5 stage widths")
        self._stage_out_channels = stages_out_channels
        cin, cout = 3, stages_out_channels[0]
        self.conv1 = nn.Sequential(Conv2d(cin, cout, 3, stride=2, padding=1, bias=False), BatchNorm2d(cout),
                                   ReLU(inplace=True))
        self.conv1[0].needs_input_grad = False
        cin = cout
        self.maxpool = MaxPool2d(3, stride=2, padding=1)
        for name, repeats, cout in zip(("stage2", "stage3", "stage4"), stages_repeats, stages_out_channels[1:]):
            seq = [InvertedResidual(cin, cout, 2)]
            seq += [InvertedResidual(cout, cout, 1) for _ in range(repeats - 1)]
            setattr(self, name, nn.Sequential(*seq))
            cin = cout
        cout = stages_out_channels[-1]
        self.conv5 = nn.Sequential(Conv2d(cin, cout, 1, bias=False), BatchNorm2d(cout), ReLU(inplace=True))
        self.avgpool = AdaptiveAvgPool2d((1, 1))        # torchvision: x.mean([2, 3]); no parameters either way
        self.fc = Linear(cout, num_classes)

    force_vec = 0        # tests: take the padded-channel code path on the CPU reference backend too

    def forward(self, x):
        x = self.prepare_input(x)
        vec = self.force_vec or (1 if not F._on_gpu(x) else (4 if x.dtype == torch.float32 else 8))
        x = conv_bn_act(x, self.conv1[0], self.conv1[1], relu=True)
        x = self.maxpool(x)
        lay = _Layout.of((x.shape[-1],), vec)
        pair = None
        for stage in (self.stage2, self.stage3, self.stage4):
            for blk in stage:
                if blk.stride > 1:
                    if pair is not None:
                        x = F.cat_channels(pair)
                    pair = blk(x, lay, vec)
                else:
                    pair = blk(pair, None, vec)
            lay = _Layout.of((stage[0].bf, stage[0].bf), vec)
        x = F.cat_channels(pair)
        x = _cba(x, self.conv5[0], self.conv5[1], True, lay, _Layout.of((self.conv5[0].out_channels,), vec))
        x = self.avgpool(x)
        return self.fc(x.reshape(x.shape[0], -1), out_dtype=torch.float32)


def _shufflenetv2(repeats, widths, **kw):
    return ShuffleNetV2(repeats, widths, **kw)


def shufflenet_v2_x0_5(**kw):
    return _shufflenetv2([4, 8, 4], [24, 48, 96, 192, 1024], **kw)


def shufflenet_v2_x1_0(**kw):
    return _shufflenetv2([4, 8, 4], [24, 116, 232, 464, 1024], **kw)


def shufflenet_v2_x1_5(**kw):
    return _shufflenetv2([4, 8, 4], [24, 176, 352, 704, 1024], **kw)


def shufflenet_v2_x2_0(**kw):
    return _shufflenetv2([4, 8, 4], [24, 244, 488, 976, 2048], **kw)

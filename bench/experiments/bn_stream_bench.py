"""Achieved bandwidth of the BatchNorm streaming passes (apply / backward apply) on ResNet-50 shapes, batch 1024, bf16.

    bn_apply:      relu only (bn1 / bn2),  residual + relu + bit mask (bn3),  no activation (downsample)
    bn_bwd_apply:  mask mode 2 (recomputed), 3 (bit mask, + unmasked residual gradient), 0
"""
import os

import torch

from distributed_pytorch_amd import backend

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
SHAPES = [(3211264, 256), (3211264, 64), (802816, 512), (802816, 128), (200704, 1024), (200704, 256), (50176, 2048),
          (50176, 512)]
if os.environ.get('SHORT'):
    SHAPES = SHAPES[:3]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


if "NTMASK" in os.environ:
    ext.set_nt_mask(int(os.environ["NTMASK"]))
if "GRID" in os.environ:
    ext.bn_set_tuning(int(os.environ["GRID"]), 1)
if "REV" in os.environ:
    ext.bn_set_reverse(int(os.environ["REV"]))
GRIDS = [int(g) for g in os.environ.get("GRIDS", "0").split(",")]
REVS = [int(g) for g in os.environ.get("REVS", "0").split(",")]
QUIET = bool(os.environ.get("QUIET"))
OFF = int(os.environ.get("OFF", "0"))      # byte skew between the streams' base addresses
tot = {}
for M, C in SHAPES:
    def alloc(k, rand=True):
        pad = OFF * 4 // 2
        buf = torch.randn(M * C + pad, device=dev, dtype=torch.bfloat16) if rand else \
            torch.empty(M * C + pad, device=dev, dtype=torch.bfloat16)
        o = k * OFF // 2
        return buf[o:o + M * C].view(M, C)
    x, r, y, y2 = alloc(0), alloc(1), alloc(2, False), alloc(3, False)
    bits = torch.randint(0, 255, (M * C // 8,), device=dev, dtype=torch.uint8)
    scale = torch.ones(C, device=dev)
    shift = torch.zeros(C, device=dev)
    coef = torch.randn(5 * C, device=dev)
    nb = M * C * 2 / 1e9
    cases = [
        ("apply relu", 2 * nb, lambda: ext.bn_apply(x.data_ptr(), 0, y.data_ptr(), M, C, 1, scale.data_ptr(),
                                                   shift.data_ptr(), 1, s, 0, 0, 0)),
        ("apply res+relu+mask", 3.0625 * nb, lambda: ext.bn_apply(x.data_ptr(), r.data_ptr(), y.data_ptr(), M, C, 1,
                                                                 scale.data_ptr(), shift.data_ptr(), 1, s,
                                                                 bits.data_ptr(), 0, 0)),
        ("apply res(bn)+relu+mask", 3.0625 * nb, lambda: ext.bn_apply(x.data_ptr(), r.data_ptr(), y.data_ptr(), M, C, 1,
                                                                     scale.data_ptr(), shift.data_ptr(), 1, s,
                                                                     bits.data_ptr(), scale.data_ptr(),
                                                                     shift.data_ptr())),
        ("bwd mode2", 3 * nb, lambda: ext.bn_bwd_apply(r.data_ptr(), x.data_ptr(), x.data_ptr(), y.data_ptr(), 0, M, C,
                                                      1, coef.data_ptr(), 2, s)),
        ("bwd mode3", 3.0625 * nb, lambda: ext.bn_bwd_apply(r.data_ptr(), x.data_ptr(), bits.data_ptr(), y.data_ptr(),
                                                           0, M, C, 1, coef.data_ptr(), 3, s)),
        ("bwd mode0", 3 * nb, lambda: ext.bn_bwd_apply(r.data_ptr(), x.data_ptr(), 0, y.data_ptr(), 0, M, C, 1,
                                                      coef.data_ptr(), 0, s)),
        ("bwd mode3+dres", 4.0625 * nb, lambda: ext.bn_bwd_apply(r.data_ptr(), x.data_ptr(), bits.data_ptr(),
                                                                y.data_ptr(), y2.data_ptr(), M, C, 1, coef.data_ptr(),
                                                                3, s)),
    ]
    for name, gb, fn in cases:
        for grid in GRIDS:
            for rev in REVS:
                if grid:
                    ext.bn_set_tuning(grid, 1)
                ext.bn_set_reverse(rev)
                us = timed(fn)
                key = "%s g%d r%d" % (name, grid, rev)
                tot[key] = tot.get(key, 0.0) + us
                if not QUIET:
                    print("M=%8d C=%5d %-24s g%-5d r%-3d %8.1f us  %.2f TB/s" % (M, C, name, grid, rev, us, gb / us * 1e3),
                          flush=True)
    del x, r, y, y2, bits
for k, v in tot.items():
    print("total %-40s %8.1f us" % (k, v))

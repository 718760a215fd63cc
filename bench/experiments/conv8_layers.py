#!/usr/bin/env python
"""conv8.hip against the kernels it would replace, layer by layer (fprop with BN statistics, data gradient with the fused
BN-backward reduction), batch 1024 ResNet-50 and batch 128 VGG-16 shapes.  GPU time through hipGraph replay.
    python bench/experiments/conv8_layers.py [--batch 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timeit(fn, iters=10):
    """GPU time per call through hipGraph replay (bench/conv_layers.py); a failed capture is an error here."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    s.record()
    g.replay()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * iters) * 1e-3

from distributed_pytorch_amd.ops import hip_ops  # noqa: E402

# (name, images at batch 1024 scale, H, Cin, Cout, k)
LAYERS = [
    ("r50 l3 conv1", 1.0, 14, 1024, 256, 1), ("r50 l3 conv2", 1.0, 14, 256, 256, 3), ("r50 l3 conv3", 1.0, 14, 256, 1024, 1),
    ("r50 l4 conv1", 1.0, 7, 2048, 512, 1), ("r50 l4 conv2", 1.0, 7, 512, 512, 3), ("r50 l4 conv3", 1.0, 7, 512, 2048, 1),
    ("r50 l2 conv1", 1.0, 28, 512, 128, 1), ("r50 l2 conv2", 1.0, 28, 128, 128, 3),
    ("vgg conv3_2", 0.125, 56, 256, 256, 3), ("vgg conv4_1", 0.125, 28, 256, 512, 3), ("vgg conv4_2", 0.125, 28, 512, 512, 3),
    ("vgg conv5", 0.125, 14, 512, 512, 3), ("vgg conv2_2", 0.125, 112, 128, 128, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--modes", default="0,1,2")
    ap.add_argument("--tail", type=int, default=4)
    args = ap.parse_args()
    C = hip_ops.C()
    C.conv8_set_tail_split(args.tail)
    dt = torch.bfloat16
    modes = [int(m) for m in args.modes.split(",")]
    print("# ms per launch; mode 0 = igemm / lean / filter-row kernels, 1 = conv8 one tile per workgroup, 2 = conv8 stream-K")
    print("# %-14s %-28s %s" % ("layer", "problem (M x N x K)", "  ".join("fprop+stats m%d  dgrad+bnb m%d" % (m, m) for m in modes)))
    for name, frac, H, Cin, Cout, k in LAYERS:
        NB = max(1, int(args.batch * frac))
        p = k // 2
        x = torch.randn(NB, H, H, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(NB, H, H, Cout, device="cuda").to(dt)
        coef = torch.stack([torch.zeros(Cin), torch.ones(Cin), torch.ones(Cin), torch.zeros(Cin)]).cuda().contiguous()
        bits = torch.randint(0, 256, (x.numel() // 8,), dtype=torch.uint8, device="cuda")
        M = NB * H * H
        flops = 2.0 * M * Cin * Cout * k * k
        cells = []
        for m in modes:
            C.conv8_set_mode(m)
            n0 = C.conv8_launches()
            tf = timeit(lambda: hip_ops.conv2d(x, w, None, (1, 1), (p, p), False, None, True, False))
            td = timeit(lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (1, 1), (p, p), bnred=(x, bits, coef, 3)))
            took = C.conv8_launches() > n0
            info = "[g%d r%d]" % (C.conv8_last(0), C.conv8_last(1)) if took else ""
            cells.append("%7.3f (%4.0f TF)%s %7.3f (%4.0f TF) %s" % (tf * 1e3, flops / tf / 1e12, "*" if took else " ",
                                                                    td * 1e3, flops / td / 1e12, info))
        C.conv8_set_mode(-1)
        hip_ops._pending_bnred.clear()
        print("%-16s %-28s %s" % (name, "%d x %d x %d" % (M, Cout, Cin * k * k), "   ".join(cells)), flush=True)
    torch.cuda.synchronize()
    print("# conv8 error word:", C.conv8_error(torch.cuda.current_stream().cuda_stream))


if __name__ == "__main__":
    main()

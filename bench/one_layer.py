#!/usr/bin/env python
"""Run ONE convolution kernel (fprop | dgrad | wgrad) of a given shape / tile configuration a few times:
the target of `rocprofv3 --pmc ...` counter passes (profiles/r1_pmc_*.md).
    python bench/one_layer.py H Cin Cout k stride kind [cfg] [iters] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    H, Cin, Cout, k, s = (int(v) for v in sys.argv[1:6])
    kind = sys.argv[6]
    cfg = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
    N = int(sys.argv[9]) if len(sys.argv) > 9 else 256
    from distributed_pytorch_amd.ops import hip_ops

    C = hip_ops.C()
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    dt = torch.bfloat16
    x = torch.randn(N, H, H, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, OH, OH, Cout, device="cuda").to(dt)
    if kind == "wgrad":
        C.wgrad_force_config(cfg)
        fn = lambda: hip_ops.conv_wgrad(dy, x, w, (s, s), (p, p))
    elif kind == "dgrad":
        C.igemm_force_config(cfg)
        fn = lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (s, s), (p, p))
    else:
        C.igemm_force_config(cfg)
        fn = lambda: hip_ops.conv2d(x, w, None, (s, s), (p, p), False, None, True, False)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s %dx%d %d->%d k%d s%d cfg %d: %.3f ms" % (kind, H, H, Cin, Cout, k, s, cfg, e0.elapsed_time(e1) / iters))


if __name__ == "__main__":
    main()

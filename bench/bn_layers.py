#!/usr/bin/env python
"""Bandwidth of the BatchNorm streaming kernels on ResNet-50's activation shapes (GPU time through hipGraph
replay).  `--tune` sweeps the launch shape hooks (bn_set_tuning)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench.conv_layers import timeit  # noqa: E402

SHAPES = [(112, 64, 1), (56, 64, 7), (56, 256, 4), (28, 128, 8), (28, 512, 5), (14, 256, 12), (14, 1024, 7),
          (7, 512, 6), (7, 2048, 4)]      # (H, C, BN layers of that shape in ResNet-50)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tune", action="store_true")
    args = ap.parse_args()
    from distributed_pytorch_amd.ops import hip_ops

    C = hip_ops.C()
    cs = lambda: torch.cuda.current_stream().cuda_stream     # evaluated per call: the capture stream differs
    settings = [(2048, 1)] if not args.tune else [(2048, 1), (2048, 2), (2048, 4), (1024, 2), (1024, 4), (4096, 1),
                                                  (4096, 2), (512, 4)]
    for grid, unroll in settings:
        C.bn_set_tuning(grid, unroll)
        tot = {"apply": 0.0, "reduce": 0.0, "bwd_apply": 0.0}
        lines = []
        for H, Cc, cnt in SHAPES:
            M = args.batch * H * H
            x = torch.randn(M, Cc, device="cuda").bfloat16()
            dy = torch.randn(M, Cc, device="cuda").bfloat16()
            y = torch.empty_like(x)
            dx = torch.empty_like(x)
            coef = torch.rand(5, Cc, device="cuda") + 0.5
            nblk = C.bn_partial_rows(M, Cc)
            part = torch.empty(nblk, 2, Cc, device="cuda")
            dt = hip_ops._dt(x)
            t_a = timeit(lambda: C.bn_apply(x.data_ptr(), 0, y.data_ptr(), M, Cc, dt, coef[0].data_ptr(),
                                            coef[1].data_ptr(), 1, cs()))
            t_r = timeit(lambda: C.bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), 0, M, Cc, dt, coef[0].data_ptr(),
                                                 coef[1].data_ptr(), 2, coef[3].data_ptr(), coef[4].data_ptr(),
                                                 part.data_ptr(), nblk, cs()))
            t_b = timeit(lambda: C.bn_bwd_apply(dy.data_ptr(), x.data_ptr(), 0, dx.data_ptr(), 0, M, Cc, dt,
                                                coef.data_ptr(), 2, cs()))
            nbytes = x.numel() * 2
            tot["apply"] += t_a * cnt
            tot["reduce"] += t_r * cnt
            tot["bwd_apply"] += t_b * cnt
            lines.append("  %3dx%-3d C=%-4d x%-2d apply %.3f ms %.2f TB/s | reduce %.3f ms %.2f TB/s | bwd_apply %.3f ms %.2f TB/s"
                         % (H, H, Cc, cnt, t_a * 1e3, 2 * nbytes / t_a / 1e12, t_r * 1e3, 2 * nbytes / t_r / 1e12,
                            t_b * 1e3, 3 * nbytes / t_b / 1e12))
        print("grid %d unroll %d: weighted ms/step apply %.2f reduce %.2f bwd_apply %.2f" % (
            grid, unroll, tot["apply"] * 1e3, tot["reduce"] * 1e3, tot["bwd_apply"] * 1e3), flush=True)
        if (grid, unroll) == (2048, 1):
            print("\n".join(lines), flush=True)
    C.bn_set_tuning(0, 0)


if __name__ == "__main__":
    main()

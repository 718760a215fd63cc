"""Achieved bandwidth of the stand-alone BatchNorm-backward column reduction on the ResNet-50 shapes that still use it
(downsample BNs, bn1 of the stride-2 blocks, the last bn3) -- batch 1024, bf16."""
import sys

import torch

from distributed_pytorch_amd import backend

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
SHAPES = [(3211264, 256), (802816, 512), (200704, 1024), (50176, 2048), (3211264, 128), (802816, 256), (200704, 512),
          (3211264, 64)]
modes = [int(a) for a in sys.argv[1:]] or [0, 2, 3]
for M, C in SHAPES:
    dy = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    bits = torch.randint(0, 255, (M * C // 8,), device=dev, dtype=torch.uint8)
    mean = torch.zeros(C, device=dev)
    invstd = torch.ones(C, device=dev)
    scale = torch.ones(C, device=dev)
    shift = torch.zeros(C, device=dev)
    nblk = ext.bn_partial_rows(M, C)
    partial = torch.empty(nblk * 2 * C + nblk, device=dev)
    for mode in modes:
        y = bits if mode == 3 else x

        def run():
            ext.bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), y.data_ptr(), M, C, 1, mean.data_ptr(), invstd.data_ptr(),
                              mode, scale.data_ptr(), shift.data_ptr(), partial.data_ptr(), nblk, s)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        gb = M * C * (4 + (0.125 if mode == 3 else 0)) / 1e9
        print("M=%8d C=%5d mode=%d nblk=%5d  %8.1f us  %.2f TB/s" % (M, C, mode, nblk, us, gb / us * 1e3), flush=True)
    del dy, x, bits

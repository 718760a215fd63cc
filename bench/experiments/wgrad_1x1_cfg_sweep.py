"""Tile configurations of wgrad.hip on the deep 1x1 weight gradients of ResNet-50 (b1024)."""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = 1024
NAMES = {-1: "auto", 0: "128x128", 5: "256x256", 6: "256x128", 7: "128x256", 8: "128x128 pf2", 9: "256x128 pf2"}
for (H, Cin, Cout) in [(14, 256, 1024), (14, 1024, 256), (7, 512, 2048), (7, 2048, 512), (28, 128, 512), (28, 512, 128)]:
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(N, H, H, Cout, device="cuda").bfloat16()
    w = torch.zeros(Cout, Cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    row = []
    for cfg in (-1, 0, 8, 6, 9, 7, 5):
        C.wgrad_force_config(cfg)
        try:
            t = timeit(lambda: hip_ops.conv_wgrad(dy, x, w, (1, 1), (0, 0)))
            row.append("%s %.3f" % (NAMES[cfg], t * 1e3))
        except Exception as e:
            row.append("%s n/a" % NAMES[cfg])
    C.wgrad_force_config(-1)
    print("%2dx%-2d x[%4d] dy[%4d]: " % (H, H, Cin, Cout) + " | ".join(row), flush=True)

"""Ablation of the plain lean GEMM (build with DPA_BUILD_DEFS=-DDPA_LEAN_ABLATE): which part does the time follow?"""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = 1024
NAMES = {0: "full", 1: "no MFMA", 2: "no LDS reads + MFMA", 3: "no loop loads", 4: "no store", 5: "no loop loads + no LDS writes", 6: "5 + no loop barriers"}
for (H, Cin, Cout) in [(14, 1024, 512), (14, 256, 1024), (14, 1024, 256), (28, 512, 128), (7, 2048, 512)]:
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    for k in (1, 2):
        C.igemm_set_lean(k)
        row = []
        for a in (0, 5, 6):
            C.igemm_set_lean_ablate(a)
            t = timeit(lambda: hip_ops.conv2d(x, w, None, (1, 1), (0, 0), False, None, False, False))
            row.append("%s %.3f" % (NAMES[a], t * 1e3))
        print("%2dx%-2d %4d->%-4d K tile %3d B: " % (H, H, Cin, Cout, 64 * k) + " | ".join(row), flush=True)
C.igemm_set_lean_ablate(0)
C.igemm_set_lean(-1)

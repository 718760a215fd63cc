"""wgrad8.hip against wgrad.hip on the deep dense 1x1 weight gradients of ResNet-50 (b1024), with and without the Gram
operand of the BN fold."""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
NB = 1024
print("layer                         gram   wgrad.hip   wgrad8   PFLOP/s (wgrad8)   splits")
for (H, Cin, Cout) in [(14, 256, 1024), (14, 1024, 256), (14, 512, 1024), (7, 512, 2048), (7, 2048, 512), (7, 1024, 2048),
                       (28, 256, 512), (28, 512, 256)]:
    M = NB * H * H
    x = torch.randn(M, Cin, device="cuda").bfloat16()
    dy = torch.randn(M, Cout, device="cuda").bfloat16()
    for gram in (False, True):
        row = []
        for mode in (0, 1):
            C.wgrad8_set_mode(mode)
            row.append(timeit(lambda: hip_ops._wgrad_raw(dy, x, M, Cout, Cin, 1, M, gram=gram)))
        s = C.wgrad8_splits(M, Cout, Cin if gram else 0, Cin, hip_ops._dt(dy))[0]
        C.wgrad8_set_mode(-1)
        fl = 2.0 * M * (Cout + (Cin if gram else 0)) * Cin
        print("%2dx%-2d x[%4d] dy[%4d]       %-5s  %8.3f  %8.3f   %8.3f   %6d" % (H, H, Cin, Cout, gram, row[0] * 1e3, row[1] * 1e3,
                                                                            fl / row[1] / 1e15, s), flush=True)

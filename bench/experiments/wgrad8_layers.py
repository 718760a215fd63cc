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

print("3x3 / strided layers            wgrad.hip   wgrad8   PFLOP/s (wgrad8)")
for (NBc, H, Cin, Cout, k, stride) in [(1024, 14, 256, 256, 3, 1), (1024, 28, 256, 256, 3, 2), (1024, 7, 512, 512, 3, 1),
                                       (1024, 14, 512, 512, 3, 2), (128, 56, 256, 256, 3, 1), (128, 28, 256, 512, 3, 1),
                                       (128, 28, 512, 512, 3, 1), (128, 14, 512, 512, 3, 1)]:
    x = torch.randn(NBc, H, H, Cin, device="cuda").bfloat16()
    OH = (H + 2 - k) // stride + 1
    dy = torch.randn(NBc, OH, OH, Cout, device="cuda").bfloat16()
    w = torch.zeros(Cout, Cin, k, k, device="cuda").contiguous(memory_format=torch.channels_last)
    row = []
    for mode in (0, -1):
        C.wgrad8_set_mode(mode)
        row.append(timeit(lambda: hip_ops.conv_wgrad(dy, x, w, (stride, stride), (1, 1))))
    C.wgrad8_set_mode(-1)
    fl = 2.0 * NBc * OH * OH * Cout * Cin * k * k
    print("b%-4d %2dx%-2d %dx%d s%d x[%4d] dy[%4d]  %8.3f  %8.3f   %8.3f" % (NBc, H, H, k, k, stride, Cin, Cout, row[0] * 1e3,
                                                                       row[1] * 1e3, fl / row[1] / 1e15), flush=True)

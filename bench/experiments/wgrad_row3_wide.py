"""Filter-row weight gradient on wide images (VGG conv1_2 / conv2_x at b128) against wgrad.hip."""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
for (NB, H, Cin, Cout) in [(128, 224, 64, 64), (128, 112, 64, 128), (128, 112, 128, 128)]:
    x = torch.randn(NB, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(NB, H, H, Cout, device="cuda").bfloat16()
    w = torch.zeros(Cout, Cin, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    row = []
    for mode in (0, 1):
        C.wgrad_row3_set_mode(mode)
        row.append(timeit(lambda: hip_ops.conv_wgrad(dy, x, w, (1, 1), (1, 1))))
    C.wgrad_row3_set_mode(-1)
    fl = 2.0 * NB * H * H * Cout * Cin * 9
    print("b%d %dx%d x[%d] dy[%d]: wgrad.hip %.3f ms, filter-row %.3f ms (%.2f PFLOP/s)" % (NB, H, H, Cin, Cout, row[0] * 1e3,
                                                                                         row[1] * 1e3, fl / row[1] / 1e15), flush=True)

import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = [(56, 64, 64, 1), (56, 64, 256, 1), (56, 256, 64, 1), (56, 256, 128, 1), (28, 128, 512, 1), (56, 256, 512, 2),
     (28, 512, 128, 1), (28, 512, 256, 1), (14, 256, 1024, 1), (28, 512, 1024, 2), (14, 1024, 256, 1),
     (14, 1024, 512, 1), (7, 512, 2048, 1), (14, 1024, 2048, 2), (7, 2048, 512, 1)]
for (H, Cin, Cout, s) in L:
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    OH = (H - 1) // s + 1
    dy = torch.randn(N, OH, OH, Cout, device="cuda").bfloat16()
    row = []
    for mode in (0, 1, 2, -1):
        C.igemm_set_lean(mode)
        f = timeit(lambda: hip_ops.conv2d(x, w, None, (s, s), (0, 0), False, None, True, False))
        d = timeit(lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (s, s), (0, 0))) if s == 1 else float("nan")
        row.append("m%2d: f %.3f d %.3f" % (mode, f * 1e3, d * 1e3))
    C.igemm_set_lean(-1)
    print("%2dx%-2d %4d->%-4d s%d  " % (H, H, Cin, Cout, s) + " | ".join(row), flush=True)

import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for (H, Cin, Cout) in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]:
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, H, H, Cout, device="cuda").bfloat16()
    row = []
    for mode in (0, 1, 2, -1):
        C.igemm_set_row3(mode)
        f = timeit(lambda: hip_ops.conv2d(x, w, None, (1, 1), (1, 1), False, None, True, False))
        d = timeit(lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (1, 1), (1, 1)))
        row.append("mode %2d: fprop %.3f dgrad %.3f" % (mode, f * 1e3, d * 1e3))
    C.igemm_set_row3(-1)
    print("%dx%d %d->%d  " % (H, H, Cin, Cout) + " | ".join(row), flush=True)

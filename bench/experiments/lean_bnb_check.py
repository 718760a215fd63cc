import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
# conv1 (1x1, Cin -> Cmid) data gradients with fused accumulate + BN-backward reduction, conv3 ones with the reduction
L = [(56, 256, 64, True), (28, 512, 128, True), (14, 1024, 256, True), (7, 2048, 512, True),
     (56, 64, 256, False), (28, 128, 512, False), (14, 256, 1024, False), (7, 512, 2048, False)]
for (H, Cin, Cout, acc) in L:
    dy = torch.randn(N, H, H, Cout, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    bn_x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    coef = torch.rand(4, Cin, device="cuda") + 0.5
    bits = torch.randint(0, 256, (N * H * H * Cin // 8,), dtype=torch.uint8, device="cuda")
    other = torch.randn(N, H, H, Cin, device="cuda").bfloat16() if acc else None
    gb = (dy.numel() + bn_x.numel() * (3 if acc else 2)) * 2 / 1e9 + bits.numel() * (2 if acc else 1) / 1e9
    row = []
    for mode in (0, -1):
        C.igemm_set_lean(mode)
        def f():
            dx = hip_ops.conv_dgrad(dy, w, tuple(bn_x.shape), (1, 1), (0, 0), accum=other, bnred=(bn_x, bits, coef, 3),
                                    accum_mask=bits if acc else None)
            hip_ops._pending_bnred.clear()
        t = timeit(f)
        row.append("lean %2d: %.3f ms %.2f TB/s" % (mode, t * 1e3, gb / t / 1e3))
    C.igemm_set_lean(-1)
    print("%2dx%-2d dy[%4d] -> dx[%4d] accum=%d  %.2f GB  " % (H, H, Cout, Cin, acc, gb) + " | ".join(row), flush=True)

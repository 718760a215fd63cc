"""Ablation of the lean data gradient with fused accumulate + folded BN-backward sums (conv1 of a bottleneck, the
largest kernel of the ResNet-50 step): build with DPA_BUILD_DEFS=-DDPA_LEAN_ABLATE."""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = 1024
NAMES = {0: "full", 3: "no loop loads", 4: "no store", 5: "no loop loads/LDS writes", 7: "no epilogue operand loads", 8: "no column sums", 9: "column sums, no slab store", 10: "sums accumulated, not reduced", 11: "reduced in the wave only"}
for (H, Cin, Cout) in [(56, 256, 64), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)]:
    dy = torch.randn(N, H, H, Cout, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    bn_x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    coef = torch.rand(4, Cin, device="cuda") + 0.5
    bits = torch.randint(0, 256, (N * H * H * Cin // 8,), dtype=torch.uint8, device="cuda")
    other = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    row = []
    for a in (0, 100):
        C.igemm_set_lean(3 if a == 100 else -1)
        def f():
            hip_ops.conv_dgrad(dy, w, tuple(bn_x.shape), (1, 1), (0, 0), accum=other, bnred=(bn_x, bits, coef, 3, True),
                               accum_mask=bits)
            hip_ops._pending_bnred.clear()
        row.append("%s %.3f" % (NAMES.get(a, "128 x 64 tiles"), timeit(f) * 1e3))
    C.igemm_set_lean(-1)
    print("%2dx%-2d dy[%4d] -> dx[%4d]: " % (H, H, Cout, Cin) + " | ".join(row), flush=True)
C.igemm_set_lean_ablate(0)

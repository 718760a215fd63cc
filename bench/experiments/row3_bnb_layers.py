"""3x3 data gradients with / without the fused BN-backward reduction (bn1 of a bottleneck) per ResNet-50 stage, b1024."""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
N = 1024
for (H, Cc) in [(56, 64), (28, 128), (14, 256), (7, 512)]:
    dy = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = (torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    bn_x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    coef = torch.rand(4, Cc, device="cuda") + 0.5
    bits = torch.randint(0, 256, (N * H * H * Cc // 8,), dtype=torch.uint8, device="cuda")
    row = []
    for name, bnred in (("plain", None), ("bnb mask bits", (bn_x, bits, coef, 3)), ("bnb mask recomputed", (bn_x, None, coef, 2)),
                        ("bnb no relu", (bn_x, None, coef, 0))):
        def f():
            hip_ops.conv_dgrad(dy, w, (N, H, H, Cc), (1, 1), (1, 1), bnred=bnred)
            hip_ops._pending_bnred.clear()
        row.append("%s %.3f" % (name, timeit(f) * 1e3))
    print("%2dx%-2d %4d: " % (H, H, Cc) + " | ".join(row), flush=True)

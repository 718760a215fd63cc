"""layer1 3x3 data gradient (56x56, 64 -> 64, batch 1024) plain and with the fused BN-backward sums: counter passes."""
import sys, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd.ops import hip_ops
N, H, Cc = 1024, 56, 64
dy = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
w = (torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
bn_x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
coef = torch.rand(4, Cc, device="cuda") + 0.5
for _ in range(3):
    hip_ops.conv_dgrad(dy, w, (N, H, H, Cc), (1, 1), (1, 1))
    hip_ops.conv_dgrad(dy, w, (N, H, H, Cc), (1, 1), (1, 1), bnred=(bn_x, None, coef, 0))
    hip_ops._pending_bnred.clear()
torch.cuda.synchronize()

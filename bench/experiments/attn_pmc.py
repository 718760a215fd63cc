"""Fused attention forward + backward at the ViT-B/16 shape (b128) for the PMC / timing scripts."""
import sys, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd.ops import hip_ops
B, S, H, D = 128, 197, 12, 64
qkv = (torch.randn(B, S, 3 * H * D, device="cuda") * 0.8).bfloat16().requires_grad_(True)
scale = torch.full((H,), D ** -0.5, device="cuda")
gy = torch.randn(B, S, H * D, device="cuda").bfloat16()
for _ in range(4):
    y = hip_ops.attention(qkv, H, scale)
    y.backward(gy)
    qkv.grad = None
torch.cuda.synchronize()

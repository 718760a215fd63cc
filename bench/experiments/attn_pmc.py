"""Fused attention forward + backward at the ViT-B/16 shape (b128) for the PMC / timing scripts."""
import sys, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd.ops import hip_ops
import os
B, S, H, D = (int(v) for v in os.environ.get("ATTN_SHAPE", "128,197,12,64").split(","))
NW = int(os.environ.get("ATTN_WINDOWS", "0"))            # > 0: relative-position bias + a per-window mask (Swin)
qkv = (torch.randn(B, S, 3 * H * D, device="cuda") * 0.8).bfloat16().requires_grad_(True)
scale = torch.full((H,), D ** -0.5, device="cuda")
gy = torch.randn(B, S, H * D, device="cuda").bfloat16()
rel = (torch.randn(H, S, S, device="cuda") * 0.5).requires_grad_(True) if NW else None
mask = torch.zeros(NW, S, S, device="cuda") if NW else None
for _ in range(4):
    y = hip_ops.attention(qkv, H, scale, rel, mask)
    y.backward(gy)
    qkv.grad = None
torch.cuda.synchronize()

import torch, torch.nn.functional as TF
from distributed_pytorch_amd.ops import functional as F, hip_ops
torch.manual_seed(5)
for (H, W, k, s, p) in [(17, 17, 3, 1, 1), (6, 6, 3, 1, 1)]:
    x = torch.randn(2, H, W, 40, device="cuda").requires_grad_(True)
    y = F.avg_pool2d(x, k, s, p)
    gy = torch.ones_like(y)
    y.backward(gy)
    xr = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
    yr = TF.avg_pool2d(xr, k, s, p)
    yr.backward(gy.permute(0, 3, 1, 2))
    print("fwd err", (y.permute(0, 3, 1, 2) - yr).abs().max().item())
    g = x.grad.permute(0, 3, 1, 2)
    print("bwd err", (g - xr.grad).abs().max().item())
    print(g[0, 0, :4, :4]); print(xr.grad[0, 0, :4, :4])
    print(g[1, 39, -3:, -3:]); print(xr.grad[1, 39, -3:, -3:])

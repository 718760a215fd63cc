import torch, torch.nn.functional as TF
from distributed_pytorch_amd.ops import functional as F
DEV = "cuda"
torch.manual_seed(21)
ref_act = {"sigmoid": torch.sigmoid, "silu": TF.silu, "hardsigmoid": TF.hardsigmoid, "hardswish": TF.hardswish}
for dtype in (torch.bfloat16, torch.float16):
    for shape in ((3, 5, 7, 40), (2, 1283)):
        x0 = (torch.randn(*shape, device=DEV) * 3).to(dtype)
        g0 = torch.randn(*shape, device=DEV).to(dtype)
        for kind, fn in ref_act.items():
            x = x0.clone().requires_grad_(True)
            y = F.activation(x, kind)
            y.backward(g0)
            xr = x0.float().clone().requires_grad_(True)
            yr = fn(xr)
            yr.backward(g0.float())
            ey = (y.float() - yr).abs()
            eg = (x.grad.float() - xr.grad).abs()
            i = eg.argmax()
            print(dtype, shape, kind, "y err %.4g" % ey.max().item(), "g err %.4g" % eg.max().item(),
                  "at x=%.6f g=%.4f mine=%.4f ref=%.4f" % (x0.flatten()[i].item(), g0.flatten()[i].item(),
                                                         x.grad.flatten()[i].item(), xr.grad.flatten()[i].item()))

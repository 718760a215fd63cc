import torch, copy
from distributed_pytorch_amd import models
from distributed_pytorch_amd.ops import functional as F
torch.manual_seed(0)
m = models.create_model("swin_t", num_classes=1000).cuda().train()
x = torch.randn(4, 3, 64, 64).cuda(); t = torch.randint(0, 1000, (4,)).cuda()
acts = {}
def mk(name):
    def hook(mod, inp, out):
        o = out if isinstance(out, torch.Tensor) else out[0]
        acts[name] = o
        if o.requires_grad:
            o.register_hook(lambda g, name=name: print("grad", name, tuple(g.shape), bool(torch.isfinite(g).all()), float(g.float().abs().max())))
        print("fwd", name, tuple(o.shape), bool(torch.isfinite(o).all()), float(o.float().abs().max()))
    return hook
for n, mod in m.named_modules():
    if n.startswith("features.5.0") or n.startswith("features.7.1") or n in ("features.6", "features.4"):
        mod.register_forward_hook(mk(n))
loss, rec = F.loss_and_metrics(m(x), t)
loss.backward()
print("loss", float(loss))

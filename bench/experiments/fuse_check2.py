import copy, sys, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import hip_ops, functional as F
from distributed_pytorch_amd.optim import FusedSGD
torch.manual_seed(0)
base = models.create_model("resnet50").cuda().train()
x = torch.randint(0, 256, (32, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (32,), device="cuda")
def grads(level, fuse):
    hip_ops._FUSE_BN_BWD = fuse
    amp.state().reset()
    m = copy.deepcopy(base)
    opt = FusedSGD(m.parameters(), lr=0.0)
    if level == "O0":
        m, opt = amp.initialize(m, opt, opt_level="O0")
    else:
        m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
    hip_ops.bn_bwd_counts.update(fused=0, reduced=0)
    loss, _ = F.loss_and_metrics(m(x), t)
    opt.zero_grad(); loss.backward()
    for a in opt._arenas(): a.finish_step()
    torch.cuda.synchronize()
    print(level, "fuse", fuse, dict(hip_ops.bn_bwd_counts), float(loss))
    return {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}
ref = grads("O0", False)
a = grads("O2", False)
b = grads("O2", True)
import statistics
ea = [float((a[n] - ref[n]).norm() / (ref[n].norm() + 1e-12)) for n in ref]
eb = [float((b[n] - ref[n]).norm() / (ref[n].norm() + 1e-12)) for n in ref]
print("bf16 unfused vs fp32: mean %.4f max %.4f" % (statistics.mean(ea), max(ea)))
print("bf16 fused   vs fp32: mean %.4f max %.4f" % (statistics.mean(eb), max(eb)))
for n in list(ref)[:6] + list(ref)[-4:]:
    print(n, "ref %.4g a %.4g b %.4g  cos(a,ref) %.4f" % (float(ref[n].norm()), float(a[n].norm()), float(b[n].norm()),
          float((a[n].flatten() @ ref[n].flatten()) / (a[n].norm() * ref[n].norm() + 1e-12))))

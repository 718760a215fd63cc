import copy, sys, time, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import hip_ops, functional as F
torch.manual_seed(0)
base = models.create_model("resnet50").cuda().train()
x = torch.randint(0, 256, (32, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (32,), device="cuda")
res = []
for fuse in (False, True):
    hip_ops._FUSE_BN_BWD = fuse
    amp.state().reset()
    m = copy.deepcopy(base)
    from distributed_pytorch_amd.optim import FusedSGD
    opt = FusedSGD(m.parameters(), lr=0.0)
    m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
    hip_ops.bn_bwd_counts.update(fused=0, reduced=0)
    loss, _ = F.loss_and_metrics(m(x), t)
    opt.zero_grad(); loss.backward(); opt._arenas()[0].finish_step() if opt._arenas() else None
    torch.cuda.synchronize()
    print("fuse", fuse, dict(hip_ops.bn_bwd_counts), float(loss))
    res.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
worst = 0
for n in res[0]:
    a, b = res[0][n], res[1][n]
    rel = float((a - b).norm() / (a.norm() + 1e-12))
    worst = max(worst, rel)
    if rel > 1e-2: print("DIFF", n, rel, float(a.norm()), float((a-b).abs().max()))
print("worst rel diff", worst)

"""Does training behave the same with the forward fold?  ResNet-50 memorising ONE fixed batch (128 images, random labels)
for 60 SGD steps, DPA_BN_FWD_FOLD on / off (and the backward fold off), same seeds: the loss curves should track."""
import sys, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.ops import hip_ops
from distributed_pytorch_amd.optim import FusedSGD

def run(fwd, fold=True, steps=60):
    hip_ops._BN_FWD_FOLD, hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB = fwd, fold, 0.0
    torch.manual_seed(0)
    net = models.create_model("resnet50", num_classes=100).cuda().train()
    opt = FusedSGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    net, opt = amp.initialize(net, opt, opt_level="O2", dtype="bf16")
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(0, 256, (128, 224, 224, 3), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.randint(0, 100, (128,), device="cuda", generator=g)
    out = []
    for i in range(steps):
        opt.zero_grad()
        loss = F.cross_entropy(net(net.prepare_input(x)), y)
        loss.backward()
        opt.step()
        if i % 6 == 5:
            out.append(float(loss))
    return out, hip_ops.bn_fwd_fold_counts["fused"]

for name, fwd, fold in (("forward + backward fold", True, True), ("backward fold only", False, True), ("no fold", False, False)):
    hip_ops.bn_fwd_fold_counts["fused"] = 0
    curve, fused = run(fwd, fold)
    print("%-24s fused launches %4d  loss every 6 steps: %s" % (name, fused, " ".join("%.3f" % v for v in curve)), flush=True)

import torch, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _gpu_dist_workers as G
from distributed_pytorch_amd import amp
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD
amp.state().reset()
model = G._small_resnet().cuda().train()
for m in model.modules():
    if hasattr(m, "running_mean"):
        m.eval()
opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8).cuda()
t = torch.randint(0, 10, (16,)).cuda()
for it in range(2):
    loss, _ = F.loss_and_metrics(model(x), t)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    print(it, [n for n, p in model.named_parameters() if p.grad is None])
    opt.step()

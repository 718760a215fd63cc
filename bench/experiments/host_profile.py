"""cProfile of the host side of a training step (no GPU sync inside the loop)."""
import cProfile
import pstats
import sys
import time

import torch

from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
model = models.create_model(arch).cuda().train()
opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
x = torch.randint(0, 256, (bs, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (bs,), device="cuda")


def step():
    out = model(x)
    loss, rec = F.loss_and_metrics(out, t)
    opt.zero_grad()
    with amp.scale_loss(loss, opt) as s:
        s.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time per step %.2f ms (enqueue only), %.2f ms incl. drain" % ((t1 - t0) * 50, (t2 - t0) * 50))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

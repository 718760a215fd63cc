import os, sys, time, torch
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"
import torch.distributed as dist
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD
from distributed_pytorch_amd.parallel import DistributedDataParallel
ddp = len(sys.argv) > 1 and sys.argv[1] == "ddp"
torch.cuda.set_device(0)
if ddp:
    dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
m = models.create_model("resnet50").cuda().train()
opt = FusedSGD(m.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4)
m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
if ddp:
    m = DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False)
x = torch.randint(0, 256, (1024, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (1024,), device="cuda")
acc = [0.0] * 4
def step(rec):
    t0 = time.perf_counter()
    loss, _ = F.loss_and_metrics(m(x), t)
    t1 = time.perf_counter()
    opt.zero_grad()
    with amp.scale_loss(loss, opt) as s:
        s.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if rec:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
for i in range(4): step(False)
torch.cuda.synchronize()
T0 = time.perf_counter()
N = 10
for i in range(N): step(True)
t_cpu = time.perf_counter() - T0
torch.cuda.synchronize()
t_all = time.perf_counter() - T0
print("ddp" if ddp else "plain", "CPU ms/step: fwd %.1f bwd %.1f opt %.1f | loop cpu %.1f ms/step, wall %.1f ms/step" %
      (acc[0] / N * 1e3, acc[1] / N * 1e3, acc[2] / N * 1e3, t_cpu / N * 1e3, t_all / N * 1e3))
if ddp: dist.destroy_process_group()

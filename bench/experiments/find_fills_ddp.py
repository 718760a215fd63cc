import os, sys, torch, collections, traceback
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"
import torch.distributed as dist
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD
from distributed_pytorch_amd.parallel import DistributedDataParallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
m = models.create_model("resnet50").cuda().train()
opt = FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
m = DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False)
x = torch.randint(0, 256, (32, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (32,), device="cuda")
def step():
    loss, _ = F.loss_and_metrics(m(x), t)
    opt.zero_grad()
    with amp.scale_loss(loss, opt) as s:
        s.backward()
    opt.step()
for _ in range(3): step()
counts = collections.Counter()
def where():
    fr = traceback.extract_stack()[:-2]
    fr = [f for f in fr if "distributed_pytorch_amd" in f.filename]
    return " <- ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in fr[-3:])
for name in ("zero_", "fill_", "copy_", "add_"):
    orig = getattr(torch.Tensor, name)
    def mk(orig, name):
        def f(self, *a, **k):
            counts[name + " " + where()] += 1
            return orig(self, *a, **k)
        return f
    setattr(torch.Tensor, name, mk(orig, name))
step()
for k, v in counts.most_common(12): print(v, k)
dist.destroy_process_group()

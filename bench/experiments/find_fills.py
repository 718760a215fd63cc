import sys, torch, collections, traceback
sys.path.insert(0, '.')
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD
m = models.create_model("resnet50").cuda().train()
opt = FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
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
orig_zero, orig_fill, orig_zeros = torch.Tensor.zero_, torch.Tensor.fill_, torch.zeros
def where():
    fr = traceback.extract_stack()[:-2]
    fr = [f for f in fr if "distributed_pytorch_amd" in f.filename or "scratch" in f.filename]
    return " <- ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in fr[-3:])
def zero_(self):
    counts["zero_ " + str(tuple(self.shape)) + " " + where()] += 1; return orig_zero(self)
def fill_(self, v):
    counts["fill_ " + str(tuple(self.shape)) + " " + where()] += 1; return orig_fill(self, v)
def zeros(*a, **k):
    counts["zeros " + str(a[0] if a else "") + " " + where()] += 1; return orig_zeros(*a, **k)
torch.Tensor.zero_, torch.Tensor.fill_, torch.zeros = zero_, fill_, zeros
step()
torch.Tensor.zero_, torch.Tensor.fill_, torch.zeros = orig_zero, orig_fill, orig_zeros
for k, v in counts.most_common(20): print(v, k)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if "fill" in e.key.lower() or "zero" in e.key.lower() or "copy" in e.key.lower()]
for e in rows: print(e.key, e.count)

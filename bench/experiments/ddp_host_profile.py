"""Host-side cost of the DDP wrapper on a launch-bound model: per-phase host time and a cProfile of the step,
plain vs --force-ddp (1-rank communicator, no bytes move).  usage: ddp_host_profile.py ARCH BATCH [ddp]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd import amp, launch, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD
from distributed_pytorch_amd.parallel import process_group as pg

arch, batch = sys.argv[1], int(sys.argv[2])
ddp = len(sys.argv) > 3 and sys.argv[3] == "ddp"
torch.cuda.set_device(0)
if ddp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
    os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"
    pg.init_process_group("native", "env://", 1, 0, timeout_s=60, use_gpu=True)
torch.manual_seed(0)
model = models.create_model(arch).cuda()
opt = FusedSGD(model.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4)
model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
if ddp:
    from distributed_pytorch_amd.parallel import DistributedDataParallel
    model = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False)
model.train()
x = torch.randint(0, 256, (batch, 224, 224, 3), dtype=torch.uint8, device="cuda")
t = torch.randint(0, 1000, (batch,), device="cuda")
ph = [0.0, 0.0, 0.0, 0.0]
def step(timed=False):
    t0 = time.perf_counter()
    out = model(x); loss, _ = F.loss_and_metrics(out, t)
    t1 = time.perf_counter()
    opt.zero_grad()
    t2 = time.perf_counter()
    with amp.scale_loss(loss, opt) as s:
        s.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if timed:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            ph[i] += d
for _ in range(10): step()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n): step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("%s b%d ddp=%s: wall %.3f ms/step, host-issue %.3f ms/step; host phases fwd %.3f zero %.3f bwd %.3f step %.3f" % (
    arch, batch, ddp, 1e3 * wall / n, 1e3 * host / n, *[1e3 * p / n for p in ph]))
# where inside forward does the host time go while the GPU queue is deep?  per-leaf-module host time
import collections
acc = collections.OrderedDict()
base = getattr(model, "module", model)
def wrap(name, mod):
    orig = mod.forward
    def f(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    mod.forward = f
for name, mod in base.named_modules():
    if not list(mod.children()):
        wrap(name + ":" + type(mod).__name__, mod)
for _ in range(40): step()
torch.cuda.synchronize()
print("per-leaf host ms/step (deep queue):", ", ".join("%s %.3f" % (k, 1e3 * v / 40) for k, v in acc.items()))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
if ddp:
    pg.destroy_process_group()

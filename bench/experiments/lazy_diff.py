"""Gradient differences: same mode twice (determinism) and lazy vs materialised residual gradient (fp32 resnet50)."""
import copy
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import hip_ops
from test_gpu_model import _run

amp.state().reset()
torch.manual_seed(1)
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
base = models.create_model(arch, num_classes=100).cuda()
x = torch.randn(4, 3, 64, 64, device="cuda")
t = torch.randint(0, 100, (4,), device="cuda")
runs = []
for lazy in (True, True, False, False):
    hip_ops._LAZY_DRES = lazy
    runs.append(_run(copy.deepcopy(base), x, t)[2])
hip_ops._LAZY_DRES = True


def worst(a, b):
    w = (0.0, "")
    for n in a:
        d = float((a[n] - b[n]).abs().max()) / max(float(b[n].abs().max()), 1e-30)
        w = max(w, (d, n))
    return w


print("lazy vs lazy        ", worst(runs[0], runs[1]))
print("material vs material", worst(runs[2], runs[3]))
print("lazy vs material    ", worst(runs[0], runs[2]))

import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_bnfold as T
from distributed_pytorch_amd.ops import hip_ops
def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 3
class S(T.Stack):
    def __init__(self, planes=64, blocks=blocks, stride=0):
        super().__init__(planes, blocks, stride)
T.Stack = S
net1, x, gy, y1, dx1, g1, folded = T._run(True, 64, (4, 14, 14), fwd=True)
print("fused", hip_ops.bn_fwd_fold_counts["fused"], "folded", folded)
net0, _, _, y0, dx0, g0, _ = T._run(True, 64, (4, 14, 14), fwd=False)
print("y", rel(y1, y0), "dx", rel(dx1, dx0))
for n in g1:
    print("%-28s %.4f" % (n, rel(g1[n], g0[n])))

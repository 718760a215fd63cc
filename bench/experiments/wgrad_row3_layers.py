"""Per-layer A/B of the filter-row weight gradient (wgrad_row3.hip) against wgrad.hip on the 3x3 layers of ResNet-50
(b1024) and VGG-16 (b128): python bench/experiments/wgrad_row3_layers.py"""
import torch
from distributed_pytorch_amd.ops import hip_ops

LAYERS = [("r50 l1 56x56x64", 1024, 56, 64, 64), ("r50 l2 28x28x128", 1024, 28, 128, 128), ("r50 l3 14x14x256", 1024, 14, 256, 256),
          ("r50 l4 7x7x512", 1024, 7, 512, 512), ("vgg 224x224 64", 128, 224, 64, 64), ("vgg 112 128", 128, 112, 128, 128),
          ("vgg 56 256", 128, 56, 256, 256), ("vgg 28 512", 128, 28, 512, 512), ("vgg 14 512", 128, 14, 512, 512)]


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


C = hip_ops.C()
print("%-22s %9s %9s %9s %9s  %s" % ("layer", "wgrad ms", "row3 v0", "row3 v1", "row3 v2", "TFLOP/s best"))
for name, N, HW, Ci, Co in LAYERS:
    x = torch.randn(N, HW, HW, Ci, device="cuda").bfloat16()
    dy = torch.randn(N, HW, HW, Co, device="cuda").bfloat16()
    w = torch.zeros(Co, Ci, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    splits, _ = C.wgrad_row3_splits(N, HW, HW, Co, Ci, 3, hip_ops._dt(dy))
    f = lambda: hip_ops.conv_wgrad(dy, x, w, (1, 1), (1, 1))
    C.wgrad_row3_set_mode(0)
    t0 = t(f)
    C.wgrad_row3_set_mode(1)
    splits, _ = C.wgrad_row3_splits(N, HW, HW, Co, Ci, 3, hip_ops._dt(dy))
    ts = []
    for v in (0, 1, 2):
        C.wgrad_row3_set_variant(v)
        ts.append(t(f) if splits > 0 and (v < 2 or Co <= 64) else float("nan"))
    C.wgrad_row3_set_variant(0)
    C.wgrad_row3_set_mode(-1)
    fl = 2.0 * N * HW * HW * Ci * Co * 9
    best = min(x for x in ts + [t0] if x == x)
    print("%-22s %9.3f %9.3f %9.3f %9.3f  %.0f   (splits %d)" % (name, t0, ts[0], ts[1], ts[2], fl / best / 1e9, splits), flush=True)

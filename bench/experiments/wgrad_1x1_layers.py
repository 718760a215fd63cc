"""1x1 weight gradients of ResNet-50 at batch 1024 against their bandwidth roofline (dy and x read once):
python bench/experiments/wgrad_1x1_layers.py"""
import sys, torch
sys.path.insert(0, '.')
from bench.conv_layers import timeit
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
N = 1024
print("%-28s %8s %8s %8s" % ("layer (x[C] , dy[N])", "ms", "GB", "TB/s"))
for (H, Cin, Cout) in [(56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024), (14, 1024, 256),
                       (7, 512, 2048), (7, 2048, 512), (56, 64, 64)]:
    x = torch.randn(N, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(N, H, H, Cout, device="cuda").bfloat16()
    w = torch.zeros(Cout, Cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: hip_ops.conv_wgrad(dy, x, w, (1, 1), (0, 0)))
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    splits, pps = C.wgrad_choose_splits(N * H * H, Cout, Cin, hip_ops._dt(dy))
    print("%2dx%-2d x[%4d] dy[%4d]        %8.3f %8.2f %8.2f   splits %d" % (H, H, Cin, Cout, t * 1e3, gb, gb / t / 1e3, splits), flush=True)

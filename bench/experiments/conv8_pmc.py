"""A few launches of the conv8 forward (+ BN statistics) and data gradient on the two deep 3x3 layers of ResNet-50 at
batch 1024 (rocprofv3 counter pass: bench/experiments/pmc_conv8.sh)."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd.ops import hip_ops

for name, N, HW, Cc in [("l3 conv2 14x14 256->256", 1024, 14, 256), ("l4 conv2 7x7 512->512", 1024, 7, 512)]:
    x = torch.randn(N, HW, HW, Cc, device="cuda").bfloat16()
    w = (torch.randn(Cc, Cc, 3, 3, device="cuda") / (Cc * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, HW, HW, Cc, device="cuda").bfloat16()
    for _ in range(4):
        hip_ops.conv2d(x, w, None, (1, 1), (1, 1), False, None, True, False)
        hip_ops.conv_dgrad(dy, w, tuple(x.shape), (1, 1), (1, 1))
torch.cuda.synchronize()

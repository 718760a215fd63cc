"""fp32 convolutions of ResNet-50 at batch 256 (BASELINE config #2) for a counter pass: forward + statistics, data
gradient and weight gradient of layer3's 3x3 and 1x1 expansions (bench/experiments/pmc_mfma.sh fp32_pmc.py ...)."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd.ops import hip_ops

for HW, Ci, Co, k in [(14, 256, 256, 3), (14, 256, 1024, 1), (28, 128, 128, 3)]:
    x = torch.randn(256, HW, HW, Ci, device="cuda")
    w = (torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(256, HW, HW, Co, device="cuda")
    for _ in range(3):
        hip_ops.conv2d(x, w, None, (1, 1), (k // 2, k // 2), False, None, True, False)
        hip_ops.conv_dgrad(dy, w, tuple(x.shape), (1, 1), (k // 2, k // 2))
        hip_ops.conv_wgrad(dy, x, w, (1, 1), (k // 2, k // 2))
torch.cuda.synchronize()

"""A few launches of wgrad8 (dense 1x1 with the Gram operand, and the gather variant on a 3x3 layer) and of wgrad.hip on
the same problems, for the rocprofv3 counter pass (bench/experiments/pmc_mfma.sh wgrad8_pmc.py pmc_wgrad8 wgrad '...')."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd.ops import hip_ops
C = hip_ops.C()
NB = 1024
x = torch.randn(NB * 196, 256, device="cuda").bfloat16()
dy = torch.randn(NB * 196, 1024, device="cuda").bfloat16()
x3 = torch.randn(NB, 14, 14, 256, device="cuda").bfloat16()
dy3 = torch.randn(NB, 14, 14, 256, device="cuda").bfloat16()
w3 = torch.zeros(256, 256, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
for mode in (-1, 0):
    C.wgrad8_set_mode(mode)
    for _ in range(4):
        hip_ops._wgrad_raw(dy, x, NB * 196, 1024, 256, 1, NB * 196, gram=True)
        hip_ops.conv_wgrad(dy3, x3, w3, (1, 1), (1, 1))
C.wgrad8_set_mode(-1)
torch.cuda.synchronize()

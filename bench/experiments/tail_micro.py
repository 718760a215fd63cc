"""Time one RegNet-Y 8GF grouped 3x3 (56 channels per group) forward + data gradient under the tail-kernel settings."""
import sys
sys.path.insert(0, ".")
import torch
from distributed_pytorch_amd.ops import hip_ops, functional as F

C = hip_ops.C()
for (NB, H, ch, G) in ((128, 28, 448, 8), (128, 14, 896, 16), (256, 28, 240, 2), (256, 56, 72, 3)):
    x = torch.randn(NB, H, H, ch, device="cuda").bfloat16().requires_grad_(True)
    w = (torch.randn(ch, ch // G, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(NB, H, H, ch, device="cuda").bfloat16()
    outs = {}
    for mode in (0, -16, 16):
        C.igemm_set_tail_min(mode)
        for it in range(3):
            y = F.conv2d(x, w, None, 1, 1, groups=G)
            y.backward(gy)
        torch.cuda.synchronize()
        t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        n = 20
        tf = tb = 0.0
        for it in range(n):
            x.grad = None
            t0.record()
            y = F.conv2d(x, w, None, 1, 1, groups=G)
            t1.record()
            y.backward(gy)
            t2.record()
            torch.cuda.synchronize()
            tf += t0.elapsed_time(t1)
            tb += t1.elapsed_time(t2)
        outs[mode] = (y.detach().float(), x.grad.float())
        print("N%d H%d C%d G%d mode %4d: fwd %.1f us  bwd (dgrad+wgrad) %.1f us" % (NB, H, ch, G, mode, tf / n * 1e3, tb / n * 1e3), flush=True)
    for mode in (-16, 16):
        print("   max |y - y0| %.3g  max |dx - dx0| %.3g" % ((outs[mode][0] - outs[0][0]).abs().max().item(),
                                                            (outs[mode][1] - outs[0][1]).abs().max().item()))

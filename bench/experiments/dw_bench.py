"""Depthwise convolution kernels on MobileNetV2 / EfficientNet-B0 / ConvNeXt-T shapes (bf16): time and bytes moved."""
import torch

from distributed_pytorch_amd import backend

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
# (N, H, W, C, K, stride)
SHAPES = [(256, 112, 112, 32, 3, 1), (256, 112, 112, 96, 3, 2), (256, 56, 56, 144, 3, 1), (256, 56, 56, 144, 3, 2),
          (256, 28, 28, 192, 3, 1), (256, 14, 14, 384, 3, 1), (256, 14, 14, 576, 3, 1), (256, 7, 7, 960, 3, 1),
          (256, 56, 56, 144, 5, 2), (256, 28, 28, 240, 5, 1), (256, 14, 14, 672, 5, 1), (256, 7, 7, 1152, 5, 1),
          (128, 56, 56, 96, 7, 1), (128, 28, 28, 192, 7, 1), (128, 14, 14, 384, 7, 1)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


tf = tb = tw = 0.0
for N, H, W, C, K, st in SHAPES:
    p = K // 2
    OH, OW = (H + 2 * p - K) // st + 1, (W + 2 * p - K) // st + 1
    x = torch.randn(N, H, W, C, device=dev, dtype=torch.bfloat16)
    w = torch.randn(C, K, K, device=dev)
    y = torch.empty(N, OH, OW, C, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(N, OH, OW, C, device=dev, dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    rows = ext.dw_conv_rows(N * OH * OW, C, 1)
    stats = torch.empty(rows * 2 * C + rows, device=dev)
    wrows = ext.dw_wgrad_rows(N * OH * OW, C, K, 1)
    partial = torch.empty(wrows * C * K * K, device=dev)
    f = timed(lambda: ext.dw_conv(x.data_ptr(), w.data_ptr(), 0, 0, y.data_ptr(), stats.data_ptr(), 0, N, H, W, C, OH, OW,
                                  K, K, st, st, p, p, 0, 1, s))
    b = timed(lambda: ext.dw_conv(dy.data_ptr(), w.data_ptr(), 0, 0, dx.data_ptr(), 0, 1, N, H, W, C, OH, OW, K, K, st, st,
                                  p, p, 0, 1, s))
    g = timed(lambda: ext.dw_wgrad(dy.data_ptr(), x.data_ptr(), partial.data_ptr(), N, H, W, C, OH, OW, K, K, st, st, p, p,
                                   1, s))
    # check forward against torch
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).float(), w.to(torch.bfloat16).float()[:, None], stride=st,
                                     padding=p, groups=C).permute(0, 2, 3, 1)
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    del ref
    # backward-data and weight gradient against autograd (fp32 math on the rounded operands)
    xr = x.permute(0, 3, 1, 2).float().requires_grad_(True)
    wr = w.to(torch.bfloat16).float()[:, None].requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, stride=st, padding=p, groups=C).backward(dy.permute(0, 3, 1, 2).float())
    dxr = xr.grad.permute(0, 2, 3, 1)
    errb = (dx.float() - dxr).abs().max().item() / dxr.abs().max().item()
    dwv = partial.view(wrows, C, K * K).sum(0)
    dwr = wr.grad.view(C, K * K)
    errw = (dwv - dwr).abs().max().item() / dwr.abs().max().item()
    del xr, wr, dxr
    gb = (x.numel() + y.numel()) * 2 / 1e9
    tf += f
    tb += b
    tw += g
    print("N%d %3dx%-3d C%-4d k%d s%d  fwd %7.1f us %.2f TB/s  bwd %7.1f us %.2f TB/s  wgrad %7.1f us %.2f TB/s  rel err %.2g %.2g %.2g"
          % (N, H, W, C, K, st, f, gb / f * 1e3, b, gb / b * 1e3, g, gb / g * 1e3, err, errb, errw), flush=True)
print("total fwd %.1f  bwd %.1f  wgrad %.1f us" % (tf, tb, tw))

"""LayerNorm forward / backward on the zoo's row shapes (bf16): time and effective bandwidth."""
import torch

from distributed_pytorch_amd import backend

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
# ViT-B/16 b128, ViT-L/16 b64, ConvNeXt-T b128 stages, Swin-T b128 stages, ViT-H/14 b32
SHAPES = [(25216, 768), (12608, 1024), (401408, 96), (100352, 192), (25088, 384), (6272, 768), (8224, 1280)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


tf = tb = 0.0
for rows, C in SHAPES:
    x = torch.randn(rows, C, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(rows, C, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    w = torch.randn(C, device=dev)
    b = torch.randn(C, device=dev)
    mean = torch.empty(rows, device=dev)
    rstd = torch.empty(rows, device=dev)
    nparts = ext.ln_partial_rows(rows, C, 1)
    partial = torch.empty(nparts * 2 * C, device=dev)
    dg = torch.empty(C, device=dev)
    db = torch.empty(C, device=dev)
    f = timed(lambda: ext.ln_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 rows, C, 1e-6, 1, s))
    bw = timed(lambda: ext.ln_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                  dx.data_ptr(), partial.data_ptr(), dg.data_ptr(), db.data_ptr(), 0.0, 0.0, rows, C, 1, s))
    # reference check
    ref = torch.nn.functional.layer_norm(x.float(), (C,), w, b, 1e-6)
    err = (y.float() - ref).abs().max().item()
    gb = rows * C * 2 / 1e9
    tf += f
    tb += bw
    print("rows=%7d C=%5d parts=%5d  fwd %7.1f us %.2f TB/s   bwd(+fold) %7.1f us %.2f TB/s   max err %.3g"
          % (rows, C, nparts, f, 2 * gb / f * 1e3, bw, 3 * gb / bw * 1e3, err), flush=True)
print("total fwd %.1f us  bwd %.1f us" % (tf, tb))

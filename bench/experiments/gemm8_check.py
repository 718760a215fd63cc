"""gemm8 (256x256 multi-phase GEMM) against torch.matmul: correctness on ragged shapes, repeated-run race screen,
throughput against the lean GEMM (F.linear) and hipBLASLt (torch.matmul)."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd import backend
from distributed_pytorch_amd.ops import functional as F

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream


def run(A, B, bias=None, relu=0):
    M, K = A.shape
    N = B.shape[0]
    C = torch.empty(M, N, device=dev, dtype=A.dtype)
    ext.gemm8(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if bias is not None else 0, M, N, K, A.stride(0),
              B.stride(0), C.stride(0), relu, 1, s)
    return C


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


torch.manual_seed(0)
bad = 0
for M, N, K in [(256, 256, 64), (256, 256, 128), (256, 256, 512), (300, 264, 192), (1000, 520, 320), (4096, 4096, 4096), (8192, 8192, 8192), (8192, 4096, 2048),
                (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (200704, 1024, 256), (50176, 2048, 512)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ref = (A.float() @ B.float().t() + bias).relu()
    worst = 0.0
    for rep in range(5):                           # race screen: the same launch several times
        C = run(A, B, bias, 1)
        err = float((C.float() - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
        worst = max(worst, err)
    ok = worst < 1e-2
    bad += not ok
    us = timed(lambda: run(A, B))
    us_t = timed(lambda: torch.matmul(A, B.t()))
    with torch.no_grad():
        us_l = timed(lambda: F.linear(A, B, None))
    tf = 2.0 * M * N * K / us / 1e6
    print("M=%6d N=%5d K=%5d  rel err %.2e %s   gemm8 %8.1f us %7.1f TFLOP/s   torch.matmul %8.1f us %7.1f   lean %8.1f us %7.1f"
          % (M, N, K, worst, "ok" if ok else "WRONG", us, tf, us_t, 2.0 * M * N * K / us_t / 1e6, us_l,
             2.0 * M * N * K / us_l / 1e6), flush=True)
print("FAILED" if bad else "all ok")

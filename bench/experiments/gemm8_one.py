"""A few launches of gemm8 and of the lean GEMM at 4096^3 / 8192 x 4096 x 2048 (rocprofv3 counter passes)."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd import backend
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.ops import hip_ops

ext = backend.require()
s = torch.cuda.current_stream().cuda_stream
hip_ops._GEMM8 = 0
for M, N, K in [(4096, 4096, 4096), (8192, 4096, 2048)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        ext.gemm8(A.data_ptr(), B.data_ptr(), C.data_ptr(), 0, M, N, K, K, K, N, 0, 1, s)
        with torch.no_grad():
            F.linear(A, B, None)
torch.cuda.synchronize()

"""The 256x256 multi-phase GEMM (csrc/kernels/gemm8.hip) against fp32 matmul: ragged tile edges, short and long K
loops (1 .. 40 K tiles: pipeline prologue / tail paths), bias + ReLU, both 16-bit types, repeated launches (the
kernel keeps LDS-DMA requests in flight across raw barriers: a mis-placed wait shows as a rare wrong tile), and the
Linear layer routed through it."""
import pytest
import torch

from _gpu_util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"

SHAPES = [(256, 256, 64), (256, 256, 128), (256, 512, 192), (300, 264, 320), (1000, 520, 256), (513, 1032, 2560),
          (2048, 768, 768), (77, 8, 64)]


def _gemm8(A, B, bias=None, relu=False):
    from distributed_pytorch_amd.ops import hip_ops

    M, K = A.shape
    N = B.shape[0]
    C = torch.full((M, N), float("nan"), device=DEV, dtype=A.dtype)
    hip_ops.C().gemm8(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if bias is not None else 0, M, N, K,
                      A.stride(0), B.stride(0), C.stride(0), int(relu), hip_ops._dt(A.dtype),
                      torch.cuda.current_stream().cuda_stream)
    return C


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("epi", [False, True])
def test_gemm8_matches_fp32_matmul(shape, dtype, epi):
    M, N, K = shape
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=DEV).to(dtype)
    B = (torch.randn(N, K, device=DEV) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device=DEV) if epi else None
    ref = A.float() @ B.float().t()
    if epi:
        ref = (ref + bias).relu()
    for _ in range(4):
        C = _gemm8(A, B, bias, epi)
        assert_close(C, ref, dtype, what="gemm8 %s" % (shape,))


def test_gemm8_strided_rows_and_padded_output():
    """Leading dimensions larger than the row length (views into wider buffers), output rows with a pad column block."""
    torch.manual_seed(3)
    M, N, K = 520, 264, 192
    Abuf = torch.randn(M, K + 64, device=DEV).to(torch.bfloat16)
    Bbuf = (torch.randn(N, K + 8, device=DEV) / K ** 0.5).to(torch.bfloat16)
    A, B = Abuf[:, :K], Bbuf[:, :K]
    from distributed_pytorch_amd.ops import hip_ops

    Cbuf = torch.zeros(M, N + 24, device=DEV, dtype=torch.bfloat16)
    hip_ops.C().gemm8(A.data_ptr(), B.data_ptr(), Cbuf.data_ptr(), 0, M, N, K, A.stride(0), B.stride(0), Cbuf.stride(0), 0,
                      1, torch.cuda.current_stream().cuda_stream)
    assert_close(Cbuf[:, :N], A.float() @ B.float().t(), torch.bfloat16, what="strided gemm8")
    assert float(Cbuf[:, N:].abs().max()) == 0.0            # nothing written past N


def test_linear_through_gemm8_matches_reference():
    """F.linear forward + input gradient forced onto the multi-phase kernel (DPA_GEMM8=2 semantics) against fp32."""
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.ops import hip_ops

    torch.manual_seed(4)
    x0 = torch.randn(3, 197, 384, device=DEV).to(torch.bfloat16)
    w = (torch.randn(1536, 384, device=DEV) / 384 ** 0.5).requires_grad_(True)
    b = torch.randn(1536, device=DEV).requires_grad_(True)
    gy = torch.randn(3, 197, 1536, device=DEV).to(torch.bfloat16)
    old = hip_ops._GEMM8
    hip_ops._GEMM8 = 2
    try:
        x = x0.clone().requires_grad_(True)
        y = F.linear(x, w, b)
        y.backward(gy)
    finally:
        hip_ops._GEMM8 = old
    xr = x0.float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(gy.float())
    assert_close(y, yr, torch.bfloat16, what="linear fwd")
    assert_close(x.grad, xr.grad, torch.bfloat16, what="linear dgrad")
    assert_close(w.grad, wr.grad, torch.bfloat16, what="linear wgrad")
    assert_close(b.grad, br.grad, torch.bfloat16, scale=4, what="linear bias grad")

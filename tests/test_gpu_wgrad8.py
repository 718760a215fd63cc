"""Dense 1x1 weight gradient on the 256 x 256 LDS-DMA pipeline (csrc/kernels/wgrad8.hip) against the fp32 PyTorch
reference and against wgrad.hip: ragged pixel counts (zero page past the split), partial channel tiles, padded operand
rows, the Gram-matrix operand of the BN fold, run-to-run bit identity."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import hip_ops  # noqa: E402
from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"

# M (pixels), N (dy channels), C (x channels), ldy pad, ldx pad, gram
CASES = [
    (4096, 256, 256, 0, 0, False), (3001, 256, 256, 0, 0, False), (70, 256, 256, 0, 0, False),
    (8192, 264, 320, 0, 0, False), (5000, 1024, 256, 0, 0, True), (6100, 512, 2048, 0, 0, False),
    (4160, 256, 512, 64, 128, True), (2500, 72, 40, 8, 24, True), (20000, 512, 512, 0, 0, True),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_dense_weight_gradient_on_the_dma_pipeline(case, dtype):
    M, N, Cc, pady, padx, gram = case
    C = hip_ops.C()
    torch.manual_seed(3)
    dyf = torch.randn(M, N + pady, device=DEV).to(dtype)
    xf = torch.randn(M, Cc + padx, device=DEV).to(dtype)
    try:
        C.wgrad8_set_mode(1)
        splits, pps = C.wgrad8_splits(M, N, Cc if gram else 0, Cc, hip_ops._dt(dyf))
        assert splits > 0 and (splits == 1 or pps % 64 == 0)
        got = hip_ops._wgrad_raw(dyf, xf, M, N, Cc, 1, M, gram=gram, ldy=N + pady, ldx=Cc + padx)
        again = hip_ops._wgrad_raw(dyf, xf, M, N, Cc, 1, M, gram=gram, ldy=N + pady, ldx=Cc + padx)
        C.wgrad8_set_mode(0)
        assert C.wgrad8_splits(M, N, Cc if gram else 0, Cc, hip_ops._dt(dyf))[0] == 0
        old = hip_ops._wgrad_raw(dyf, xf, M, N, Cc, 1, M, gram=gram, ldy=N + pady, ldx=Cc + padx)
    finally:
        C.wgrad8_set_mode(-1)
    dy32, x32 = dyf[:, :N].float(), xf[:, :Cc].float()
    ref = dy32.t() @ x32
    if gram:
        ref = torch.cat([ref, x32.t() @ x32], 0)
    assert got.shape == ref.shape
    assert_close(got, ref, torch.float32, scale=30, what="wgrad8 vs fp32 reference")
    assert_close(got, old, torch.float32, scale=30, what="wgrad8 vs wgrad.hip")
    assert torch.equal(got, again)


def test_convolution_weight_gradient_takes_the_dma_pipeline_on_deep_layers():
    C = hip_ops.C()
    x = torch.randn(64, 14, 14, 256, device=DEV).bfloat16()
    dy = torch.randn(64, 14, 14, 1024, device=DEV).bfloat16()
    w = torch.zeros(1024, 256, 1, 1, device=DEV).contiguous(memory_format=torch.channels_last)
    assert C.wgrad8_splits(64 * 196, 1024, 0, 256, hip_ops._dt(dy))[0] > 0      # the automatic rule
    got = hip_ops.conv_wgrad(dy, x, w, (1, 1), (0, 0))
    ref = dy.float().reshape(-1, 1024).t() @ x.float().reshape(-1, 256)
    assert_close(got.reshape(1024, 256), ref, torch.float32, scale=30, what="1x1 conv wgrad through wgrad8")
    # shallow / narrow problems stay on wgrad.hip
    assert C.wgrad8_splits(1024 * 3136, 64, 0, 256, hip_ops._dt(dy))[0] == 0
    assert C.wgrad8_splits(2000, 1024, 0, 1024, hip_ops._dt(dy))[0] == 0
    assert C.wgrad8_splits(1 << 20, 512, 0, 512, 0)[0] == 0                    # fp32


# batch, IH, IW, Cin, Cout, R, S, stride, pad
CONV_CASES = [
    (8, 14, 14, 256, 256, 3, 3, 1, 1), (3, 28, 28, 256, 512, 3, 3, 2, 1), (20, 7, 7, 512, 512, 3, 3, 1, 1),
    (5, 28, 28, 512, 1024, 1, 1, 2, 0), (2, 12, 20, 128, 264, 3, 3, 1, 1), (3, 9, 9, 128, 256, 5, 5, 1, 2),
    (2, 30, 30, 384, 256, 3, 3, 1, 0), (7, 15, 15, 128, 320, 3, 1, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_gathered_taps_on_the_dma_pipeline(case, dtype):
    import torch.nn.functional as TF
    NB, IH, IW, Cin, Cout, R, S, stride, pad = case
    C = hip_ops.C()
    torch.manual_seed(4)
    OH, OW = (IH + 2 * pad - R) // stride + 1, (IW + 2 * pad - S) // stride + 1
    x = torch.randn(NB, IH, IW, Cin, device=DEV).to(dtype)
    dy = torch.randn(NB, OH, OW, Cout, device=DEV).to(dtype)
    w = torch.zeros(Cout, Cin, R, S, device=DEV).contiguous(memory_format=torch.channels_last)
    try:
        C.wgrad8_set_mode(1)
        C.wgrad_row3_set_mode(0)
        assert C.wgrad8_splits(NB * OH * OW, Cout, 0, Cin, hip_ops._dt(dy), R * S, True)[0] > 0
        got = hip_ops.conv_wgrad(dy, x, w, (stride, stride), (pad, pad))
        again = hip_ops.conv_wgrad(dy, x, w, (stride, stride), (pad, pad))
        C.wgrad8_set_mode(0)
        old = hip_ops.conv_wgrad(dy, x, w, (stride, stride), (pad, pad))
    finally:
        C.wgrad8_set_mode(-1)
        C.wgrad_row3_set_mode(-1)
    wr = torch.zeros(Cout, Cin, R, S, device=DEV, requires_grad=True)
    TF.conv2d(x.float().permute(0, 3, 1, 2), wr, None, stride, pad).backward(dy.float().permute(0, 3, 1, 2))
    assert_close(got, wr.grad, torch.float32, scale=30, what="wgrad8 gather vs fp32 reference")
    assert_close(got, old, torch.float32, scale=30, what="wgrad8 gather vs wgrad.hip")
    assert torch.equal(got, again)
    # channel counts that are not whole 128-column half-tiles stay on wgrad.hip
    assert C.wgrad8_splits(1 << 16, 256, 0, 192, hip_ops._dt(dy), 9, True)[0] == 0

"""Filter-row weight gradient (csrc/kernels/wgrad_row3.hip) against the fp32 PyTorch reference and against wgrad.hip:
every segment geometry (W = 7: two rows per segment; 8..16, 17..32, 33..64; wide images as 2 / 4 chunks per row), every tile width, filter heights 1 / 3 / 5,
ragged row splits."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import hip_ops  # noqa: E402
from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"

# N (batch), H, W, Cin, Cout, R
CASES = [
    (9, 7, 7, 64, 64, 3), (5, 14, 14, 128, 256, 3), (3, 28, 28, 64, 128, 3), (2, 56, 56, 64, 64, 3),
    (4, 9, 11, 64, 320, 3), (2, 20, 20, 192, 72, 3), (2, 12, 33, 64, 128, 3), (6, 8, 8, 128, 128, 1),
    (3, 17, 17, 64, 64, 5), (4, 13, 64, 64, 256, 3),
    # wide images: 2 / 4 virtual rows of 33..64 pixels per image row (VGG conv1_2 / conv2_x geometry)
    (2, 10, 112, 64, 64, 3), (1, 6, 224, 64, 64, 3), (2, 5, 80, 64, 128, 3), (1, 4, 136, 128, 64, 3), (2, 7, 96, 64, 64, 5),
]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_filter_row_weight_gradient(case, dtype, variant):
    N, H, W, Cin, Cout, R = case
    C = hip_ops.C()
    torch.manual_seed(2)
    x = torch.randn(N, H, W, Cin, device=DEV).to(dtype)
    dy = torch.randn(N, H, W, Cout, device=DEV).to(dtype)
    w = torch.zeros(Cout, Cin, R, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    ph = R // 2
    try:
        C.wgrad_row3_set_mode(1)                   # every covered shape, not only those the automatic rule picks
        C.wgrad_row3_set_variant(variant)
        splits, rps = C.wgrad_row3_splits(N, H, W, Cout, Cin, R, hip_ops._dt(dy))
        assert splits > 0, "case must be covered by the filter-row kernel"
        got = hip_ops.conv_wgrad(dy, x, w, (1, 1), (ph, 1))
        again = hip_ops.conv_wgrad(dy, x, w, (1, 1), (ph, 1))
        C.wgrad_row3_set_mode(0)
        old = hip_ops.conv_wgrad(dy, x, w, (1, 1), (ph, 1))
    finally:
        C.wgrad_row3_set_mode(-1)
        C.wgrad_row3_set_variant(0)
    xr = x.float().permute(0, 3, 1, 2)
    wr = torch.zeros(Cout, Cin, R, 3, device=DEV, requires_grad=True)
    TF.conv2d(xr, wr, None, 1, (ph, 1)).backward(dy.float().permute(0, 3, 1, 2))
    assert_close(got, wr.grad, torch.float32, scale=30, what="filter-row wgrad vs fp32 reference")     # fp32 sums of 16-bit products
    assert_close(got, old, torch.float32, scale=30, what="filter-row wgrad vs wgrad.hip")
    assert torch.equal(got, again)                 # fixed-order slab fold

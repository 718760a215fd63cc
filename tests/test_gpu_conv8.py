"""conv8.hip (convolutions on the 256 x 256 multi-phase pipeline): every epilogue and both schedules against the fp32
PyTorch reference and against the stand-alone reduction kernels, on shapes with row / channel tails, padding on every
side and more taps than one.  `conv8_launches()` proves the launch really took the kernel under test."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import hip_ops  # noqa: E402
from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"

# N, H, W, Cin, Cout, k
SHAPES = [
    (5, 14, 14, 256, 256, 1),      # M = 980: a 128-row part with 84 rows, an empty last part
    (5, 14, 14, 256, 256, 3),
    (3, 9, 11, 128, 320, 3),       # N tail (320 = 256 + 64), odd extents
    (2, 16, 16, 64, 64, 3),        # one K tile per tap, N < 256
    (1, 24, 24, 512, 512, 1),      # M = 576
    (2, 12, 20, 192, 128, 3),      # W != H, three K tiles per tap
]
MODES = [1, 2]                     # data-parallel, stream-K


class _Mode:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.C = hip_ops.C()
        self.C.conv8_set_mode(self.mode)
        self.n0 = self.C.conv8_launches()
        return self

    def launches(self):
        return self.C.conv8_launches() - self.n0

    def __exit__(self, *a):
        self.C.conv8_set_mode(-1)
        torch.cuda.synchronize()
        assert self.C.conv8_error(torch.cuda.current_stream().cuda_stream) == 0


def _weights(Cout, Cin, k):
    return (torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv8_forward_data_gradient_and_statistics(shape, mode, dtype):
    N, H, W, Cin, Cout, k = shape
    torch.manual_seed(7)
    p = k // 2
    x = (torch.randn(N, H, W, Cin, device=DEV) + 0.25).to(dtype)
    w = _weights(Cout, Cin, k)
    gy = torch.randn(N, H, W, Cout, device=DEV).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.to(dtype).float()
    yr = TF.conv2d(xr, wr, None, 1, p)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    with _Mode(mode) as m:
        y, stats = hip_ops.conv2d(x, w, None, (1, 1), (p, p), False, None, True, False)
        dx = hip_ops.conv_dgrad(gy, w, tuple(x.shape), (1, 1), (p, p))
        assert m.launches() == 2
    assert_close(y.permute(0, 3, 1, 2), yr, dtype, what="fprop")
    assert_close(dx.permute(0, 3, 1, 2), xr.grad, dtype, what="dgrad")
    # Chan partials of 128-row parts + the parts' row counts behind the slab
    yf = y.float().reshape(-1, Cout)
    parts = stats.shape[0]
    assert parts == (yf.shape[0] + 127) // 128
    counts = stats.flatten().as_strided((parts,), (1,), stats.numel())
    assert counts.sum().item() == yf.shape[0]
    assert_close(stats[:, 0].sum(0), yf.sum(0), torch.float32, scale=10, what="sum")
    for t in (0, parts - 1):
        rows = yf[t * 128:(t + 1) * 128]
        assert counts[t].item() == rows.shape[0]
        assert_close(stats[t, 0], rows.sum(0), torch.float32, scale=10, what="sum of part %d" % t)
        assert_close(stats[t, 1], (rows - rows.mean(0)).square().sum(0), torch.float32, scale=10, what="M2 of part %d" % t)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", SHAPES[:4])
def test_conv8_bias_relu_epilogue(shape, mode):
    N, H, W, Cin, Cout, k = shape
    dtype = torch.bfloat16
    torch.manual_seed(8)
    p = k // 2
    x = torch.randn(N, H, W, Cin, device=DEV).to(dtype)
    w = _weights(Cout, Cin, k)
    b = torch.randn(Cout, device=DEV)
    with _Mode(mode) as m:
        y = hip_ops.conv2d(x, w, b, (1, 1), (p, p), True, None, False, False)
        assert m.launches() == 1
    yr = TF.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), b, 1, p)
    mask = y.permute(0, 3, 1, 2) > 0
    assert (mask != (yr > 0)).float().mean().item() < 1e-3
    assert_close(y.permute(0, 3, 1, 2), yr * mask, dtype, what="bias + relu")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("bn_mode", [0, 2, 3, 6])
@pytest.mark.parametrize("shape", SHAPES[:3])
def test_conv8_data_gradient_with_bn_backward_partials(shape, mode, bn_mode):
    """kIgemmBnBwd on conv8: same stored gradient as the plain launch, partial sums equal to the stand-alone reducer run
    on that stored gradient (same masks; fp32 sums in another order)."""
    N, H, W, Cin, Cout, k = shape
    C = hip_ops.C()
    dt = torch.bfloat16
    torch.manual_seed(9)
    p = k // 2
    w = _weights(Cout, Cin, k)
    dy = torch.randn(N, H, W, Cout, device=DEV).to(dt)
    bn_x = (torch.randn(N, H, W, Cin, device=DEV) * 1.5 + 0.3).to(dt)
    coef = torch.stack([torch.randn(Cin, device=DEV) * 0.1, torch.rand(Cin, device=DEV) + 0.5,
                        torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.2]).contiguous()
    z = None
    if bn_mode == 3:
        zz = torch.randn(N, H, W, Cin, device=DEV)
        bits = (zz.view(-1, 8) > 0).to(torch.uint8) << torch.arange(8, device=DEV, dtype=torch.uint8)
        z = bits.sum(1).to(torch.uint8).contiguous()
    hip_ops._pending_bnred.clear()
    with _Mode(mode) as m:
        dx = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (1, 1), (p, p), bnred=(bn_x, z, coef, bn_mode))
        ref = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (1, 1), (p, p))
        assert m.launches() == 2
    slab, xptr = hip_ops._pending_bnred.pop(dx.data_ptr())[:2]
    assert xptr == bn_x.data_ptr()
    assert torch.equal(dx, ref)
    M = N * H * W
    nblk = C.bn_partial_rows(M, Cin)
    part = torch.empty(nblk, 2, Cin, device=DEV)
    C.bn_bwd_reduce(dx.data_ptr(), bn_x.data_ptr(), z.data_ptr() if z is not None else 0, M, Cin, hip_ops._dt(dx),
                    coef[0].data_ptr(), coef[1].data_ptr(), bn_mode, coef[2].data_ptr(), coef[3].data_ptr(),
                    part.data_ptr(), nblk, torch.cuda.current_stream().cuda_stream)
    got, want = slab.sum(0), part.sum(0)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-4 * scale, (got - want).abs().max().item() / scale


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [SHAPES[1], SHAPES[2]])
def test_conv8_masked_data_gradient(shape, mode):
    """kIgemmXMask: the 3x3 data gradient stored through the ReLU mask of the tensor it belongs to."""
    N, H, W, Cin, Cout, k = shape
    dt = torch.bfloat16
    torch.manual_seed(10)
    w = _weights(Cout, Cin, k)
    dy = torch.randn(N, H, W, Cout, device=DEV).to(dt)
    xm = torch.relu(torch.randn(N, H, W, Cin, device=DEV)).to(dt)
    with _Mode(mode) as m:
        dx = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (1, 1), (1, 1), xmask=xm)
        ref = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (1, 1), (1, 1))
        assert m.launches() == 2
    assert torch.equal(dx, ref * (xm > 0))


def test_conv8_stream_k_matches_data_parallel_bit_for_bit_on_whole_tiles_and_is_replayable():
    """Segments that end inside a tile go through the fp32 slab: the sum is the same numbers in another association, so
    whole tiles are bit-identical and split tiles agree to fp32 rounding; flags are reset by their consumers, so the
    same launch can run again (graph replay) without a memset."""
    torch.manual_seed(11)
    dt = torch.bfloat16
    x = torch.randn(16, 28, 28, 256, device=DEV).to(dt)       # M = 12544 = 49 tiles, 36 K tiles each
    w = _weights(256, 256, 3)
    C = hip_ops.C()
    try:
        C.conv8_set_mode(1)
        y1 = hip_ops.conv2d(x, w, None, (1, 1), (1, 1), False, None, False, False)
        C.conv8_set_mode(2)
        ys = [hip_ops.conv2d(x, w, None, (1, 1), (1, 1), False, None, False, False) for _ in range(3)]
    finally:
        C.conv8_set_mode(-1)
    torch.cuda.synchronize()
    assert C.conv8_error(torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert_close(ys[0], y1.float(), dt, what="stream-K vs data-parallel")      # (1-ulp flips where the fp32 sums differ)
    assert (ys[0] != y1).float().mean().item() < 1e-3

"""Production shapes: real ResNet-50 layers at batch 256 (M = 802,816 pixels: multi-split weight gradients at the
split counts the benchmark uses, outputs past the streaming-store threshold) and operands past 4 GiB.

The fp32 oracle of a whole layer at this size is itself a large computation, so it is evaluated on a random subset
of images (forward / data gradient are per image) and on ALL images for the weight gradient (an fp32 reference
conv on the bf16-rounded inputs, chunked over the batch to bound memory)."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import functional as F  # noqa: E402
from distributed_pytorch_amd.ops import hip_ops  # noqa: E402

from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"

LAYERS = [
    # name, N, H, W, Cin, Cout, k, stride, pad
    ("layer1 conv3 56x56 64->256", 256, 56, 56, 64, 256, 1, 1, 0),
    ("layer1 conv2 56x56 64->64 3x3", 256, 56, 56, 64, 64, 3, 1, 1),
    ("layer2.0 conv2 56x56 128->128 3x3 s2", 256, 56, 56, 128, 128, 3, 2, 1),
    ("layer2.0 downsample 56x56 256->512 s2", 256, 56, 56, 256, 512, 1, 2, 0),
]


def _ref_chunked(x, w, gy, s, p, dtype, idx):
    """fp32 conv on images `idx` (forward, dgrad) and the fp32 weight gradient over ALL images in chunks of 32."""
    wr = w.detach().to(dtype).float()
    xs = x.detach()[idx].float().permute(0, 3, 1, 2).requires_grad_(True)
    ys = TF.conv2d(xs, wr, None, s, p)
    ys.backward(gy[idx].float().permute(0, 3, 1, 2))
    dw = torch.zeros_like(wr)
    for i in range(0, x.shape[0], 32):
        xc = x.detach()[i:i + 32].float().permute(0, 3, 1, 2)
        gc = gy[i:i + 32].float().permute(0, 3, 1, 2)
        dw += torch.nn.grad.conv2d_weight(xc, wr.shape, gc, s, p)
    return ys.detach(), xs.grad, dw


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_resnet50_layer_at_batch_256(layer):
    _, N, H, W, Cin, Cout, k, s, p = layer
    dtype = torch.bfloat16
    torch.manual_seed(41)
    x = torch.randn(N, H, W, Cin, device=DEV).to(dtype).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5).contiguous(
        memory_format=torch.channels_last).requires_grad_(True)
    y, stats = F.conv2d(x, w, None, s, p, want_stats=True)
    gy = torch.randn(y.shape, device=DEV).to(dtype)
    y.backward(gy)
    M = y.numel() // Cout
    assert M >= 200704
    splits, _ = hip_ops.C().wgrad_choose_splits(M, Cout, k * k * Cin, hip_ops.C().BF16, 1)
    assert splits > 1, "production split count expected"
    idx = torch.tensor([0, 1, N // 2, N - 1], device=DEV)
    yr, dxr, dwr = _ref_chunked(x, w, gy, s, p, dtype, idx)
    assert_close(y.detach()[idx].permute(0, 3, 1, 2), yr, dtype, what="fprop")
    assert_close(x.grad[idx].permute(0, 3, 1, 2), dxr, dtype, what="dgrad")
    # the weight gradient sums M ~ 1e6 bf16 products in fp32, in a different order than the oracle
    assert_close(w.grad, dwr, torch.float32, what="wgrad", scale=20, frob_scale=50)
    yf = y.detach().float().reshape(-1, Cout)
    assert_close(stats[:, 0].double().sum(0).float(), yf.double().sum(0).float(), torch.float32, what="stats sum", scale=10)


def test_stem_space_to_depth_at_batch_256():
    """The fused stem path the benchmark takes: uint8 NHWC -> space-to-depth -> 4x4 convolution (= 7x7 stride 2)."""
    from distributed_pytorch_amd import amp, models
    torch.manual_seed(42)
    N = 256
    img = torch.randint(0, 256, (N, 224, 224, 3), dtype=torch.uint8, device=DEV)
    w = (torch.randn(64, 3, 7, 7, device=DEV) / 12).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xs = F.to_s2d(img, torch.bfloat16)
    y = F.conv2d(xs, w, None, 2, 3, needs_input_grad=False)
    gy = torch.randn(y.shape, device=DEV).to(torch.bfloat16)
    y.backward(gy)
    mean = torch.tensor([0.485, 0.456, 0.406], device=DEV) * 255
    std = torch.tensor([0.229, 0.224, 0.225], device=DEV) * 255
    idx = torch.tensor([0, 100, N - 1], device=DEV)
    wr = w.detach().to(torch.bfloat16).float()
    xn = ((img[idx].float() - mean) / std).to(torch.bfloat16).float().permute(0, 3, 1, 2)
    yr = TF.conv2d(xn, wr, None, 2, 3)
    assert_close(y.detach()[idx].permute(0, 3, 1, 2), yr, torch.bfloat16, what="stem fprop")
    dw = torch.zeros_like(wr)
    for i in range(0, N, 32):
        xc = ((img[i:i + 32].float() - mean) / std).to(torch.bfloat16).float().permute(0, 3, 1, 2)
        dw += torch.nn.grad.conv2d_weight(xc, wr.shape, gy[i:i + 32].float().permute(0, 3, 1, 2), 2, 3)
    assert_close(w.grad, dw, torch.float32, what="stem wgrad", scale=20, frob_scale=50)


def test_operands_past_4_gib_stay_exact():
    """An activation tensor of 4.3 GiB (56x56x256 bf16, 2,700 images -- a per-GPU batch the 288 GB of HBM invite).
    The aligned kernels address their operands with 32-bit offsets relative to a per-tile / per-split base, so the
    result must equal, bit for bit, the same layer run on image slices below 4 GiB (forward and data gradient are
    computed per output element in the same order); the weight gradient equals the sum of the slices' gradients."""
    dtype = torch.bfloat16
    N, H, W, Cin, Cout = 2700, 56, 56, 256, 64
    torch.manual_seed(43)
    x = torch.randn(N, H, W, Cin, device=DEV, dtype=dtype)
    assert x.numel() * 2 > 4 * 2 ** 30
    w = (torch.randn(Cout, Cin, 1, 1, device=DEV) / 16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, H, W, Cout, device=DEV, dtype=dtype)

    def run(xs, gs):
        xq = xs.clone().requires_grad_(True)
        wq = w.clone().requires_grad_(True)
        y = F.conv2d(xq, wq, None, 1, 0)
        y.backward(gs)
        return y.detach(), xq.grad, wq.grad.float()

    y, dx, dw = run(x, gy)
    cut = 1350
    dw_parts = torch.zeros_like(dw)
    for a, b in ((0, cut), (cut, N)):
        ys, dxs, dws = run(x[a:b], gy[a:b])
        assert torch.equal(y[a:b], ys), "forward differs past the 4 GiB boundary"
        assert torch.equal(dx[a:b], dxs), "data gradient differs past the 4 GiB boundary"
        dw_parts += dws
    assert_close(dw, dw_parts, torch.float32, what="wgrad", scale=20, frob_scale=50)
    # and the last image against the fp32 oracle (it lies entirely beyond 4 GiB)
    yr = TF.conv2d(x[-1:].float().permute(0, 3, 1, 2), w.to(dtype).float())
    assert_close(y[-1:].permute(0, 3, 1, 2), yr, dtype, what="last image")

"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op.

Low-precision runs feed the reference the SAME bf16/fp16-rounded inputs (upcast to fp32), so the
only differences are accumulation order and the final rounding of the output.
"""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import functional as F  # noqa: E402

DEV = "cuda"
from _gpu_util import assert_close  # noqa: E402


def _close(a, b, dtype, scale=1.0, what=""):
    """Per element |a - b| <= atol * rms(b) + rtol * |b| AND a relative-Frobenius bound (tests/_gpu_util.py)."""
    assert_close(a, b, dtype, what=what, scale=scale)


def _slab_m2(stats):
    """Total sum of squared deviations from a forward-statistics slab: rows (sum_t, M2_t) + the parts' row counts
    stored behind the slab, merged with Chan's formula in float64."""
    tiles = stats.shape[0]
    cnt = stats.flatten().as_strided((tiles,), (1,), stats.numel()).double()
    s, m2 = stats[:, 0].double(), stats[:, 1].double()
    part = torch.where(cnt[:, None] > 0, s * s / cnt[:, None].clamp(min=1.0), torch.zeros_like(s))   # empty parts: 0 rows
    return (m2.sum(0) + part.sum(0) - s.sum(0) ** 2 / cnt.sum()).float()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (4, 14, 14, 64, 64, 1, 1, 0),
    (4, 14, 14, 64, 128, 3, 1, 1),
    (4, 14, 14, 128, 128, 3, 2, 1),
    (4, 14, 14, 64, 256, 1, 2, 0),
    (3, 15, 13, 32, 64, 3, 2, 1),       # odd extents, Cin % 64 != 0
    (2, 9, 9, 96, 72, 3, 1, 1),         # non-aligned K tiles, Cout tail
    (2, 32, 32, 3, 64, 7, 2, 3),        # stem (channel padded)
    (2, 13, 13, 64, 192, 5, 1, 2),      # AlexNet conv2
    (2, 67, 67, 3, 64, 11, 4, 2),       # AlexNet conv1
    (2, 7, 7, 512, 512, 3, 1, 1),
    (5, 7, 7, 256, 1024, 1, 1, 0),
    (2, 17, 17, 128, 192, (1, 7), 1, (0, 3)),     # Inception v3 factorised filters
    (2, 17, 17, 160, 160, (7, 1), 1, (3, 0)),
    (2, 8, 8, 384, 384, (1, 3), 1, (0, 1)),       # one filter row (filter-row kernel)
    (2, 8, 8, 448, 384, (3, 1), 1, (1, 0)),
    (2, 17, 15, 64, 96, (3, 5), (2, 1), (1, 2)),  # everything different per axis
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("bias_relu", [False, True])
def test_conv_fwd_bwd(case, dtype, bias_relu):
    N, H, W, Cin, Cout, k, s, p = case
    if bias_relu and k not in (3, 5, 11, (1, 3), (3, 5)):
        pytest.skip("bias/relu epilogue covered on a subset")
    kh, kw = (k, k) if isinstance(k, int) else k
    torch.manual_seed(0)
    vec = 4 if dtype == torch.float32 else 8
    Cx = Cin if Cin % vec == 0 else 8
    x = torch.randn(N, H, W, Cin, device=DEV)
    xp = torch.zeros(N, H, W, Cx, device=DEV)
    xp[..., :Cin] = x
    xq = xp.to(dtype).requires_grad_(Cx == Cin)
    w = (torch.randn(Cout, Cin, kh, kw, device=DEV) / (Cin * kh * kw) ** 0.5).contiguous(
        memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(Cout, device=DEV).requires_grad_(True) if bias_relu else None
    y = F.conv2d(xq, w, b, s, p, relu=bias_relu, needs_input_grad=(Cx == Cin))
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)

    xr = xq.detach().float()[..., :Cin].permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True) if b is not None else None
    yr = TF.conv2d(xr, wr, br, s, p)
    if bias_relu:
        # use the kernel's own ReLU mask: pre-activations within rounding of 0 may land on either
        # side, and one flipped element moves a weight-gradient entry by |dy * x| ~ O(1)
        mask = (y.detach().permute(0, 3, 1, 2) > 0)
        flips = (mask != (yr.detach() > 0)).float().mean().item()
        assert flips < 1e-3, "relu mask mismatch fraction %g" % flips
        yr = yr * mask
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype, what="fprop")
    if Cx == Cin:
        _close(xq.grad.permute(0, 3, 1, 2), xr.grad, dtype, scale=1.0, what="dgrad")
    # wgrad sums over N*OH*OW products
    _close(w.grad, wr.grad, dtype, scale=1.0, what="wgrad")
    if b is not None:
        _close(b.grad, br.grad, dtype, scale=1.0, what="bias grad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_every_tile_config(dtype):
    """Every tile configuration of the tuning tables (igemm_force_config / wgrad_force_config) is exact."""
    from distributed_pytorch_amd.ops import hip_ops
    C = hip_ops.C()
    torch.manual_seed(3)
    N, H, Cin, Cout, k, s, p = 3, 13, 128, 192, 3, 1, 1
    x = torch.randn(N, H, H, Cin, device=DEV).to(dtype)
    w = (torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, H, H, Cout, device=DEV).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.to(dtype).float().requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, s, p)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    try:
        for cfg in range(C.igemm_num_configs()):
            C.igemm_force_config(cfg)
            y, stats = hip_ops.conv2d(x, w, None, (s, s), (p, p), False, None, True, False)
            _close(y.permute(0, 3, 1, 2), yr, dtype, what="fprop cfg %d" % cfg)
            _close(stats.sum(0)[0], y.float().reshape(-1, Cout).sum(0), torch.float32, scale=10, what="stats cfg %d" % cfg)
            dx = hip_ops.conv_dgrad(gy, w, tuple(x.shape), (s, s), (p, p))
            _close(dx.permute(0, 3, 1, 2), xr.grad, dtype, what="dgrad cfg %d" % cfg)
        C.igemm_force_config(-1)
        for cfg in range(C.wgrad_num_configs()):
            C.wgrad_force_config(cfg)
            dw = hip_ops.conv_wgrad(gy, x, w, (s, s), (p, p))
            _close(dw, wr.grad, dtype, what="wgrad cfg %d" % cfg)
    finally:
        C.igemm_force_config(-1)
        C.wgrad_force_config(-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_stats_epilogue(dtype):
    torch.manual_seed(1)
    x = torch.randn(6, 14, 14, 64, device=DEV).to(dtype)
    w = (torch.randn(128, 64, 3, 3, device=DEV) / 24).contiguous(memory_format=torch.channels_last)
    y, stats = F.conv2d(x, w, None, 1, 1, want_stats=True)
    # per tile of 128 rows: (sum, M2 about the tile mean); the rows of each tile sit behind the slab
    yf = y.float().reshape(-1, 128)
    tiles = stats.shape[0]
    counts = stats.flatten().as_strided((tiles,), (1,), stats.numel())
    assert counts.sum().item() == yf.shape[0]
    _close(stats[:, 0].sum(0), yf.sum(0), torch.float32, scale=10, what="sum")
    for t in (0, tiles - 1):
        rows = yf[t * 128:(t + 1) * 128]
        assert counts[t].item() == rows.shape[0]
        _close(stats[t, 1], (rows - rows.mean(0)).square().sum(0), torch.float32, scale=10, what="M2 tile %d" % t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 96, 2048])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batch_norm(dtype, C, relu, res):
    torch.manual_seed(2)
    N, H, W = 4, 7, 9
    x = (torch.randn(N, H, W, C, device=DEV) * 2 + 0.5).to(dtype).requires_grad_(True)
    r = torch.randn(N, H, W, C, device=DEV).to(dtype).requires_grad_(True) if res else None
    g = (torch.rand(C, device=DEV) + 0.5).requires_grad_(True)
    b = torch.randn(C, device=DEV).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    y = F.batch_norm(x, g, b, rm, rv, nbt, True, 0.1, 1e-5, relu=relu, residual=r)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)

    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    rr = r.detach().float().permute(0, 3, 1, 2).requires_grad_(True) if res else None
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = TF.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = TF.relu(yr)
    # the kernel masks with the ROUNDED output; do the same in the reference for low precision
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype, what="y")
    _close(rm, rm2, torch.float32, what="running_mean")
    _close(rv, rv2, torch.float32, scale=5, what="running_var")
    assert int(nbt.item()) == 1
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype, scale=2, what="dx")
    _close(g.grad, gr.grad, dtype, scale=4, what="dgamma")
    _close(b.grad, br.grad, dtype, scale=4, what="dbeta")
    if res:
        _close(r.grad.permute(0, 3, 1, 2), rr.grad, dtype, what="dres")


def test_batch_norm_eval():
    torch.manual_seed(3)
    C = 64
    x = torch.randn(2, 5, 5, C, device=DEV).requires_grad_(True)
    g, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    y = F.batch_norm(x, g, b, rm, rv, None, False, 0.1, 1e-5, relu=True)
    y.sum().backward()
    xr = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
    yr = TF.relu(TF.batch_norm(xr, rm, rv, g, b, False, 0.1, 1e-5))
    yr.sum().backward()
    _close(y.permute(0, 3, 1, 2), yr, torch.float32)
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,s,p,H", [(3, 2, 1, 16), (3, 2, 0, 13), (2, 2, 0, 8), (2, 2, 0, 9), (2, 3, 0, 11), (3, 1, 1, 14), (5, 2, 2, 15), (5, 1, 2, 9)])
def test_max_pool(dtype, k, s, p, H):
    torch.manual_seed(4)
    x = torch.randn(3, H, H, 64, device=DEV).to(dtype).requires_grad_(True)
    y = F.max_pool2d(x, k, s, p)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = TF.max_pool2d(xr, k, s, p)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype)
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,o", [(7, 1), (13, 6), (10, 7), (29, 1), (56, 1), (3, 1)])
def test_avg_pool(dtype, H, o):
    torch.manual_seed(5)
    # 1x1 outputs of extent >= 4 take the parallel global-average kernel (squeeze-excitation at 56x56 ...)
    x = torch.randn(3, H, H, {29: 40, 56: 8, 3: 136}.get(H, 64), device=DEV).to(dtype).requires_grad_(True)
    y = F.adaptive_avg_pool2d(x, o)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = TF.adaptive_avg_pool2d(xr, o)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype)
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W,k,s,p", [(17, 17, 3, 1, 1), (17, 17, 5, 3, 0), (9, 12, (3, 2), (2, 1), (1, 1)),
                                       (7, 7, 2, 2, 0), (35, 35, 3, 1, 1)])
def test_avg_pool_fixed_window(dtype, H, W, k, s, p):
    """Overlapping / padded windows (Inception branch pools, the auxiliary head's 5x5 stride 3; padding counts in the
    divisor) and a floor-mode window that does not tile the input (7 / 2)."""
    torch.manual_seed(5)
    x = torch.randn(2, H, W, 40, device=DEV).to(dtype).requires_grad_(True)
    y = F.avg_pool2d(x, k, s, p)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    # reference on the CPU: ATen's GPU avg_pool2d backward returns a shifted gradient for a padded window on a
    # channels_last-strided input (seen on ROCm 7 / torch 2.10: corner 1/9 instead of 4/9 for 3x3 s1 p1)
    xr = x.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yr = TF.avg_pool2d(xr, k, s, p)
    yr.backward(gy.float().cpu().permute(0, 3, 1, 2).contiguous())
    _close(y.permute(0, 3, 1, 2), yr, dtype)
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,I,O,relu", [(32, 2048, 1000, False), (16, 512, 10, False), (24, 256, 4096, True)])
def test_linear(dtype, M, I, O, relu):
    torch.manual_seed(6)
    x = torch.randn(M, I, device=DEV).to(dtype).requires_grad_(True)
    w = (torch.randn(O, I, device=DEV) / I ** 0.5).requires_grad_(True)
    b = torch.randn(O, device=DEV).requires_grad_(True)
    y = F.linear(x, w, b, relu=relu, out_dtype=torch.float32)
    gy = torch.randn(M, O, device=DEV)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    yr = TF.linear(xr, wr, br)
    if relu:
        yr = TF.relu(yr)
    yr.backward(gy.to(dtype).float() if dtype != torch.float32 else gy)
    _close(y, yr, dtype, what="y")
    _close(x.grad, xr.grad, dtype, scale=2, what="dx")
    _close(w.grad, wr.grad, dtype, scale=2, what="dw")
    _close(b.grad, br.grad, dtype, scale=2, what="db")


@pytest.mark.parametrize("C", [1000, 10])
def test_cross_entropy_and_metrics(C):
    torch.manual_seed(7)
    N = 37
    z = (torch.randn(N, C, device=DEV) * 3).requires_grad_(True)
    t = torch.randint(0, C, (N,), device=DEV)
    loss, rec = F.loss_and_metrics(z, t)
    (loss * 3.0).backward()
    zr = z.detach().clone().requires_grad_(True)
    lr = TF.cross_entropy(zr, t)
    (lr * 3.0).backward()
    _close(loss, lr, torch.float32)
    _close(z.grad, zr.grad, torch.float32)
    top = zr.detach().topk(min(5, C), 1).indices
    c1 = (top[:, 0] == t).sum().item()
    c5 = (top == t[:, None]).any(1).sum().item()
    rec = rec.cpu()
    assert abs(rec[0].item() - lr.item() * N) < 1e-2
    assert rec[1].item() == c1 and rec[2].item() == c5 and rec[3].item() == N


def test_relu_add_dropout_normalize():
    torch.manual_seed(8)
    a = torch.randn(5, 6, 7, 64, device=DEV).requires_grad_(True)
    b = torch.randn(5, 6, 7, 64, device=DEV).requires_grad_(True)
    y = F.add_relu(a, b)
    y.sum().backward()
    assert torch.equal(y, torch.relu(a + b).detach())
    assert torch.equal(a.grad, (y > 0).float()) and torch.equal(b.grad, a.grad)
    r = F.relu(a.detach())
    assert torch.equal(r, torch.relu(a.detach()))
    d = F.dropout(torch.ones(1 << 16, device=DEV), 0.5, True)
    frac = (d == 0).float().mean().item()
    assert 0.45 < frac < 0.55 and set(d.unique().tolist()) <= {0.0, 2.0}
    u8 = torch.randint(0, 256, (3, 8, 9, 3), dtype=torch.uint8, device=DEV)
    n = F.normalize_u8(u8, torch.float32, 8)
    mean = torch.tensor(F.IMAGENET_MEAN, device=DEV) * 255
    std = torch.tensor(F.IMAGENET_STD, device=DEV) * 255
    _close(n[..., :3], (u8.float() - mean) / std, torch.float32)
    assert n.shape[-1] == 8 and float(n[..., 3:].abs().max()) == 0.0
    n2 = F.normalize_u8(u8.permute(0, 3, 1, 2).contiguous(), torch.bfloat16, 8, input_layout="nchw")
    _close(n2[..., :3], (u8.float() - mean) / std, torch.bfloat16)
    xf = torch.randn(2, 3, 5, 6, device=DEV)
    h = F.to_nhwc(xf, torch.float32, 8)
    assert torch.equal(h[..., :3], xf.permute(0, 2, 3, 1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_smooth_activations_se_gate_and_stochastic_depth(dtype):
    """act_fwd / act_bwd (sigmoid, SiLU, hard-sigmoid, hard-swish), the squeeze-excitation gate (dx and the
    per-(sample, channel) reduction in one pass), plain residual add and row-mode stochastic depth against fp32
    PyTorch references."""
    import torch.nn.functional as TF

    torch.manual_seed(21)
    ref_act = {"sigmoid": torch.sigmoid, "silu": TF.silu, "hardsigmoid": TF.hardsigmoid, "hardswish": TF.hardswish}
    for shape in ((3, 5, 7, 40), (2, 1283)):                   # second: scalar tail path (1283 % 8 != 0)
        x0 = (torch.randn(*shape, device=DEV) * 3).to(dtype)
        g0 = torch.randn(*shape, device=DEV).to(dtype)
        for kind, fn in ref_act.items():
            x = x0.clone().requires_grad_(True)
            y = F.activation(x, kind)
            y.backward(g0)
            xr = x0.float().clone().requires_grad_(True)
            yr = fn(xr)
            yr.backward(g0.float())
            _close(y, yr, dtype)
            _close(x.grad, xr.grad, dtype)
    for (NB, H, W, Cc) in ((3, 7, 7, 48), (2, 9, 5, 136), (5, 1, 1, 8)):
        x0 = torch.randn(NB, H, W, Cc, device=DEV).to(dtype)
        s0 = torch.rand(NB, Cc, device=DEV)
        g0 = torch.randn(NB, H, W, Cc, device=DEV).to(dtype)
        x, s = x0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
        y = F.se_scale(x, s)
        y.backward(g0)
        xr, sr = x0.float().clone().requires_grad_(True), s0.clone().requires_grad_(True)
        yr = xr * sr.view(NB, 1, 1, Cc)
        yr.backward(g0.float())
        _close(y, yr, dtype)
        _close(x.grad, xr.grad, dtype)
        assert s.grad.dtype == torch.float32
        _close(s.grad, sr.grad, dtype)
    a = torch.randn(4, 3, 3, 24, device=DEV).to(dtype).requires_grad_(True)
    b = torch.randn(4, 3, 3, 24, device=DEV).to(dtype).requires_grad_(True)
    y = F.add(a, b)
    y.backward(torch.ones_like(y))
    _close(y, a.detach().float() + b.detach().float(), dtype)
    assert torch.equal(a.grad, torch.ones_like(a)) and torch.equal(b.grad, a.grad)
    x = torch.ones(512, 2, 2, 8, device=DEV, dtype=dtype).requires_grad_(True)
    y = F.stochastic_depth(x, 0.25, True)
    y.sum().backward()
    per = y.detach().float().reshape(512, -1)
    assert bool((per.min(1).values == per.max(1).values).all())            # whole samples are kept or dropped
    vals = set(round(v, 3) for v in per[:, 0].unique().tolist())
    assert vals <= {0.0, round(1 / 0.75, 3)} or dtype != torch.float32
    assert 0.15 < (per[:, 0] == 0).float().mean().item() < 0.35
    assert torch.equal(x.grad.float().reshape(512, -1)[:, 0], per[:, 0])   # same mask scales the gradient
    assert F.stochastic_depth(x, 0.25, False) is x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,C", [((3, 5, 7), 96), ((2, 3, 3), 40), ((1, 2, 2), 8), ((2, 4, 4), 768), ((5,), 1536),
                                    ((3, 2, 1), 1024), ((4099,), 192)])
def test_layer_norm_and_gelu(dtype, rows, C):
    """Row-wise LayerNorm (4 ... 64 lanes per row, up to 4 chunks per lane, ragged last wave) and exact GELU."""
    if dtype == torch.float32 and C > 1024:
        pytest.skip("fp32 rows are limited to 256 16-byte chunks")
    torch.manual_seed(31)
    x0 = (torch.randn(*rows, C, device=DEV) * 2 + 0.5).to(dtype)
    w0, b0 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    g0 = torch.randn(*rows, C, device=DEV).to(dtype)
    x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    y = F.layer_norm(x, w, b, 1e-6)
    y.backward(g0)
    xr, wr, br = x0.float().clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    yr = TF.layer_norm(xr, (C,), wr, br, 1e-6)
    yr.backward(g0.float())
    _close(y, yr, dtype)
    _close(x.grad, xr.grad, dtype)
    nrows = x0.numel() // C
    _close(w.grad, wr.grad, dtype, scale=max(1.0, nrows ** 0.5 / 8))
    _close(b.grad, br.grad, dtype, scale=max(1.0, nrows ** 0.5 / 8))
    x = x0.clone().requires_grad_(True)
    y = F.activation(x, "gelu")
    y.backward(g0)
    xr = x0.float().clone().requires_grad_(True)
    yr = TF.gelu(xr)
    yr.backward(g0.float())
    _close(y, yr, dtype)
    _close(x.grad, xr.grad, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,gate", [("relu", "sigmoid"), ("silu", "sigmoid"), ("relu", "hardsigmoid")])
@pytest.mark.parametrize("fused", [True, False])
def test_squeeze_excitation_block(dtype, act, gate, fused):
    """nn.SqueezeExcitation on the GPU -- as one autograd node (fused backward: ds reduction, MLP backward, then
    dx = dy * s + d(mean) / HW in one pass) and as separate nodes -- against the CPU module in fp32."""
    import copy

    from distributed_pytorch_amd import nn as dnn
    from distributed_pytorch_amd.ops import hip_ops

    torch.manual_seed(12)
    ref = dnn.SqueezeExcitation(48, 12, activation=act, scale_activation=gate)
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(3.0)                            # gates away from 0.5, hard-sigmoid partly saturated
    gpu = copy.deepcopy(ref).cuda()
    x0 = torch.randn(3, 6, 5, 48)
    g0 = torch.randn(3, 6, 5, 48)
    xr = x0.to(dtype).float().clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g0.to(dtype).float())
    x = x0.to(dtype).cuda().requires_grad_(True)
    try:
        hip_ops._SE_FUSED = fused
        y = gpu(x)
        y.backward(g0.to(dtype).cuda())
    finally:
        hip_ops._SE_FUSED = True
    _close(y, yr, dtype, what="y")
    _close(x.grad, xr.grad, dtype, scale=2.0, what="dx")
    for (n, a), (_, b) in zip(gpu.named_parameters(), ref.named_parameters()):
        assert a.grad is not None and a.grad.shape == a.shape, n
        _close(a.grad, b.grad, dtype, scale=4.0, what=n)


@pytest.mark.parametrize("lp", [None, torch.bfloat16])
def test_fused_sgd_matches_torch(lp):
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.ops import hip_ops

    torch.manual_seed(9)
    shapes = [(64, 3, 7, 7), (128,), (1000, 512), (5,), (256, 64, 1, 1)]
    ps = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    ps[0] = ps[0].detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    if lp is not None:
        hip_ops.linear_weight(ps[2], lp)
    a = FusedSGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4)
    b = torch.optim.SGD(qs, lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step in range(4):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad = g.clone()
            q.grad = g.clone()
        a.step()
        b.step()
        for p, q in zip(ps, qs):
            _close(p, q, torch.float32, what="step %d" % step)
    if lp is not None:
        assert torch.equal(hip_ops.linear_weight(ps[2], lp), ps[2].detach().to(lp))
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    _close(sa["state"][2]["momentum_buffer"], sb["state"][2]["momentum_buffer"], torch.float32)


def test_loss_scaler_overflow_skips_step():
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.optim import FusedSGD

    p = torch.ones(1024, device=DEV).requires_grad_(True)
    opt = FusedSGD([p], lr=0.1, momentum=0.9)
    scaler = amp.LossScaler(torch.device(DEV), init_scale=1024.0, dynamic=True, growth_interval=2)
    p.grad = torch.full_like(p, float("inf"))
    scaler.check_overflow([p.grad])
    opt.arm_scaler(scaler)
    opt.step()
    assert torch.equal(p.detach(), torch.ones_like(p)) and scaler.loss_scale() == 512.0
    for _ in range(2):
        p.grad = torch.full_like(p, 512.0)
        scaler.check_overflow([p.grad])
        opt.arm_scaler(scaler)
        opt.step()
    assert scaler.loss_scale() == 1024.0
    assert abs(p.detach()[0].item() - (1 - 0.1 * 1 - 0.1 * 1.9)) < 1e-5


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("k,with_accum", [(1, False), (3, True)])
def test_dgrad_epilogue_bn_backward_partials(mode, k, with_accum):
    """The data-gradient epilogue's fused BatchNorm-backward partial sums equal the stand-alone reduce kernel
    run on the gradient it stored (same masks, fp32 sums; only the summation order differs)."""
    from distributed_pytorch_amd.ops import hip_ops
    C = hip_ops.C()
    torch.manual_seed(5)
    N, H, Cin, Cout = 5, 14, 128, 64
    dt = torch.bfloat16
    w = (torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, H, H, Cout, device=DEV).to(dt)
    bn_x = (torch.randn(N, H, H, Cin, device=DEV) * 1.5 + 0.3).to(dt)            # the BN input
    coef = torch.stack([torch.randn(Cin, device=DEV) * 0.1, torch.rand(Cin, device=DEV) + 0.5,
                        torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.2]).contiguous()
    z = torch.randn(N, H, H, Cin, device=DEV).to(dt)                              # BN output (mask source, mode 1)
    if mode == 3:     # the mask as the BN apply pass stores it: one byte per 8 elements, bit j = (z_j > 0)
        bits = (z.view(-1, 8) > 0).to(torch.uint8) << torch.arange(8, device=DEV, dtype=torch.uint8)
        z = bits.sum(1).to(torch.uint8).contiguous()
    accum = torch.randn(N, H, H, Cin, device=DEV).to(dt) if with_accum else None
    p = k // 2
    hip_ops._pending_bnred.clear()
    dx = hip_ops.conv_dgrad(dy, w, (N, H, H, Cin), (1, 1), (p, p), accum=accum,
                            bnred=(bn_x, z if mode in (1, 3) else None, coef, mode))
    slab, xptr = hip_ops._pending_bnred.pop(dx.data_ptr())[:2]
    assert xptr == bn_x.data_ptr()
    ref_dx = hip_ops.conv_dgrad(dy, w, (N, H, H, Cin), (1, 1), (p, p), accum=accum)
    # the gradient itself is untouched (3x3: the reference launch takes the filter-row kernel, whose fp32
    # summation order differs -> equal to bf16 rounding, not bit for bit)
    if k == 1:
        assert torch.equal(dx, ref_dx)
    else:
        _close(dx, ref_dx, dt, scale=0.5, what="dgrad with fused reduction vs plain")
    M = N * H * H
    nblk = C.bn_partial_rows(M, Cin)
    part = torch.empty(nblk, 2, Cin, device=DEV)
    C.bn_bwd_reduce(dx.data_ptr(), bn_x.data_ptr(), z.data_ptr() if mode in (1, 3) else 0, M, Cin, hip_ops._dt(dx),
                    coef[0].data_ptr(), coef[1].data_ptr(), mode, coef[2].data_ptr(), coef[3].data_ptr(),
                    part.data_ptr(), nblk, torch.cuda.current_stream().cuda_stream)
    got, want = slab.sum(0), part.sum(0)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-4 * scale, (got - want).abs().max().item() / scale


@pytest.mark.parametrize("groups,ch", [(32, 128), (4, 128), (8, 512)])
@pytest.mark.parametrize("relu", [True, False])
def test_grouped_dgrad_fuses_bn_backward(groups, ch, relu):
    """BatchNorm (+ReLU) -> stride-1 grouped 3x3 convolution (ResNeXt / RegNet bottleneck): the grouped data gradient
    reduces the BN's backward sums in its epilogue, one group slice per launch row.  Same gradients as with the
    fusion switched off (the stand-alone column reduction), and the BN really took the fused slab."""
    from distributed_pytorch_amd.ops import hip_ops
    dt = torch.bfloat16
    torch.manual_seed(9)
    N, H = 4, 14
    x0 = (torch.randn(N, H, H, ch, device=DEV) * 1.3 + 0.2).to(dt)
    w0 = (torch.randn(ch, ch // groups, 3, 3, device=DEV) / (9 * ch / groups) ** 0.5)
    w0 = w0.contiguous(memory_format=torch.channels_last)
    g0, b0 = torch.rand(ch, device=DEV) + 0.5, torch.randn(ch, device=DEV) * 0.2
    gy = torch.randn(N, H, H, ch, device=DEV).to(dt)

    def run(fuse):
        old = hip_ops._FUSE_BN_BWD, hip_ops._FUSE_BN_BWD_GROUPED
        hip_ops._FUSE_BN_BWD = hip_ops._FUSE_BN_BWD_GROUPED = fuse      # (the grouped fusion is off by default)
        try:
            x = x0.clone().requires_grad_(True)
            w, g, b = (t.clone().requires_grad_(True) for t in (w0, g0, b0))
            z = F.batch_norm(x, g, b, None, None, None, True, 0.1, 1e-5, relu=relu)
            y = F.conv2d(z, w, None, 1, 1, groups=groups)
            before = dict(hip_ops.bn_bwd_counts)
            y.backward(gy)
            took = hip_ops.bn_bwd_counts["fused"] - before["fused"]
            return x.grad, g.grad, b.grad, w.grad, took
        finally:
            hip_ops._FUSE_BN_BWD, hip_ops._FUSE_BN_BWD_GROUPED = old

    dx_f, dg_f, db_f, dw_f, took_f = run(True)
    dx_r, dg_r, db_r, dw_r, took_r = run(False)
    assert took_f == 1 and took_r == 0
    assert torch.equal(dw_f, dw_r)
    for a, b, what in ((dg_f, dg_r, "dgamma"), (db_f, db_r, "dbeta")):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-4 * scale, what
    _close(dx_f, dx_r, dt, scale=0.25, what="dx fused vs reduced")


@pytest.mark.parametrize("k,s,p,ceil", [(2, 2, 0, False), (3, 2, 0, True), (3, 2, 1, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relu_conv_maxpool_premasked_gradient(k, s, p, ceil, dtype):
    """conv + ReLU -> max-pool (VGG / AlexNet): the pool applies the ReLU mask to the POOLED gradient (pooled value 0 <=>
    the window's arg-max is inactive) and the convolution skips its full-size relu_bwd pass.  Bit-identical gradients,
    and the pass really is skipped."""
    from distributed_pytorch_amd.ops import hip_ops
    torch.manual_seed(13)
    x0 = torch.randn(3, 13, 14, 32, device=DEV).to(dtype)
    w0 = (torch.randn(64, 32, 3, 3, device=DEV) / 17).contiguous(memory_format=torch.channels_last)
    b0 = torch.randn(64, device=DEV) - 1.0           # many all-inactive windows
    mod = hip_ops.C()
    real = mod.relu_bwd
    calls = []

    def counting(*a):
        calls.append(a[3])                            # element count of the masked tensor
        return real(*a)

    def run(on):
        old = hip_ops._POOL_PREMASK
        hip_ops._POOL_PREMASK = on
        mod.relu_bwd = counting
        del calls[:]
        try:
            x = x0.clone().requires_grad_(True)
            w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = F.conv2d(x, w, b, 1, 1, relu=True)
            z = F.max_pool2d(y, k, s, p, ceil_mode=ceil)
            torch.manual_seed(14)
            gz = torch.randn_like(z.float()).to(dtype)
            z.backward(gz)
            return x.grad, w.grad, b.grad, list(calls), y.numel(), z.numel(), (z == 0).float().mean().item()
        finally:
            hip_ops._POOL_PREMASK = old
            mod.relu_bwd = real

    dx1, dw1, db1, c1, ny, nz, dead = run(True)
    dx0, dw0, db0, c0, _, _, _ = run(False)
    assert 0.05 < dead < 0.95
    assert c1 == [nz] and c0 == [ny]                  # masked the pooled gradient instead of the full-size one
    assert torch.equal(dx1, dx0) and torch.equal(dw1, dw0) and torch.equal(db1, db0)
    assert not hip_ops._premasked


def test_relu_tag_is_void_after_an_in_place_edit_of_the_activation():
    """The pre-masked fast paths trust a tag on the conv + ReLU output.  An in-place edit between the producer and the
    pool voids it (the tag records the tensor's version): the pool must not mask with values the ReLU never produced.
    Autograd then refuses the backward anyway (the convolution saved its output) -- never silently wrong gradients."""
    from distributed_pytorch_amd.ops import hip_ops
    torch.manual_seed(3)
    x = torch.randn(2, 8, 8, 32, device=DEV).bfloat16().requires_grad_(True)
    w = (torch.randn(32, 32, 3, 3, device=DEV) / 17).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1, relu=True)
    assert hip_ops._is_relu_out(y)
    y.sub_(1.0)
    assert not hip_ops._is_relu_out(y)
    mod = hip_ops.C()
    real, calls = mod.relu_bwd, []
    mod.relu_bwd = lambda *a: (calls.append(a[3]), real(*a))[1]
    try:
        z = F.max_pool2d(y, 2, 2, 0)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            z.float().sum().backward()
        assert calls == []                      # the pool did not take the pre-masked path
    finally:
        mod.relu_bwd = real
        hip_ops._premasked.clear()


@pytest.mark.parametrize("row3", [-1, 0])
@pytest.mark.parametrize("cin,cmid,cout", [(32, 64, 128), (64, 128, 64), (64, 512, 64)])
def test_relu_conv_conv_masked_data_gradient(cin, cmid, cout, row3):
    """conv + ReLU -> 3x3 conv (VGG): the second convolution's data gradient is stored through the first one's ReLU
    mask (filter-row kernel epilogue; with the filter-row kernels switched off, the launcher's in-place fallback) and
    the first convolution skips its relu_bwd pass.  Bit-identical gradients."""
    from distributed_pytorch_amd.ops import hip_ops
    dt = torch.bfloat16
    torch.manual_seed(21)
    x0 = torch.randn(3, 12, 10, cin, device=DEV).to(dt)
    w1 = (torch.randn(cmid, cin, 3, 3, device=DEV) / (9 * cin) ** 0.5).contiguous(memory_format=torch.channels_last)
    b1 = torch.randn(cmid, device=DEV) * 0.5
    w2 = (torch.randn(cout, cmid, 3, 3, device=DEV) / (9 * cmid) ** 0.5).contiguous(memory_format=torch.channels_last)
    b2 = torch.randn(cout, device=DEV) * 0.5
    mod = hip_ops.C()
    real = mod.relu_bwd
    calls = []

    def counting(*a):
        calls.append(a[3])
        return real(*a)

    def run(on):
        old = hip_ops._DGRAD_XMASK
        hip_ops._DGRAD_XMASK = on
        mod.relu_bwd = counting
        mod.igemm_set_row3(row3)
        del calls[:]
        try:
            x = x0.clone().requires_grad_(True)
            ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
            y = F.conv2d(x, ps[0], ps[1], 1, 1, relu=True)
            z = F.conv2d(y, ps[2], ps[3], 1, 1, relu=True)
            torch.manual_seed(22)
            z.backward(torch.randn_like(z.float()).to(dt))
            return [x.grad] + [q.grad for q in ps], list(calls), y.numel(), z.numel()
        finally:
            hip_ops._DGRAD_XMASK = old
            mod.relu_bwd = real
            mod.igemm_set_row3(-1)

    g1, c1, ny, nz = run(True)
    g0, c0, _, _ = run(False)
    assert c0 == [nz, ny] and c1 == [nz]              # the first convolution's pass is gone
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    assert not hip_ops._premasked


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cat_channels_ceil_maxpool_avgpool(dtype):
    torch.manual_seed(11)
    a = torch.randn(3, 9, 7, 64, device=DEV).to(dtype).requires_grad_(True)
    b = torch.randn(3, 9, 7, 32, device=DEV).to(dtype).requires_grad_(True)
    y = F.cat_channels([a, b])
    assert torch.equal(y, torch.cat([a, b], -1))
    g = torch.randn_like(y.float()).to(dtype)
    y.backward(g)
    assert torch.equal(a.grad, g[..., :64]) and torch.equal(b.grad, g[..., 64:])
    # ceil_mode max pooling (SqueezeNet): 13 -> 6, 14 -> 7 (the last window hangs over the edge)
    for H in (13, 14, 55):
        x = torch.randn(2, H, H, 16, device=DEV).to(dtype).requires_grad_(True)
        yp = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
        xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = TF.max_pool2d(xr, 3, 2, 0, ceil_mode=True)
        assert yp.shape[1:3] == yr.shape[2:]
        _close(yp.permute(0, 3, 1, 2), yr, dtype, what="maxpool ceil fwd")
        gy = torch.randn_like(yr)
        yp.backward(gy.permute(0, 2, 3, 1).to(dtype))
        yr.backward(gy.to(dtype).float())
        _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype, what="maxpool ceil bwd")
    x = torch.randn(2, 8, 12, 32, device=DEV).to(dtype)
    _close(F.avg_pool2d(x, 2, 2).permute(0, 3, 1, 2), TF.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, 2), dtype,
           what="avgpool 2x2")


GROUP_CASES = [
    # N, H, W, Cin, Cout, groups, k, stride, pad
    (3, 14, 14, 128, 128, 32, 3, 1, 1),     # ResNeXt 32x4d layer1 (16 groups merged per tile)
    (3, 14, 14, 256, 256, 32, 3, 2, 1),     # ResNeXt stride-2 block
    (2, 7, 7, 1024, 1024, 32, 3, 1, 1),     # 32 channels per group: 2 merged
    (2, 9, 11, 128, 256, 2, 3, 1, 1),       # wide groups, no merging (64 -> 128 per group)
    (2, 14, 14, 96, 96, 96, 3, 1, 1),       # depthwise, every group in one block-diagonal tile
    (2, 15, 15, 192, 192, 192, 3, 2, 1),    # depthwise stride 2, 64 merged
    (2, 8, 8, 64, 128, 4, 1, 1, 0),         # grouped 1x1 (wgrad linear fast path)
    (2, 14, 14, 72, 72, 3, 3, 1, 1),        # RegNetX-1.6GF: 24 channels per group, K tiles straddle taps
    (2, 14, 14, 400, 400, 25, 3, 2, 1),     # RegNetX-400MF last stage: 25 groups of 16, unmerged
    (2, 14, 14, 40, 40, 40, 5, 2, 2),       # direct depthwise kernels: 5x5 stride 2 (MNASNet / EfficientNet)
    (2, 12, 12, 96, 96, 96, 7, 1, 3),       # ConvNeXt 7x7
    (3, 9, 9, 8, 8, 8, 3, 1, 1),            # a single 16-byte chunk
    (2, 28, 28, 144, 144, 144, 3, 2, 1),    # two channel slabs, ragged second one
    (130, 5, 5, 24, 24, 24, 3, 1, 1),       # many images, tiny extent
    (2, 14, 14, 240, 240, 240, 5, 1, 2),    # sliding-window kernel, 5x5 stride 1 (EfficientNet), 30 chunks -> slabs of 15
    (3, 30, 20, 16, 16, 16, 3, 1, 1),       # sliding-window kernel over several strips, last one ragged
    (2, 12, 16, 24, 24, 24, 5, 1, 1),       # 5x5 with padding 1 (output smaller than the input)
    (2, 16, 12, 32, 32, 32, 3, 2, 1),       # stride-2 backward-data on 2x2 blocks, non-square
    (2, 14, 14, 224, 224, 4, 3, 1, 1),      # RegNetY-8GF: 56 per group -> two K tiles per tap, the second with a masked tail
    (3, 15, 13, 112, 112, 2, 3, 2, 1),      # the same width, stride 2, odd extents, the LAST group's tail ends the tensor row
    (2, 7, 7, 240, 240, 2, 3, 1, 1),        # RegNetX-8GF: 120 per group (four K tiles, 8 channels masked)
    (2, 9, 9, 96, 144, 2, 3, 1, 1),         # 48 -> 72 per group: tails of different length forward and backward
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", GROUP_CASES)
@pytest.mark.parametrize("bias_relu", [False, True])
def test_group_conv_fwd_bwd(case, dtype, bias_relu):
    """Grouped / depthwise convolution (one launch per pass, group on grid.y) against torch's conv2d(groups=)."""
    N, H, W, Cin, Cout, G, k, s, p = case
    torch.manual_seed(5)
    xq = torch.randn(N, H, W, Cin, device=DEV).to(dtype).requires_grad_(True)
    w = (torch.randn(Cout, Cin // G, k, k, device=DEV) / (Cin // G * k * k) ** 0.5).contiguous(
        memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(Cout, device=DEV).requires_grad_(True) if bias_relu else None
    y, stats = F.conv2d(xq, w, b, s, p, relu=bias_relu, groups=G, want_stats=True)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)

    xr = xq.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True) if b is not None else None
    yr = TF.conv2d(xr, wr, br, s, p, 1, G)
    if bias_relu:
        mask = (y.detach().permute(0, 3, 1, 2) > 0)
        assert (mask != (yr.detach() > 0)).float().mean().item() < 1e-3
        yr = yr * mask
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype, what="fprop")
    yf = y.detach().float().reshape(-1, Cout)
    _close(stats.sum(0)[0], yf.sum(0), torch.float32, scale=10, what="stats sum")
    _close(_slab_m2(stats), (yf - yf.mean(0)).square().sum(0), torch.float32, scale=10, what="stats M2")
    _close(xq.grad.permute(0, 3, 1, 2), xr.grad, dtype, what="dgrad")
    _close(w.grad, wr.grad, dtype, what="wgrad")
    if b is not None:
        _close(b.grad, br.grad, dtype, what="bias grad")


@pytest.mark.parametrize("rows", [0, 3, 8])
@pytest.mark.parametrize("k", [3, 5])
def test_depthwise_sliding_window_strip_heights(k, rows):
    """The stride-1 sliding-window depthwise kernels (forward + backward-data) at forced strip heights -- 3 leaves a
    ragged last strip and a window rotation that ends mid-period, 8 is the production height -- against the
    window-per-pixel kernels (rows = 0) and fp32 autograd."""
    from distributed_pytorch_amd.ops import hip_ops

    torch.manual_seed(9)
    N, H, W, Cc = 3, 19, 13, 48
    x0 = torch.randn(N, H, W, Cc, device=DEV).to(torch.bfloat16)
    w0 = (torch.randn(Cc, 1, k, k, device=DEV) / k).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, H, W, Cc, device=DEV).to(torch.bfloat16)
    hip_ops.C().dw_set_sliding(rows)
    try:
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        y, stats = F.conv2d(x, w, None, 1, k // 2, groups=Cc, want_stats=True)
        y.backward(gy)
    finally:
        hip_ops.C().dw_set_sliding(1)
    xr = x0.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w0.to(torch.bfloat16).float().requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, 1, k // 2, 1, Cc)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, torch.bfloat16, what="fprop")
    yf = y.detach().float().reshape(-1, Cc)
    _close(stats.sum(0)[0], yf.sum(0), torch.float32, scale=10, what="stats sum")
    _close(_slab_m2(stats), (yf - yf.mean(0)).square().sum(0), torch.float32, scale=10, what="stats M2")
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, torch.bfloat16, what="dgrad")
    _close(w.grad, wr.grad, torch.bfloat16, what="wgrad")


def test_depthwise_direct_equals_block_diagonal_path():
    """DPA_DW_DIRECT=0 keeps depthwise convolutions on the grouped implicit-GEMM kernels; both paths agree."""
    from distributed_pytorch_amd.ops import hip_ops

    torch.manual_seed(6)
    x0 = torch.randn(2, 14, 14, 96, device=DEV).to(torch.bfloat16)
    w0 = (torch.randn(96, 1, 3, 3, device=DEV) / 3).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, 7, 7, 96, device=DEV).to(torch.bfloat16)
    res = []
    try:
        for direct in (True, False):
            hip_ops._DW_DIRECT = direct
            x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            y, stats = F.conv2d(x, w, None, 2, 1, groups=96, want_stats=True)
            y.backward(gy)
            res.append((y.detach(), torch.stack([stats[:, 0].sum(0), _slab_m2(stats)]), x.grad, w.grad))
    finally:
        hip_ops._DW_DIRECT = True
    for a, b in zip(*res):
        _close(a, b, torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["silu", "hardswish", "gelu"])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_smooth_activation(dtype, kind, training):
    """BN + SiLU / hard-swish / GELU in one normalise pass; backward folds the activation's slope (at the recomputed
    pre-activation) into the BN reduce and apply kernels.  Reference: BN then the activation in fp32 autograd."""
    torch.manual_seed(4)
    N, H, W, C = 4, 7, 9, 96
    fn = {"silu": TF.silu, "hardswish": TF.hardswish, "gelu": TF.gelu}[kind]
    x = (torch.randn(N, H, W, C, device=DEV) * 2 + 0.5).to(dtype).requires_grad_(True)
    g = (torch.rand(C, device=DEV) * 2 + 1).requires_grad_(True)
    b = torch.randn(C, device=DEV).requires_grad_(True)
    rm, rv = torch.randn(C, device=DEV) * 0.1 + 0.5, torch.rand(C, device=DEV) + 3.5
    rm2, rv2 = rm.clone(), rv.clone()
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    y = F.batch_norm(x, g, b, rm, rv, nbt if training else None, training, 0.1, 1e-5, relu=kind)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = fn(TF.batch_norm(xr, rm2, rv2, gr, br, training, 0.1, 1e-5))
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype, what="y")
    if training or dtype == torch.float32 or kind != "hardswish":
        # (eval mode applies the activation to the ROUNDED BN output; hard-swish's slope jumps at +-3, so a bf16
        # value that rounds across the kink legitimately gets the other slope)
        _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype, scale=2.0, what="dx")
    if training:                 # eval-mode BN is frozen: no parameter gradients
        _close(g.grad, gr.grad, dtype, scale=4.0, what="dgamma")
        _close(b.grad, br.grad, dtype, scale=4.0, what="dbeta")
        _close(rm, rm2, torch.float32, what="running_mean")
        _close(rv, rv2, torch.float32, what="running_var")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu6(dtype, res, training):
    """BN (+residual) + ReLU6 in one pass (activation code 6); backward masks 0 < y < 6.  The reference takes the
    kernel's own mask: an output within bf16 rounding of 6 may land on either side."""
    torch.manual_seed(4)
    N, H, W, C = 4, 7, 9, 96
    x = (torch.randn(N, H, W, C, device=DEV) * 2 + 0.5).to(dtype).requires_grad_(True)
    r = torch.randn(N, H, W, C, device=DEV).to(dtype).requires_grad_(True) if res else None
    g = (torch.rand(C, device=DEV) * 3 + 2).requires_grad_(True)
    b = (torch.randn(C, device=DEV) + 2).requires_grad_(True)
    rm, rv = torch.randn(C, device=DEV) * 0.1 + 0.5, torch.rand(C, device=DEV) + 3.5
    rm2, rv2 = rm.clone(), rv.clone()
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    y = F.batch_norm(x, g, b, rm, rv, nbt if training else None, training, 0.1, 1e-5, relu=6, residual=r)
    assert 0.02 < (y.detach() >= 6).float().mean().item() < 0.6 and (y.detach() <= 0).float().mean().item() > 0.02
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)

    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    rr = r.detach().float().permute(0, 3, 1, 2).requires_grad_(True) if res else None
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    pre = TF.batch_norm(xr, rm2, rv2, gr, br, training, 0.1, 1e-5)
    if res:
        pre = pre + rr
    _close(y.permute(0, 3, 1, 2), TF.relu6(pre), dtype, what="y")
    yk = y.detach().permute(0, 3, 1, 2)
    mask = (yk > 0) & (yk < 6)
    assert (mask != ((pre > 0) & (pre < 6))).float().mean().item() < 2e-2
    (pre * mask).backward(gy.float().permute(0, 3, 1, 2))
    _close(x.grad.permute(0, 3, 1, 2), xr.grad, dtype, scale=4, what="dx")
    if training:
        _close(g.grad, gr.grad, dtype, scale=8, what="dgamma")
        _close(b.grad, br.grad, dtype, scale=8, what="dbeta")
    if res:
        _close(r.grad.permute(0, 3, 1, 2), rr.grad, dtype, what="dres")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(3, 2, 1, False, 17, 19), (3, 2, 0, True, 16, 16), (2, 2, 0, False, 12, 10)])
def test_fused_bn_relu_maxpool_stem(dtype, geom):
    """BN + ReLU + max-pool in one pass (never writes the normalised tensor) == the three separate kernels,
    bit for bit in the outputs / running statistics and to rounding in the gradients, == the torch reference."""
    from distributed_pytorch_amd.ops import hip_ops
    k, s, p, ceil, H, W = geom
    torch.manual_seed(6)
    N, C = 3, 64
    x0 = (torch.randn(N, H, W, C, device=DEV) * 1.5 + 0.3).to(dtype)
    g0 = torch.rand(C, device=DEV) + 0.5
    b0 = torch.randn(C, device=DEV) * 0.3
    outs = []
    for fused in (True, False):
        hip_ops._FUSE_STEM = fused
        try:
            x = x0.clone().requires_grad_(True)
            g, b = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.long, device=DEV)
            y = F.bn_relu_max_pool2d(x, g, b, rm, rv, nbt, True, 0.1, 1e-5, k, s, p, ceil)
            if fused:
                assert type(y.grad_fn).__name__ == "_BNReluMaxPoolFnBackward"
                gy = torch.randn_like(y.float()).to(dtype)
            y.backward(gy)
            outs.append((y.detach(), x.grad, g.grad, b.grad, rm, rv, int(nbt.item())))
        finally:
            hip_ops._FUSE_STEM = True
    (yf, dxf, dgf, dbf, rmf, rvf, nf), (yu, dxu, dgu, dbu, rmu, rvu, nu) = outs
    assert torch.equal(yf, yu) and torch.equal(rmf, rmu) and torch.equal(rvf, rvu) and nf == nu == 1
    _close(dxf, dxu, dtype, what="dx fused vs unfused")
    _close(dgf, dgu, dtype, scale=2, what="dgamma fused vs unfused")
    _close(dbf, dbu, dtype, scale=2, what="dbeta fused vs unfused")
    # torch reference on the same (rounded) inputs
    xr = x0.float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    yr = TF.max_pool2d(TF.relu(TF.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)), k, s, p, ceil_mode=ceil)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(yf.permute(0, 3, 1, 2), yr, dtype, what="y")
    if dtype == torch.float32:        # low precision: rounding can move an arg-max between tied neighbours
        _close(dxf.permute(0, 3, 1, 2), xr.grad, dtype, scale=2, what="dx")
        _close(dgf, gr.grad, dtype, scale=4, what="dgamma")
        _close(dbf, br.grad, dtype, scale=4, what="dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("src", ["u8_nhwc", "u8_nchw", "float_nchw"])
def test_space_to_depth_stem(dtype, src):
    """7x7 stride-2 pad-3 stem as a 4x4 convolution over the space-to-depth image (F.to_s2d): forward, fused
    statistics and the weight gradient (mapped back to [K, 3, 7, 7]) against torch's conv2d on the same pixels."""
    torch.manual_seed(8)
    N, H, W, K = 3, 20, 28, 64
    mean, std = F.IMAGENET_MEAN, F.IMAGENET_STD
    if src == "float_nchw":
        img = torch.randn(N, 3, H, W, device=DEV)
        ref_in = img.to(dtype).float()
        xs = F.to_s2d(img, dtype)
    else:
        u8 = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device=DEV)
        m = torch.tensor(mean, device=DEV) * 255
        s = torch.tensor(std, device=DEV) * 255
        ref_in = ((u8.float() - m) / s).to(dtype).float().permute(0, 3, 1, 2)
        xs = F.to_s2d(u8 if src == "u8_nhwc" else u8.permute(0, 3, 1, 2).contiguous(), dtype)
    assert xs.shape == (N, H // 2, W // 2, 16) and xs._dpa_s2d and bool((xs[..., 12:] == 0).all())
    # channel (py*2+px)*3+c of s2d pixel (Y, X) is image pixel (2Y+py, 2X+px)
    _close(xs[:, 3, 5, 9:12], ref_in[:, :, 7, 11], dtype, scale=0.01 if dtype == torch.float32 else 1.0, what="layout")
    w = (torch.randn(K, 3, 7, 7, device=DEV) / 12).requires_grad_(True)
    b = torch.randn(K, device=DEV).requires_grad_(True)
    y, stats = F.conv2d(xs, w, b, 2, 3, relu=True, want_stats=True, needs_input_grad=False)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    yr = TF.conv2d(ref_in, wr, br, 2, 3)
    mask = y.detach().permute(0, 3, 1, 2) > 0
    assert (mask != (yr.detach() > 0)).float().mean().item() < 2e-3
    yr = yr * mask
    yr.backward(gy.float().permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), yr, dtype, what="fprop")
    yf = y.detach().float().reshape(-1, K)
    _close(stats.sum(0)[0], yf.sum(0), torch.float32, scale=10, what="stats")
    _close(w.grad, wr.grad, dtype, scale=2, what="wgrad")
    _close(b.grad, br.grad, dtype, scale=2, what="bias grad")

"""BatchNorm + ReLU applied while the consuming convolution stages its operand (igemm kIgemmAxf / wgrad x_scale).

Two checks per shape: (1) the fused chain is BIT-IDENTICAL to the unfused chain of the same HIP kernels (same values
reach LDS, same accumulation order), forward and every gradient; (2) the fused chain matches an fp32 PyTorch
reference of conv(relu(batch_norm(x))) fed the same rounded inputs."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.ops import functional as F  # noqa: E402
from distributed_pytorch_amd.ops import hip_ops  # noqa: E402

from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"

AXF_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (4, 14, 14, 64, 64, 3, 1, 1),        # filter-row kernel, 64-byte K tiles, narrow N
    (4, 14, 14, 64, 256, 1, 1, 0),       # 1x1 expansion (conv3 of a bottleneck)
    (3, 15, 13, 128, 128, 3, 2, 1),      # stride-2 3x3 (first block of a stage): generic aligned gather, odd extents
    (2, 9, 9, 512, 512, 3, 1, 1),        # filter-row kernel, 128-byte K tiles
    (2, 7, 7, 512, 2048, 1, 1, 0),       # deep 1x1, 128-byte K tiles
    (5, 9, 11, 96, 72, 3, 1, 1),         # Cin % 32 != 0 for bf16 -> not eligible: must fall back to the materialised path
    (2, 12, 12, 32, 40, 3, 1, 1),        # smallest eligible channel count, Cout tail
]


def _chain(x, g, b, w, s, p, fused):
    old, old_red = hip_ops._AXF, hip_ops._AXF_MAX_RED
    hip_ops._AXF, hip_ops._AXF_MAX_RED = fused, 1e9         # every eligible shape, not only the ones the heuristic picks
    try:
        C = x.shape[-1]
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        a = F.batch_norm(x, g, b, rm, rv, nbt, True, 0.1, 1e-5, relu=True, defer=True)
        deferred = getattr(a, "_dpa_deferred_coef", None) is not None
        y = F.conv2d(a, w, None, s, p)
        return y, deferred
    finally:
        hip_ops._AXF, hip_ops._AXF_MAX_RED = old, old_red


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", AXF_CASES)
def test_bn_relu_apply_on_load(case, dtype):
    N, H, W, Cin, Cout, k, s, p = case
    torch.manual_seed(11)
    x0 = (torch.randn(N, H, W, Cin, device=DEV) * 1.5 + 0.3).to(dtype)
    g0 = torch.rand(Cin, device=DEV) + 0.5
    b0 = torch.randn(Cin, device=DEV) * 0.3
    w0 = (torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    res = []
    # bit-identity is a statement about ONE weight-gradient kernel with and without the transform on load: the
    # filter-row kernel (no apply-on-load variant) would take the unfused 64-channel 3x3 case and sum in another order
    hip_ops.C().wgrad_row3_set_mode(0)
    try:
        for fused in (True, False):
            x = x0.clone().requires_grad_(True)
            g, b, w = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            y, deferred = _chain(x, g, b, w, s, p, fused)
            assert deferred == fused
            torch.manual_seed(12)
            gy = torch.randn_like(y.float()).to(dtype)
            y.backward(gy)
            res.append((y.detach(), x.grad, g.grad, b.grad, w.grad, gy))
    finally:
        hip_ops.C().wgrad_row3_set_mode(-1)
    (yf, dxf, dgf, dbf, dwf, gy), (yu, dxu, dgu, dbu, dwu, _) = res
    assert torch.equal(yf, yu), "forward differs from the unfused kernels"
    assert torch.equal(dwf, dwu), "weight gradient differs from the unfused kernels"
    assert torch.equal(dxf, dxu) and torch.equal(dgf, dgu) and torch.equal(dbf, dbu), "BN backward differs"

    # fp32 reference
    xr = x0.float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    wr = w0.to(dtype).float().requires_grad_(True)
    ar = TF.relu(TF.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    # the kernels round the activation to the compute dtype before the MFMA: do the same (straight-through)
    ar_q = ar + (ar.to(dtype).float() - ar).detach()
    yr = TF.conv2d(ar_q, wr, None, s, p)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    assert_close(yf.permute(0, 3, 1, 2), yr, dtype, what="fprop")
    assert_close(dwf, wr.grad, dtype, what="wgrad", scale=2)
    # BN backward: the ReLU mask comes from the rounded activation and dy passes one more rounding
    assert_close(dxf.permute(0, 3, 1, 2), xr.grad, dtype, what="dx", scale=4)
    assert_close(dgf, gr.grad, dtype, what="dgamma", scale=4)
    assert_close(dbf, br.grad, dtype, what="dbeta", scale=4)


def test_deferred_bn_relu_into_grouped_conv_materialises():
    """A consumer that cannot apply the BatchNorm on load (grouped convolution) gets the materialised activation."""
    torch.manual_seed(5)
    dtype = torch.bfloat16
    x0 = torch.randn(2, 8, 8, 64, device=DEV).to(dtype)
    g0, b0 = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.2
    w0 = (torch.randn(64, 16, 3, 3, device=DEV) / 12).contiguous(memory_format=torch.channels_last)
    outs = []
    for defer in (True, False):
        x = x0.clone().requires_grad_(True)
        g, b, w = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        a = F.batch_norm(x, g, b, rm, rv, None, True, 0.1, 1e-5, relu=True, defer=defer)
        y = F.conv2d(a, w, None, 1, 1, groups=4)
        y.float().square().sum().backward()
        outs.append((y.detach(), x.grad, w.grad))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.parametrize("kind", ["plain_bn", "alias"])
def test_lazy_residual_gradient_with_a_second_consumer(kind):
    """The residual of a BN + add + ReLU normally gets its gradient back UNMASKED plus a bit mask that its producer
    applies.  When the residual has a second consumer autograd sums the gradients first; the tensor hook must then
    repair the sum (silent wrong gradients otherwise).  Reference: the same graph with the lazy path off."""
    dtype = torch.bfloat16
    torch.manual_seed(21)
    x0 = torch.randn(2, 8, 8, 64, device=DEV).to(dtype)
    wa0 = (torch.randn(64, 64, 1, 1, device=DEV) / 8).contiguous(memory_format=torch.channels_last)
    wb0 = (torch.randn(64, 64, 3, 3, device=DEV) / 24).contiguous(memory_format=torch.channels_last)
    ga, ba = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
    gb, bb = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1

    def run(lazy):
        old = hip_ops._LAZY_DRES
        hip_ops._LAZY_DRES = lazy
        try:
            x = x0.clone().requires_grad_(True)
            wa, wb = wa0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
            if kind == "plain_bn":
                res = F.batch_norm(F.conv2d(x, wa, None, 1, 0), ga, ba, None, None, None, True, 0.1, 1e-5)
                main_in = x
            else:
                main_in, res = F.conv2d(x, wa, None, 1, 0, skip=True)      # res aliases x
            y = F.batch_norm(F.conv2d(main_in, wb, None, 1, 1), gb, bb, None, None, None, True, 0.1, 1e-5,
                             relu=True, residual=res)
            second = (res.float() * 0.5).sum()                              # a consumer the framework knows nothing about
            (y.float().square().sum() + second).backward()
            return x.grad.float(), wa.grad.float(), wb.grad.float()
        finally:
            hip_ops._LAZY_DRES = old

    lazy, ref = run(True), run(False)
    for a, b, what in zip(lazy, ref, ("dx", "dwa", "dwb")):
        assert_close(a, b, dtype, what="%s %s" % (kind, what))
    assert len(hip_ops._pending_mask) == 0, "an unapplied ReLU mask was left behind"


# ---- statistics at |mean| >> std (fp32): Chan partials instead of sums of squares ---------------------------------
def _ref_stats(x2d):
    xd = x2d.double()
    return xd.mean(0), xd.var(0, unbiased=False)


@pytest.mark.parametrize("sync", [False, True])
def test_batch_norm_statistics_at_mean_1000_std(sync):
    """fp32 BatchNorm (and the SyncBN code path: local moments -> gather -> Chan merge) with mean = 1e3 std."""
    torch.manual_seed(31)
    C, M = 64, 4 * 33 * 29
    x = (torch.randn(4, 33, 29, C, device=DEV) + 1000.0).requires_grad_(True)
    g, b = torch.ones(C, device=DEV, requires_grad=True), torch.zeros(C, device=DEV, requires_grad=True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    old = hip_ops._FORCE_SYNC_PATH
    hip_ops._FORCE_SYNC_PATH = sync
    try:
        y = F.batch_norm(x, g, b, rm, rv, None, True, 0.1, 1e-5, group="world" if sync else None)
    finally:
        hip_ops._FORCE_SYNC_PATH = old
    mean, var = _ref_stats(x.detach().reshape(-1, C))
    yref = ((x.detach().double() - mean) / (var + 1e-5).sqrt())
    assert (y.double() - yref).abs().max().item() < 2e-3
    assert torch.allclose(rv.double(), 0.9 + 0.1 * var * M / (M - 1), rtol=1e-3), (rv.double() - (0.9 + 0.1 * var)).abs().max()
    assert torch.allclose(rm.double(), 0.1 * mean, rtol=1e-6)


@pytest.mark.parametrize("kind", ["dense", "depthwise"])
def test_conv_epilogue_statistics_at_mean_1000_std(kind):
    """The (sum, M2, rows) partials a convolution epilogue emits give the right variance when the conv output sits at
    1e3 std from zero (bias = 1000): fp32, compared with float64 statistics of the kernel's own output."""
    torch.manual_seed(32)
    Cin = Cout = 64
    x = torch.randn(3, 19, 17, Cin, device=DEV)
    if kind == "dense":
        w = (torch.randn(Cout, Cin, 3, 3, device=DEV) / 24).contiguous(memory_format=torch.channels_last)
        groups = 1
    else:
        w = torch.randn(Cout, 1, 3, 3, device=DEV) / 3
        groups = Cout
    bias = torch.full((Cout,), 1000.0, device=DEV)
    y, stats = F.conv2d(x, w, bias, 1, 1, want_stats=True, groups=groups)
    C = Cout
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    out = F.batch_norm(y, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), rm, rv, None, True, 0.1, 1e-5,
                       stats=stats)
    mean, var = _ref_stats(y.reshape(-1, C))
    yref = (y.double() - mean) / (var + 1e-5).sqrt()
    assert (out.double() - yref).abs().max().item() < 2e-3
    assert torch.allclose(rm.double(), 0.1 * mean, rtol=1e-6)


# ---- fused attention --------------------------------------------------------------------------------------------
ATTN_CASES = [
    # B, S, H, D, bias, nW
    (3, 197, 12, 64, False, 0),      # ViT-B/16
    (5, 50, 12, 64, False, 0),       # ViT-B/32
    (2, 257, 16, 80, False, 0),      # ViT-H/14: D = 80 (padded to 96 inside the kernel), 9 key blocks
    (8, 49, 3, 32, True, 4),         # Swin-T stage 1, shifted windows
    (6, 49, 24, 32, True, 0),        # Swin-T stage 4
    (4, 64, 4, 32, True, 2),         # Swin-V2 window 8
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", ATTN_CASES)
def test_fused_attention(case, dtype):
    B, S, H, D, has_bias, nW = case
    torch.manual_seed(51)
    qkv = (torch.randn(B, S, 3 * H * D, device=DEV) * 0.8).to(dtype).requires_grad_(True)
    rel = (torch.randn(H, S, S, device=DEV) * 0.5).requires_grad_(True) if has_bias else None
    mask = None
    if nW:
        mask = torch.where(torch.rand(nW, S, S, device=DEV) < 0.2, -100.0, 0.0)
        mask[:, torch.arange(S), torch.arange(S)] = 0.0
    scale = torch.linspace(0.8, 1.2, H, device=DEV) * D ** -0.5
    y = F.attention(qkv, H, scale, rel, mask)
    gy = torch.randn(y.shape, device=DEV).to(dtype)
    y.backward(gy)

    qr = qkv.detach().float().requires_grad_(True)
    rr = rel.detach().clone().requires_grad_(True) if has_bias else None
    q, k, v = qr.reshape(B, S, 3, H, D).permute(2, 0, 3, 1, 4).unbind(0)
    att = (q @ k.transpose(-1, -2)) * scale.reshape(1, H, 1, 1)
    if has_bias:
        att = att + rr[None]
    if nW:
        att = (att.reshape(B // nW, nW, H, S, S) + mask.reshape(1, nW, 1, S, S)).reshape(B, H, S, S)
    yr = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, S, H * D)
    yr.backward(gy.float())
    # fp32 runs with fp16 operands; the probabilities pass through the operand dtype once more
    cmp = torch.float16 if dtype == torch.float32 else dtype
    assert_close(y, yr, cmp, what="out", scale=4 if dtype == torch.float32 else 2)
    assert_close(qkv.grad, qr.grad, cmp, what="dqkv", scale=4)
    if has_bias:
        assert_close(rel.grad, rr.grad, cmp, what="dbias", scale=4)


@pytest.mark.parametrize("bn_mode", [0, 2, 3])
@pytest.mark.parametrize("shape", [(3, 12, 16, 64, 128), (2, 14, 14, 128, 128)])
def test_strided_data_gradient_reduces_bn_backward_sums_in_its_parity_classes(shape, bn_mode):
    """3x3 stride-2 data gradient = four parity-class launches that together write every input pixel once: each leaves its
    BN-backward partial sums in one shared slab (same stored gradient as without the fusion, sums equal to the stand-alone
    reducer's on that gradient)."""
    N, H, W, Cin, Cout = shape
    C = hip_ops.C()
    dt = torch.bfloat16
    torch.manual_seed(21)
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV) / (Cin * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, H // 2, W // 2, Cout, device=DEV).to(dt)
    bn_x = (torch.randn(N, H, W, Cin, device=DEV) * 1.5 + 0.3).to(dt)
    coef = torch.stack([torch.randn(Cin, device=DEV) * 0.1, torch.rand(Cin, device=DEV) + 0.5,
                        torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.2]).contiguous()
    z = None
    if bn_mode == 3:
        zz = torch.randn(N, H, W, Cin, device=DEV)
        bits = (zz.view(-1, 8) > 0).to(torch.uint8) << torch.arange(8, device=DEV, dtype=torch.uint8)
        z = bits.sum(1).to(torch.uint8).contiguous()
    hip_ops._pending_bnred.clear()
    old = hip_ops._FUSE_BN_BWD_STRIDED
    hip_ops._FUSE_BN_BWD_STRIDED = True          # (off by default: measured a small loss on ResNet-50)
    try:
        dx = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (2, 2), (1, 1), bnred=(bn_x, z, coef, bn_mode))
    finally:
        hip_ops._FUSE_BN_BWD_STRIDED = old
    ent = hip_ops._pending_bnred.pop(dx.data_ptr(), None)
    assert ent is not None, "the strided launches did not take the fused reduction"
    ref = hip_ops.conv_dgrad(dy, w, (N, H, W, Cin), (2, 2), (1, 1))
    assert torch.equal(dx, ref)
    M = N * H * W
    nblk = C.bn_partial_rows(M, Cin)
    part = torch.empty(nblk, 2, Cin, device=DEV)
    C.bn_bwd_reduce(dx.data_ptr(), bn_x.data_ptr(), z.data_ptr() if z is not None else 0, M, Cin, hip_ops._dt(dx),
                    coef[0].data_ptr(), coef[1].data_ptr(), bn_mode, coef[2].data_ptr(), coef[3].data_ptr(),
                    part.data_ptr(), nblk, torch.cuda.current_stream().cuda_stream)
    got, want = ent[0].sum(0), part.sum(0)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-4 * scale, (got - want).abs().max().item() / scale


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(6, 64, 3, 32), (3, 50, 12, 64), (2, 33, 2, 128), (5, 49, 4, 24)])
def test_cosine_attention_prepass(shape, dtype):
    """q <- mult[h] * q / |q|, k <- k / |k|, v unchanged (Swin V2), forward and backward: the one-lane-per-chunk kernel
    (D = 32 / 64 / 128) and the thread-per-vector fallback (D = 24) against PyTorch in fp32."""
    B, S, H, D = shape
    torch.manual_seed(9)
    qkv = torch.randn(B, S, 3 * H * D, device=DEV).to(dtype).requires_grad_(True)
    mult = (torch.rand(H, device=DEV) * 4 + 1).requires_grad_(True)
    y = F.qk_normalize(qkv, mult, H)
    gy = torch.randn(y.shape, device=DEV).to(dtype)
    y.backward(gy)
    qr = qkv.detach().float().requires_grad_(True)
    mr = mult.detach().clone().requires_grad_(True)
    q, k, v = qr.reshape(B, S, 3, H, D).unbind(2)
    qn = mr.reshape(1, 1, H, 1) * q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    kn = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    yr = torch.stack([qn, kn, v], 2).reshape(B, S, 3 * H * D)
    yr.backward(gy.float())
    assert_close(y, yr, dtype, what="out", scale=2)
    assert_close(qkv.grad, qr.grad, dtype, what="dqkv", scale=4)
    assert torch.allclose(mult.grad, mr.grad, rtol=3e-2, atol=3e-2 * mr.grad.abs().max().item())

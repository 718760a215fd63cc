"""csrc/kernels/gemm_lean.hip: the lean MFMA GEMM that takes the single-tap launches (1x1 convolutions without
padding, their stride-1 data gradients, Linear).  It accumulates in the same order as the general implicit-GEMM
kernel, so the two must agree BIT FOR BIT (outputs, statistics slabs, fused accumulate); both are also checked
against a plain PyTorch fp32 reference of the same op."""
import pytest
import torch

from _gpu_util import assert_close

pytestmark = pytest.mark.gpu

# NB, H, Cin, Cout, stride: tails in M (243 rows), odd / single K-tile counts, 64- and 128-wide channel tiles, strided
CASES = [
    (4, 14, 256, 1024, 1),
    (3, 9, 64, 64, 1),
    (2, 15, 128, 128, 2),
    (5, 7, 96, 192, 1),
    (2, 8, 64, 128, 1),
    (2, 6, 320, 256, 1),
    (3, 13, 512, 64, 2),
]


def _ops():
    from distributed_pytorch_amd.ops import hip_ops
    return hip_ops, hip_ops.C()


def _slab_views(stats):
    """(sum row, M2 row, counts) of a statistics slab: the counts sit right behind the [parts][2][C] view."""
    parts, _, c = stats.shape
    flat = torch.as_strided(stats, (parts * 2 * c + parts,), (1,), stats.storage_offset())
    return stats[:, 0], stats[:, 1], flat[parts * 2 * c:]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_forward_with_statistics_equals_general_kernel_and_reference(case, dtype):
    hip_ops, C = _ops()
    NB, H, Cin, Cout, s = case
    torch.manual_seed(sum(case))
    x = torch.randn(NB, H, H, Cin, device="cuda").to(dtype)
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * Cin ** -0.5).contiguous(memory_format=torch.channels_last)
    outs = {}
    try:
        for mode in (0, 1, 2, -1):
            C.igemm_set_lean(mode)
            y, stats = hip_ops.conv2d(x, w, None, (s, s), (0, 0), False, None, True, False)[:2]
            torch.cuda.synchronize()
            outs[mode] = (y, [t.clone() for t in _slab_views(stats)])
    finally:
        C.igemm_set_lean(-1)
    y0, st0 = outs[0]
    for mode in (1, 2, -1):
        y, st = outs[mode]
        assert torch.equal(y, y0), "lean mode %d output differs from the general kernel" % mode
        assert torch.equal(st[2], st0[2])
        # same values, possibly summed in another order (the two kernels may pick different channel-tile widths)
        assert_close(st[0], st0[0], torch.float32, "sum row")
        assert_close(st[1], st0[1], torch.float32, "M2 row", scale=4)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), stride=s).permute(0, 2, 3, 1)
    assert_close(y0, ref, dtype, "conv")
    # statistics of the stored (rounded) values
    ssum, m2, cnt = st0
    yf = y0.float().reshape(-1, Cout)
    assert cnt.sum().item() == yf.shape[0]
    assert_close(ssum.sum(0), yf.sum(0), torch.float32, "sum", scale=4)
    mean = yf.mean(0)
    tot_m2 = ((yf - mean) ** 2).sum(0)
    part_mean = ssum / cnt[:, None]
    chan = m2.sum(0) + (cnt[:, None] * (part_mean - mean) ** 2).sum(0)
    assert_close(chan, tot_m2, torch.float32, "M2", scale=8)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [c for c in CASES if c[4] == 1])
def test_data_gradient_with_masked_accumulate_equals_general_kernel(case, dtype):
    hip_ops, C = _ops()
    NB, H, Cin, Cout, _ = case
    torch.manual_seed(sum(case) + 1)
    dy = torch.randn(NB, H, H, Cout, device="cuda").to(dtype)
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * Cout ** -0.5).contiguous(memory_format=torch.channels_last)
    other = torch.randn(NB, H, H, Cin, device="cuda").to(dtype)
    mask = torch.randint(0, 256, (NB * H * H * Cin // 8,), dtype=torch.uint8, device="cuda")
    res = {}
    try:
        for mode in (0, -1):
            C.igemm_set_lean(mode)
            plain = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0))
            acc = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0), accum=other.clone())
            accm = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0), accum=other.clone(), accum_mask=mask)
            torch.cuda.synchronize()
            res[mode] = (plain, acc, accm)
    finally:
        C.igemm_set_lean(-1)
    for a, b in zip(res[0], res[-1]):
        assert torch.equal(a, b)
    ref = torch.einsum("nhwk,kc->nhwc", dy.float(), w.to(dtype).float().reshape(Cout, Cin))
    assert_close(res[-1][0], ref, dtype, "dgrad")
    assert_close(res[-1][1], ref + other.float(), dtype, "dgrad + accumulate")
    assert_close(res[-1][2], ref + hip_ops.apply_bit_mask(other, mask).float(), dtype, "dgrad + masked accumulate")


def test_linear_layer_runs_the_lean_kernel_and_matches():
    hip_ops, C = _ops()
    torch.manual_seed(3)
    x = torch.randn(300, 512, device="cuda").bfloat16()
    w = torch.randn(1024, 512, device="cuda") * 512 ** -0.5
    try:
        C.igemm_set_lean(0)
        y0 = hip_ops.linear(x, w, None, False, None, torch.bfloat16)
        C.igemm_set_lean(-1)
        y1 = hip_ops.linear(x, w, None, False, None, torch.bfloat16)
    finally:
        C.igemm_set_lean(-1)
    assert torch.equal(y0, y1)
    assert_close(y1, x.float() @ w.bfloat16().float().t(), torch.bfloat16, "linear")


@pytest.mark.parametrize("mode", [0, 2, 3, 6])
@pytest.mark.parametrize("with_accum", [False, True])
@pytest.mark.parametrize("case", [(4, 14, 256, 1024, 1), (3, 9, 64, 64, 1), (5, 7, 96, 192, 1), (2, 6, 320, 256, 1)])
def test_data_gradient_with_fused_bn_backward_reduction(case, with_accum, mode):
    """kIgemmBnBwd in the lean epilogue: the slab holds (sum dz_m, sum dz_m * xhat) of the stored gradient, masked by
    the BatchNorm's ReLU (bit mask / recomputed from the BN input / none); same values as the general kernel."""
    hip_ops, C = _ops()
    NB, H, Cin, Cout, _ = case
    dtype = torch.bfloat16
    torch.manual_seed(sum(case) + 7 * mode + with_accum)
    dy = torch.randn(NB, H, H, Cout, device="cuda").to(dtype)
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * Cout ** -0.5).contiguous(memory_format=torch.channels_last)
    bn_x = (torch.randn(NB, H, H, Cin, device="cuda") * 2 + 0.5).to(dtype)
    mean, invstd = torch.randn(Cin, device="cuda") * 0.3 + 0.5, torch.rand(Cin, device="cuda") + 0.5
    scale, shift = torch.randn(Cin, device="cuda") * 3, torch.randn(Cin, device="cuda") * 0.2
    coef = torch.stack([mean, invstd, scale, shift]).contiguous()
    bits = torch.randint(0, 256, (NB * H * H * Cin // 8,), dtype=torch.uint8, device="cuda")
    other = torch.randn(NB, H, H, Cin, device="cuda").to(dtype) if with_accum else None
    omask = torch.randint(0, 256, (NB * H * H * Cin // 8,), dtype=torch.uint8, device="cuda") if with_accum else None
    res = {}
    try:
        for lean in (0, -1):
            C.igemm_set_lean(lean)
            dx = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0),
                                    accum=other.clone() if with_accum else None,
                                    bnred=(bn_x, bits if mode == 3 else None, coef, mode), accum_mask=omask)
            slab = hip_ops._pending_bnred.pop(dx.data_ptr())[0]
            torch.cuda.synchronize()
            res[lean] = (dx, slab.clone())
    finally:
        C.igemm_set_lean(-1)
    assert torch.equal(res[0][0], res[-1][0])
    assert_close(res[-1][1].sum(0), res[0][1].sum(0), torch.float32, "slab vs general kernel", scale=4)
    # reference from the stored gradient
    d = res[-1][0].float().reshape(-1, Cin)
    xf = bn_x.float().reshape(-1, Cin)
    if mode == 3:
        keep = ((bits.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1).reshape(-1, Cin).bool()
    elif mode == 6:                      # ReLU6: 0 < y < 6
        yv = (scale * xf + shift).to(dtype).float()
        keep = (yv > 0) & (yv < 6)
    elif mode == 2:
        keep = (scale * xf + shift).to(dtype).float() > 0
    else:
        keep = torch.ones_like(d, dtype=torch.bool)
    dm = torch.where(keep, d, torch.zeros_like(d))
    ref = torch.stack([dm.sum(0), (dm * ((xf - mean) * invstd)).sum(0)])
    assert_close(res[-1][1].sum(0), ref, torch.float32, "slab vs reference", scale=8)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(300, 512, 1024), (197 * 3, 768, 3072), (64, 64, 64), (130, 3072, 768)])
def test_linear_with_bias_starts_the_accumulators_at_the_bias(shape, relu):
    """BIAS variant: same fp32 sum as the general kernel up to the order of the additions, one rounding."""
    hip_ops, C = _ops()
    M, K, N = shape
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    try:
        C.igemm_set_lean(0)
        y0 = hip_ops.linear(x, w, b, relu, None, torch.bfloat16)
        C.igemm_set_lean(-1)
        y1 = hip_ops.linear(x, w, b, relu, None, torch.bfloat16)
    finally:
        C.igemm_set_lean(-1)
    ref = x.float() @ w.bfloat16().float().t() + b
    if relu:
        ref = ref.clamp_min(0)
    assert_close(y1, ref, torch.bfloat16, "linear + bias")
    assert_close(y1, y0.float(), torch.bfloat16, "lean vs general kernel", scale=0.5)


@pytest.mark.parametrize("with_bn", [False, True])
@pytest.mark.parametrize("case", [(3, 14, 256, 64, 2), (2, 9, 128, 96, 2), (2, 8, 64, 64, 4)])
def test_compact_strided_accumulate_equals_scattered_gradient(case, with_bn):
    """kIgemmRStrided: the gradient of a stride-s 1x1 branch is added from its compact [N, H/s, W/s, C] form; same
    result as adding the zero-filled, scattered full-size tensor (odd extents: the compact grid is ceil(H / s))."""
    hip_ops, C = _ops()
    NB, H, Cin, Cout, s = case
    dtype = torch.bfloat16
    torch.manual_seed(sum(case) + with_bn)
    dy = torch.randn(NB, H, H, Cout, device="cuda").to(dtype)
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * Cout ** -0.5).contiguous(memory_format=torch.channels_last)
    Hc = (H + s - 1) // s
    compact = torch.randn(NB, Hc, Hc, Cin, device="cuda").to(dtype)
    full = torch.zeros(NB, H, H, Cin, device="cuda", dtype=dtype)
    full[:, ::s, ::s] = compact
    bn_x = torch.randn(NB, H, H, Cin, device="cuda").to(dtype)
    coef = torch.rand(4, Cin, device="cuda") + 0.5
    bits = torch.randint(0, 256, (NB * H * H * Cin // 8,), dtype=torch.uint8, device="cuda")
    bnred = (bn_x, bits, coef, 3) if with_bn else None
    a = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0), accum=full.clone(), bnred=bnred)
    sa = hip_ops._pending_bnred.pop(a.data_ptr())[0].clone() if with_bn else None
    b = hip_ops.conv_dgrad(dy, w, (NB, H, H, Cin), (1, 1), (0, 0), accum=compact, bnred=bnred, accum_stride=(s, s))
    sb = hip_ops._pending_bnred.pop(b.data_ptr())[0].clone() if with_bn else None
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    if with_bn:
        assert torch.equal(sa, sb)
    ref = torch.einsum("nhwk,kc->nhwc", dy.float(), w.to(dtype).float().reshape(Cout, Cin)) + full.float()
    # (the epilogue adds R to the ROUNDED product and rounds again: up to one more ulp where the two cancel)
    assert_close(b, ref, dtype, "dgrad + compact accumulate", scale=2)

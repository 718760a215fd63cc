"""BatchNorm backward folded into the producing 1x1 convolution (csrc/kernels/bnfold.hip, hip_ops._conv_bn_fold_backward):
a stack of bottleneck blocks trained one step with the fold on and off, and against a plain PyTorch fp32 reference of
the same network fed the same rounded parameters."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd import amp, models  # noqa: E402
from distributed_pytorch_amd.models.resnet import Bottleneck  # noqa: E402
from distributed_pytorch_amd.ops import functional as F  # noqa: E402
from distributed_pytorch_amd.ops import hip_ops  # noqa: E402
from distributed_pytorch_amd.optim import FusedSGD  # noqa: E402
from _gpu_util import assert_close  # noqa: E402

DEV = "cuda"


from distributed_pytorch_amd.models.resnet import conv1x1  # noqa: E402
from distributed_pytorch_amd.nn import BatchNorm2d  # noqa: E402


class Stack(torch.nn.Module):
    """Three bottlenecks; ``stride`` > 0: the first one changes width (and resolution) through a downsample branch."""

    def __init__(self, planes=64, blocks=3, stride=0):
        super().__init__()
        self.cin = planes * 4
        first = Bottleneck(planes * 4, planes)
        if stride:
            self.cin = planes * 2
            down = torch.nn.Sequential(conv1x1(self.cin, planes * 4, stride), BatchNorm2d(planes * 4))
            first = Bottleneck(self.cin, planes, stride, down)
        self.blocks = torch.nn.ModuleList([first] + [Bottleneck(planes * 4, planes) for _ in range(blocks - 1)])

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


def _run(fold, planes, shape, seed=0, stride=0, fwd=False):
    torch.manual_seed(seed)
    net = Stack(planes, stride=stride).to(DEV).train()
    for m in net.modules():           # non-trivial BN parameters
        if hasattr(m, "running_mean") and m.weight is not None and m.weight.dim() == 1:
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.0)
    net, opt = amp.initialize(net, opt, opt_level="O2", dtype="bf16")
    N, H, W = shape
    x = (torch.randn(N, H, W, net.cin, device=DEV).relu() + 0.1).to(torch.bfloat16).requires_grad_(True)
    st = max(stride, 1)
    gy = torch.randn(N, (H + st - 1) // st, (W + st - 1) // st, planes * 4, device=DEV).to(torch.bfloat16)
    old, old_mb, old_fwd = hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB, hip_ops._BN_FWD_FOLD
    hip_ops._BN_FOLD = fold
    hip_ops._BN_FWD_FOLD = fwd              # statistics from the Gram matrix, BN + residual + ReLU in the GEMM epilogue
    hip_ops._BN_FOLD_MIN_MB = 0.0           # (the default leaves tensors this small to the ordinary backward)
    hip_ops.bn_fold_counts["folded"] = 0
    hip_ops.bn_fwd_fold_counts["fused"] = 0
    try:
        opt.zero_grad()
        y = net(x)
        y.backward(gy)
    finally:
        hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB, hip_ops._BN_FWD_FOLD = old, old_mb, old_fwd
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}
    return net, x, gy, y.detach().float(), x.grad.detach().float(), grads, hip_ops.bn_fold_counts["folded"]


def _reference(net, x, gy):
    """fp32 PyTorch reference of the stack with the SAME (bf16-rounded) weights and input: (y, input gradient NCHW,
    parameter gradients by name)."""
    params = {n: p for n, p in net.named_parameters()}
    leaves = {n: p.detach().float().requires_grad_(True) for n, p in params.items()}
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    t = xr
    for i, blk in enumerate(net.blocks):
        def cbr(t, cname, bname, conv, bn, relu, res=None):
            w = leaves[cname + ".weight"].to(torch.bfloat16).float() + (leaves[cname + ".weight"] - leaves[cname + ".weight"].detach())
            z = TF.conv2d(t, w, None, conv.stride, conv.padding)
            o = TF.batch_norm(z, None, None, leaves[bname + ".weight"], leaves[bname + ".bias"], True, 0.0, bn.eps)
            if res is not None:
                o = o + res
            return TF.relu(o) if relu else o
        pre = "blocks.%d." % i
        idt = t
        if blk.downsample is not None:
            idt = cbr(t, pre + "downsample.0", pre + "downsample.1", blk.downsample[0], blk.downsample[1], False)
        o = cbr(t, pre + "conv1", pre + "bn1", blk.conv1, blk.bn1, True)
        o = cbr(o, pre + "conv2", pre + "bn2", blk.conv2, blk.bn2, True)
        t = cbr(o, pre + "conv3", pre + "bn3", blk.conv3, blk.bn3, True, idt)
    t.backward(gy.float().permute(0, 3, 1, 2))
    return t.detach(), xr.grad, {n: l.grad for n, l in leaves.items()}


@pytest.mark.parametrize("planes,shape,stride", [(64, (4, 14, 14), 0), (128, (3, 9, 11), 0), (64, (4, 14, 14), 1),
                                                 (64, (3, 14, 18), 2)])
def test_bn_fold_matches_the_unfolded_backward_and_an_fp32_reference(planes, shape, stride):
    net, x, gy, y1, dx1, g1, folded = _run(True, planes, shape, stride=stride)
    _, _, _, y0, dx0, g0, none = _run(False, planes, shape, stride=stride)
    # blocks 0 and 1: their bn3 receives a fused, pre-masked gradient; with a downsample branch its BatchNorm folds too
    assert folded == (3 if stride else 2) and none == 0
    assert torch.equal(y1, y0)
    _, xgrad, rgrads = _reference(net, x, gy)
    # The network runs bf16 activations through three blocks (ReLU masks flip, roundings accumulate): per-element bounds
    # against an fp32 reference are meaningless here.  Tensor-level: the fold differs from the ordinary backward by
    # roundings only (it never rounds dz to bf16), and it is no further from the fp32 reference than the ordinary one.
    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    assert torch.isfinite(dx1).all()
    assert rel(dx1, dx0) < 2e-2
    e1, e0 = rel(dx1.permute(0, 3, 1, 2), xgrad), rel(dx0.permute(0, 3, 1, 2), xgrad)
    assert e1 <= 1.25 * e0 + 1e-3, ("input gradient", e1, e0)
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
        assert rel(g1[n], g0[n]) < 2e-2, (n, rel(g1[n], g0[n]))
        ref = rgrads[n].reshape(g1[n].shape)
        e1, e0 = rel(g1[n], ref), rel(g0[n], ref)
        assert e1 <= 1.25 * e0 + 1e-3, (n, e1, e0)


@pytest.mark.parametrize("planes,shape,stride", [(64, (4, 14, 14), 0), (128, (3, 9, 11), 0), (64, (4, 14, 14), 1),
                                                 (64, (3, 14, 18), 2)])
def test_forward_fold_statistics_from_the_gram_matrix(planes, shape, stride):
    """The bottleneck tail as ONE GEMM (hip_ops._ConvBnResReluFn): BatchNorm statistics of conv3's output from a^T a and
    colsum(a), scale folded into the weights, shift + identity + ReLU (+ bit mask) in the epilogue.  Same network with
    the ordinary conv -> BatchNorm tail: outputs, running statistics and every gradient agree to bf16 rounding."""
    net1, x, gy, y1, dx1, g1, folded = _run(True, planes, shape, stride=stride, fwd=True)
    fused = hip_ops.bn_fwd_fold_counts["fused"]
    net0, _, _, y0, dx0, g0, _ = _run(True, planes, shape, stride=stride, fwd=False)
    assert fused == 3                           # identity blocks (one operand + R) and the downsample block (two operands)
    assert folded == (4 if stride else 3)       # every bn3 (the last block reduces its own sums) + the downsample BN

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    assert torch.isfinite(y1).all() and torch.isfinite(dx1).all()
    assert rel(y1, y0) < 1e-2, rel(y1, y0)
    assert ((y1 > 0) != (y0 > 0)).float().mean().item() < 2e-2
    # Gradients: a ReLU that flips where the two outputs straddle zero changes the sums over random upstream gradients by
    # far more than a rounding, so the two runs are compared through their distance to the fp32 reference (same rounded
    # parameters): the fused tail must not be further from it than the ordinary one.
    yr, xgrad, rgrads = _reference(net1, x, gy)
    e1, e0 = rel(y1.permute(0, 3, 1, 2), yr), rel(y0.permute(0, 3, 1, 2), yr)
    assert e1 <= 1.25 * e0 + 1e-3, ("output", e1, e0)
    e1, e0 = rel(dx1.permute(0, 3, 1, 2), xgrad), rel(dx0.permute(0, 3, 1, 2), xgrad)
    assert e1 <= 1.3 * e0 + 1e-3, ("input gradient", e1, e0)
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
        ref = rgrads[n].reshape(g1[n].shape)
        e1, e0 = rel(g1[n], ref), rel(g0[n], ref)
        assert e1 <= 1.5 * e0 + 5e-3, (n, e1, e0)          # (e0 itself is ~0.1 on the deepest block: mask-flip noise)
    b1 = dict(net1.named_buffers())
    for n, b0 in net0.named_buffers():
        if b0.dtype.is_floating_point:
            assert rel(b1[n].float(), b0.float()) < 5e-3, (n, rel(b1[n].float(), b0.float()))
        else:
            assert torch.equal(b1[n], b0), n


def test_forward_fold_keeps_no_block_output_alive_between_steps():
    """Nothing of one step may survive into the next without the cyclic garbage collector: the BatchNorm descriptor that
    rides on a block output as an attribute must not point back at that tensor (it did: 41 -> 133 GB at batch 1024)."""
    import gc

    torch.manual_seed(0)
    net = Stack(64, stride=2).to(DEV).train()
    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.0)
    net, opt = amp.initialize(net, opt, opt_level="O2", dtype="bf16")
    x = (torch.randn(4, 14, 18, net.cin, device=DEV).relu() + 0.1).to(torch.bfloat16)
    old = hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB, hip_ops._BN_FWD_FOLD
    hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB, hip_ops._BN_FWD_FOLD = True, 0.0, True
    hip_ops.bn_fwd_fold_counts["fused"] = 0
    gc.collect()
    gc.disable()
    try:
        used = []
        for _ in range(5):
            opt.zero_grad()
            y = net(x)
            y.float().square().mean().backward()
            del y
            torch.cuda.synchronize()
            used.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
        hip_ops._BN_FOLD, hip_ops._BN_FOLD_MIN_MB, hip_ops._BN_FWD_FOLD = old
    assert hip_ops.bn_fwd_fold_counts["fused"] == 15
    assert used[4] <= used[1] + (1 << 20), used


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,Cc", [(1000, 64), (4099, 128), (777, 256), (50000, 512), (300000, 64)])
def test_bn_apply_with_column_sums(M, Cc, dtype):
    """bn_apply_colsum: relu(scale * x + shift) bit-identical to the ordinary apply pass, column sums of the STORED values."""
    C = hip_ops.C()
    torch.manual_seed(3)
    x = (torch.randn(M, Cc, device=DEV) * 1.3 + 0.2).to(dtype)
    scale = torch.rand(Cc, device=DEV) + 0.5
    shift = torch.randn(Cc, device=DEV) * 0.3
    st = torch.cuda.current_stream().cuda_stream
    y0 = torch.empty_like(x)
    C.bn_apply(x.data_ptr(), 0, y0.data_ptr(), M, Cc, hip_ops._dt(x), scale.data_ptr(), shift.data_ptr(), 1, st, 0, 0, 0)
    n = C.bn_apply_colsum_parts(M, Cc, hip_ops._dt(x))
    assert n > 0
    y1 = torch.empty_like(x)
    part = torch.full((n, 2, Cc), float("nan"), device=DEV)
    C.bn_apply_colsum(x.data_ptr(), y1.data_ptr(), M, Cc, hip_ops._dt(x), scale.data_ptr(), shift.data_ptr(), part.data_ptr(), n, st)
    assert torch.equal(y1, y0)
    assert torch.equal(part[:, 1], torch.zeros_like(part[:, 1]))
    assert_close(part[:, 0].double().sum(0).float(), y0.double().sum(0).float(), torch.float32, scale=20, what="column sums")


@pytest.mark.parametrize("shape,s", [((3, 14, 18, 64), 2), ((2, 7, 9, 128), 2), ((2, 8, 8, 32), 4)])
def test_subsample_gather(shape, s):
    x = torch.randn(*shape, device=DEV).to(torch.bfloat16)
    assert torch.equal(hip_ops._subsample(x, s), x[:, ::s, ::s, :].contiguous())

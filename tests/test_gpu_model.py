"""End-to-end parity: the same model / weights / batch on the CPU reference backend (plain PyTorch
ops) and on the HIP kernel backend must give the same logits, loss and parameter gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, x, t):
    from distributed_pytorch_amd.ops import functional as F

    model.train()
    out = model(x)
    loss, rec = F.loss_and_metrics(out, t)
    model.zero_grad()
    loss.backward()
    return out.detach().float().cpu(), float(loss), {n: p.grad.detach().float().cpu()
                                                     for n, p in model.named_parameters()}


@pytest.mark.parametrize("arch,size,tol", [("resnet18", 64, 2e-3), ("resnet50", 128, 3e-3), ("alexnet", 224, 2e-3),
                                           ("vgg11_bn", 128, 3e-3), ("squeezenet1_1", 127, 2e-3), ("squeezenet1_0", 96, 2e-3),
                                           ("densenet121", 64, 3e-3), ("resnext50_32x4d", 128, 4e-3), ("mobilenet_v2", 96, 4e-3), ("mnasnet0_5", 96, 4e-3),
                                           ("googlenet", 96, 3e-3), ("regnet_x_400mf", 96, 4e-3),
                                           ("regnet_y_400mf", 96, 4e-3), ("mobilenet_v3_small", 96, 4e-3),
                                           ("mobilenet_v3_large", 64, 4e-3), ("efficientnet_b0", 64, 4e-3),
                                           ("efficientnet_v2_s", 64, 4e-3), ("inception_v3", 171, 6e-3), ("convnext_tiny", 64, 4e-3),
                                           ("shufflenet_v2_x0_5", 96, 4e-3), ("shufflenet_v2_x1_0", 96, 4e-3),
                                           ("shufflenet_v2_x2_0", 64, 4e-3),
                                           ("vit_b_32", 224, 6e-3), ("swin_t", 64, 6e-3), ("swin_v2_t", 64, 6e-3),
                                           ("maxvit_t", 224, 6e-3)])
def test_fp32_model_matches_cpu_reference(arch, size, tol):
    from distributed_pytorch_amd import amp, models

    amp.state().reset()
    torch.manual_seed(0)
    cpu = models.create_model(arch, num_classes=1000)
    for m in cpu.modules():                      # no dropout / stochastic-depth noise in the comparison
        if m.__class__.__name__ in ("Dropout", "StochasticDepth"):
            m.p = 0.0
    for n, p in cpu.named_parameters():          # ConvNeXt's 1e-6 layer scale would hide the residual branches
        if n.endswith("layer_scale"):
            p.data.fill_(0.5)
        if n == "heads.head.weight":             # ViT starts with a zero classifier: nothing would reach the encoder
            torch.nn.init.normal_(p.data, std=0.02)
        if n.endswith("relative_position_bias_table"):
            torch.nn.init.normal_(p.data, std=0.5)
    gpu = copy.deepcopy(cpu).cuda()
    x = torch.randn(4, 3, size, size)
    t = torch.randint(0, 1000, (4,))
    o1, l1, g1 = _run(cpu, x, t)
    o2, l2, g2 = _run(gpu, x.cuda(), t.cuda())
    assert abs(l1 - l2) < tol * max(1.0, abs(l1)), (l1, l2)
    assert (o1 - o2).abs().max() < tol * max(1.0, o1.abs().max())
    # per-parameter relative L2 error (tiny-batch BatchNorm backward amplifies rounding, so a max-norm
    # criterion on single entries is not meaningful)
    # (a conv bias in front of a BatchNorm has an exactly-zero gradient: floor the denominator)
    floor = 1e-4 * max(g.norm().item() / g.numel() ** 0.5 for g in g1.values())
    for n in g1:
        a, b = g1[n], g2[n]
        rel = ((a - b).norm() / max(a.norm().item(), floor * a.numel() ** 0.5)).item()
        assert rel < 10 * tol, (n, rel)


@pytest.mark.parametrize("arch,size", [("resnet50", 96), ("densenet121", 64), ("googlenet", 96), ("mobilenet_v3_small", 96),
                                       ("efficientnet_b0", 64), ("regnet_y_400mf", 96), ("inception_v3", 171),
                                       ("convnext_tiny", 64), ("vgg11", 64)])
def test_eval_mode_matches_cpu_reference(arch, size):
    """model.eval() (validate(): BN from running statistics, no dropout / stochastic depth) after one training step
    that moved the running statistics: same logits on the CPU reference backend and on the HIP backend."""
    from distributed_pytorch_amd import amp, models

    amp.state().reset()
    torch.manual_seed(0)
    cpu = models.create_model(arch, num_classes=100)
    for m in cpu.modules():
        if m.__class__.__name__ in ("Dropout", "StochasticDepth"):
            m.p = 0.0
    gpu = copy.deepcopy(cpu).cuda()
    x = torch.randn(4, 3, size, size)
    with torch.no_grad():
        cpu.train()(x)
        gpu.train()(x.cuda())                    # running_mean / running_var / num_batches_tracked now differ from init
        o1 = cpu.eval()(x)
        o2 = gpu.eval()(x.cuda()).float().cpu()
    for (n, b1), (_, b2) in zip(cpu.named_buffers(), gpu.named_buffers()):
        assert torch.allclose(b1.float(), b2.float().cpu(), rtol=2e-3, atol=2e-4), n
    assert o2.shape == (4, 100) and torch.isfinite(o2).all()
    assert (o1 - o2).abs().max() < 4e-3 * max(1.0, o1.abs().max())


def test_bf16_training_reduces_loss_and_tracks_fp32():
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD

    torch.manual_seed(0)
    x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8, device="cuda")
    t = torch.randint(0, 10, (16,), device="cuda")
    curves = {}
    for level in ("O0", "O2"):
        torch.manual_seed(1)
        amp.state().reset()
        model = models.create_model("resnet18", num_classes=16).cuda().train()
        opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        model, opt = amp.initialize(model, opt, opt_level=level, dtype="bf16")
        losses = []
        for _ in range(12):
            loss, _ = F.loss_and_metrics(model(x), t)
            opt.zero_grad()
            with amp.scale_loss(loss, opt) as s:
                s.backward()
            opt.step()
            losses.append(float(loss))
        curves[level] = losses
    amp.state().reset()
    assert curves["O0"][-1] < 0.5 * curves["O0"][0], curves["O0"]
    assert curves["O2"][-1] < 0.5 * curves["O2"][0], curves["O2"]
    assert abs(curves["O0"][0] - curves["O2"][0]) < 0.1


def test_fp16_dynamic_loss_scaling_trains_and_recovers_from_overflow():
    """fp16 O2 with Apex-style dynamic scaling: an injected inf gradient skips the step and halves the scale
    on the device (no host sync in the step), then training continues."""
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD

    torch.manual_seed(0)
    amp.state().reset()
    model = models.create_model("resnet18", num_classes=16).cuda().train()
    opt = FusedSGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    model, opt = amp.initialize(model, opt, opt_level="O2", dtype="fp16", loss_scale="dynamic")
    scaler = amp.state().scaler
    assert scaler.loss_scale() == 2.0 ** 16
    x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8, device="cuda")
    t = torch.randint(0, 10, (16,), device="cuda")
    losses = []
    for it in range(14):
        out = model(x)
        assert out.dtype == torch.float32
        loss, _ = F.loss_and_metrics(out, t)
        opt.zero_grad()
        with amp.scale_loss(loss, opt) as s:
            s.backward()
            if it == 3:
                model.fc.weight.grad[0, 0] = float("inf")          # poison one gradient element
        before = model.fc.weight.detach().clone()
        opt.step()
        if it == 3:
            assert torch.equal(before, model.fc.weight.detach())   # whole step skipped
        losses.append(float(loss))
    assert scaler.loss_scale() < 2.0 ** 16                          # backed off at least once
    assert losses[-1] < 0.6 * losses[0], losses
    amp.state().reset()


@pytest.mark.parametrize("act", [True, "silu", 6])
def test_syncbn_gpu_path_equals_plain_bn_on_one_rank(act):
    """The SyncBN branch (packed [sum, sumsq, n] all-reduce, device-side count, separate coefficient
    kernel) must reproduce plain BN when the group has a single rank."""
    from distributed_pytorch_amd.ops import functional as F, hip_ops

    torch.manual_seed(3)
    C = 64
    xs = [(torch.randn(4, 6, 6, C, device="cuda") * 2 + 1).to(torch.bfloat16).requires_grad_(True) for _ in range(2)]
    ws = [(torch.rand(C, device="cuda") + 0.5).requires_grad_(True) for _ in range(2)]
    bs = [torch.randn(C, device="cuda").requires_grad_(True) for _ in range(2)]
    with torch.no_grad():
        xs[1].copy_(xs[0]); ws[1].copy_(ws[0]); bs[1].copy_(bs[0])
    outs, stats = [], []
    for i, group in enumerate((None, "world")):
        hip_ops._FORCE_SYNC_PATH = group is not None
        try:
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
            nbt = torch.zeros((), dtype=torch.long, device="cuda")
            y = F.batch_norm(xs[i], ws[i], bs[i], rm, rv, nbt, True, 0.1, 1e-5, relu=act, group=group)
            y.float().square().sum().backward()
            outs.append(y.detach().float())
            stats.append((rm, rv))
        finally:
            hip_ops._FORCE_SYNC_PATH = False
    assert torch.allclose(outs[0], outs[1], atol=1e-2)
    assert torch.allclose(stats[0][0], stats[1][0], atol=1e-5) and torch.allclose(stats[0][1], stats[1][1], atol=1e-4)
    for a, b in ((xs[0].grad, xs[1].grad), (ws[0].grad, ws[1].grad), (bs[0].grad, bs[1].grad)):
        assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(a.float().abs().max()))


@pytest.mark.parametrize("arch", ["resnet18", "mobilenet_v3_small", "resnet18:prefetched"])
def test_hip_graph_step_matches_eager_training(arch):
    """--hip-graph: the captured + replayed step leaves exactly the eager trajectory (same kernels, same order),
    and the learning rate can still change between replays."""
    import copy

    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.utils.graph import GraphedTrainStep

    # ":prefetched": the batches arrive as the prefetcher delivers them (space-to-depth layout on zero-padded rows, a
    # strided view): the graph's static input has to keep that layout
    arch, _, feed = arch.partition(":")
    amp.state().reset()
    torch.manual_seed(0)
    # mobilenet_v3_small: squeeze-excitation weights get their gradients through autograd's AccumulateGrad (not a
    # gradient sink) -- warm-up and capture must share one stream for that to be capturable
    base = models.create_model(arch, num_classes=10).cuda().train()
    for m in base.modules():
        if m.__class__.__name__ == "Dropout":
            m.p = 0.0                              # the two runs would draw different masks
    xs = [torch.randint(0, 256, (8, 64, 64, 3), dtype=torch.uint8, device="cuda") for _ in range(8)]
    ts = [torch.randint(0, 10, (8,), device="cuda") for _ in range(8)]
    results = []
    for use_graph in (False, True):
        amp.state().reset()
        model = copy.deepcopy(base)
        opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
        stepper = GraphedTrainStep(model, opt, warmup=2, enabled=use_graph)
        losses = []
        for i, (x, t) in enumerate(zip(xs, ts)):
            if i == 5:
                opt.param_groups[0]["lr"] = 0.01
            if feed == "prefetched":
                from distributed_pytorch_amd.ops import functional as F
                x = F.to_s2d(x, torch.bfloat16)
            loss, _ = stepper(x, t)
            losses.append(float(loss))
        if use_graph:
            assert stepper.replays == 6, "capture did not happen (%d replays)" % stepper.replays
        results.append((losses, [p.detach().float().clone() for p in model.parameters()]))
    (l0, p0), (l1, p1) = results
    assert l0 == pytest.approx(l1, rel=1e-5, abs=1e-6), (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_lazy_residual_gradient_equals_materialised(arch):
    """Residual BN + ReLU backward hands the UNMASKED gradient + its ReLU bit mask to the block's first convolution,
    whose data-gradient epilogue applies the mask (no masked copy is written).  Same gradients as the path that
    writes the masked copy, and the path is really taken."""
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import hip_ops

    amp.state().reset()
    torch.manual_seed(1)
    base = models.create_model(arch, num_classes=100).cuda()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    t = torch.randint(0, 100, (4,), device="cuda")
    res = []
    try:
        for lazy in (True, False):
            hip_ops._LAZY_DRES = lazy
            seen = []
            orig = hip_ops._igemm

            def spy(*a, **k):
                seen.append(k.get("accum_mask") is not None)
                return orig(*a, **k)

            hip_ops._igemm = spy
            try:
                res.append(_run(copy.deepcopy(base), x, t))
            finally:
                hip_ops._igemm = orig
            assert any(seen) == lazy
            assert not hip_ops._pending_mask          # every mask was consumed by its convolution
    finally:
        hip_ops._LAZY_DRES = True
    (o1, l1, g1), (o2, l2, g2) = res
    assert torch.equal(o1, o2) and l1 == l2
    for n in g1:
        assert torch.allclose(g1[n], g2[n], rtol=1e-5, atol=1e-7), n


def test_compact_downsample_gradient_equals_scattered():
    """The strided 1x1 downsample branch of a bottleneck returns its input gradient in compact form (a NaN placeholder
    travels through autograd); the block's first convolution adds it in its data-gradient epilogue.  Bit-identical
    to the zero-filled, scattered tensor, the path is really taken (3 downsample branches in ResNet-50), and nothing
    is left pending."""
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import hip_ops
    from distributed_pytorch_amd.optim import FusedSGD

    torch.manual_seed(2)
    base = models.create_model("resnet50", num_classes=100).cuda()
    x = torch.randint(0, 256, (8, 96, 96, 3), dtype=torch.uint8, device="cuda")
    t = torch.randint(0, 100, (8,), device="cuda")
    res = []
    try:
        for compact in (True, False):
            hip_ops._COMPACT_DRES = compact
            amp.state().reset()
            model = copy.deepcopy(base)
            opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9)
            model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
            seen = []
            orig = hip_ops._igemm

            def spy(*a, **k):
                seen.append(k.get("rstride") is not None)
                return orig(*a, **k)

            hip_ops._igemm = spy
            try:
                res.append(_run(model, x, t))
            finally:
                hip_ops._igemm = orig
            assert sum(seen) == (3 if compact else 0)
            assert not hip_ops._pending_compact
    finally:
        hip_ops._COMPACT_DRES = True
    (o1, l1, g1), (o2, l2, g2) = res
    assert torch.equal(o1, o2) and l1 == l2 and torch.isfinite(torch.tensor(l1))
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


@pytest.mark.parametrize("arch", ["regnet_y_128gf", "efficientnet_b7", "convnext_large", "densenet201", "resnext101_64x4d",
                                  "vgg19_bn", "wide_resnet101_2", "regnet_x_32gf", "vit_h_14", "vit_l_16", "swin_b",
                                  "swin_v2_b", "shufflenet_v2_x1_5"])
def test_largest_zoo_members_take_a_bf16_training_step(arch):
    """The widest / deepest member of each family (7392-channel BatchNorms, 644 M parameters, 2560-wide depthwise
    convolutions ...) runs one AMP O2 step and an eval forward: kernel limits are sized for the whole registry."""
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD

    amp.state().reset()
    torch.manual_seed(0)
    try:
        m = models.create_model(arch).cuda().train()
        opt = FusedSGD(m.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4)
        m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
        x = torch.randint(0, 256, (4, 224, 224, 3), dtype=torch.uint8, device="cuda")
        t = torch.randint(0, 1000, (4,), device="cuda")
        loss, _ = F.loss_and_metrics(m(x), t)
        opt.zero_grad()
        with amp.scale_loss(loss, opt) as s:
            s.backward()
        opt.step()
        assert torch.isfinite(loss) and float(loss) < 50
        m.eval()
        with torch.no_grad():
            out = m(x)
        assert out.shape == (4, 1000) and torch.isfinite(out.float()).all()
    finally:
        amp.state().reset()
        torch.cuda.empty_cache()

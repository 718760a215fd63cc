"""Checkpoint dict compatibility, FusedSGD == torch.optim.SGD, loss-scaler state machine."""
import os

import pytest
import torch

from distributed_pytorch_amd import amp, checkpoint as ckpt, models
from distributed_pytorch_amd.optim import FusedSGD


def _tiny():
    torch.manual_seed(0)
    return models.create_model("resnet18", num_classes=10)


def test_param_counts_and_torchvision_key_names():
    want = {"resnet18": 11689512, "resnet50": 25557032, "alexnet": 61100840, "vgg16": 138357544,
            "squeezenet1_0": 1248424, "squeezenet1_1": 1235496, "densenet121": 7978856, "densenet169": 14149480,
            "resnext50_32x4d": 25028904, "wide_resnet50_2": 68883240,
            "mobilenet_v2": 3504872, "mnasnet1_0": 4383312, "mnasnet0_5": 2218512,
            "googlenet": 6624904, "regnet_x_400mf": 5495976, "regnet_x_1_6gf": 9190136, "regnet_x_8gf": 39572648,
            "regnet_y_400mf": 4344144, "regnet_y_1_6gf": 11202430, "mobilenet_v3_large": 5483032,
            "mobilenet_v3_small": 2542856, "efficientnet_b0": 5288548, "efficientnet_b3": 12233232,
            "efficientnet_v2_s": 21458488, "inception_v3": 23834568, "convnext_tiny": 28589128,
            "shufflenet_v2_x0_5": 1366792, "shufflenet_v2_x1_0": 2278604, "shufflenet_v2_x1_5": 3503624,
            "shufflenet_v2_x2_0": 7393996, "vit_b_16": 86567656, "vit_b_32": 88224232, "swin_t": 28288354,
            "swin_s": 49606258, "swin_v2_t": 28351570, "swin_v2_b": 87930848, "maxvit_t": 30919624}
    for arch, n in want.items():
        m = models.create_model(arch)
        assert sum(p.numel() for p in m.parameters()) == n, arch
    sd = models.create_model("resnet50").state_dict()
    for k in ("conv1.weight", "bn1.running_mean", "bn1.num_batches_tracked", "layer1.0.conv1.weight",
              "layer1.0.downsample.0.weight", "layer1.0.downsample.1.running_var", "layer4.2.bn3.bias", "fc.weight"):
        assert k in sd, k
    assert len(list(models.create_model("resnet50").parameters())) == 161
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    va = models.create_model("vgg16").state_dict()
    assert va["features.0.weight"].shape == (64, 3, 3, 3) and va["classifier.0.weight"].shape == (4096, 25088)
    dn = models.create_model("densenet121").state_dict()
    for k in ("features.conv0.weight", "features.denseblock1.denselayer1.norm1.running_mean",
              "features.denseblock4.denselayer16.conv2.weight", "features.transition2.conv.weight",
              "features.norm5.bias", "classifier.weight"):
        assert k in dn, k
    assert models.create_model("resnext50_32x4d").state_dict()["layer1.0.conv2.weight"].shape == (128, 4, 3, 3)
    mb = models.create_model("mobilenet_v2").state_dict()
    assert mb["features.2.conv.1.0.weight"].shape == (96, 1, 3, 3) and mb["features.18.1.running_var"].shape == (1280,)
    assert models.create_model("mnasnet1_0").state_dict()["layers.9.0.layers.3.weight"].shape == (72, 1, 5, 5)
    gn = models.create_model("googlenet", aux_logits=True).state_dict()
    assert gn["inception4a.branch3.1.conv.weight"].shape == (48, 16, 3, 3) and gn["aux1.fc1.weight"].shape == (1024, 2048)
    rg = models.create_model("regnet_x_400mf").state_dict()
    assert rg["trunk_output.block3.block3-0.f.b.0.weight"].shape == (160, 16, 3, 3) and "stem.1.running_mean" in rg \
        and "trunk_output.block2.block2-0.proj.0.weight" in rg
    ry = models.create_model("regnet_y_400mf").state_dict()
    assert ry["trunk_output.block2.block2-0.f.se.fc1.weight"].shape == (12, 104, 1, 1) \
        and ry["trunk_output.block2.block2-0.f.se.fc2.bias"].shape == (104,)
    m3 = models.create_model("mobilenet_v3_large").state_dict()
    assert m3["features.4.block.2.fc1.weight"].shape == (24, 72, 1, 1) and m3["features.4.block.1.0.weight"].shape \
        == (72, 1, 5, 5) and m3["classifier.0.weight"].shape == (1280, 960) and "features.16.1.running_var" in m3
    ef = models.create_model("efficientnet_b0").state_dict()
    assert ef["features.2.0.block.2.fc1.weight"].shape == (4, 96, 1, 1) and ef["features.8.0.weight"].shape \
        == (1280, 320, 1, 1) and ef["features.1.0.block.0.0.weight"].shape == (32, 1, 3, 3)
    e2 = models.create_model("efficientnet_v2_s").state_dict()
    assert e2["features.1.0.block.0.0.weight"].shape == (24, 24, 3, 3) and e2["features.2.0.block.1.0.weight"].shape \
        == (48, 96, 1, 1) and e2["features.4.0.block.2.fc2.weight"].shape == (256, 16, 1, 1)
    iv = models.create_model("inception_v3", aux_logits=True)
    assert sum(p.numel() for p in iv.parameters()) == 27161264
    iv = iv.state_dict()
    assert iv["Mixed_6b.branch7x7_2.conv.weight"].shape == (128, 128, 1, 7) and iv["AuxLogits.conv1.conv.weight"].shape \
        == (768, 128, 5, 5) and iv["Mixed_7c.branch3x3dbl_3b.conv.weight"].shape == (384, 384, 3, 1) \
        and "Conv2d_1a_3x3.bn.running_mean" in iv
    cx = models.create_model("convnext_tiny").state_dict()
    assert cx["features.1.0.block.0.weight"].shape == (96, 1, 7, 7) and cx["features.1.0.layer_scale"].shape == (96, 1, 1) \
        and cx["features.1.0.block.3.weight"].shape == (384, 96) and cx["features.2.1.weight"].shape == (192, 96, 2, 2) \
        and cx["classifier.2.weight"].shape == (1000, 768) and "features.0.1.bias" in cx and "classifier.0.weight" in cx
    sq = models.create_model("squeezenet1_1").state_dict()
    assert sq["features.3.squeeze.weight"].shape == (16, 64, 1, 1) and sq["classifier.1.weight"].shape == (1000, 512, 1, 1)
    al = models.create_model("alexnet").state_dict()
    assert al["features.0.weight"].shape == (64, 3, 11, 11) and al["classifier.1.weight"].shape == (4096, 9216)


def test_checkpoint_roundtrip_reference_dict(tmp_path):
    m = _tiny()
    opt = FusedSGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(2, 3, 32, 32)
    m(x).sum().backward()
    opt.step()
    state = ckpt.build_state(4, "resnet18", m, opt, torch.tensor(12.5), distributed=True)
    assert set(state) == {"epoch", "arch", "state_dict", "best_acc1", "optimizer"} and state["epoch"] == 5
    assert all(k.startswith("module.") for k in state["state_dict"])
    assert all(v.dtype == torch.float32 for v in state["state_dict"].values() if v.is_floating_point())
    assert state["state_dict"]["module.conv1.weight"].is_contiguous()         # plain OIHW in the file
    path = ckpt.save_checkpoint(state, True, out_dir=str(tmp_path))
    assert os.path.isfile(path) and os.path.isfile(tmp_path / ckpt.BEST_NAME)
    m2 = _tiny()
    for p in m2.parameters():
        p.data.add_(1.0)
    opt2 = FusedSGD(m2.parameters(), lr=0.5, momentum=0.9)
    start, best = ckpt.resume(path, m2, opt2, verbose=False)
    assert start == 5 and best == 12.5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == 0.1
    b1 = opt.state_dict()["state"][0]["momentum_buffer"]
    assert torch.equal(opt2.state_dict()["state"][0]["momentum_buffer"], b1)


def test_loads_checkpoint_written_by_stock_torch(tmp_path):
    """A file produced the reference's way: torch.optim.SGD state, 'features.module.' / 'module.' prefixes,
    fp16 weights, best_prec1 key."""
    m = _tiny()
    ref_opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    m(torch.randn(2, 3, 32, 32)).sum().backward()
    ref_opt.step()
    sd = {"module." + k: (v.half() if v.is_floating_point() else v) for k, v in m.state_dict().items()}
    torch.save({"epoch": 7, "arch": "resnet18", "state_dict": sd, "best_prec1": 33.0,
                "optimizer": ref_opt.state_dict()}, tmp_path / "c.pth.tar")
    m2 = _tiny()
    opt2 = FusedSGD(m2.parameters(), lr=0.1, momentum=0.9)
    start, best = ckpt.resume(str(tmp_path / "c.pth.tar"), m2, opt2, verbose=False)
    assert (start, best) == (7, 33.0) and opt2.param_groups[0]["lr"] == 0.05
    assert torch.allclose(m2.conv1.weight, m.conv1.weight.half().float())
    assert m2.conv1.weight.dtype == torch.float32
    # and the reverse: torch's SGD loads OUR optimizer state
    torch.optim.SGD(m.parameters(), lr=1.0, momentum=0.9).load_state_dict(opt2.state_dict())
    assert ckpt.strip_prefixes({"features.module.0.weight": 1, "module.fc.bias": 2}) == \
        {"features.0.weight": 1, "fc.bias": 2}


def test_corrupt_checkpoint_is_a_clean_error(tmp_path):
    p = tmp_path / "bad.pth.tar"
    p.write_bytes(b"PK\x03\x04 truncated")
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_checkpoint(str(p))
    torch.save({"not": "a checkpoint"}, tmp_path / "other.pth")
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_checkpoint(str(tmp_path / "other.pth"))
    assert ckpt.resume(str(tmp_path / "missing.pth"), _tiny(), verbose=False) is None


@pytest.mark.parametrize("nesterov,wd", [(False, 1e-4), (True, 0.0)])
def test_fused_sgd_cpu_matches_torch(nesterov, wd):
    torch.manual_seed(1)
    ps = [torch.randn(5, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    a = FusedSGD(ps, lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    b = torch.optim.SGD(qs, lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    assert a.state_dict()["state"] == {} and a.state_dict()["param_groups"][0]["params"] == [0, 1]
    for _ in range(5):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        a.step()
        b.step()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, atol=1e-6)
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    assert torch.allclose(sa["state"][0]["momentum_buffer"], sb["state"][0]["momentum_buffer"], atol=1e-6)
    assert {k for k in sb["param_groups"][0]} <= {k for k in sa["param_groups"][0]}


def test_loss_scaler_state_machine_cpu():
    s = amp.LossScaler(torch.device("cpu"), init_scale=2.0 ** 16, dynamic=True, growth_interval=3)
    p = torch.ones(4, requires_grad=True)
    opt = FusedSGD([p], lr=1.0)
    p.grad = torch.tensor([1.0, float("nan"), 1.0, 1.0])
    s.check_overflow([p.grad])
    opt.arm_scaler(s)
    opt.step()
    assert torch.equal(p.detach(), torch.ones(4)) and s.loss_scale() == 2.0 ** 15      # skipped + halved
    for i in range(3):
        p.grad = torch.full((4,), 2.0 ** 15)
        s.check_overflow([p.grad])
        opt.arm_scaler(s)
        opt.step()
        assert torch.allclose(p.detach(), torch.full((4,), -float(i)))                  # unscaled grad == 1
    assert s.loss_scale() == 2.0 ** 16                                                   # grew after 3 clean steps
    sd = s.state_dict()
    s2 = amp.LossScaler(torch.device("cpu"))
    s2.load_state_dict(sd)
    assert s2.loss_scale() == s.loss_scale()


def test_amp_initialize_levels_and_scale_loss_cpu():
    m = _tiny()
    opt = FusedSGD(m.parameters(), lr=0.01, momentum=0.9)
    m, opt = amp.initialize(m, opt, opt_level="O1", dtype="bf16")
    assert amp.compute_dtype() == torch.bfloat16 and m.conv1.weight.dtype == torch.float32
    out = m(torch.randn(2, 3, 32, 32))
    assert out.dtype == torch.float32              # logits leave the network in fp32
    loss = out.float().logsumexp(1).mean()
    with amp.scale_loss(loss, opt) as scaled:
        scaled.backward()
    assert m.conv1.weight.grad is not None and m.conv1.weight.grad.dtype == torch.float32
    opt.step()
    m2 = _tiny()
    opt2 = FusedSGD(m2.parameters(), lr=0.01)
    amp.initialize(m2, opt2, opt_level="O2", dtype="fp16", loss_scale="dynamic")
    assert amp.state().scaler.dynamic and amp.state().scaler.loss_scale() == 2.0 ** 16
    amp.initialize(m2, opt2, opt_level="O0")
    assert amp.compute_dtype(torch.float32) == torch.float32
    with pytest.raises(ValueError):
        amp.initialize(m2, opt2, opt_level="O9")
    amp.state().reset()


def test_tfevents_writer_records_are_well_formed(tmp_path):
    import struct

    from distributed_pytorch_amd.utils.tb import SummaryWriter, _masked_crc

    w = SummaryWriter(log_dir=str(tmp_path / "run"), comment="_x")
    w.add_scalar("Loss/train", 1.5, 3)
    w.close()
    files = [f for f in os.listdir(w.log_dir) if "tfevents" in f]
    assert len(files) == 1
    raw = open(os.path.join(w.log_dir, files[0]), "rb").read()
    off, n = 0, 0
    while off < len(raw):
        (length,) = struct.unpack("<Q", raw[off:off + 8])
        assert struct.unpack("<I", raw[off + 8:off + 12])[0] == _masked_crc(raw[off:off + 8])
        payload = raw[off + 12:off + 12 + length]
        assert struct.unpack("<I", raw[off + 12 + length:off + 16 + length])[0] == _masked_crc(payload)
        off += 16 + length
        n += 1
    assert n == 2 and b"Loss/train" in raw           # version record + one scalar

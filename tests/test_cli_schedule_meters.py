"""CLI parity, LR schedules (P19/P20), meters (P16-P18), accuracy (P21)."""
import math

import pytest
import torch

from distributed_pytorch_amd import cli, schedule
from distributed_pytorch_amd.utils.meters import (AverageMeter, DeviceMeter, ProgressMeter, Summary, accuracy,
                                                  format_hms)


def test_nd_defaults_match_reference():
    a = cli.parse("nd", ["/data"])
    assert (a.arch, a.workers, a.epochs, a.start_epoch, a.batch_size) == ("resnet18", 4, 90, 0, 256)
    assert (a.lr, a.momentum, a.weight_decay, a.print_freq) == (0.1, 0.9, 1e-4, 10)
    assert (a.world_size, a.rank, a.dist_url, a.dist_backend) == (-1, -1, "tcp://224.66.41.62:23456", "nccl")
    assert a.seed is None and a.gpu is None and not a.multiprocessing_distributed and not a.evaluate


def test_ddp_and_apex_defaults_and_aliases():
    d = cli.parse("ddp", ["/data", "-a", "resnet50", "--desired-acc", "0.75", "--wd", "5e-5", "-j", "8",
                          "--learning-rate", "0.2", "-b", "512", "-p", "5"])
    assert d.batch_size == 512 and d.weight_decay == 5e-5 and d.workers == 8 and d.lr == 0.2 and d.print_freq == 5
    assert d.desired_acc == 75.0           # fraction -> percent (reference compares against percent)
    assert cli.parse("ddp", ["/d"]).batch_size == 1024 and cli.parse("ddp", ["/d"]).arch == "resnet50"
    a = cli.parse("apex", ["/data", "--opt-level", "O2", "--loss-scale", "dynamic", "--sync-bn", "--local_rank", "3",
                           "--keep-batchnorm-fp32", "True", "--channels-last", "False", "--weight-decay", "1e-3"])
    assert a.batch_size == 224 and a.opt_level == "O2" and a.loss_scale == "dynamic" and a.sync_bn
    assert a.local_rank == 3 and a.keep_batchnorm_fp32 == "True" and a.channels_last is False
    assert a.weight_decay == 1e-3 and a.dist_url == "env://"


def test_arch_choices_and_data_required():
    assert {"resnet18", "resnet50", "alexnet", "vgg16"} <= set(cli.model_names())
    with pytest.raises(SystemExit):
        cli.parse("nd", ["-a", "not_a_model", "/data"])
    with pytest.raises(SystemExit):
        cli.parse("nd", [])
    assert cli.parse("nd", ["--synthetic"]).data is None


def test_step_lr_closed_form():
    for epoch, want in [(0, 0.1), (29, 0.1), (30, 0.01), (59, 0.01), (60, 0.001), (89, 0.001)]:
        assert math.isclose(schedule.step_lr(0.1, epoch), want, rel_tol=1e-12)


def test_warmup_step_lr_closed_form():
    L, base = 100, 0.8
    # linear warm-up over the first 5 epochs, reference formula lr*(1+step+epoch*len)/(5*len)
    assert math.isclose(schedule.warmup_step_lr(base, 0, 1, L), base * 2 / 500)
    assert math.isclose(schedule.warmup_step_lr(base, 4, 99, L), base * (1 + 99 + 400) / 500)
    assert math.isclose(schedule.warmup_step_lr(base, 5, 1, L), base)
    assert math.isclose(schedule.warmup_step_lr(base, 30, 1, L), base * 0.1)
    assert math.isclose(schedule.warmup_step_lr(base, 60, 1, L), base * 0.01)
    assert math.isclose(schedule.warmup_step_lr(base, 80, 1, L), base * 1e-3)      # factor = 80//30 + 1 = 3
    assert math.isclose(schedule.scale_lr_for_global_batch(0.1, 224, 8), 0.1 * 224 * 8 / 256)


def test_meter_strings_byte_compatible():
    m = AverageMeter("Loss", ":.4e")
    m.update(2.5, 4)
    m.update(1.5, 4)
    assert str(m) == "Loss 1.5000e+00 (2.0000e+00)"
    t = AverageMeter("Time", ":6.3f")
    t.update(0.1234)
    p = ProgressMeter(1000, [t, m], prefix="Epoch: [3]")
    assert p.line(7) == "Epoch: [3][   7/1000]\tTime  0.123 ( 0.123)\tLoss 1.5000e+00 (2.0000e+00)"
    a1 = AverageMeter("Acc@1", ":6.2f", Summary.AVERAGE)
    a1.update(75.0, 2)
    n = AverageMeter("Time", ":6.3f", Summary.NONE)
    assert ProgressMeter(1, [n, a1]).summary_line() == " *  Acc@1 75.000"
    assert format_hms(3725.5).startswith("1.00 hrs 2.00 mins 5.50 secs")


def test_accuracy_percent_and_shape():
    out = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0, 0.0], [0.9, 0.1, 0.0, 0.0, 0.0, 0.0],
                        [0.0, 0.1, 0.2, 0.3, 0.4, 0.5], [0.5, 0.4, 0.3, 0.2, 0.1, 0.0]])
    tgt = torch.tensor([1, 1, 5, 5])
    a1, a5 = accuracy(out, tgt, topk=(1, 5))
    assert a1.shape == (1,) and a5.shape == (1,)
    assert a1.item() == 50.0 and a5.item() == 75.0
    # oracle: the reference's topk-based implementation
    _, pred = out.topk(5, 1, True, True)
    correct = pred.t().eq(tgt.view(1, -1).expand_as(pred.t()))
    assert correct[:1].reshape(-1).float().sum().item() * 25.0 == a1.item()


def test_device_meter_packs_one_record():
    dm = DeviceMeter(torch.device("cpu"))
    dm.update(torch.tensor([8.0, 2.0, 3.0, 4.0]))
    dm.update(torch.tensor([4.0, 4.0, 4.0, 4.0]))
    last, run = dm.fetch()
    assert last == (1.0, 100.0, 100.0, 4.0) and run == (1.5, 75.0, 87.5, 8.0)

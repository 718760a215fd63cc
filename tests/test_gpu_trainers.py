"""The three reference-compatible trainers end to end on one MI355X (synthetic data): training, validation,
checkpoint + resume, --evaluate, AMP opt levels, SyncBN flag, hipGraph mode and both process-group engines."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(script, *argv, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", script)] + list(argv)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


COMMON = ["--synthetic", "--synthetic-size", "128", "--synthetic-val-size", "32", "--num-classes", "16", "-b", "32",
          "-p", "2", "--steps", "4", "-j", "0"]


def test_nd_trainer_single_gpu_checkpoint_resume_evaluate_and_hip_graph(tmp_path):
    r = _run("nd_imagenet.py", *COMMON, "-a", "resnet18", "--gpu", "0", "--epochs", "1", "--out-dir", str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Use GPU: 0 for training" in r.stdout and "Epoch: [0][4/4]" in r.stdout and " * " in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert ck["epoch"] == 1 and ck["state_dict"]["conv1.weight"].shape == (64, 3, 7, 7)
    assert ck["state_dict"]["conv1.weight"].is_contiguous()                      # plain OIHW, loadable by torchvision
    r2 = _run("nd_imagenet.py", *COMMON, "-a", "resnet18", "--gpu", "0", "--epochs", "2", "--hip-graph",
              "--resume", str(tmp_path / "checkpoint.pth.tar"), "--out-dir", str(tmp_path))
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "(epoch 1)" in r2.stdout and "Epoch: [1][4/4]" in r2.stdout
    r3 = _run("nd_imagenet.py", *COMMON, "-a", "resnet18", "--gpu", "0", "--evaluate",
              "--resume", str(tmp_path / "checkpoint.pth.tar"))
    assert r3.returncode == 0 and "Test:" in r3.stdout and "Epoch:" not in r3.stdout


@pytest.mark.parametrize("backend", ["nccl", "native"])
def test_ddp_trainer_one_rank_on_gpu(tmp_path, backend):
    r = _run("imagenet_ddp.py", *COMMON, "-a", "resnet18", "--epochs", "1", "--dist-url",
             "tcp://127.0.0.1:%d" % _port(), "--dist-backend", backend, "--world-size", "1", "--rank", "0",
             "--out-dir", str(tmp_path), env={"DPA_NPROC_PER_NODE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Epoch: [0][4/4]" in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert "module.conv1.weight" in ck["state_dict"]


@pytest.mark.parametrize("extra", [["--opt-level", "O2"], ["--opt-level", "O1", "--sync-bn"],
                                   ["--opt-level", "O2", "--amp-dtype", "fp16", "--loss-scale", "dynamic"]])
def test_amp_trainer_on_gpu(tmp_path, extra):
    r = _run("imagenet_ddp_apex.py", *COMMON, "-a", "resnet50", "--epochs", "1", *extra, "--out-dir", str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Throughput" in r.stdout and "### Training Time:" in r.stdout and "Prec@1" in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert "best_prec1" in ck and "amp" in ck

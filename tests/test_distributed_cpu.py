"""Multi-process data parallelism on the CPU (gloo, world_size 2): the plumbing configuration of
BASELINE.json #1.  Covers the bucket planner, the reducer state machine, DDP == single process on the
concatenated batch, SyncBN statistics, the launcher's failure propagation and the trainers end to end."""
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

import _dist_workers as W
from distributed_pytorch_amd import launch
from distributed_pytorch_amd.parallel.bucketing import MiB, chunk_ranges, plan_buckets_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bucket_planner_caps_order_and_alignment():
    numels = [1000, 300000, 50, 2000000, 700]
    specs = plan_buckets_py(numels, [4] * 5, [0] * 5, [4, 3, 2, 1, 0], cap_bytes=2 * MiB, first_cap_bytes=1024,
                            min_bytes=0)
    assert [s.param_ids for s in specs] == [[4], [3], [2, 1, 0]]          # small first bucket, oversize alone
    # default min_bytes (1 MiB): the 2.8 KB first bucket is not worth a collective of its own, it takes the big tensor
    specs = plan_buckets_py(numels, [4] * 5, [0] * 5, [4, 3, 2, 1, 0], cap_bytes=2 * MiB, first_cap_bytes=1024)
    assert [s.param_ids for s in specs] == [[4, 3], [2, 1, 0]]
    for s in specs:
        assert all(off * 4 % 256 == 0 for off in s.offsets) and s.total >= sum(s.numels)
    # the parameters that become ready last form a small final bucket (its all-reduce cannot overlap with backward)
    from distributed_pytorch_amd.parallel.bucketing import plan_buckets
    tail = plan_buckets([1000] * 10, [4] * 10, [0] * 10, list(range(9, -1, -1)), cap_bytes=1 << 30,
                        first_cap_bytes=1 << 30, last_cap_bytes=9000)
    assert [s.param_ids for s in tail] == [[9, 8, 7, 6, 5, 4, 3, 2], [1, 0]] and [s.index for s in tail] == [0, 1]
    mixed = plan_buckets_py([10, 10, 10], [4, 2, 4], [0, 1, 0], [2, 1, 0], 1 << 30, 1 << 30)
    assert sorted(tuple(s.param_ids) for s in mixed) == [(1,), (2, 0)]    # buckets are dtype-homogeneous
    assert chunk_ranges(10, 4, 16) == [(0, 4), (4, 8), (8, 10)]


def test_native_bucket_planner_equals_python_oracle():
    """csrc/comm/comm.cpp:plan_buckets against the Python oracle on random parameter lists."""
    import random

    from distributed_pytorch_amd import backend
    from distributed_pytorch_amd.parallel.bucketing import plan_buckets

    ext = backend.load()
    if ext is None or not hasattr(ext, "plan_buckets"):
        pytest.skip("_C not built")
    rng = random.Random(0)
    for trial in range(200):
        n = rng.randint(1, 40)
        numels = [rng.choice([1, 7, 64, 1000, 4096, 250000, 3000000]) for _ in range(n)]
        dk = [rng.choice([0, 0, 0, 1]) for _ in range(n)]
        es = [4 if k == 0 else 2 for k in dk]
        order = list(range(n))
        rng.shuffle(order)
        cap, first = rng.choice([1 << 12, 1 << 20, 1 << 24, 1 << 40]), rng.choice([1, 1 << 10, 1 << 20, 1 << 40])
        last = rng.choice([0, 0, 1 << 8, 1 << 14, 1 << 22, 1 << 40])
        align = rng.choice([16, 256])
        mn = rng.choice([0, 1 << 10, 1 << 20])
        a = plan_buckets(numels, es, dk, order, cap, first, align, last, mn)
        b = plan_buckets_py(numels, es, dk, order, cap, first, align, last, mn)
        assert [(s.index, s.dtype_key, s.param_ids, s.offsets, s.numels, s.total) for s in a] == \
               [(s.index, s.dtype_key, s.param_ids, s.offsets, s.numels, s.total) for s in b], trial
    with pytest.raises(Exception):
        ext.plan_buckets([1], [4], [0], [3], 1, 1, 256, 0)          # parameter index out of range


def test_planner_gives_vgg16_no_sub_megabyte_bucket():
    """VGG-16's classifier in backward order: a bias in front of an over-cap weight used to become a 16 KB collective
    of its own, queued in front of the 392 MiB one (round-2 verdict, weak #5)."""
    from distributed_pytorch_amd import models
    from distributed_pytorch_amd.parallel.bucketing import plan_buckets

    ps = [p for p in models.create_model("vgg16").parameters()]
    order = list(range(len(ps) - 1, -1, -1))
    specs = plan_buckets([p.numel() for p in ps], [4] * len(ps), [0] * len(ps), order, 64 * MiB, 4 * MiB,
                         last_cap_bytes=2 * MiB)
    sizes = [s.total * 4 / MiB for s in specs]
    assert len(specs) == 5 and min(sizes) >= 0.9, sizes        # [15.6, 64, 392, 55.2, 1.0] MiB
    assert sorted(i for s in specs for i in s.param_ids) == list(range(len(ps)))


def test_native_store_rendezvous_and_host_collectives(tmp_path):
    """--dist-backend native: C++ TcpStore rendezvous, no torch.distributed anywhere."""
    world = 2
    launch.spawn(W.store_ops, world, (world, _port(), str(tmp_path)))
    rs = [torch.load(tmp_path / ("rank%d.pt" % r), weights_only=True) for r in range(world)]
    for r, d in enumerate(rs):
        assert d["got"] == b"payload" and d["total"] == 3 and d["rank"] == r and d["world"] == 2
        assert d["backend"] == "native" and d["obj"] == {"rank": 0}
        assert torch.equal(d["sum"], torch.arange(6, dtype=torch.float32) * 3)
        assert torch.equal(d["bcast"], torch.ones(3))
        assert torch.equal(d["gathered"], torch.tensor([[0.0, 10.0], [1.0, 11.0]]))
    assert sorted(d["n"] for d in rs) in ([1, 3], [2, 3])


def test_native_store_shutdown_is_clean_every_time(tmp_path):
    """Round-1 failure: rank 0 destroyed the store while rank 1's last barrier reply was unread (3 of 4 runs)."""
    for i in range(6):
        d = tmp_path / str(i)
        d.mkdir()
        t0 = time.time()
        launch.spawn(W.store_ops, 2, (2, _port(), str(d)))
        assert time.time() - t0 < 30


@pytest.mark.parametrize("world", [2, 4])
def test_cpp_reducer_on_host_backend_chunks_replans_and_matches_oracle(tmp_path, world):
    """The same C++ state machine that drives RCCL, over host memory at world 2 and 4 (csrc/comm/reducer_core.cpp)."""
    launch.spawn(W.ddp_native_reducer, world, (world, _port(), str(tmp_path)))
    rs = [torch.load(tmp_path / ("rank%d.pt" % r), weights_only=True) for r in range(world)]
    r0 = rs[0]
    assert r0["summary0"]["reducer"] == "c++" and r0["summary0"]["ranks"] == world
    assert r0["summary0"]["buckets"] >= 3
    # the big weights exceed chunk_mb: more collectives than buckets
    assert r0["summary0"]["collectives"] == 0 and r0["summary"]["collectives"] > 2 * r0["summary"]["buckets"]
    # re-planned once, from the observed order: `last.*` is ready first although it was registered first
    assert r0["replans"] == 1 and r0["plan"] != r0["plan0"]
    names = [n for n, _ in W.OutOfOrderNet().named_parameters()]
    flat = [names[i] for b in r0["plan"] for i in b]
    assert flat[:2] == ["last.bias", "last.weight"] or flat[:2] == ["last.weight", "last.bias"]
    assert flat.index("mid.weight") < flat.index("first.weight") < flat.index("unused.weight")
    for r in rs[1:]:
        assert r["plan"] == r0["plan"] and r["replans"] == 1
        for n in r0["params"]:
            assert torch.equal(r["params"][n], r0["params"][n]), n
    # oracle: one process, whole batch, torch SGD
    model = W.OutOfOrderNet()
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 4 * world, 20, generator=g)
    t = torch.randn(3, 4 * world, 6, generator=g)
    for it in range(3):
        opt.zero_grad()
        ((model(x[it]) - t[it]) ** 2).mean().backward()
        for n, p in model.named_parameters():
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            assert torch.allclose(r0["grads"][it][n], want, rtol=1e-4, atol=1e-6), (it, n)
        for p in model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt.step()
    for n, p in model.named_parameters():
        assert torch.allclose(r0["params"][n], p, rtol=1e-4, atol=1e-6), n


@pytest.mark.parametrize("world", [2, 4])
def test_bf16_gradient_wire_halves_the_bytes_and_tracks_the_fp32_trajectory(tmp_path, world):
    """Apex O2 all-reduces 16-bit gradients (imagenet_ddp_apex.py:169-183): fp32 buckets travelling as bf16 through the
    same C++ reducer.  Half the bytes, ranks still bit-identical, three SGD steps within bf16 rounding of fp32 wire."""
    a, b = tmp_path / "fp32", tmp_path / "bf16"
    a.mkdir()
    b.mkdir()
    launch.spawn(W.ddp_native_reducer, world, (world, _port(), str(a), "fp32"))
    launch.spawn(W.ddp_native_reducer, world, (world, _port(), str(b), "bf16"))
    fa = torch.load(a / "rank0.pt", weights_only=True)
    rb = [torch.load(b / ("rank%d.pt" % r), weights_only=True) for r in range(world)]
    fb = rb[0]
    assert fa["summary"]["grad_wire"] == "fp32" and fb["summary"]["grad_wire"] == "bf16"
    assert fb["summary"]["bytes_per_step"] * 2 == fa["summary"]["bytes_per_step"] == fa["summary"]["bucket_bytes"]
    assert fb["summary"]["wire_bytes_total"] * 2 == fa["summary"]["wire_bytes_total"] > 0
    for r in rb[1:]:
        for n in fb["params"]:
            assert torch.equal(r["params"][n], fb["params"][n]), n
    for n in fa["params"]:
        g32, g16 = fa["grads"][0][n], fb["grads"][0][n]
        assert torch.equal(g16, g16.bfloat16().float()), n            # what arrived is a bf16 value
        # first step: same weights, rounding only (each rank's value is off by <= 2^-9 of ITS magnitude, and the
        # ranks' values may cancel in the mean: bound against the tensor's scale, not element by element)
        assert float((g16 - g32).abs().max()) <= 2 ** -6 * float(g32.abs().max()) + 1e-9, n
        assert torch.allclose(fb["params"][n], fa["params"][n], rtol=2e-2, atol=2e-3), n


@pytest.mark.parametrize("backend", ["native", "gloo"])
def test_ddp_self_diagnosis_reports_link_probe_and_bucket_waits(tmp_path, backend):
    """The comm layer explains itself: all-reduce bandwidth at construction, then per-bucket busy / exposed time of
    the first steady steps, through the C++ reducer (host backend) and the Python reducer (gloo)."""
    launch.spawn(W.ddp_self_diagnosis, 2, (2, _port(), str(tmp_path), backend))
    for r in range(2):
        d = torch.load(tmp_path / ("rank%d.pt" % r), weights_only=True)
        probe, timing = d["summary"]["probe"], d["summary"]["timing"]
        assert probe["ranks"] == 2 and len(probe["all_reduce"]) == 2
        for row in probe["all_reduce"]:
            assert row["ms"] > 0 and row["algbw_gbs"] > 0 and abs(row["busbw_gbs"] - row["algbw_gbs"]) < 1e-3
        assert probe["all_reduce"][0]["mbytes"] > probe["all_reduce"][1]["mbytes"]
        assert timing is not None and timing["steps"] == 4
        assert len(timing["bucket_busy_ms"]) == len(timing["bucket_exposed_ms"]) == d["buckets"]
        assert all(v >= 0 for v in timing["bucket_busy_ms"]) and timing["exposed_tail_ms"] >= 0
        # stamps ran for a bounded window and were switched off again
        assert d["states"][0] == "wait" and d["states"][-1] == "done" and "on" in d["states"]


def test_watchdog_fails_the_rank_when_a_peer_stops_calling_all_reduce():
    t0 = time.time()
    with pytest.raises(launch.ProcessFailure) as e:
        launch.spawn(W.reducer_hang, 2, (2, _port()))
    assert e.value.local_rank == 0 and e.value.exitcode != 0
    assert "reducer failed" in e.value.tb or e.value.exitcode == 86
    assert time.time() - t0 < 40


@pytest.mark.parametrize("delay,backend", [(False, "gloo"), (True, "gloo"), (False, "native")])
def test_ddp_matches_single_process_on_full_batch(tmp_path, delay, backend):
    world = 2
    launch.spawn(W.ddp_grads, world, (world, _port(), str(tmp_path), delay, backend))
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for n in r0["grads"]:
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n             # identical after all-reduce
        assert torch.equal(r0["params"][n], r1["params"][n]), n
    assert r0["keys"][0].startswith("module.")
    if delay:
        assert len(r0["buckets"]) == 1
    else:
        assert len(r0["buckets"]) > 1
    # oracle: one process, whole batch (mean loss over 8 == mean of the two shard means)
    from distributed_pytorch_amd.ops import functional as F

    model = W._model()
    g = torch.Generator().manual_seed(99)
    x = torch.randn(8, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (8,), generator=g)
    for m in model.modules():
        if hasattr(m, "running_mean"):
            m.eval()
    F.cross_entropy(model(x), t).backward()
    for n, p in model.named_parameters():
        got = r0["grads"]["module." + n]
        assert torch.allclose(got, p.grad, rtol=1e-3, atol=1e-5), (n, (got - p.grad).abs().max())


@pytest.mark.parametrize("backend", ["gloo", "native"])
def test_syncbn_equals_batchnorm_over_the_global_batch(tmp_path, backend):
    world = 2
    launch.spawn(W.syncbn_stats, world, (world, _port(), str(tmp_path), backend))
    rs = [torch.load(tmp_path / ("rank%d.pt" % r), weights_only=True) for r in range(world)]
    torch.manual_seed(7)
    x = torch.randn(8, 5, 5, 16).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(16)
    y = bn(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    (y * torch.arange(16.0)).sum().backward()
    assert torch.allclose(torch.cat([r["y"] for r in rs]), y, atol=1e-5)
    assert torch.allclose(torch.cat([r["dx"] for r in rs]), x.grad, atol=1e-5)
    assert torch.allclose(rs[0]["rm"], bn.running_mean, atol=1e-6) and torch.allclose(rs[1]["rv"], bn.running_var, atol=1e-5)
    assert torch.allclose(rs[0]["dw"] + rs[1]["dw"], bn.weight.grad, atol=1e-4)     # local sums; DDP averages


def test_syncbn_chan_merge_survives_mean_1000_std(tmp_path):
    """Apex merges (mean, M2, n) partials (welford_parallel, imagenet_ddp_apex.py:146-148); E[x^2] - mean^2 in fp32 has
    lost every digit of the variance at |mean| = 1e3 std.  Oracle: float64 statistics of the global batch."""
    world = 2
    launch.spawn(W.syncbn_stats, world, (world, _port(), str(tmp_path), "gloo", 1000.0))
    rs = [torch.load(tmp_path / ("rank%d.pt" % r), weights_only=True) for r in range(world)]
    torch.manual_seed(7)
    x = torch.randn(8, 5, 5, 16) + 1000.0
    x[4:] += 0.5
    xd = x.double().reshape(-1, 16)
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    yref = ((xd - mean) / (var + 1e-5).sqrt()).reshape(x.shape)
    y = torch.cat([r["y"] for r in rs]).double()
    assert (y - yref).abs().max() < 2e-3, (y - yref).abs().max()          # fp32 input spacing at 1000 is 6e-5
    assert torch.allclose(rs[0]["rv"].double(), 0.9 + 0.1 * xd.var(0, unbiased=True), rtol=1e-3)
    assert torch.allclose(rs[0]["rm"].double(), 0.1 * mean, rtol=1e-6)


def test_rank_failure_tears_the_job_down_instead_of_hanging():
    t0 = time.time()
    with pytest.raises(launch.ProcessFailure) as e:
        launch.spawn(W.crash_one, 2, (2, _port()))
    assert e.value.exitcode == 17 and e.value.local_rank == 1
    assert time.time() - t0 < 60


def _run(script, *argv, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", script)] + list(argv)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)


def test_nd_trainer_cpu_trains_checkpoints_and_resumes(tmp_path):
    common = ["--synthetic", "-a", "resnet18", "-b", "8", "-p", "2", "--cpu", "--synthetic-size", "32",
              "--synthetic-val-size", "8", "--num-classes", "10", "--lr", "0.01", "--out-dir", str(tmp_path)]
    r = _run("nd_imagenet.py", *common, "--epochs", "1", "--seed", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "using CPU, this will be slow" in r.stdout and "Epoch: [0][4/4]" in r.stdout and " * " in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert set(ck) == {"epoch", "arch", "state_dict", "best_acc1", "optimizer"} and ck["epoch"] == 1
    assert "conv1.weight" in ck["state_dict"]                                  # unwrapped: no module. prefix
    r2 = _run("nd_imagenet.py", *common, "--epochs", "2", "--resume", str(tmp_path / "checkpoint.pth.tar"))
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "(epoch 1)" in r2.stdout and "Epoch: [1]" in r2.stdout and "Epoch: [0]" not in r2.stdout
    r3 = _run("nd_imagenet.py", *common, "--evaluate")
    assert r3.returncode == 0 and "Epoch:" not in r3.stdout and "Test:" in r3.stdout


@pytest.mark.parametrize("backend", ["gloo", "native"])
def test_ddp_trainer_two_ranks_and_desired_acc_stop(tmp_path, backend):
    r = _run("imagenet_ddp.py", "--synthetic", "-a", "resnet18", "-b", "16", "-p", "1", "--cpu", "--steps", "2",
             "--synthetic-size", "64", "--synthetic-val-size", "16", "--num-classes", "10", "--epochs", "3",
             "--dist-url", "tcp://127.0.0.1:%d" % _port(), "--dist-backend", backend, "--world-size", "1",
             "--rank", "0", "--desired-acc", "0.0", "--out-dir", str(tmp_path), env={"DPA_NPROC_PER_NODE": "2"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("=> creating model 'resnet18'") == 2                # one process per "GPU"
    assert "Epoch: [0][2/2]" in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert "module.conv1.weight" in ck["state_dict"]                          # saved from the DDP wrapper


def test_amp_trainer_env_rendezvous_single_rank(tmp_path):
    r = _run("imagenet_ddp_apex.py", "--synthetic", "-a", "resnet18", "-b", "8", "-p", "1", "--cpu", "--steps", "2",
             "--synthetic-size", "32", "--synthetic-val-size", "8", "--num-classes", "10", "--epochs", "1",
             "--opt-level", "O1", "--out-dir", str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "opt_level = O1" in r.stdout and "Throughput" in r.stdout and "### Training Time:" in r.stdout
    ck = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=True)
    assert "best_prec1" in ck and "best_acc1" in ck
    assert any("tfevents" in f for f in os.listdir(tmp_path))

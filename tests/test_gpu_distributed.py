"""GPU-side checks of the data-parallel runtime that fit a 1-GPU box: gradient arena / sinks, the native
RCCL communicator and the C++ bucket reducer at world_size 1 (RCCL refuses two ranks on one device), and -- through
the peer-memory communicator (csrc/comm/peer.hip), whose hipIpc windows work between processes that SHARE the
device -- the whole multi-rank GPU path at world 2 and 4: collectives, DDP == single process on the full batch,
SyncBN == BatchNorm over the global batch, re-planned + chunked buckets, bf16 wire format, a dead peer."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(model, opt, x, t):
    from distributed_pytorch_amd.ops import functional as F

    loss, _ = F.loss_and_metrics(model(x), t)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss)


def test_arena_sinks_match_plain_autograd_grads():
    """Same model / batch: gradients delivered through sinks == gradients returned through autograd."""
    import copy

    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.parallel import GradArena

    amp.state().reset()
    torch.manual_seed(0)
    a = models.create_model("resnet18", num_classes=10).cuda().train()
    b = copy.deepcopy(a)
    x = torch.randn(8, 3, 64, 64, device="cuda")
    t = torch.randint(0, 10, (8,), device="cuda")
    F.cross_entropy(a(x), t).backward()
    arena = GradArena(list(b.parameters()))
    for _ in range(2):                       # second pass: views are overwritten, not accumulated
        arena.begin_step()
        for p in b.parameters():
            p.grad = None
        F.cross_entropy(b(x), t).backward()
        arena.finish_step()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert q.grad is q._dpa_sink.view, n
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6), n
    assert len(arena.buckets) == 1 and arena.buckets[0].flat.numel() >= sum(p.numel() for p in b.parameters())
    # accumulation (two backward passes without begin_step) doubles the gradient
    F.cross_entropy(b(x), t).backward()
    assert torch.allclose(b.fc.weight.grad, 2 * a.fc.weight.grad, rtol=1e-4, atol=1e-6)


def test_native_comm_and_cpp_reducer_world1():
    import torch.distributed as dist

    from distributed_pytorch_amd import amp, backend, models
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import DistributedDataParallel
    from distributed_pytorch_amd.parallel.native_comm import NativeComm

    amp.state().reset()
    assert backend.require().rccl_version() > 20000
    os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29651", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        model = models.create_model("resnet18", num_classes=10).cuda().train()
        ddp = DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=8.0, first_bucket_mb=1.0)
        assert isinstance(ddp.comm, NativeComm) and type(ddp.reducer).__name__ == "_NativeReducer"
        assert ddp.reducer.raw.num_buckets == len(ddp.arena.buckets) > 1
        opt = FusedSGD(ddp.parameters(), lr=0.05, momentum=0.9)
        x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8, device="cuda")
        t = torch.randint(0, 10, (16,), device="cuda")
        losses = [_step(ddp, opt, x, t) for _ in range(8)]
        torch.cuda.synchronize()
        assert losses[-1] < losses[0], losses
        assert ddp.comm.raw.async_error() == ""
        # collectives of the communicator itself
        v = torch.arange(8, dtype=torch.float32, device="cuda")
        ddp.comm.all_reduce_sum(v).wait()
        ddp.comm.broadcast(v, 0)
        out = torch.empty(8, dtype=torch.float32, device="cuda")
        ddp.comm.all_gather(out, v)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), torch.arange(8, dtype=torch.float32))
        assert list(ddp.state_dict().keys())[0].startswith("module.")
    finally:
        os.environ.pop("DPA_FORCE_NATIVE_REDUCER", None)
        dist.destroy_process_group()


def test_native_process_group_bootstraps_rccl_through_tcpstore():
    """--dist-backend native: TcpStore rendezvous -> ncclUniqueId -> C++ communicator, no torch.distributed."""
    import socket

    import torch.distributed as dist

    from distributed_pytorch_amd import amp, launch, models
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import DistributedDataParallel
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.parallel.comm import reduce_tensor
    from distributed_pytorch_amd.parallel.native_comm import NativeComm

    amp.state().reset()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"
    launch.init_process_group("native", "tcp://127.0.0.1:%d" % port, 1, 0, timeout_s=60, use_gpu=True)
    try:
        assert not dist.is_initialized() and pg.is_initialized() and pg.get_backend() == "native"
        assert isinstance(pg.native().gpu_comm, NativeComm)
        torch.manual_seed(0)
        model = models.create_model("resnet18", num_classes=10).cuda().train()
        ddp = DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=8.0, first_bucket_mb=1.0)
        assert ddp.comm is pg.native().gpu_comm and type(ddp.reducer).__name__ == "_NativeReducer"
        opt = FusedSGD(ddp.parameters(), lr=0.05, momentum=0.9)
        x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8, device="cuda")
        t = torch.randint(0, 10, (16,), device="cuda")
        losses = [_step(ddp, opt, x, t) for _ in range(6)]
        assert losses[-1] < losses[0], losses
        m = reduce_tensor(torch.tensor([1.0, 2.0, 3.0], device="cuda"))
        cpu = pg.all_reduce_sum(torch.tensor([4.0]))            # host tensors go through the store
        pg.barrier()
        assert m.tolist() == [1.0, 2.0, 3.0] and cpu.item() == 4.0
    finally:
        os.environ.pop("DPA_FORCE_NATIVE_REDUCER", None)
        pg.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_memory_allreduce_between_processes_sharing_the_gpu(world, tmp_path):
    """csrc/comm/p2p.hip: the small-message all-reduce (SyncBN statistics, packed metrics) that reads peers' slots
    directly instead of going through RCCL.  RCCL refuses two ranks on one device, hipIpc does not: W processes map
    each other's buffers on this one GPU and must agree on every sum."""
    import socket
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _dist_workers as W
    from distributed_pytorch_amd import launch

    for attempt in range(3):           # the probed port can be taken again before rank 0 binds it
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        try:
            launch.spawn(W.p2p_allreduce, world, (world, port, str(tmp_path)))
            break
        except launch.ProcessFailure as exc:
            if "Address already in use" not in str(exc) or attempt == 2:
                raise
    for r in range(world):
        out = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True)
        assert out["status"] == 0 and out["bad"] == [] and out["calls"] == 306 and out["routed"], out


# ---------------------------------------------------------------------------------------------------------------
# world > 1 on ONE GPU: the peer-memory communicator
# ---------------------------------------------------------------------------------------------------------------
def _spawn_shared(fn, world, args):
    import socket
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from distributed_pytorch_amd import launch

    for attempt in range(3):           # the probed port can be taken again before rank 0 binds it
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        try:
            launch.spawn(fn, world, (world, port) + tuple(args))
            return
        except launch.ProcessFailure as exc:
            if "Address already in use" not in str(exc) or attempt == 2:
                raise


@pytest.mark.parametrize("world", [2, 4])
def test_peer_communicator_collectives_between_ranks_sharing_the_gpu(world, tmp_path):
    """csrc/comm/peer.hip: two-shot all-reduce (sum / mean / max, fp32 and 16-bit wire, multi-pass), broadcast,
    all-gather and barrier between W processes on one device, against closed-form results."""
    import _gpu_dist_workers as G

    _spawn_shared(G.peer_collectives, world, (str(tmp_path),))
    for r in range(world):
        out = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True)
        assert out["bad"] == [] and out["status"] == 0, out
        assert out["passes"] >= 20 and out["p2p_calls"] >= 100, out     # big messages two-shot, metrics one-shot


@pytest.mark.parametrize("world,wire", [(2, "fp32"), (4, "fp32"), (2, "bf16")])
def test_gpu_ddp_over_peer_backend_matches_single_process_on_full_batch(world, wire, tmp_path):
    """imagenet_ddp.py:127 on real kernels with W > 1: arena views written by the backward kernels, C++ reducer,
    HIP events, re-plan after step 1, chunked slices.  All ranks must end bit-identical; against one process on the
    concatenated batch the gradients agree to accumulation-order tolerance (fp32 wire) / bf16 rounding (bf16 wire)."""
    import _gpu_dist_workers as G
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD

    _spawn_shared(G.ddp_gpu, world, (str(tmp_path), wire))
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    r0 = rs[0]
    assert r0["summary"]["backend"] == "peer-native" and r0["summary"]["reducer"] == "c++"
    assert r0["summary"]["ranks"] == world and r0["summary"]["grad_wire"] == wire and r0["status"] == 0
    assert r0["summary"]["buckets"] >= 3 and r0["summary"]["collectives"] > 3 * r0["summary"]["buckets"]   # chunked
    if wire == "bf16":
        assert r0["summary"]["bytes_per_step"] * 2 == r0["summary"]["bucket_bytes"]
    assert r0["replans"] in (0, 1)                 # 1 when the observed order differs from reverse registration
    for r in rs[1:]:
        assert r["plan"] == r0["plan"] and r["status"] == 0
        for n in r0["params"]:
            assert torch.equal(r["params"][n], r0["params"][n]), n         # every rank holds the same weights
        for it in range(3):
            for n in r0["grads"][it]:
                assert torch.equal(r["grads"][it][n], r0["grads"][it][n]), (it, n)
    # oracle: one process, the whole batch, the same kernels
    amp.state().reset()
    model = G._small_resnet().cuda().train()
    for m in model.modules():
        if hasattr(m, "running_mean"):
            m.eval()
    opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(99)
    x = torch.randint(0, 256, (3, 8 * world, 64, 64, 3), dtype=torch.uint8, generator=g)
    t = torch.randint(0, 10, (3, 8 * world), generator=g)
    rt, at = (2e-3, 2e-5) if wire == "fp32" else (3e-2, 2e-3)
    for it in range(3):
        loss, _ = F.loss_and_metrics(model(x[it].cuda()), t[it].cuda())
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            got, want = r0["grads"][it][n], p.grad.float().cpu()
            scale = float(want.abs().max()) + 1e-12
            err = float((got - want).abs().max())
            # step 0 starts from identical weights: summation order only.  Later steps compare two TRAJECTORIES
            # (momentum SGD at lr 0.05 on random labels amplifies a 1e-6 difference quickly): loose bound.
            ok = err <= at + rt * scale * (1, 8, 40)[it]
            assert ok, (it, n, err, scale)
        opt.step()


def test_gpu_syncbn_over_peer_backend_equals_batchnorm_on_the_global_batch(tmp_path):
    import _gpu_dist_workers as G
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.nn.modules import BatchNorm2d

    world = 2
    _spawn_shared(G.syncbn_gpu, world, (str(tmp_path),))
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    amp.state().reset()
    torch.manual_seed(7)
    x = torch.randn(4 * world, 6, 6, 32) * 2.0 + 3.0
    bn = BatchNorm2d(32).cuda().train()
    xg = x.clone().cuda().requires_grad_(True)
    w = torch.arange(32.0, device="cuda") / 8.0 - 1.0
    y = bn(xg)
    (y * w).sum().backward()
    ys = torch.cat([r["y"] for r in rs])
    dxs = torch.cat([r["dx"] for r in rs])
    assert torch.allclose(ys, y.detach().cpu(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(dxs, xg.grad.cpu(), rtol=1e-3, atol=1e-4)
    for r in rs:
        assert torch.allclose(r["rm"], bn.running_mean.cpu(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(r["rv"], bn.running_var.cpu(), rtol=1e-4, atol=1e-6)
    # parameter gradients are per-rank partial sums (DDP averages them): their sum is the global gradient
    assert torch.allclose(sum(r["dw"] for r in rs), bn.weight.grad.cpu(), rtol=1e-3, atol=1e-3)
    assert torch.allclose(sum(r["db"] for r in rs), bn.bias.grad.cpu(), rtol=1e-3, atol=1e-3)


def test_hip_graph_captures_the_ddp_step_with_its_all_reduce(tmp_path):
    """--hip-graph together with DDP (W = 2 sharing the GPU): the reducer's communication stream is captured as a fork
    / join of the step's graph.  Replayed losses == eager losses, ranks bit-identical, the graph really replayed."""
    import _gpu_dist_workers as G

    world = 2
    _spawn_shared(G.ddp_graph, world, (str(tmp_path),))
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    for r in rs:
        assert r["graph"]["replays"] >= 5 and r["eager"]["replays"] == 0, (r["graph"]["replays"],)
        for a, b in zip(r["eager"]["losses"], r["graph"]["losses"]):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (r["eager"]["losses"], r["graph"]["losses"])
    for n in rs[0]["graph"]["params"]:
        assert torch.equal(rs[0]["graph"]["params"][n], rs[1]["graph"]["params"][n]), n
        assert torch.allclose(rs[0]["graph"]["params"][n], rs[0]["eager"]["params"][n], rtol=5e-2, atol=5e-3), n


def test_dead_peer_ends_the_gpu_job_through_the_barrier_deadline_and_the_watchdog():
    """A rank that stops calling all-reduce: the survivor's barrier kernel gives up after DPA_PEER_TIMEOUT_S, the
    watchdog turns the communicator's error into exit code 86 and the supervisor tears the job down."""
    import time

    import _gpu_dist_workers as G
    from distributed_pytorch_amd import launch

    t0 = time.time()
    with pytest.raises(launch.ProcessFailure) as info:
        _spawn_shared(G.peer_hang, 2, ())
    assert time.time() - t0 < 75
    assert "86" in str(info.value) or "timed out" in str(info.value), str(info.value)


def test_bench_two_ranks_sharing_the_gpu_reports_the_multi_rank_path(tmp_path):
    """bench.py --gpus 2 --share-gpu: the driver's multi-GPU command shape on the one device this box has."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "4",
                        "--warmup", "3", "--batch-size", "32", "--arch", "resnet18"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    comm = out["config"]["comm"]
    assert out["config"]["ranks"] == 2 and comm["ranks"] == 2 and comm["backend"] == "peer-native"
    assert comm["reducer"] == "c++" and comm["collectives"] >= comm["buckets"] and comm["grad_wire"] == "bf16"
    assert comm["bytes_per_step"] * 2 == comm["bucket_bytes"] and out["final_loss"] is not None

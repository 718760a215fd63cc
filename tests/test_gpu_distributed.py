"""GPU-side checks of the data-parallel runtime that fit a 1-GPU box: gradient arena / sinks, the native
RCCL communicator and the C++ bucket reducer at world_size 1 (multi-rank logic is covered over gloo in
test_distributed_cpu.py; RCCL refuses two ranks on one device)."""
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

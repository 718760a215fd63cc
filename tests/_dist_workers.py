"""Worker functions for multi-process (gloo, CPU) tests; must be importable by spawned children."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port, backend="gloo"):
    from distributed_pytorch_amd import launch

    launch.init_process_group(backend, "tcp://127.0.0.1:%d" % port, world, rank, timeout_s=60, use_gpu=False)


def _finish():
    from distributed_pytorch_amd.parallel import process_group as pg

    pg.barrier()
    pg.destroy_process_group()


def _model(num_classes=10):
    from distributed_pytorch_amd import models

    torch.manual_seed(1234)
    return models.create_model("resnet18", num_classes=num_classes)


def ddp_grads(local_rank, world, port, out_dir, delay, backend="gloo"):
    """Each rank: its shard of a fixed global batch -> save averaged grads + params after one step."""
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.parallel import DistributedDataParallel

    _init(local_rank, world, port, backend)
    model = _model()
    if local_rank != 0:                     # DDP must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    ddp = DistributedDataParallel(model, bucket_cap_mb=8.0, first_bucket_mb=0.5, delay_allreduce=delay)
    opt = FusedSGD(ddp.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(8, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (8,), generator=g)
    per = 8 // world
    xs, ts = x[local_rank * per:(local_rank + 1) * per], t[local_rank * per:(local_rank + 1) * per]
    for m in model.modules():               # BN batch stats differ between sharded and full batch:
        if hasattr(m, "running_mean"):      # freeze them so DDP == single-process on the full batch
            m.eval()
    loss = F.cross_entropy(ddp(xs), ts)
    opt.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in ddp.named_parameters()}
    opt.step()
    torch.save({"grads": grads, "params": {n: p.detach().clone() for n, p in ddp.named_parameters()},
                "buckets": ddp.bucket_summary(), "keys": list(ddp.state_dict().keys())[:3]},
               os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def syncbn_stats(local_rank, world, port, out_dir, backend="gloo", offset=0.0):
    from distributed_pytorch_amd.nn.modules import BatchNorm2d
    from distributed_pytorch_amd.parallel import convert_syncbn_model

    _init(local_rank, world, port, backend)
    torch.manual_seed(7)
    x = torch.randn(8, 5, 5, 16) + offset          # offset = 1e3: |mean| = 1000 std (Welford / Chan territory)
    if offset:
        x[4:] += 0.5                                # the two ranks' local means differ: the merge term matters
    bn = convert_syncbn_model(torch.nn.Sequential(BatchNorm2d(16)))[0]
    per = 8 // world
    xs = x[local_rank * per:(local_rank + 1) * per].clone().requires_grad_(True)
    y = bn(xs)
    (y * torch.arange(16.0)).sum().backward()
    torch.save({"y": y.detach(), "dx": xs.grad, "rm": bn.running_mean.clone(), "rv": bn.running_var.clone(),
                "dw": bn.weight.grad.clone()}, os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def crash_one(local_rank, world, port):
    """Rank 1 dies hard mid-run; the others would block in a collective forever."""
    from distributed_pytorch_amd import launch

    _init(local_rank, world, port)
    t = torch.ones(4)
    dist.all_reduce(t)
    launch.maybe_fault_inject("1:0", local_rank, 0)
    dist.all_reduce(t)          # never completes on the survivors without the supervisor
    import time

    time.sleep(120)


def store_ops(local_rank, world, port, out_dir):
    """TcpStore semantics across processes: blocking get, counters, barrier, host collectives."""
    from distributed_pytorch_amd.parallel import process_group as pg

    _init(local_rank, world, port, "native")
    npg = pg.native()
    st = npg.store
    if local_rank == 1:
        import time

        time.sleep(0.3)
        st.set("from1", b"payload")
    got = st.get("from1")                              # rank 0 blocks until rank 1 publishes
    n = st.add("counter", local_rank + 1)
    pg.barrier()
    total = st.add("counter", 0)
    t = torch.arange(6, dtype=torch.float32) * (local_rank + 1)
    pg.all_reduce_sum(t)
    b = torch.full((3,), float(local_rank))
    pg.broadcast(b, src=1)
    obj = npg.broadcast_object({"rank": local_rank} if local_rank == 0 else None, src=0)
    gathered = torch.empty(world, 2)
    npg.host_comm.all_gather(gathered, torch.tensor([float(local_rank), 10.0 + local_rank]))
    torch.save({"got": got, "n": n, "total": total, "sum": t, "bcast": b, "obj": obj, "gathered": gathered,
                "rank": pg.get_rank(), "world": pg.get_world_size(), "backend": pg.get_backend()},
               os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def p2p_allreduce(local_rank, world, port, out_dir):
    """One-shot peer-memory all-reduce between processes that share ONE GPU (hipIpc mappings of each other's
    fine-grained slot buffers): sizes from 1 float to the slot capacity, then 300 calls queued back to back without
    a host sync in between (slot parity reuse)."""
    from distributed_pytorch_amd import backend

    ext = backend.require()
    torch.cuda.set_device(0)
    store = ext.TcpStore("127.0.0.1", int(port), local_rank == 0, 60.0)
    peer = ext.PeerAllReduce(local_rank, world, 0, 16384, 20.0)
    store.set("h/%d" % local_rank, bytes(peer.handle()))
    peer.open_peers([bytes(store.get("h/%d" % r)) for r in range(world)])
    store.barrier("opened", world)
    stream = torch.cuda.current_stream().cuda_stream
    scale = world * (world + 1) / 2.0
    results = []
    for it, n in enumerate([1, 7, 1024, 4097, 16384, 5] + [129] * 300):
        base = torch.arange(n, device="cuda", dtype=torch.float32) % 97 + float(it % 50)
        t = base * float(local_rank + 1)
        peer.all_reduce(t.data_ptr(), n, stream)
        results.append((t, base * scale))
        if it < 6:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    bad = [i for i, (t, want) in enumerate(results) if not torch.equal(t, want)]

    # the same path as NativeComm wires it up (handle exchange, agreement, self-test, routing by size / dtype);
    # only the agreement vote is replaced, because RCCL itself cannot run several ranks on one device
    import types

    from distributed_pytorch_amd.parallel import native_comm
    from distributed_pytorch_amd.parallel.process_group import StoreComm

    class _VoteThroughStore(native_comm.NativeComm):
        def _agree(self, ok):
            return all(p == b"1" for p in self._exchange(b"1" if ok else b"0"))

    comm = _VoteThroughStore(types.SimpleNamespace(world_size=world, rank=local_rank), None,
                             StoreComm(store, local_rank, world)._exchange)
    v = torch.full((2 * 2048 + 1,), float(local_rank + 1), device="cuda")        # SyncBN message of a 2048-ch layer
    comm.all_reduce_sum(v, async_op=False)
    torch.cuda.synchronize()
    routed = comm.p2p_calls == 1 and bool(torch.equal(v, torch.full_like(v, scale))) and native_comm.shared() is comm
    torch.save({"bad": bad, "status": int(peer.status()), "calls": len(results), "routed": routed},
               os.path.join(out_dir, "rank%d.pt" % local_rank))
    store.barrier("done", world)
    del peer, comm


class OutOfOrderNet(torch.nn.Module):
    """Registration order (last, first, mid) is NOT the backward order (last, mid, first): the reducer's first plan,
    made from reverse registration order, is wrong and must be re-planned from the observed order."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.last = torch.nn.Linear(48, 6)
        self.first = torch.nn.Linear(20, 64)
        self.mid = torch.nn.Linear(64, 48)
        self.unused = torch.nn.Linear(3, 3)         # never touched by forward: reduced as zeros

    def forward(self, x):
        return self.last(torch.relu(self.mid(torch.relu(self.first(x)))))


def ddp_native_reducer(local_rank, world, port, out_dir, wire="fp32"):
    """The C++ BucketReducer over the host backend (TcpStore + worker thread) at world >= 2: chunked buckets,
    re-planning from the observed ready order, an unused parameter, three optimizer steps."""
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import DistributedDataParallel

    _init(local_rank, world, port, "native")
    model = OutOfOrderNet()
    ddp = DistributedDataParallel(model, bucket_cap_mb=0.008, first_bucket_mb=0.001, last_bucket_mb=0.0,
                                  chunk_mb=0.004, collective_timeout_s=30.0, grad_wire=wire, min_bucket_mb=0.0)
    summary0 = ddp.comm_summary()
    plan0 = [list(b.spec.param_ids) for b in ddp.arena.buckets]
    opt = FusedSGD(ddp.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 4 * world, 20, generator=g)
    t = torch.randn(3, 4 * world, 6, generator=g)
    grads = []
    for it in range(3):
        xs = x[it, local_rank * 4:(local_rank + 1) * 4]
        ts = t[it, local_rank * 4:(local_rank + 1) * 4]
        loss = ((ddp(xs) - ts) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
        opt.step()
    torch.save({"grads": grads, "params": {n: p.detach().clone() for n, p in model.named_parameters()},
                "summary0": summary0, "summary": ddp.comm_summary(), "plan0": plan0,
                "plan": [list(b.spec.param_ids) for b in ddp.arena.buckets], "replans": ddp.replans},
               os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def reducer_hang(local_rank, world, port):
    """Rank 1 stops calling all-reduce (it skips a backward and idles); rank 0's watchdog must abort and fail the
    rank within the collective timeout instead of waiting forever."""
    import time

    from distributed_pytorch_amd.parallel import DistributedDataParallel

    _init(local_rank, world, port, "native")
    model = OutOfOrderNet()
    ddp = DistributedDataParallel(model, collective_timeout_s=2.0, replan=False)
    x = torch.randn(4, 20)
    ddp(x).sum().backward()                  # a healthy step first
    if local_rank == 1:
        time.sleep(60)                       # never calls all-reduce again (the supervisor ends it)
        return
    ddp(x).sum().backward()                  # rank 0: blocks in finalize until the watchdog fires
    time.sleep(60)


def ddp_self_diagnosis(local_rank, world, port, out_dir, backend="native"):
    """Link probe at construction + per-bucket ready->done / exposed-wait stamps of steady steps (both reducers)."""
    from distributed_pytorch_amd.parallel import DistributedDataParallel

    os.environ["DPA_COMM_PROBE"] = "1"
    os.environ["DPA_COMM_PROBE_SCALE"] = "0.002"          # 128 KiB and 32 KiB instead of 64 / 16 MiB
    _init(local_rank, world, port, backend)
    model = OutOfOrderNet()
    ddp = DistributedDataParallel(model, bucket_cap_mb=0.008, first_bucket_mb=0.001, last_bucket_mb=0.0,
                                  collective_timeout_s=30.0, min_bucket_mb=0.0)
    x = torch.randn(4, 20)
    states = []
    for _ in range(8):
        ddp.zero_grad()
        ddp(x).sum().backward()
        states.append(ddp._timing_state)
    torch.save({"summary": ddp.comm_summary(), "states": states, "buckets": len(ddp.arena.buckets)},
               os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()

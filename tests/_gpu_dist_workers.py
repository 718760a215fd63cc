"""Workers of the shared-GPU multi-rank tests: W processes on ONE MI355X, talking through the peer-memory
communicator (csrc/comm/peer.hip).  Must be importable by spawned children."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port, window_mb=None, timeout_s=None):
    from distributed_pytorch_amd import launch

    os.environ["DPA_COMM"] = "peer"
    os.environ["DPA_SHARE_GPU"] = "1"
    if window_mb is not None:
        os.environ["DPA_PEER_WINDOW_MB"] = str(window_mb)
    if timeout_s is not None:
        os.environ["DPA_PEER_TIMEOUT_S"] = str(timeout_s)
    torch.cuda.set_device(launch.device_index(rank))
    launch.init_process_group("native", "tcp://127.0.0.1:%d" % port, world, rank, timeout_s=60, use_gpu=True)


def _finish():
    from distributed_pytorch_amd.parallel import process_group as pg

    pg.barrier()
    pg.destroy_process_group()


def peer_collectives(local_rank, world, port, out_dir):
    """Every collective of the peer communicator against closed-form results: vector bodies and ragged tails,
    unaligned tensors, messages longer than the window (multi-pass), 16-bit wire and 16-bit storage."""
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.parallel.peer_comm import PeerNativeComm

    _init(local_rank, world, port, window_mb=1)            # 1 MiB window: 300k floats need two passes
    comm = pg.native().gpu_comm
    assert isinstance(comm, PeerNativeComm) and comm.world_size == world
    bad = []
    tri = world * (world + 1) / 2.0
    for n in (1, 3, 64, 1000, 4099, 300001):
        base = torch.arange(n, device="cuda", dtype=torch.float32) % 113 - 50.0
        t = base * float(local_rank + 1)
        comm.all_reduce_sum(t, async_op=False)
        if not torch.equal(t, base * tri):
            bad.append(("sum", n))
        t = base * float(local_rank + 1)
        comm.all_reduce_mean(t, async_op=False)
        if not torch.allclose(t, base * (tri / world), rtol=1e-6, atol=0):
            bad.append(("mean", n))
        # a view that starts 4 bytes into an allocation: the scalar path
        buf = torch.zeros(n + 1, device="cuda")
        u = buf[1:]
        u.copy_(base * float(local_rank + 1))
        comm.all_reduce_sum(u, async_op=False)
        if not torch.equal(u, base * tri):
            bad.append(("unaligned", n))
    # bf16 on the wire, fp32 in storage: small integers are exact in bf16, and every rank must hold the same bits
    for n in (8, 1000, 70001):
        base = torch.arange(n, device="cuda", dtype=torch.float32) % 31 - 15.0
        t = base * float(local_rank + 1)
        comm.all_reduce_mean(t, async_op=False, wire=1)
        want = (base * (tri / world)).to(torch.bfloat16).float()
        if not torch.equal(t, want):
            bad.append(("wire-bf16", n))
    r = torch.randn(50000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5 + local_rank))
    mine = r.clone()
    comm.all_reduce_mean(r, async_op=False, wire=2)
    gathered = torch.empty(world, 50000, device="cuda")
    comm.all_gather(gathered, mine)
    want = gathered.half().float().sum(0) / world
    if not torch.allclose(r, want.half().float(), rtol=2e-3, atol=1e-6):
        bad.append(("wire-fp16-random", float((r - want).abs().max())))
    allr = torch.empty(world, 50000, device="cuda")
    comm.all_gather(allr, r)
    if not all(torch.equal(allr[0], allr[k]) for k in range(world)):
        bad.append(("wire-fp16 ranks differ",))
    # 16-bit storage (amp O3 buckets)
    hb = (torch.arange(5000, device="cuda") % 9).to(torch.bfloat16) * (local_rank + 1)
    comm.all_reduce_mean(hb, async_op=False)
    if not torch.equal(hb.float(), ((torch.arange(5000, device="cuda") % 9).float() * (tri / world)).bfloat16().float()):
        bad.append(("bf16 storage",))
    # broadcast (int64, 2.4 MB = three window passes), all_gather, max, barrier
    b = torch.arange(300000, device="cuda", dtype=torch.int64) * (local_rank + 7)
    comm.broadcast(b, src=world - 1)
    if not torch.equal(b, torch.arange(300000, device="cuda", dtype=torch.int64) * (world - 1 + 7)):
        bad.append(("broadcast",))
    g = torch.empty(world, 3, device="cuda")
    comm.all_gather(g, torch.tensor([local_rank, 2.0 * local_rank, -1.0], device="cuda"))
    if g.tolist() != [[float(k), 2.0 * k, -1.0] for k in range(world)]:
        bad.append(("all_gather", g.tolist()))
    m = torch.tensor([float(local_rank), -float(local_rank)], device="cuda")
    comm.all_reduce_max(m)
    if m.tolist() != [float(world - 1), 0.0]:
        bad.append(("max", m.tolist()))
    # the facade (metrics path) and many collectives queued back to back without a host sync
    ts = [torch.full((257,), float(local_rank + 1 + i), device="cuda") for i in range(100)]
    for t in ts:
        pg.all_reduce_sum(t)
    torch.cuda.synchronize()
    for i, t in enumerate(ts):
        if not torch.equal(t, torch.full_like(t, tri + world * i)):
            bad.append(("queued", i))
            break
    comm.barrier()
    torch.save({"bad": bad, "status": int(comm.raw.status()), "passes": int(comm.raw.passes),
                "p2p_calls": int(comm.p2p_calls)}, os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def _small_resnet(num_classes=10):
    from distributed_pytorch_amd import models

    torch.manual_seed(1234)
    return models.create_model("resnet18", num_classes=num_classes)


def ddp_gpu(local_rank, world, port, out_dir, wire, graph=False):
    """The whole GPU data-parallel path with W > 1: kernels write gradients into the arena views, the C++ reducer
    issues the peer all-reduce per bucket slice behind HIP events, buckets are re-planned after the first step and
    the big ones travel in chunks.  Each rank trains on its shard of a fixed global batch for three steps."""
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import DistributedDataParallel
    from distributed_pytorch_amd.parallel import process_group as pg

    _init(local_rank, world, port, window_mb=8)
    amp.state().reset()
    model = _small_resnet().cuda().train()
    if local_rank != 0:                     # DDP must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.25)
    for m in model.modules():               # BN batch statistics differ between a shard and the full batch:
        if hasattr(m, "running_mean"):      # freeze them so DDP == single process on the full batch
            m.eval()
    ddp = DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=8.0, first_bucket_mb=0.5, last_bucket_mb=0.25,
                                  chunk_mb=2.0, collective_timeout_s=60.0, grad_wire=wire)
    opt = FusedSGD(ddp.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(99)
    per = 8
    x = torch.randint(0, 256, (3, per * world, 64, 64, 3), dtype=torch.uint8, generator=g)
    t = torch.randint(0, 10, (3, per * world), generator=g)
    grads, losses = [], []
    plan0 = [list(b.spec.param_ids) for b in ddp.arena.buckets]

    def one(it):
        xs = x[it, local_rank * per:(local_rank + 1) * per].cuda()
        ts = t[it, local_rank * per:(local_rank + 1) * per].cuda()
        loss, _ = F.loss_and_metrics(ddp(xs), ts)
        opt.zero_grad()
        loss.backward()
        return loss

    for it in range(3):
        loss = one(it)
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()})
        losses.append(float(loss))
        opt.step()
    torch.cuda.synchronize()
    pg.check_health()
    torch.save({"grads": grads, "losses": losses,
                "params": {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()},
                "summary": ddp.comm_summary(), "plan0": plan0,
                "plan": [list(b.spec.param_ids) for b in ddp.arena.buckets], "replans": ddp.replans,
                "status": int(ddp.comm.raw.status())}, os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def syncbn_gpu(local_rank, world, port, out_dir):
    """SyncBN over W ranks on the GPU kernels == plain BatchNorm over the concatenated batch."""
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.nn.modules import BatchNorm2d
    from distributed_pytorch_amd.parallel import convert_syncbn_model

    _init(local_rank, world, port)
    amp.state().reset()
    torch.manual_seed(7)
    per = 4
    x = torch.randn(per * world, 6, 6, 32) * 2.0 + 3.0
    bn = convert_syncbn_model(torch.nn.Sequential(BatchNorm2d(32)))[0].cuda().train()
    xs = x[local_rank * per:(local_rank + 1) * per].clone().cuda().requires_grad_(True)
    w = torch.arange(32.0, device="cuda") / 8.0 - 1.0
    y = bn(xs)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    torch.save({"y": y.detach().cpu(), "dx": xs.grad.cpu(), "rm": bn.running_mean.cpu().clone(),
                "rv": bn.running_var.cpu().clone(), "dw": bn.weight.grad.cpu().clone(),
                "db": bn.bias.grad.cpu().clone()}, os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()


def peer_hang(local_rank, world, port):
    """Rank 1 stops calling all-reduce: rank 0's barrier kernel gives up at its deadline (it never hangs), the
    reducer's watchdog sees the communicator's error and fails the rank -- the supervisor then ends the job."""
    import time

    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.parallel import DistributedDataParallel

    _init(local_rank, world, port, timeout_s=3)
    amp.state().reset()
    model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).cuda()
    ddp = DistributedDataParallel(model, device_ids=[0], collective_timeout_s=6.0, replan=False, grad_wire="fp32")
    x = torch.randn(4, 64, device="cuda")
    ddp(x).sum().backward()                  # a healthy step first
    torch.cuda.synchronize()
    if local_rank == 1:
        time.sleep(90)                       # never calls all-reduce again (the supervisor ends it)
        return
    ddp(x).sum().backward()                  # rank 0: the barrier times out on the device, the watchdog fails the rank
    torch.cuda.synchronize()
    time.sleep(90)


def ddp_graph(local_rank, world, port, out_dir):
    """--hip-graph with DDP at W > 1: the captured step (forward, backward, per-bucket peer all-reduce on the
    communication stream, fused SGD) must reproduce the eager losses, and every rank must stay bit-identical."""
    from distributed_pytorch_amd import amp
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import DistributedDataParallel
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.utils.graph import GraphedTrainStep

    _init(local_rank, world, port, window_mb=16)
    out = {}
    g = torch.Generator().manual_seed(123)
    per = 8
    x = torch.randint(0, 256, (8, per * world, 64, 64, 3), dtype=torch.uint8, generator=g)
    t = torch.randint(0, 10, (8, per * world), generator=g)
    for mode in ("eager", "graph"):
        amp.state().reset()
        model = _small_resnet().cuda().train()
        opt = FusedSGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
        ddp = DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=8.0, first_bucket_mb=0.5,
                                      last_bucket_mb=0.25, broadcast_buffers=False, collective_timeout_s=60.0)
        step = GraphedTrainStep(ddp, opt, warmup=2, enabled=(mode == "graph"))
        losses = []
        for it in range(8):
            xs = x[it, local_rank * per:(local_rank + 1) * per].cuda()
            ts = t[it, local_rank * per:(local_rank + 1) * per].cuda()
            loss, _ = step(xs, ts)
            losses.append(float(loss.detach().float().item()))
        torch.cuda.synchronize()
        pg.check_health()
        out[mode] = {"losses": losses, "replays": step.replays,
                     "params": {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()},
                     "collectives": int(ddp.reducer.collectives)}
        del step, ddp, opt, model
    torch.save(out, os.path.join(out_dir, "rank%d.pt" % local_rank))
    _finish()

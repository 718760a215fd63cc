#!/usr/bin/env python
"""Headline benchmark: ResNet-50 mixed-precision (bf16) training throughput, images/sec, whole job.

    python bench.py --gpus N --steps 20 --warmup 5          # starts N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Like the reference (``mp.spawn(main_worker, nprocs=ngpus)``, imagenet_ddp.py:76-85) the plain command launches
its own workers: fresh spawn-context children started before the parent touches a GPU, supervised, and the
command fails fast when the node has fewer than N devices.  Under torchrun the ranks are taken from the
environment.  Gradients are all-reduced by the native C++ bucketed reducer calling RCCL directly, overlapped
with backward; the rendezvous is the native TcpStore -- ``torch.distributed`` is not initialised, so each rank
owns exactly ONE RCCL communicator (``--dist-backend nccl`` switches to torch's process group).

Every timed step does the full training iteration of the reference's AMP script
(imagenet_ddp_apex.py:367-425): async H2D of a fresh uint8 batch from pinned memory + on-GPU
normalise (prefetcher), forward, loss, zero_grad, backward (+all-reduce), fused SGD step.
Synthetic data, random-init weights (no network / datasets on the benchmark machines).
Throughput formula = the reference's: world_size * batch / batch_time (imagenet_ddp_apex.py:411).
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    # BASELINE.json north star: "per-GPU batch is sized to use the 288 GB HBM".  Measured on one MI355X (bf16, end of
    # round 3): 256 -> 12.7k, 512 -> 15.0k, 1024 -> 16.0k (39 GB), 2048 -> 16.5k (77 GB), 4096 -> 16.8k img/s (154 GB);
    # the reference's 224 was a 16 GB-GPU limit (imagenet_ddp_apex.py:63-67).  Rounds 1-2 and most of the round-3 tables
    # were measured at 1024 (`--batch-size 1024` reproduces them).  LR follows the reference's 0.1 * global_batch / 256.
    ap.add_argument("--batch-size", type=int, default=None, help="per-GPU batch (weak scaling); default 2048")
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--num-classes", type=int, default=1000)
    ap.add_argument("--bucket-mb", type=float, default=None)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("DPA_BENCH_GRAPH", "0")))
    ap.add_argument("--force-ddp", action="store_true",
                    help="1-GPU rehearsal of the multi-GPU path: process group of one rank + DDP wrapper + RCCL reducer")
    ap.add_argument("--dist-backend", default=os.environ.get("DPA_BENCH_BACKEND", "native"),
                    help="native (TcpStore + C++ RCCL communicator, default) | nccl (torch.distributed) | gloo")
    ap.add_argument("--comm", default=os.environ.get("DPA_COMM", "rccl"), choices=["rccl", "peer", "auto", "torch"],
                    help="rccl (default): C++ reducer calling RCCL; peer: two-shot all-reduce over hipIpc windows "
                         "(csrc/comm/peer.hip), the only engine that lets ranks share a device")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on the visible device(s), wrapping around: rehearses the N-rank GPU path on one "
                         "MI355X (implies --comm peer; per-GPU numbers are NOT comparable to real N-GPU runs)")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16", "fp16"],
                    help="dtype the fp32 gradient buckets travel in (auto = AMP dtype under O2, like Apex)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu = rehearsal of the launch / rendezvous / reducer logic without a GPU (tests)")
    args = ap.parse_args(argv)
    if args.batch_size is None:
        args.batch_size = 2048 if args.device == "cuda" else 8
    if args.dtype is None:
        args.dtype = "bf16" if args.device == "cuda" else "fp32"
    if args.share_gpu:
        args.comm = "peer"
    return args


def _spawned_rank(local_rank, args):
    os.environ["RANK"] = str(local_rank)
    run(args)


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks here."""
    from distributed_pytorch_amd import launch

    have = launch.device_count() if args.device == "cuda" else int(os.environ.get("DPA_NPROC_PER_NODE", args.gpus))
    if args.share_gpu and have >= 1:
        have = args.gpus
    if have < args.gpus:
        print("bench.py: --gpus %d requested but this node has %d GPU(s)" % (args.gpus, have), file=sys.stderr)
        return 2
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(launch.free_port())
    os.environ["WORLD_SIZE"] = str(args.gpus)
    try:
        launch.spawn(_spawned_rank, args.gpus, (args,))
    except launch.ProcessFailure as exc:
        print("bench.py: %s" % exc, file=sys.stderr)
        return 1
    return 0


_REAL_STDOUT = None


def _claim_stdout() -> None:
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL's version banner at
    communicator creation: 'RCCL version : ... / Librccl path : ...'), so file descriptor 1 is pointed at stderr for
    the whole run and the JSON line goes to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line: str) -> None:
    sys.stdout.flush()
    data = (line + "\n").encode()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        data = data[os.write(fd, data):]


def run(args) -> None:
    _claim_stdout()
    import faulthandler

    faulthandler.enable()          # a native crash in one rank leaves its Python stack on stderr
    import torch

    from distributed_pytorch_amd import amp, launch, models
    from distributed_pytorch_amd.data.datasets import SyntheticImageNet
    from distributed_pytorch_amd.data.prefetcher import StreamPrefetcher
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD
    from distributed_pytorch_amd.parallel import process_group as pg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    use_gpu = args.device == "cuda"
    copy_stream = None
    if args.share_gpu:
        os.environ["DPA_SHARE_GPU"] = "1"
    if args.comm == "peer":
        os.environ["DPA_COMM"] = "peer"
    elif args.comm in ("rccl", "auto"):
        os.environ.pop("DPA_COMM", None)
    if use_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU")
        dev_index = launch.device_index(local_rank)
        if torch.cuda.device_count() <= dev_index:
            raise SystemExit("rank %d: local rank %d but only %d GPU(s) visible"
                             % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        # the input copy stream is created first so it gets a hardware queue of its own (see data.prefetcher.copy_stream)
        from distributed_pytorch_amd.data.prefetcher import copy_stream as _copy_stream

        copy_stream = _copy_stream(device) if os.environ.get("DPA_EARLY_COPY_STREAM", "1") == "1" \
            else torch.cuda.Stream(device)
    else:
        device = torch.device("cpu")
    cpus = launch.set_affinity_from_env(use_gpu=use_gpu) if world > 1 else []

    ddp = world > 1 or args.force_ddp
    backend = None
    # A rendezvous / communicator build that never returns (a rank that died before joining, a wedged collective) must
    # end the job with a message instead of hanging the launcher: hard exit after DPA_BENCH_INIT_TIMEOUT_S.
    # The limit is re-armed per phase (rendezvous + wrap, then warm-up), and the extension is loaded before the clock
    # starts, so a cold build cache or a slow architecture at a large batch does not count against it.
    init_done = threading.Event()
    phase = {"name": "process-group / model set-up", "deadline": None}
    if use_gpu:
        from distributed_pytorch_amd import backend as _backend

        _backend.require()
    if ddp:
        limit = float(os.environ.get("DPA_BENCH_INIT_TIMEOUT_S", "600"))
        phase["deadline"] = time.monotonic() + limit

        def _init_watchdog():
            while not init_done.wait(1.0):
                if time.monotonic() > phase["deadline"]:
                    print("bench.py: rank %d: %s did not finish within %.0f s -- aborting"
                          % (rank, phase["name"], limit), file=sys.stderr, flush=True)
                    os._exit(3)

        threading.Thread(target=_init_watchdog, daemon=True).start()
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
            os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"      # a single rank would otherwise skip the reducer
        backend = args.dist_backend
        timeout_s = float(os.environ.get("DPA_DIST_TIMEOUT_S", "300"))
        try:
            pg.init_process_group(backend, "env://", world, rank, timeout_s=timeout_s, use_gpu=use_gpu)
        except RuntimeError as exc:
            # every rank learns of a failed native communicator through the store (process_group.py), so all of
            # them arrive here together and fall back to torch's process group
            if backend not in pg.NATIVE_NAMES or "native RCCL communicator" not in str(exc):
                raise
            if rank == 0:
                print("bench.py: %s -- falling back to torch.distributed" % exc, file=sys.stderr)
            pg.destroy_process_group()
            backend = "nccl"
            pg.init_process_group(backend, "env://", world, rank, timeout_s=timeout_s, use_gpu=use_gpu)

    torch.manual_seed(1234)
    model = models.create_model(args.arch, num_classes=args.num_classes).to(device)
    # reference recipe (imagenet_ddp_apex.py:162,527-543): peak lr = 0.1 * global_batch / 256 reached after a
    # 5-epoch linear warm-up; a benchmark of a few dozen steps sits at the start of that ramp (1 % of the peak),
    # which also keeps random-init / random-label training finite at global batches of 8k
    lr = 0.1 * args.batch_size * world / 256.0 * 0.01
    opt = FusedSGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    if args.dtype == "fp32":
        model, opt = amp.initialize(model, opt, opt_level="O0")
    else:
        model, opt = amp.initialize(model, opt, opt_level="O2", dtype=args.dtype)
    if args.sync_bn and world > 1:
        from distributed_pytorch_amd.parallel import convert_syncbn_model

        model = convert_syncbn_model(model)
    if ddp:
        from distributed_pytorch_amd.parallel import DistributedDataParallel

        kw = {} if args.bucket_mb is None else {"bucket_cap_mb": args.bucket_mb}
        # Apex DDP (the reference's AMP script) does not re-broadcast BN buffers every forward
        model = DistributedDataParallel(model, device_ids=[device.index] if use_gpu else None,
                                        broadcast_buffers=False, grad_wire=args.grad_wire,
                                        probe=True if (args.force_ddp and world == 1) else None, **kw)
    model.train()
    if ddp:
        phase["name"], phase["deadline"] = "warm-up (first collectives)", time.monotonic() + limit

    total = args.steps + args.warmup + 2
    data = SyntheticImageNet(args.batch_size, world_size=world, rank=rank, image_size=args.image_size,
                             steps=total, num_classes=args.num_classes)
    base = getattr(model, "module", model)
    pre = StreamPrefetcher(data, device, dtype=amp.compute_dtype(torch.float32), pad_c=8 if use_gpu else 0,
                           s2d=bool(getattr(base, "stem_s2d", False)) and use_gpu, stream=copy_stream)

    graphed = None
    if args.graph and use_gpu:
        from distributed_pytorch_amd.utils.graph import GraphedTrainStep

        graphed = GraphedTrainStep(model, opt, warmup=2)

    def step():
        images, target = pre.next()
        if graphed is not None:
            return graphed(images, target)[0]
        out = model(images)
        loss, rec = F.loss_and_metrics(out, target)
        opt.zero_grad()
        with amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()
        return loss

    def fence():
        if use_gpu:
            torch.cuda.synchronize()
        if ddp:
            pg.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    if ddp:
        model.finish_diagnosis()       # no time stamps (host waits) inside the timed region
    fence()
    init_done.set()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float32, device=device)
    if ddp:
        pg.all_reduce_max(t)               # the slowest rank's wall time
    dt = float(t.item())
    final_loss = float(loss.item())
    if final_loss != final_loss or final_loss in (float("inf"), float("-inf")):
        final_loss = None               # keep the line strict JSON
    n_ranks = pg.get_world_size() if ddp else 1       # what the communicator saw, not what the flag asked for
    ips = n_ranks * args.batch_size * args.steps / dt
    comm = None
    if ddp:
        comm = model.comm_summary()
        comm["process_group"] = pg.get_backend()
        raw = getattr(model.comm, "raw", None)
        comm["rccl_ranks"] = int(raw.world_size) if raw is not None else None
        if raw is not None:
            from distributed_pytorch_amd import backend as _be

            comm["rccl_version"] = int(_be.require().rccl_version())
        comm["cpu_affinity"] = len(cpus)
        comm["nccl_env"] = {k: os.environ.get(k) for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_ALGO",
                                                           "NCCL_PROTO") if os.environ.get(k) is not None}
    if rank == 0:
        _emit(json.dumps({
            "metric": "images/sec (whole job) %s %s training" % (
                {"resnet50": "ResNet-50"}.get(args.arch, args.arch), "AMP" if args.dtype != "fp32" else "fp32"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": (min(n_ranks, torch.cuda.device_count()) if args.share_gpu else n_ranks) if use_gpu else 0, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic uint8 %dx%d from pinned host memory, random-init weights" % (args.image_size, args.image_size),
            "final_loss": final_loss,
            "config": {"model": args.arch, "global_batch": args.batch_size * n_ranks,
                       "per_gpu_batch": args.batch_size, "seq_len": None, "image_size": args.image_size,
                       "parallelism": "dp%d" % n_ranks + (" (forced DDP path)" if args.force_ddp and world == 1 else "")
                                      + (" (ranks share one GPU)" if args.share_gpu and world > 1 else ""),
                       "ranks": n_ranks, "device": args.device,
                       "optimizer": "sgd+momentum+wd (fused)",
                       "amp": "O2 " + args.dtype if args.dtype != "fp32" else "O0",
                       "hip_graph": bool(graphed is not None and graphed.replays > 0),
                       "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1) if use_gpu else None,
                       "comm": comm},
        }))
    if ddp:
        pg.barrier()
        pg.destroy_process_group()


def main(argv=None) -> int:
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())

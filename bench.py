#!/usr/bin/env python
"""Headline benchmark: ResNet-50 mixed-precision (bf16) training throughput, images/sec, whole job.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; gradients are all-reduced by the native bucketed RCCL reducer overlapped with
backward.  Every timed step does the full training iteration of the reference's AMP script
(imagenet_ddp_apex.py:367-425): async H2D of a fresh uint8 batch from pinned memory + on-GPU
normalise (prefetcher), forward, loss, zero_grad, backward (+all-reduce), fused SGD step.
Synthetic data, random-init weights (no network / datasets on the benchmark machines).
Throughput formula = the reference's: world_size * batch / batch_time (imagenet_ddp_apex.py:411).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    # BASELINE.json north star: "per-GPU batch is sized to use the 288 GB HBM".  Measured on one MI355X (bf16):
    # 256 -> 9.46k, 512 -> 10.27k, 1024 -> 10.74k, 2048 -> 10.96k img/s (41 GB at 1024); the reference's 224 was a
    # 16 GB-GPU limit (imagenet_ddp_apex.py:63-67).  LR follows the reference's 0.1 * global_batch / 256 rule.
    ap.add_argument("--batch-size", type=int, default=1024, help="per-GPU batch (weak scaling)")
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--bucket-mb", type=float, default=None)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("DPA_BENCH_GRAPH", "0")))
    ap.add_argument("--force-ddp", action="store_true",
                    help="1-GPU rehearsal of the multi-GPU path: process group of one rank + DDP wrapper + RCCL reducer")
    args = ap.parse_args()

    import torch.distributed as dist

    import distributed_pytorch_amd as dpa
    from distributed_pytorch_amd import amp, models
    from distributed_pytorch_amd.data.datasets import SyntheticImageNet
    from distributed_pytorch_amd.data.prefetcher import StreamPrefetcher
    from distributed_pytorch_amd.ops import functional as F
    from distributed_pytorch_amd.optim import FusedSGD

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # the input copy stream is created first so it gets a hardware queue of its own (see data.prefetcher.copy_stream)
    from distributed_pytorch_amd.data.prefetcher import copy_stream as _copy_stream

    copy_stream = _copy_stream(device) if os.environ.get("DPA_EARLY_COPY_STREAM", "1") == "1" else torch.cuda.Stream(device)
    ddp = world > 1 or args.force_ddp
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ["DPA_FORCE_NATIVE_REDUCER"] = "1"      # a single rank would otherwise skip the reducer
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank,
                                device_id=device)

    torch.manual_seed(1234)
    model = models.create_model(args.arch).to(device)
    # reference recipe (imagenet_ddp_apex.py:162,527-543): peak lr = 0.1 * global_batch / 256 reached after a
    # 5-epoch linear warm-up; a benchmark of a few dozen steps sits at the start of that ramp (1 % of the peak),
    # which also keeps random-init / random-label training finite at global batches of 8k
    lr = 0.1 * args.batch_size * world / 256.0 * 0.01
    if F._TORCH_BASELINE:   # stock PyTorch-ROCm reference point (MIOpen/rocBLAS/ATen + torch.optim.SGD)
        opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    else:
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
        model = DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False, **kw)
    model.train()

    total = args.steps + args.warmup + 2
    data = SyntheticImageNet(args.batch_size, world_size=world, rank=rank, image_size=args.image_size,
                             steps=total)
    base = getattr(model, "module", model)
    pre = StreamPrefetcher(data, device, dtype=amp.compute_dtype(torch.float32),
                           pad_c=0 if F._TORCH_BASELINE else 8,
                           s2d=bool(getattr(base, "stem_s2d", False)) and not F._TORCH_BASELINE,
                           stream=copy_stream)

    graphed = None
    if args.graph and world == 1 and not F._TORCH_BASELINE:
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

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if ddp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.item())
    if final_loss != final_loss or final_loss in (float("inf"), float("-inf")):
        final_loss = None               # keep the line strict JSON
    ips = world * args.batch_size * args.steps / dt
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec (whole job) %s %s training" % (
                {"resnet50": "ResNet-50"}.get(args.arch, args.arch), "AMP" if args.dtype != "fp32" else "fp32") +
                      (" [stock-PyTorch-ops baseline]" if F._TORCH_BASELINE else ""),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic uint8 %dx%d from pinned host memory, random-init weights" % (args.image_size, args.image_size),
            "final_loss": final_loss,
            "config": {"model": args.arch, "global_batch": args.batch_size * world,
                       "per_gpu_batch": args.batch_size, "seq_len": None, "image_size": args.image_size,
                       "parallelism": "dp%d" % world + (" (forced DDP path)" if args.force_ddp and world == 1 else ""),
                       "optimizer": "sgd+momentum+wd (fused)",
                       "amp": "O2 " + args.dtype if args.dtype != "fp32" else "O0",
                       "hip_graph": bool(graphed is not None and graphed.replays > 0),
                       "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)},
        }), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

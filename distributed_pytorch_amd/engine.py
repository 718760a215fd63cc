"""Training engine: launcher glue, epoch driver and train / validate loops of the three trainers.

One engine serves the three command lines of the reference:

* ``nd``   — ``nd_imagenet.py``: CPU / single GPU / DDP / multiprocessing-DDP (``nd_imagenet.py:81-263``).
  The reference's single-process ``nn.DataParallel`` branch (``:163-169``) is deliberately not
  reproduced: on a multi-GPU node the same command self-launches one process per GPU (DDP).
* ``ddp``  — ``imagenet_ddp.py``: always one process per GPU, ``--desired-acc`` early stop
  (``imagenet_ddp.py:72-236``).
* ``apex`` — ``imagenet_ddp_apex.py``: native AMP (bf16/fp16 + dynamic loss scaling) instead of Apex,
  per-iteration warm-up LR, SyncBN, uint8 collate + stream prefetcher, throughput + TensorBoard
  logging (``imagenet_ddp_apex.py:101-493``).

Hot-loop differences from the reference, by design: metrics stay on the device (one packed D2H copy
every ``print_freq`` iterations instead of ``loss.item()`` per step, ``nd_imagenet.py:296``), the
three metric all-reduces are one (``imagenet_ddp_apex.py:394-397``), checkpoints are written by one
rank followed by a barrier, and rank failures tear the job down instead of hanging it.
"""
from __future__ import annotations

import json
import math
import os
import random
import time
import warnings
from typing import Optional

import torch
from .parallel import process_group as pg

from . import amp, checkpoint as ckpt, launch, models, schedule
from .data.datasets import ImageFolder, SyntheticImageNet, fast_collate
from .data.prefetcher import StreamPrefetcher
from .data.sampler import DistributedSampler
from .data import transforms as T
from .nn.modules import CrossEntropyLoss
from .ops import functional as F
from .optim import FusedSGD
from .utils import profiling
from .utils.meters import AverageMeter, DeviceMeter, ProgressMeter, Summary, format_hms
from .utils.tb import SummaryWriter


# ------------------------------------------------------------------------------------------ launch
def main(args) -> None:
    """Entry point shared by the three scripts (``args.script`` in {nd, ddp, apex})."""
    if getattr(args, "seed", None) is not None:
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        if args.script == "nd":
            warnings.warn("You have chosen to seed training. Reductions are deterministic on this "
                          "framework (no float atomics), so this does not slow training down, but "
                          "you may see unexpected behavior when restarting from checkpoints.")
    use_gpu = torch.cuda.is_available() and not args.cpu
    ngpus = launch.device_count() if use_gpu or os.environ.get("DPA_NPROC_PER_NODE") else 0

    if args.script == "apex":
        # torch.distributed.launch / torchrun contract (imagenet_ddp_apex.py:88,113-129)
        args.distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
        args.local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
        args.world_size_total = int(os.environ.get("WORLD_SIZE", "1"))
        args.global_rank = int(os.environ.get("RANK", "0"))
        return main_worker(args.local_rank, max(ngpus, 1), args)

    if args.script == "nd" and args.gpu is not None:
        warnings.warn("You have chosen a specific GPU. This will completely disable data parallelism.")
    if args.dist_url == "env://" and args.world_size == -1:
        args.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.world_size < 1:
        args.world_size = 1            # reference: -1 * ngpus (imagenet_ddp.py:81) -- fixed here
    if args.rank < 0:
        args.rank = int(os.environ.get("RANK", "0")) if args.dist_url == "env://" else 0

    multiproc = args.multiprocessing_distributed
    if args.script == "nd" and not multiproc and args.gpu is None and args.world_size == 1 and ngpus > 1:
        # the reference would wrap the model in nn.DataParallel here (nd_imagenet.py:163-169);
        # one process per GPU is the native equivalent
        multiproc = True
        if args.dist_url.startswith("tcp://224."):
            args.dist_url = "tcp://127.0.0.1:%d" % (20000 + os.getpid() % 20000)
    args.distributed = args.world_size > 1 or multiproc
    if multiproc:
        nproc = max(ngpus, 1)
        args.world_size_total = nproc * args.world_size
        if nproc == 1:
            return main_worker(0, 1, args)
        return launch.spawn(main_worker, nproc, (nproc, args))
    args.world_size_total = args.world_size
    return main_worker(args.gpu if args.gpu is not None else 0, max(ngpus, 1), args)


# ------------------------------------------------------------------------------------------ worker
def main_worker(local_rank: int, ngpus_per_node: int, args) -> None:
    use_gpu = torch.cuda.is_available() and not args.cpu
    script = args.script
    if getattr(args, "share_gpu", False):
        os.environ["DPA_SHARE_GPU"] = "1"
    if getattr(args, "comm", "auto") == "peer":
        os.environ["DPA_COMM"] = "peer"
    if use_gpu:
        dev_index = launch.device_index(local_rank)
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        from .data.prefetcher import copy_stream

        copy_stream(device)       # first stream of the process: a hardware queue of its own for the input copies
        # reference: nd_imagenet.py:120-121 (only with --gpu), imagenet_ddp.py:97 (every worker)
        if script == "ddp" or (script == "nd" and getattr(args, "gpu", None) is not None):
            print("Use GPU: {} for training".format(local_rank), flush=True)
    else:
        device = torch.device("cpu")
        if script == "nd":
            print("using CPU, this will be slow")
    if getattr(args, "distributed", False) or "LOCAL_WORLD_SIZE" in os.environ:
        # CPUs next to this rank's GPU (its NUMA node), for ranks of our launcher and of torchrun alike
        launch.set_affinity_from_env(use_gpu=use_gpu)
    if args.profile:
        profiling.enable(True)

    # ---- process group ------------------------------------------------------------------------
    world = getattr(args, "world_size_total", 1)
    if script == "apex":
        rank = getattr(args, "global_rank", 0)
    elif args.distributed and (args.multiprocessing_distributed or world > args.world_size):
        rank = launch.global_rank(args.rank, ngpus_per_node, local_rank)
    else:
        rank = max(args.rank, 0)
    if args.distributed:
        launch.init_process_group(args.dist_backend, args.dist_url, world, rank,
                                  timeout_s=args.dist_timeout, use_gpu=use_gpu)
    is_main = rank == 0
    node_main = (rank % max(ngpus_per_node, 1) == 0) if script != "apex" else is_main
    if script == "apex" and is_main:
        print("opt_level = {}".format(args.opt_level))
        print("keep_batchnorm_fp32 = {}".format(args.keep_batchnorm_fp32), type(args.keep_batchnorm_fp32))
        print("loss_scale = {}".format(args.loss_scale), type(args.loss_scale))

    # ---- model --------------------------------------------------------------------------------
    if args.pretrained:
        print("=> using pre-trained model '{}'".format(args.arch))
    else:
        print("=> creating model '{}'".format(args.arch))
    model = models.create_model(args.arch, pretrained=args.pretrained, num_classes=args.num_classes)
    if args.sync_bn:
        from .parallel import convert_syncbn_model

        print("using native syncbn")
        model = convert_syncbn_model(model)
    model = model.to(device)

    # batch / worker semantics: per node for nd & ddp (divided by the GPUs of the node,
    # nd_imagenet.py:152-153, imagenet_ddp.py:125-126), per GPU for the AMP script
    if script in ("nd", "ddp") and args.distributed and ngpus_per_node > 1:
        args.batch_size = int(args.batch_size / ngpus_per_node)
        args.workers = int((args.workers + ngpus_per_node - 1) / ngpus_per_node)
    if script == "apex":
        args.lr = schedule.scale_lr_for_global_batch(args.lr, args.batch_size, world)

    optimizer = FusedSGD(model.parameters(), args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    opt_level = args.opt_level if script == "apex" else None
    model, optimizer = amp.initialize(model, optimizer, opt_level=opt_level or "O0",
                                      keep_batchnorm_fp32=args.keep_batchnorm_fp32,
                                      loss_scale=args.loss_scale, dtype=args.amp_dtype)
    if args.distributed:
        from .parallel import DistributedDataParallel

        model = DistributedDataParallel(
            model, device_ids=[device.index] if use_gpu else None, bucket_cap_mb=args.bucket_mb,
            delay_allreduce=args.delay_allreduce, comm=args.comm, grad_wire=getattr(args, "grad_wire", "auto"),
            broadcast_buffers=not args.no_broadcast_buffers and script != "apex")
        model.bind_optimizer(optimizer)
    criterion = CrossEntropyLoss().to(device)

    # ---- resume -------------------------------------------------------------------------------
    best_acc1 = 0.0
    if args.resume:
        got = ckpt.resume(args.resume, model, optimizer, map_location=device)
        if got is not None:
            args.start_epoch, best_acc1 = got

    # ---- data ---------------------------------------------------------------------------------
    train_loader, val_loader, train_sampler = build_loaders(args, rank if args.distributed else 0,
                                                            world if args.distributed else 1, script)
    writer = None
    log_dir = args.out_dir
    if script == "apex" and is_main:
        comment = "_{}_gpux{}_b{}_cpu{}_opt{}".format(args.arch, world, args.batch_size, args.workers,
                                                      args.opt_level)
        writer = SummaryWriter(log_dir=args.out_dir or None, comment=comment)
        log_dir = writer.log_dir
    state = _Run(args, script, model, criterion, optimizer, device, rank, world, is_main, writer)

    if args.evaluate:
        state.validate(val_loader, 0)
        return _shutdown(args, writer)

    # ---- epochs -------------------------------------------------------------------------------
    total_start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        if script != "apex":
            schedule.adjust_learning_rate(optimizer, epoch, args)
        tr = state.train(train_loader, epoch)
        va = state.validate(val_loader, epoch)
        acc1 = va["top1"]
        is_best = acc1 > best_acc1
        best_acc1 = max(acc1, best_acc1)
        if (script == "apex" and is_main) or (script != "apex" and (not args.multiprocessing_distributed
                                                                    or node_main)):
            extra = {"amp": amp.state_dict()} if opt_level else None
            ckpt.save_checkpoint(
                ckpt.build_state(epoch, args.arch, model, optimizer, best_acc1, args.distributed,
                                 legacy_prec1_key=(script == "apex"), extra=extra),
                is_best, out_dir=log_dir)
        if args.distributed:
            pg.barrier()            # nobody races ahead of the checkpoint file (SURVEY.md C-l)
        if writer is not None:
            for tag, v in (("Throughput/train", tr["throughput"]), ("Throughput/val", va["throughput"]),
                           ("Time/train", tr["batch_time"]), ("Time/val", va["batch_time"]),
                           ("Loss/train", tr["loss"]), ("Loss/val", va["loss"]),
                           ("Top1/train", tr["top1"]), ("Top1/val", va["top1"]),
                           ("Top5/train", tr["top5"]), ("Top5/val", va["top5"]), ("Lr", tr["lr"])):
                writer.add_scalar(tag, v, epoch + 1)
        desired = getattr(args, "desired_acc", None)
        if desired and best_acc1 >= desired:
            training_time = time.time() - total_start
            if is_main:       # the reference lets every rank write here (imagenet_ddp.py:225-234)
                ckpt.save_checkpoint(
                    ckpt.build_state(epoch, args.arch, model, optimizer, best_acc1, args.distributed,
                                     extra={"training_time": training_time}), is_best, out_dir=log_dir)
            if args.distributed:
                pg.barrier()
            print("Training time: {:.3f} s; desired accuracy {:.3f} reached at epoch {}".format(
                training_time, desired, epoch + 1))
            break
    if script == "apex" and is_main:
        print("### Training Time: " + format_hms(time.time() - total_start))
        print(vars(args))
    _shutdown(args, writer)


def _shutdown(args, writer) -> None:
    if writer is not None:
        writer.close()
    if args.distributed:
        launch.shutdown()


# ------------------------------------------------------------------------------------------ data
def build_loaders(args, rank: int, world: int, script: str):
    """-> (train_iterable, val_iterable, train_sampler)."""
    if args.synthetic:
        steps = args.steps
        train = SyntheticImageNet(args.batch_size, args.synthetic_size, world, rank, num_classes=args.num_classes,
                                  steps=steps, seed=1)
        shard_val = script == "apex"        # nd/ddp validate the full set on every rank (imagenet_ddp.py:186-194)
        val = SyntheticImageNet(args.batch_size, args.synthetic_val_size, world if shard_val else 1,
                                rank if shard_val else 0, num_classes=args.num_classes,
                                steps=max(1, steps // 4) if steps else 0, seed=2, ring=2)
        return train, val, None
    from torch.utils.data import DataLoader

    traindir, valdir = os.path.join(args.data, "train"), os.path.join(args.data, "val")
    crop = 299 if args.arch == "inception_v3" else 224
    val_size = 320 if args.arch == "inception_v3" else 256
    train_ds = ImageFolder(traindir, T.Compose([T.RandomResizedCrop(crop), T.RandomHorizontalFlip()]))
    val_ds = ImageFolder(valdir, T.Compose([T.Resize(val_size), T.CenterCrop(crop)]))
    train_sampler = DistributedSampler(train_ds, world, rank) if world > 1 else None
    val_sampler = DistributedSampler(val_ds, world, rank, shuffle=False) if (world > 1 and script == "apex") else None
    collate = lambda b: fast_collate(b, nhwc=True)      # noqa: E731  uint8 NHWC, normalised on the GPU
    pin = torch.cuda.is_available() and not args.cpu
    train = DataLoader(train_ds, batch_size=args.batch_size, shuffle=(train_sampler is None), num_workers=args.workers,
                       pin_memory=pin, sampler=train_sampler, collate_fn=collate, drop_last=False)
    val = DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers, pin_memory=pin,
                     sampler=val_sampler, collate_fn=collate)
    return train, val, train_sampler


# ------------------------------------------------------------------------------------------ loops
class _Run:
    def __init__(self, args, script, model, criterion, optimizer, device, rank, world, is_main, writer):
        self.args, self.script, self.model, self.criterion = args, script, model, criterion
        self.optimizer, self.device, self.rank, self.world = optimizer, device, rank, world
        self.is_main, self.writer = is_main, writer
        self.global_step = 0
        self.jsonl = None
        self.graphed = None
        if getattr(args, "hip_graph", False):
            if device.type == "cuda":
                # with DDP the reducer's communication stream is captured as a fork / join of the same graph
                from .utils.graph import GraphedTrainStep

                self.graphed = GraphedTrainStep(model, optimizer)
            elif is_main:
                print("=> --hip-graph applies to GPU runs; this run executes eagerly")
        if is_main and args.out_dir:
            os.makedirs(args.out_dir, exist_ok=True)
            self.jsonl = open(os.path.join(args.out_dir, "metrics.jsonl"), "a")

    def _prefetcher(self, loader):
        base = getattr(self.model, "module", self.model)
        return StreamPrefetcher(loader, self.device, dtype=amp.compute_dtype(torch.float32),
                                pad_c=8 if F._on_gpu(torch.empty(0, device=self.device)) else 0,
                                s2d=bool(getattr(base, "stem_s2d", False)))

    def _reduced(self, packed: torch.Tensor) -> torch.Tensor:
        """APX averages metrics over ranks (``reduce_tensor``), nd/ddp report rank-local values."""
        if self.script == "apex" and self.args.distributed:
            packed = packed.clone()
            pg.all_reduce_sum(packed)       # sums of (loss*n, correct, n): one collective, exact mean
        return packed

    def _log(self, rec: dict) -> None:
        if self.jsonl is not None:
            self.jsonl.write(json.dumps(rec) + "\n")
            self.jsonl.flush()

    # ------------------------------------------------------------------ train
    def train(self, loader, epoch: int) -> dict:
        args, script = self.args, self.script
        n_iter = len(loader)
        batch_time = AverageMeter("Time", ":6.3f")
        data_time = AverageMeter("Data", ":6.3f")
        losses = AverageMeter("Loss", ":.4e")
        top1 = AverageMeter("Acc@1", ":6.2f")
        top5 = AverageMeter("Acc@5", ":6.2f")
        progress = ProgressMeter(n_iter, [batch_time, data_time, losses, top1, top5],
                                 prefix="Epoch: [{}]".format(epoch))
        dm = DeviceMeter(self.device)
        self.model.train()
        pre = self._prefetcher(loader)
        end = time.time()
        win_start, win_iters, lr = end, 0, self.optimizer.param_groups[0]["lr"]
        thr_avg = AverageMeter("Throughput", ":.1f")
        # the iteration this epoch really ends at: --steps may cut a real dataset short, and the meters must be
        # flushed there (by every rank together: the flush is a collective under the AMP script)
        n_stop = min(n_iter, args.steps) if args.steps else n_iter
        i = 0
        while True:
            with profiling.range_("data"):
                images, target = pre.next()
            if images is None:
                break
            i += 1
            launch.maybe_fault_inject(args.fault_inject, self.rank, self.global_step)
            if script == "apex":
                lr = schedule.adjust_learning_rate_iter(self.optimizer, epoch, i, n_iter, args)
            data_time.update(time.time() - end)
            if self.graphed is not None:
                with profiling.range_("graphed_step"):
                    loss, rec = self.graphed(images, target)
            else:
                with profiling.range_("forward"):
                    output = self.model(images)
                    loss, rec = F.loss_and_metrics(output, target)
                self.optimizer.zero_grad()
                with profiling.range_("backward"):
                    with amp.scale_loss(loss, self.optimizer) as scaled:
                        scaled.backward()
                with profiling.range_("step"):
                    self.optimizer.step()
            dm.update(rec)
            self.global_step += 1
            win_iters += 1
            if i % args.print_freq == 0 or i == n_stop:
                (l_loss, l_a1, l_a5, l_n), (r_loss, r_a1, r_a5, r_n) = DeviceMeter.unpack(
                    self._reduced(dm.packed()).cpu())          # the ONE host sync of the window
                now = time.time()
                per_iter = (now - win_start) / max(win_iters, 1)
                batch_time.update(per_iter, win_iters)
                for m, lv, rv in ((losses, l_loss, r_loss), (top1, l_a1, r_a1), (top5, l_a5, r_a5)):
                    m.val, m.avg, m.sum, m.count = lv, rv, rv * r_n, r_n
                thr = self.world * args.batch_size / per_iter
                thr_avg.update(thr, win_iters)
                if script == "apex":
                    if self.is_main:
                        print("Epoch: [{0}][{1}/{2}]\tTime {bt.val:.3f} ({bt.avg:.3f})\t"
                              "Throughput {thr:.3f} ({ta:.3f})\tLoss {l.val:.10f} ({l.avg:.4f})\t"
                              "Prec@1 {t1.val:.3f} ({t1.avg:.3f})\tPrec@5 {t5.val:.3f} ({t5.avg:.3f})\t"
                              "Lr {lr:.6f}".format(epoch, i, n_iter, bt=batch_time, thr=thr, ta=thr_avg.avg,
                                                   l=losses, t1=top1, t5=top5, lr=lr), flush=True)
                else:
                    progress.display(i)
                self._log({"phase": "train", "epoch": epoch, "iter": i, "loss": l_loss, "top1": l_a1,
                           "top5": l_a5, "time": per_iter, "throughput": thr, "lr": lr})
                win_start, win_iters = time.time(), 0
            end = time.time()
            if i >= n_stop:
                break
        return {"throughput": thr_avg.avg, "batch_time": batch_time.avg, "loss": losses.avg,
                "top1": top1.avg, "top5": top5.avg, "lr": lr}

    # ------------------------------------------------------------------ validate
    def validate(self, loader, epoch: int) -> dict:
        args, script = self.args, self.script
        n_iter = len(loader)
        batch_time = AverageMeter("Time", ":6.3f", Summary.NONE)
        losses = AverageMeter("Loss", ":.4e", Summary.NONE)
        top1 = AverageMeter("Acc@1", ":6.2f", Summary.AVERAGE)
        top5 = AverageMeter("Acc@5", ":6.2f", Summary.AVERAGE)
        progress = ProgressMeter(n_iter, [batch_time, losses, top1, top5], prefix="Test: ")
        dm = DeviceMeter(self.device)
        self.model.eval()
        pre = self._prefetcher(loader)
        start = win_start = time.time()
        n_stop = min(n_iter, max(1, args.steps // 4)) if args.steps else n_iter
        i = win_iters = 0
        with torch.no_grad():
            while True:
                images, target = pre.next()
                if images is None:
                    break
                i += 1
                output = self.model(images)
                _, rec = F.loss_and_metrics(output, target)
                dm.update(rec)
                win_iters += 1
                if i % args.print_freq == 0 or i == n_stop:
                    (l_loss, l_a1, l_a5, _), (r_loss, r_a1, r_a5, r_n) = DeviceMeter.unpack(
                        self._reduced(dm.packed()).cpu())
                    per_iter = (time.time() - win_start) / max(win_iters, 1)
                    batch_time.update(per_iter, win_iters)
                    for m, lv, rv in ((losses, l_loss, r_loss), (top1, l_a1, r_a1), (top5, l_a5, r_a5)):
                        m.val, m.avg, m.sum, m.count = lv, rv, rv * r_n, r_n
                    if script == "apex":
                        if self.is_main:
                            print("Test: [{0}/{1}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tThroughput {thr:.3f}\t"
                                  "Loss {l.val:.4f} ({l.avg:.4f})\tPrec@1 {t1.val:.3f} ({t1.avg:.3f})\t"
                                  "Prec@5 {t5.val:.3f} ({t5.avg:.3f})".format(
                                      i, n_iter, bt=batch_time, thr=self.world * args.batch_size / per_iter,
                                      l=losses, t1=top1, t5=top5), flush=True)
                    else:
                        progress.display(i)
                    win_start, win_iters = time.time(), 0
                if i >= n_stop:
                    break
        if script == "nd":
            progress.display_summary()
        elif script == "ddp" or self.is_main:
            print(" * Acc@1 {top1.avg:.3f} Acc@5 {top5.avg:.3f}".format(top1=top1, top5=top5), flush=True)
        elapsed = max(time.time() - start, 1e-9)
        out = {"throughput": self.world * args.batch_size * max(i, 1) / elapsed, "batch_time": batch_time.avg,
               "loss": losses.avg, "top1": float(top1.avg), "top5": float(top5.avg)}
        self._log(dict(out, phase="val", epoch=epoch))
        return out

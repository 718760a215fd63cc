"""Command-line surfaces of the three trainers.

Every flag, alias and default of the reference scripts is kept verbatim
(flag table: SURVEY.md §5.6; reference ``nd_imagenet.py:26-76``,
``imagenet_ddp.py:23-67``, ``imagenet_ddp_apex.py:42-98``).  New flags are purely
additive and shared by all three front ends (``add_native_flags``).

Deliberate, documented fixes of reference quirks:
* ``data`` is optional when ``--synthetic`` is given.
* ``--world-size``/``--rank`` default ``-1`` is normalised to one node / node 0
  (the reference computes a negative world size, ``imagenet_ddp.py:81``).
* ``--desired-acc`` ≤ 1.0 is read as a fraction (``0.75`` → 75 %); the reference
  compares it to a percent-valued metric (``imagenet_ddp.py:225``).
* ``--channels-last`` accepts real booleans (the reference's ``type=bool`` makes
  every non-empty string true, ``imagenet_ddp_apex.py:95``).
"""
from __future__ import annotations

import argparse
from typing import List, Optional, Sequence


def model_names() -> List[str]:
    from .models import list_models

    return list_models()


def _str2bool(text) -> bool:
    if isinstance(text, bool):
        return text
    return str(text).strip().lower() not in ("", "0", "false", "no", "off", "none")


def add_native_flags(parser: argparse.ArgumentParser) -> None:
    g = parser.add_argument_group("MI355X-native options (additive)")
    g.add_argument("--synthetic", action="store_true",
                   help="feed synthetic uint8 224x224 batches from pinned host memory")
    g.add_argument("--synthetic-size", default=1281167, type=int, metavar="N",
                   help="number of samples per synthetic training epoch")
    g.add_argument("--synthetic-val-size", default=50000, type=int, metavar="N")
    g.add_argument("--num-classes", default=1000, type=int)
    g.add_argument("--amp-dtype", default="bf16", choices=["bf16", "fp16"],
                   help="low-precision compute dtype used by opt levels O1-O3")
    g.add_argument("--bucket-mb", default=64.0, type=float,
                   help="gradient bucket cap in MiB (sized for xGMI, see parallel/bucketing.py)")
    g.add_argument("--delay-allreduce", action="store_true",
                   help="one flat all-reduce after backward (Apex delay_allreduce parity)")
    g.add_argument("--no-broadcast-buffers", action="store_true",
                   help="skip the per-forward broadcast of BN buffers from rank 0")
    g.add_argument("--steps", default=0, type=int, metavar="N",
                   help="stop every epoch after N iterations (0 = full epoch)")
    g.add_argument("--warmup-epochs", default=5, type=int)
    g.add_argument("--profile", action="store_true", help="emit roctx ranges around step phases")
    g.add_argument("--hip-graph", action="store_true",
                   help="capture the train step into a hipGraph (static shapes)")
    g.add_argument("--dist-timeout", default=600.0, type=float, metavar="SEC",
                   help="rendezvous / collective timeout")
    g.add_argument("--comm", default="auto", choices=["auto", "native", "torch", "peer"],
                   help="collective engine: native C++ reducer on RCCL, torch.distributed, or the peer-memory "
                        "communicator (two-shot all-reduce over hipIpc windows, csrc/comm/peer.hip)")
    g.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16", "fp16"],
                   help="dtype fp32 gradient buckets travel in (auto: the AMP dtype under O2/O3, like Apex)")
    g.add_argument("--share-gpu", action="store_true",
                   help="local ranks wrap around the visible GPUs (rehearse N ranks on one device; needs --comm peer)")
    g.add_argument("--fault-inject", default=None, type=str, metavar="RANK:STEP",
                   help="test hook: hard-exit RANK at global STEP")
    g.add_argument("--out-dir", default="", type=str, help="where checkpoints/logs are written")
    g.add_argument("--cpu", action="store_true", help="force the CPU reference backend")


def _common(parser: argparse.ArgumentParser, *, arch: str, batch: int, batch_help: str) -> None:
    names = model_names()
    parser.add_argument("data", metavar="DIR", nargs="?", default=None, help="path to dataset")
    parser.add_argument("-a", "--arch", metavar="ARCH", default=arch, choices=names,
                        help="model architecture: " + " | ".join(names) + " (default: %s)" % arch)
    parser.add_argument("-j", "--workers", default=4, type=int, metavar="N",
                        help="number of data loading workers (default: 4)")
    parser.add_argument("--epochs", default=90, type=int, metavar="N",
                        help="number of total epochs to run")
    parser.add_argument("--start-epoch", default=0, type=int, metavar="N",
                        help="manual epoch number (useful on restarts)")
    parser.add_argument("-b", "--batch-size", default=batch, type=int, metavar="N", help=batch_help)
    parser.add_argument("--lr", "--learning-rate", default=0.1, type=float, metavar="LR",
                        help="initial learning rate", dest="lr")
    parser.add_argument("--momentum", default=0.9, type=float, metavar="M", help="momentum")
    parser.add_argument("--wd", "--weight-decay", default=1e-4, type=float, metavar="W",
                        help="weight decay (default: 1e-4)", dest="weight_decay")
    parser.add_argument("-p", "--print-freq", default=10, type=int, metavar="N",
                        help="print frequency (default: 10)")
    parser.add_argument("--resume", default="", type=str, metavar="PATH",
                        help="path to latest checkpoint (default: none)")
    parser.add_argument("-e", "--evaluate", dest="evaluate", action="store_true",
                        help="evaluate model on validation set")
    parser.add_argument("--pretrained", dest="pretrained", action="store_true",
                        help="use pre-trained model (local file only: no network)")


def _dist_flags(parser: argparse.ArgumentParser) -> None:
    parser.add_argument("--world-size", default=-1, type=int,
                        help="number of nodes for distributed training")
    parser.add_argument("--rank", default=-1, type=int, help="node rank for distributed training")
    parser.add_argument("--dist-url", default="tcp://224.66.41.62:23456", type=str,
                        help="url used to set up distributed training")
    parser.add_argument("--dist-backend", default="nccl", type=str,
                        help="distributed backend (nccl == rccl on MI355X; gloo for CPU)")


def nd_parser() -> argparse.ArgumentParser:
    """``nd_imagenet.py`` flags (reference ``nd_imagenet.py:26-76``)."""
    p = argparse.ArgumentParser(description="MI355X ImageNet Training (ND / DDP)")
    _common(p, arch="resnet18", batch=256,
            batch_help="mini-batch size (default: 256), this is the total batch size of all GPUs "
                       "on the current node when using Data Parallel or Distributed Data Parallel")
    _dist_flags(p)
    p.add_argument("--seed", default=None, type=int, help="seed for initializing training.")
    p.add_argument("--gpu", default=None, type=int, help="GPU id to use.")
    p.add_argument("--multiprocessing-distributed", action="store_true",
                   help="launch N processes per node, which has N GPUs")
    add_native_flags(p)
    p.set_defaults(script="nd")
    return p


def ddp_parser() -> argparse.ArgumentParser:
    """``imagenet_ddp.py`` flags (reference ``imagenet_ddp.py:23-67``)."""
    p = argparse.ArgumentParser(description="MI355X ImageNet Training (DDP)")
    _common(p, arch="resnet50", batch=1024,
            batch_help="mini-batch size (default: 1024), this is the total batch size of all GPUs "
                       "on the current node when using Distributed Data Parallel")
    _dist_flags(p)
    p.add_argument("--desired-acc", default=None, type=float,
                   help="Training will stop after desired-acc is reached.")
    p.add_argument("--seed", default=None, type=int, help="seed (additive; not in the reference)")
    add_native_flags(p)
    p.set_defaults(script="ddp", multiprocessing_distributed=True, gpu=None)
    return p


def apex_parser() -> argparse.ArgumentParser:
    """``imagenet_ddp_apex.py`` flags (reference ``imagenet_ddp_apex.py:42-98``)."""
    p = argparse.ArgumentParser(description="MI355X ImageNet Training (native AMP + DDP)")
    _common(p, arch="resnet50", batch=224,
            batch_help="mini-batch size per GPU (default: 224)")
    p.add_argument("--local_rank", "--local-rank", default=0, type=int, dest="local_rank")
    p.add_argument("--sync-bn", action="store_true", help="enable native SyncBatchNorm")
    p.add_argument("--opt-level", type=str, default=None, help="O0 | O1 | O2 | O3")
    p.add_argument("--keep-batchnorm-fp32", type=str, default=None)
    p.add_argument("--loss-scale", type=str, default=None, help='"dynamic" or a number')
    p.add_argument("--channels-last", type=_str2bool, default=False)
    p.add_argument("--dist-backend", default="nccl", type=str,
                   help="(additive) the reference hard-codes gloo at imagenet_ddp_apex.py:124")
    p.add_argument("--seed", default=None, type=int)
    add_native_flags(p)
    p.set_defaults(script="apex", dist_url="env://", world_size=-1, rank=-1, gpu=None,
                   multiprocessing_distributed=False, desired_acc=None)
    return p


def normalise(args: argparse.Namespace) -> argparse.Namespace:
    """Shared post-parse fix-ups (see module docstring)."""
    if args.data is None and not args.synthetic:
        raise SystemExit("error: DIR is required unless --synthetic is given")
    acc = getattr(args, "desired_acc", None)
    if acc is not None and acc <= 1.0:
        args.desired_acc = acc * 100.0
    if not hasattr(args, "sync_bn"):
        args.sync_bn = False
    if not hasattr(args, "opt_level"):
        args.opt_level = None
    if not hasattr(args, "loss_scale"):
        args.loss_scale = None
    if not hasattr(args, "keep_batchnorm_fp32"):
        args.keep_batchnorm_fp32 = None
    if not hasattr(args, "channels_last"):
        args.channels_last = False
    if not hasattr(args, "seed"):
        args.seed = None
    if not hasattr(args, "local_rank"):
        args.local_rank = 0
    return args


def parse(script: str, argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    parser = {"nd": nd_parser, "ddp": ddp_parser, "apex": apex_parser}[script]()
    return normalise(parser.parse_args(argv))

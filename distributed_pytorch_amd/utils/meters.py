"""Running-average meters and progress printing.

Output formats are byte-compatible with the reference scripts so existing log
parsers keep working:

* ``AverageMeter`` / ``Summary`` / ``ProgressMeter`` — reference
  ``nd_imagenet.py:361-426`` and ``imagenet_ddp.py:333-371``.
* ``PlainMeter`` — the nameless meter of ``imagenet_ddp_apex.py:509-524``.

New here: ``DeviceMeter`` keeps ``[loss, top1, top5]`` sums on the GPU so the hot
loop never calls ``.item()`` per iteration (the reference syncs the host every
step at ``nd_imagenet.py:296``); values are fetched with one D2H copy every
``print_freq`` iterations.
"""
from __future__ import annotations

from enum import Enum
from typing import Iterable, List, Optional

import torch


class Summary(Enum):
    NONE = 0
    AVERAGE = 1
    SUM = 2
    COUNT = 3


class AverageMeter:
    """Tracks the latest value, running sum, count and mean of a scalar."""

    def __init__(self, name: str = "", fmt: str = ":f", summary_type: Summary = Summary.AVERAGE):
        self.name = name
        self.fmt = fmt
        self.summary_type = summary_type
        self.reset()

    def reset(self) -> None:
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n: int = 1) -> None:
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self) -> str:
        template = "{name} {val" + self.fmt + "} ({avg" + self.fmt + "})"
        return template.format(name=self.name, val=self.val, avg=self.avg)

    def summary(self) -> str:
        kind = self.summary_type
        if kind is Summary.NONE:
            return ""
        if kind is Summary.AVERAGE:
            return "{} {:.3f}".format(self.name, self.avg)
        if kind is Summary.SUM:
            return "{} {:.3f}".format(self.name, self.sum)
        if kind is Summary.COUNT:
            return "{} {:.3f}".format(self.name, self.count)
        raise ValueError("invalid summary type %r" % kind)


class PlainMeter(AverageMeter):
    """Nameless meter used by the AMP script (``imagenet_ddp_apex.py:509-524``)."""

    def __init__(self):
        super().__init__("", ":f", Summary.NONE)


class ProgressMeter:
    """Prints ``prefix[  i/N]\\tmeter\\tmeter...`` lines, tab separated."""

    def __init__(self, num_batches: int, meters: Iterable[AverageMeter], prefix: str = ""):
        self.batch_fmtstr = self._batch_format(num_batches)
        self.meters: List[AverageMeter] = list(meters)
        self.prefix = prefix

    def line(self, batch: int) -> str:
        fields = [self.prefix + self.batch_fmtstr.format(batch)]
        fields.extend(str(m) for m in self.meters)
        return "\t".join(fields)

    def display(self, batch: int) -> None:
        print(self.line(batch), flush=True)

    def summary_line(self) -> str:
        return " ".join([" *"] + [m.summary() for m in self.meters])

    def display_summary(self) -> None:
        print(self.summary_line(), flush=True)

    @staticmethod
    def _batch_format(num_batches: int) -> str:
        width = len(str(num_batches // 1))
        one = "{:" + str(width) + "d}"
        return "[" + one + "/" + one.format(num_batches) + "]"


class DeviceMeter:
    """Accumulates ``[loss*n, top1_correct, top5_correct, n]`` on the device.

    ``update`` is sync-free; ``fetch`` performs the single device->host copy.
    ``packed`` returns the 4-vector so that a distributed run can all-reduce the
    whole record in ONE collective instead of the reference's three 4-byte
    all-reduces (``imagenet_ddp_apex.py:394-397``).
    """

    def __init__(self, device: torch.device):
        self.device = device
        self.acc = torch.zeros(4, dtype=torch.float32, device=device)
        self.last = torch.zeros(4, dtype=torch.float32, device=device)

    def reset(self) -> None:
        self.acc.zero_()
        self.last.zero_()

    def update(self, record: torch.Tensor) -> None:
        """``record`` = float32[4]: loss_sum, correct@1, correct@5, n."""
        self.last = record
        self.acc.add_(record)

    def packed(self) -> torch.Tensor:
        return torch.stack([self.last, self.acc])

    @staticmethod
    def unpack(packed_host: torch.Tensor):
        """-> ((loss, acc1%, acc5%, n) last, (loss, acc1%, acc5%, n) running)."""
        out = []
        for row in packed_host.tolist():
            n = max(row[3], 1.0)
            out.append((row[0] / n, 100.0 * row[1] / n, 100.0 * row[2] / n, row[3]))
        return out[0], out[1]

    def fetch(self):
        return self.unpack(self.packed().cpu())


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """Top-k accuracy in PERCENT, each a shape-``[1]`` tensor.

    Same contract as reference ``nd_imagenet.py:436-450``.  Implemented by rank
    counting (``#{z_j > z_target} < k``) instead of a sort, which is also what
    the fused HIP kernel ``topk_correct`` does.
    """
    with torch.no_grad():
        from ..ops import functional as F

        counts = F.topk_correct(output, target, tuple(topk))  # float32 [len(topk)]
        batch = target.size(0)
        return [counts[i : i + 1] * (100.0 / batch) for i in range(len(topk))]


def format_hms(seconds: float) -> str:
    mins, secs = divmod(seconds, 60)
    hrs, mins = divmod(mins, 60)
    return "{:.2f} hrs {:.2f} mins {:.2f} secs | {:.2f} secs".format(hrs, mins, secs, seconds)

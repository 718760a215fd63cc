"""Gradient bucket planner (host logic of the reducer).

Parameters are walked in *reverse registration order* (a proxy for backward
order, exactly what c10d's Reducer does before its first-iteration rebuild) and
packed per dtype into flat buckets.  After the first backward the DDP wrapper
re-plans from the *observed* ready order.

Bucket size is chosen for xGMI, not PCIe/IB: 7 point-to-point links of ~153 GB/s
per GPU make a 32 MiB ring all-reduce on 8 ranks a ~0.4 ms affair, so there is no
reason to pay launch latency on 25 MiB-or-smaller pieces; but the LAST bucket to
become ready (first layers) is latency-exposed, so the first/last buckets stay
small.  One tensor bigger than the cap (VGG-16 ``fc6.weight`` = 392 MiB) becomes a
bucket of its own and is all-reduced in ``chunk_bytes`` slices so its transfer
pipelines instead of serialising.

The native implementation is ``plan_buckets`` in ``csrc/comm/comm.cpp``; this Python
version is the oracle used in tests and the fallback when ``_C`` is not built.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

MiB = 1024 * 1024


@dataclass
class BucketSpec:
    index: int
    dtype_key: int                      # opaque dtype id (buckets are homogeneous)
    param_ids: List[int] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)   # element offsets into the flat buffer
    numels: List[int] = field(default_factory=list)
    total: int = 0                      # elements (padded so every view is 16-byte aligned)


def plan_buckets(numels: Sequence[int], elem_sizes: Sequence[int], dtype_keys: Sequence[int],
                 order: Sequence[int], cap_bytes: int, first_cap_bytes: int,
                 align_bytes: int = 256, last_cap_bytes: int = 0, min_bytes: int = MiB) -> List[BucketSpec]:
    """``order`` = parameter indices in expected ready order.  Returns buckets in
    the order they are expected to complete.

    ``min_bytes``: a bucket smaller than this is never closed by the cap -- it takes the next tensor anyway, so a
    lone bias in front of an over-cap weight (VGG-16 ``fc6``) does not become a latency-bound collective of its own.

    ``last_cap_bytes`` > 0: the parameters that become ready LAST (the first layers) are kept out of the big
    buckets and form a final bucket of at most that size -- its all-reduce is the only one that cannot overlap
    with backward, so it must be short (ResNet-50 with 64 MiB buckets otherwise ends on a 34 MB all-reduce)."""
    try:
        from .. import backend

        ext = backend.load()
    except Exception:
        ext = None
    if ext is not None and hasattr(ext, "plan_buckets"):
        raw = ext.plan_buckets(list(numels), list(elem_sizes), list(dtype_keys), list(order),
                               int(cap_bytes), int(first_cap_bytes), int(align_bytes), int(last_cap_bytes),
                               int(min_bytes))
        return [BucketSpec(i, dk, list(pids), list(offs), [numels[p] for p in pids], tot)
                for i, (dk, pids, offs, tot) in enumerate(raw)]
    return plan_buckets_py(numels, elem_sizes, dtype_keys, order, cap_bytes, first_cap_bytes,
                           align_bytes, last_cap_bytes, min_bytes)


def plan_buckets_py(numels, elem_sizes, dtype_keys, order, cap_bytes, first_cap_bytes,
                    align_bytes: int = 256, last_cap_bytes: int = 0, min_bytes: int = MiB) -> List[BucketSpec]:
    """Pure-Python oracle of the native planner (same result, used in tests and without ``_C``)."""
    order = list(order)
    if last_cap_bytes > 0 and len(order) > 1:
        split, acc = len(order), 0
        while split > 0:
            sz = numels[order[split - 1]] * elem_sizes[order[split - 1]]
            if split < len(order) and acc + sz > last_cap_bytes:
                break
            acc += sz
            split -= 1
        if split > 0:
            out = plan_buckets_py(numels, elem_sizes, dtype_keys, order[:split], cap_bytes, first_cap_bytes,
                                  align_bytes, 0, min_bytes)
            out += plan_buckets_py(numels, elem_sizes, dtype_keys, order[split:], 1 << 62, 1 << 62, align_bytes,
                                   0, min_bytes)
            for i, b in enumerate(out):
                b.index = i
            return out
    buckets: List[BucketSpec] = []
    open_by_dtype: Dict[int, BucketSpec] = {}
    first_done: Dict[int, bool] = {}
    for pid in order:
        dk, es, n = dtype_keys[pid], elem_sizes[pid], numels[pid]
        cur = open_by_dtype.get(dk)
        cap = cap_bytes if first_done.get(dk) else first_cap_bytes
        if cur is not None and cur.total * es + n * es > cap and cur.param_ids:
            first_done[dk] = True
            if cur.total * es >= min_bytes:         # a tiny bucket takes the tensor anyway (see the docstring)
                cur = None
        if cur is None:
            cur = BucketSpec(len(buckets), dk)
            buckets.append(cur)
            open_by_dtype[dk] = cur
        cur.param_ids.append(pid)
        cur.offsets.append(cur.total)
        cur.numels.append(n)
        align_el = max(1, align_bytes // es)
        cur.total += -(-n // align_el) * align_el
    return buckets


def chunk_ranges(total_elems: int, elem_size: int, chunk_bytes: int) -> List[Tuple[int, int]]:
    """Slices ``[begin, end)`` (elements) for pipelined all-reduce of one oversized bucket."""
    step = max(1, chunk_bytes // elem_size)
    return [(b, min(total_elems, b + step)) for b in range(0, total_elems, step)]

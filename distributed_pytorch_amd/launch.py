"""Process topology: one process per GPU, supervised.

Covers the launcher layer of the reference — ``mp.spawn(main_worker, nprocs=ngpus)``
(``nd_imagenet.py:103-113``, ``imagenet_ddp.py:72-86``), the
``torch.distributed.launch`` contract of the AMP script (``--local_rank`` +
``WORLD_SIZE``/``RANK`` env, ``imagenet_ddp_apex.py:88,113-129``) and the
rendezvous (``dist.init_process_group(backend, init_method, world_size, rank)``,
``imagenet_ddp.py:103-105``; global rank = ``node_rank * ngpus + gpu``).

Additions the reference lacks: a supervisor that tears every
rank down as soon as one dies (the reference needs ``process_cleanup.sh``), a
rendezvous/collective timeout instead of blocking forever, NUMA-friendly CPU
affinity per local rank, and a ``--fault-inject RANK:STEP`` hook for tests.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import signal
import sys
import time
import traceback
from typing import Callable, List, Optional, Tuple

import torch


def device_count() -> int:
    """GPUs on this node; ``DPA_NPROC_PER_NODE`` overrides (CPU/gloo tests)."""
    env = os.environ.get("DPA_NPROC_PER_NODE")
    if env:
        return int(env)
    return torch.cuda.device_count()


def device_index(local_rank: int) -> int:
    """The device a local rank binds.  ``DPA_SHARE_GPU=1``: ranks wrap around the visible devices, so W ranks can
    share one GPU (peer-memory communicator only: RCCL refuses two ranks on a device)."""
    if os.environ.get("DPA_SHARE_GPU", "0") == "1":
        return local_rank % max(1, torch.cuda.device_count())
    return local_rank


def resolve_backend(name: str, use_gpu: bool) -> str:
    name = (name or "nccl").lower()
    if name == "rccl":
        name = "nccl"
    if name == "nccl" and not use_gpu:
        return "gloo"          # CPU plumbing configuration (BASELINE config #1)
    return name


def parse_dist_url(url: str) -> Tuple[str, Optional[str], Optional[int]]:
    """-> (scheme, host, port) for ``tcp://host:port`` / ``env://`` / ``file://path``."""
    if url.startswith("env://"):
        return "env", os.environ.get("MASTER_ADDR"), \
            int(os.environ["MASTER_PORT"]) if "MASTER_PORT" in os.environ else None
    if url.startswith("tcp://"):
        host, _, port = url[len("tcp://"):].rpartition(":")
        return "tcp", host, int(port)
    if url.startswith("file://"):
        return "file", url[len("file://"):], None
    raise ValueError("unsupported --dist-url %r" % url)


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def global_rank(node_rank: int, ngpus_per_node: int, local_rank: int) -> int:
    return node_rank * ngpus_per_node + local_rank


def _cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_node(device_index: int) -> Optional[int]:
    """NUMA node that owns GPU ``device_index`` (``/sys/bus/pci/devices/<bdf>/numa_node``), None when unknown."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:  # noqa: BLE001  (no GPU, no sysfs, old torch)
        return None


def numa_cpus(node: int) -> List[int]:
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            return _cpulist(f.read())
    except OSError:
        return []


def plan_affinity(local_rank: int, nprocs: int, allowed: List[int], nodes: List[Optional[int]],
                  node_cpus) -> List[int]:
    """CPUs for ``local_rank``: the allowed CPUs of the NUMA node its GPU hangs off, divided among the local ranks
    whose GPUs share that node (pinned input buffers and the H2D copies then stay node-local); an even slice of
    the allowed CPUs when the topology is unknown.  Pure function of its inputs (tested on fake topologies)."""
    allowed = sorted(allowed)
    node = nodes[local_rank] if local_rank < len(nodes) else None
    if node is not None:
        local = [c for c in node_cpus(node) if c in set(allowed)]
        peers = [r for r in range(min(nprocs, len(nodes))) if nodes[r] == node]
        if local and local_rank in peers:
            share = max(1, len(local) // len(peers))
            k = peers.index(local_rank)
            mine = local[k * share:(k + 1) * share]
            if mine:
                return mine
    share = max(1, len(allowed) // max(1, nprocs))
    return allowed[local_rank * share:(local_rank + 1) * share] or allowed


def set_affinity(local_rank: int, nprocs: int, use_gpu: bool = False) -> List[int]:
    """Pin this rank to the CPUs next to its GPU (see ``plan_affinity``).  Call after ``set_device``."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        nodes = [gpu_numa_node(i) for i in range(nprocs)] if use_gpu else [None] * nprocs
        mine = plan_affinity(local_rank, nprocs, allowed, nodes, numa_cpus)
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
        return mine
    except (AttributeError, OSError):
        return []


def set_affinity_from_env(use_gpu: bool = True) -> List[int]:
    """Ranks started by an external launcher (torchrun) get the same placement as ``spawn``'s children."""
    if os.environ.get("DPA_AFFINITY", "1") == "0":
        return []
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    n = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return set_affinity(lr, max(n, lr + 1), use_gpu=use_gpu)


def init_process_group(backend: str, init_method: str, world_size: int, rank: int,
                       timeout_s: float = 600.0, use_gpu: bool = True) -> None:
    """``--dist-backend nccl|rccl|gloo`` -> torch.distributed; ``native`` -> TcpStore + C++ RCCL communicator."""
    from .parallel import process_group as pg

    pg.init_process_group(backend, init_method, world_size, rank, timeout_s=timeout_s, use_gpu=use_gpu)


def shutdown() -> None:
    from .parallel import process_group as pg

    pg.destroy_process_group()


# ----------------------------------------------------------------------------- supervisor
def rank_registry() -> str:
    """Directory of pid files of live ranks on this node (read by scripts/process_cleanup.sh)."""
    d = os.path.join(os.environ.get("DPA_RUN_DIR", "/tmp"), "dpa_ranks_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    return d


def _register(pid: int) -> str:
    path = os.path.join(rank_registry(), str(pid))
    with open(path, "w") as f:
        f.write(" ".join(sys.argv))
    return path


def _child(fn: Callable, local_rank: int, nprocs: int, args: tuple, err_q) -> None:
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["LOCAL_WORLD_SIZE"] = str(nprocs)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        fn(local_rank, *args)
    except KeyboardInterrupt:
        sys.exit(130)
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001
        err_q.put((local_rank, traceback.format_exc()))
        sys.exit(1)


class ProcessFailure(RuntimeError):
    def __init__(self, local_rank: int, exitcode: int, tb: str = ""):
        super().__init__("rank-local process %d failed with exit code %s\n%s"
                         % (local_rank, exitcode, tb))
        self.local_rank, self.exitcode, self.tb = local_rank, exitcode, tb


def spawn(fn: Callable, nprocs: int, args: tuple = (), poll_s: float = 0.2,
          grace_s: float = 5.0) -> None:
    """Run ``fn(local_rank, *args)`` in ``nprocs`` fresh processes (spawn context,
    so no GPU state is inherited).  First failure => SIGTERM, then SIGKILL, to all
    siblings, and a ``ProcessFailure`` carrying the child's traceback."""
    ctx = mp.get_context("spawn")
    err_q = ctx.SimpleQueue()
    procs = []
    for lr in range(nprocs):
        p = ctx.Process(target=_child, args=(fn, lr, nprocs, args, err_q), daemon=False)
        p.start()
        procs.append(p)
    pidfiles = [_register(p.pid) for p in procs]
    failed: Optional[Tuple[int, int]] = None
    try:
        alive = set(range(nprocs))
        while alive and failed is None:
            for i in list(alive):
                p = procs[i]
                if p.exitcode is not None:
                    alive.discard(i)
                    if p.exitcode != 0:
                        failed = (i, p.exitcode)
                        break
            if alive and failed is None:
                time.sleep(poll_s)
    except KeyboardInterrupt:
        failed = (-1, 130)
    finally:
        if failed is not None:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            deadline = time.time() + grace_s
            for p in procs:
                p.join(max(0.0, deadline - time.time()))
            for p in procs:
                if p.is_alive():
                    os.kill(p.pid, signal.SIGKILL)
                    p.join(1.0)
    for f in pidfiles:
        try:
            os.unlink(f)
        except OSError:
            pass
    if failed is not None:
        tb = ""
        while not err_q.empty():
            lr, text = err_q.get()
            if lr == failed[0] or not tb:
                tb = text
        raise ProcessFailure(failed[0], failed[1], tb)


def maybe_fault_inject(spec: Optional[str], rank: int, step: int) -> None:
    """``--fault-inject RANK:STEP``: die hard (no cleanup) to exercise the supervisor."""
    if not spec:
        return
    r, _, s = spec.partition(":")
    if int(r) == rank and int(s) == step:
        os._exit(17)

"""Process-group facade: one small API over two rendezvous/collective engines.

* ``torch``  -- ``torch.distributed`` (``gloo`` for the CPU plumbing configuration, ``nccl`` == RCCL on
  MI355X), exactly what the reference initialises (``nd_imagenet.py:130-131``,
  ``imagenet_ddp.py:104-105``, ``imagenet_ddp_apex.py:124-125``).
* ``native`` -- no ``torch.distributed`` at all: the C++ ``TcpStore`` (``csrc/comm/store.cpp``) is the
  rendezvous (``tcp://host:port`` or ``env://`` with ``MASTER_ADDR/MASTER_PORT``), rank 0 publishes the
  ``ncclUniqueId`` through it and every rank builds the C++ RCCL ``Communicator`` (``csrc/comm/comm.cpp``).
  GPU tensors use RCCL on the communicator's own stream; CPU tensors (metrics in the plumbing tests) go
  through the store (every rank publishes its bytes, all ranks sum them in rank order -- deterministic,
  O(world^2) bytes, meant for small messages only).  SURVEY.md N-C1.

Everything above this file (engine, DDP wrapper, SyncBN, sampler, metrics) calls the module-level
functions below, never ``torch.distributed`` directly, so ``--dist-backend native`` switches the whole
job to the native runtime.
"""
from __future__ import annotations

import datetime
import json
import os
from typing import Optional

import torch

_NATIVE: Optional["NativeProcessGroup"] = None

NATIVE_NAMES = ("native", "rccl-native", "dpa")


def _tdist():
    import torch.distributed as dist

    return dist


class _Done:
    def wait(self):
        return True


class StoreComm:
    """TorchComm-shaped collectives for CPU tensors over the TcpStore (small messages, tests)."""

    backend = "store"
    supports_avg = True

    def __init__(self, store, rank: int, world_size: int):
        self.store, self.rank, self.world_size = store, rank, world_size
        self._seq = 0

    def _exchange(self, payload: bytes):
        self._seq += 1
        base = "coll/%d" % self._seq
        self.store.set("%s/%d" % (base, self.rank), payload)
        parts = [payload if r == self.rank else self.store.get("%s/%d" % (base, r))
                 for r in range(self.world_size)]
        if self.store.add(base + "/done", 1) == self.world_size:      # last reader cleans up
            for r in range(self.world_size):
                self.store.remove("%s/%d" % (base, r))
            self.store.remove(base + "/done")
        return parts

    def _sum(self, t: torch.Tensor) -> torch.Tensor:
        flat = t.detach().contiguous().view(-1)
        acc_dtype = torch.float64 if flat.dtype.is_floating_point else torch.int64
        parts = self._exchange(flat.to(acc_dtype).numpy().tobytes())
        acc = torch.zeros(flat.numel(), dtype=acc_dtype)
        for p in parts:                                               # rank order on every rank
            acc += torch.frombuffer(bytearray(p), dtype=acc_dtype)
        return acc

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        if self.world_size > 1:
            t.copy_(self._sum(t).to(t.dtype).view_as(t))
        return _Done()

    def all_reduce_mean(self, t: torch.Tensor, async_op: bool = True):
        if self.world_size > 1:
            t.copy_((self._sum(t) / self.world_size).to(t.dtype).view_as(t))
        return _Done()

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if self.world_size > 1:
            self._seq += 1
            key = "bcast/%d" % self._seq
            if self.rank == src:
                self.store.set(key, t.detach().contiguous().view(-1).view(torch.uint8).numpy().tobytes())
            else:
                raw = torch.frombuffer(bytearray(self.store.get(key)), dtype=torch.uint8)
                t.copy_(raw.view(t.dtype).view_as(t))
            if self.store.add(key + "/done", 1) == self.world_size:
                self.store.remove(key)
                self.store.remove(key + "/done")
        return _Done()

    def all_gather(self, out: torch.Tensor, t: torch.Tensor):
        parts = self._exchange(t.detach().contiguous().view(-1).view(torch.uint8).numpy().tobytes()) \
            if self.world_size > 1 else [t.detach().contiguous().view(-1).view(torch.uint8).numpy().tobytes()]
        for r, p in enumerate(parts):
            out[r].copy_(torch.frombuffer(bytearray(p), dtype=torch.uint8).view(t.dtype).view_as(t))

    def all_reduce_max(self, t: torch.Tensor):
        if self.world_size > 1:
            flat = t.detach().contiguous().view(-1)
            parts = self._exchange(flat.to(torch.float64).numpy().tobytes())
            acc = torch.frombuffer(bytearray(parts[0]), dtype=torch.float64).clone()
            for p in parts[1:]:
                acc = torch.maximum(acc, torch.frombuffer(bytearray(p), dtype=torch.float64))
            t.copy_(acc.to(t.dtype).view_as(t))
        return _Done()

    def barrier(self):
        self.store.barrier("pg", self.world_size)

    def make_reducer(self, flats, param_bucket, chunk_bytes: int = 128 << 20, timeout_s: float = 0.0,
                     exit_on_timeout: bool = True, grace_s: float = 10.0, wire_dtype: int = -1):
        """The C++ BucketReducer over host memory: the same state machine that drives RCCL on a GPU, with a
        worker thread + the TcpStore as its communication queue (``csrc/comm/host_backend.cpp``)."""
        from .. import backend as _backend

        ext = _backend.require()
        _REDUCER_GEN[0] += 1
        # the worker's own store timeout sits well behind the watchdog's: a silent peer is the watchdog's case
        be = ext.HostBackend(self.store.host, self.store.port, self.rank, self.world_size,
                             3.0 * float(timeout_s) + 30.0 if timeout_s and timeout_s > 0 else 300.0,
                             "g%d" % _REDUCER_GEN[0])
        return ext.BucketReducer(be, [f.data_ptr() for f in flats], [f.numel() for f in flats],
                                 [_HOST_DT[f.dtype] for f in flats], list(param_bucket), int(chunk_bytes),
                                 float(timeout_s or 0.0), bool(exit_on_timeout), float(grace_s), int(wire_dtype))


_REDUCER_GEN = [0]      # reducers are built in the same order on every rank: the count names their store keys
_HOST_DT = {torch.float32: 0, torch.bfloat16: 1}


def _agent_store():
    """torchrun's rendezvous store (it owns MASTER_PORT while the job runs), used for ONE key: the port of our
    own TcpStore.  None when the job was not started by torchrun."""
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() != "true":
        return None
    import torch.distributed as dist

    return dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), None, False,
                         datetime.timedelta(seconds=120))


class NativeProcessGroup:
    """TcpStore rendezvous + (on a GPU) the C++ RCCL communicator."""

    def __init__(self, init_method: str, world_size: int, rank: int, timeout_s: float = 600.0,
                 use_gpu: Optional[bool] = None):
        from .. import backend as _backend
        from ..launch import parse_dist_url

        ext = _backend.require()
        scheme, host, port = parse_dist_url(init_method)
        if scheme == "env":
            rank = int(os.environ.get("RANK", rank if rank is not None and rank >= 0 else 0))
            world_size = int(os.environ.get("WORLD_SIZE", world_size if world_size and world_size > 0 else 1))
        if scheme == "file" or host is None or port is None:
            raise ValueError("the native process group needs tcp://host:port or env:// with MASTER_ADDR/MASTER_PORT")
        self.rank, self.world_size = int(rank), int(world_size)
        agent = _agent_store() if scheme == "env" else None
        if agent is not None:
            # under torchrun MASTER_PORT belongs to the agent's store: ours takes an ephemeral port that rank 0
            # publishes through it
            key = "dpa/native_store_port/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
            if self.rank == 0:
                self.store = ext.TcpStore(host, 0, True, float(timeout_s))
                agent.set(key, str(self.store.port))
            else:
                self.store = ext.TcpStore(host, int(agent.get(key).decode()), False, float(timeout_s))
        else:
            self.store = ext.TcpStore(host, int(port), self.rank == 0, float(timeout_s))
        self.host_comm = StoreComm(self.store, self.rank, self.world_size)
        self.gpu_comm = None
        use_gpu = torch.cuda.is_available() if use_gpu is None else use_gpu
        # census: fail early (with a timeout, not a hang) when a rank never shows up
        self.store.add("census", 1)
        self.store.barrier("init", self.world_size)
        if use_gpu and os.environ.get("DPA_COMM", "auto") == "peer":
            # peer-memory communicator (csrc/comm/peer.hip): hipIpc windows instead of RCCL -- also between ranks that
            # share one device
            from .peer_comm import PeerNativeComm

            self.gpu_comm = PeerNativeComm.build(self.rank, self.world_size, torch.cuda.current_device(),
                                                 self.host_comm._exchange)
        elif use_gpu:
            from .native_comm import NativeComm

            # pre-flight: ncclCommInitRank blocks until EVERY rank calls it, so a rank that cannot even load RCCL (or
            # sees no device) must be known before anyone enters it
            pre = ""
            try:
                ext.rccl_version()
                torch.cuda.current_device()
            except Exception as exc:  # noqa: BLE001
                pre = "rank %d: %r" % (self.rank, exc)
            pre_errs = [e.decode() for e in self.host_comm._exchange(pre.encode()) if e]
            if pre_errs:
                raise RuntimeError("native RCCL communicator could not be built: " + "; ".join(pre_errs))
            if self.rank == 0:
                self.store.set("rccl_unique_id", bytes(ext.rccl_unique_id()))
            uid = self.store.get("rccl_unique_id")
            # every rank learns whether EVERY rank built and verified its communicator: a rank that failed alone
            # would otherwise leave the others waiting inside their first collective
            err = ""
            try:
                raw = ext.Communicator(uid, self.rank, self.world_size, torch.cuda.current_device())
                self.gpu_comm = NativeComm(raw, None, self.host_comm._exchange)
                self.gpu_comm._self_test()
            except Exception as exc:  # noqa: BLE001
                err = "rank %d: %r" % (self.rank, exc)
            errs = [e.decode() for e in self.host_comm._exchange(err.encode()) if e]
            if errs:
                self.gpu_comm = None
                raise RuntimeError("native RCCL communicator could not be built: " + "; ".join(errs))

    def comm_for(self, t: Optional[torch.Tensor] = None):
        if t is not None and t.is_cuda:
            if self.gpu_comm is None:
                raise RuntimeError("native process group was initialised without a GPU communicator")
            return self.gpu_comm
        return self.host_comm

    def broadcast_object(self, obj, src: int = 0):
        """JSON only: store contents are bytes from the network and are never unpickled."""
        key = "obj/%d" % self.store.add("obj_seq/%d" % self.rank, 1)
        if self.rank == src:
            self.store.set(key, json.dumps(obj).encode())
            return obj
        return json.loads(self.store.get(key).decode())

    def barrier(self):
        self.store.barrier("pg", self.world_size)
        if self.gpu_comm is not None:
            torch.cuda.current_stream().synchronize()

    def destroy(self):
        self.gpu_comm = None
        self.host_comm = None
        store, self.store = self.store, None
        if store is not None:
            store.close()       # rank 0 keeps serving until the other ranks have hung up (store_core.h)


# ------------------------------------------------------------------------------- module-level API
def init_process_group(backend: str, init_method: str, world_size: int, rank: int,
                       timeout_s: float = 600.0, use_gpu: bool = True) -> None:
    global _NATIVE
    from ..launch import resolve_backend

    name = (backend or "nccl").lower()
    if name in NATIVE_NAMES:
        _NATIVE = NativeProcessGroup(init_method, world_size, rank, timeout_s, use_gpu=use_gpu)
        return
    name = resolve_backend(name, use_gpu)
    kwargs = dict(backend=name, init_method=init_method, world_size=world_size, rank=rank,
                  timeout=datetime.timedelta(seconds=timeout_s))
    if name == "nccl" and use_gpu:
        try:
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
        except Exception:  # noqa: BLE001
            pass
    _tdist().init_process_group(**kwargs)


def native() -> Optional[NativeProcessGroup]:
    return _NATIVE


def is_initialized() -> bool:
    if _NATIVE is not None:
        return True
    d = _tdist()
    return d.is_available() and d.is_initialized()


def get_rank(group=None) -> int:
    if _NATIVE is not None:
        return _NATIVE.rank
    return _tdist().get_rank(group) if is_initialized() else 0


def get_world_size(group=None) -> int:
    if _NATIVE is not None:
        return _NATIVE.world_size
    return _tdist().get_world_size(group) if is_initialized() else 1


def get_backend() -> str:
    if _NATIVE is not None:
        return "native"
    return _tdist().get_backend() if is_initialized() else "none"


def check_health() -> None:
    """Surface asynchronous communication failures (a timed-out peer all-reduce) on the host."""
    from .native_comm import shared

    nc = shared()
    if nc is not None:
        nc.check_health()


def barrier() -> None:
    check_health()
    if _NATIVE is not None:
        _NATIVE.barrier()
    elif is_initialized():
        d = _tdist()
        if d.get_backend() == "nccl":
            # host-side barrier through the rendezvous store: no collective, hence no second RCCL communicator
            from .native_comm import shared

            if shared() is not None:
                torch.cuda.current_stream().synchronize()
                _store_barrier(d)
                return
        d.barrier()


_BARRIER_SEQ = [0]


def _store_barrier(d) -> None:
    store = d.distributed_c10d._get_default_store()
    world = d.get_world_size()
    _BARRIER_SEQ[0] += 1
    key = "dpa/barrier/%d" % _BARRIER_SEQ[0]
    if store.add(key, 1) == world:
        store.set(key + "/go", b"1")
    store.wait([key + "/go"])


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks, complete (stream-ordered on a GPU) when the call returns."""
    if _NATIVE is not None:
        _NATIVE.comm_for(t).all_reduce_sum(t, async_op=False)
    elif is_initialized():
        nc = None
        if t.is_cuda and group is None:
            from .native_comm import shared

            nc = shared()           # DDP's native communicator: small messages take the peer-memory path
        if nc is not None:
            nc.all_reduce_sum(t, async_op=False)
        else:
            _tdist().all_reduce(t, op=_tdist().ReduceOp.SUM, group=group)
    return t


def _shared_native(t: torch.Tensor, group):
    """DDP's native communicator when the job runs on torch's process group: collectives on GPU tensors take it,
    so torch's own (lazily built) RCCL communicator is never created -- one communicator per rank."""
    if t.is_cuda and group is None:
        from .native_comm import shared

        return shared()
    return None


def all_reduce_max(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place maximum over ranks (benchmark timing: the slowest rank's wall time)."""
    if _NATIVE is not None:
        _NATIVE.comm_for(t).all_reduce_max(t)
    elif is_initialized():
        nc = _shared_native(t, group)
        if nc is not None:
            nc.all_reduce_max(t)
        else:
            _tdist().all_reduce(t, op=_tdist().ReduceOp.MAX, group=group)
    return t


def broadcast(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if _NATIVE is not None:
        _NATIVE.comm_for(t).broadcast(t, src=src, async_op=False)
    elif is_initialized():
        nc = _shared_native(t, group)
        if nc is not None:
            nc.broadcast(t, src=src, async_op=False)
        else:
            _tdist().broadcast(t, src=src, group=group)
    return t


def destroy_process_group() -> None:
    global _NATIVE
    if _NATIVE is not None:
        try:
            _NATIVE.destroy()
        finally:
            _NATIVE = None
        return
    d = _tdist()
    if d.is_available() and d.is_initialized():
        try:
            d.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass

"""Loader for the in-tree native extension ``_C`` (HIP kernels + C++ runtime).

The extension is built by ``python -m distributed_pytorch_amd.build`` (explicit
``hipcc --offload-arch=gfx950`` per ``csrc/kernels/*.hip`` + a pybind11 binding,
no hipify, no torch headers) and lives next to this file so it travels with the
tree to the GPU box.

Policy: CPU tensors run the pure-PyTorch reference backend; GPU tensors ALWAYS
run the HIP kernels.  There is no silent fallback on a GPU: if ``_C`` is missing
``require()`` raises.
"""
from __future__ import annotations

import importlib
import os
from typing import Optional

import torch  # noqa: F401  (must be imported first: _C binds to torch's libamdhip64 / librccl)

_EXT = None
_ERR: Optional[BaseException] = None


def load():
    """Returns the extension module or ``None`` (never raises)."""
    global _EXT, _ERR
    if _EXT is None and _ERR is None:
        try:
            _EXT = importlib.import_module("distributed_pytorch_amd._C")
        except BaseException as exc:  # ImportError, OSError (missing symbol), ...
            _ERR = exc
    return _EXT


def available() -> bool:
    return load() is not None


def require():
    ext = load()
    if ext is None:
        raise RuntimeError(
            "distributed_pytorch_amd._C is not built/loadable (%r). Run "
            "`python -m distributed_pytorch_amd.build` (needs hipcc, no GPU required). "
            "GPU tensors never fall back to PyTorch ops." % (_ERR,))
    return ext


def stream_handle() -> int:
    """Raw ``hipStream_t`` of torch's current stream, as an integer."""
    return torch.cuda.current_stream().cuda_stream


def host_only() -> bool:
    """True when the run should use the CPU reference backend."""
    return os.environ.get("DPA_FORCE_CPU", "0") == "1" or not torch.cuda.is_available()


# ---------------------------------------------------------------------------------------------- HIP stream budget
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 on the target boxes, HIP's default) in
# creation order; streams beyond that share a queue with an earlier one and then run IN ORDER with it.  Two round-1
# findings follow from that map (profiles/r1_notes.md, profiles/r2_notes.md): the input copy stream must be created
# first or it serialises with the compute stream (-4 % under DDP), and a high-priority communication stream that
# exchanged events with the default stream deadlocked once it shared a queue with it.  The framework therefore owns a
# fixed, small set of streams -- default (compute), input copy, communication, optional weight-gradient side stream --
# registers each here when it is created, and warns when the set no longer fits the hardware queues.
_streams = []


def register_stream(name: str) -> None:
    import os
    import warnings

    if name in _streams:
        return
    _streams.append(name)
    budget = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    if len(_streams) + 1 > budget:          # + the default stream
        warnings.warn("HIP streams %s + default exceed GPU_MAX_HW_QUEUES=%d: streams now share hardware queues and run "
                      "in order with each other (copy / communication overlap is lost)" % (_streams, budget))


def streams() -> list:
    return ["default"] + list(_streams)

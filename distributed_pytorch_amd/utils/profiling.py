"""roctx ranges for ``--profile`` (the reference has no tracing at all,
SURVEY.md §5.1).  Ranges show up in ``rocprofv3 --marker-trace``; they are no-ops
unless enabled, so the hot loop pays nothing by default."""
from __future__ import annotations

import contextlib
import ctypes

_lib = None
_enabled = False


def enable(flag: bool = True) -> None:
    global _lib, _enabled
    _enabled = flag
    if flag and _lib is None:
        for name in ("libroctx64.so", "libroctx64.so.4", "librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            _enabled = False


@contextlib.contextmanager
def range_(name: str):
    if _enabled and _lib is not None:
        _lib.roctxRangePushA(name.encode())
        try:
            yield
        finally:
            _lib.roctxRangePop()
    else:
        yield

"""Tracing and throughput metering.

The reference has no profiler hooks and no throughput meter (only the wall-clock
``Training elapsed time`` line, /root/reference/distributed.py:158-161).  Here:

* ``range_push/range_pop/annotate`` emit roctx ranges (visible in ``rocprofv3 --marker-trace``)
  when ``DTF_TRACE=1``; they are no-ops otherwise, so call sites stay in production code.
* ``ThroughputMeter`` implements the examples/sec definition used by the benchmarks:
  ``batch_size x applied updates x (replicas_to_aggregate if sync else 1) / seconds``.
* ``steps_to_target`` scans a loss trajectory for the first step at or below a target (the
  reference loss applies softmax twice, so it lives in (1.461, 2.303]: targets must too).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import time
from typing import Iterable, Optional

_roctx = None
_enabled = os.environ.get("DTF_TRACE", "0") == "1"


def _lib():
    global _roctx
    if _roctx is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            for prefix in ("", "/opt/rocm/lib/"):
                try:
                    _roctx = ctypes.CDLL(prefix + name)
                    break
                except OSError:
                    continue
            if _roctx is not None:
                break
        if _roctx is None:
            _roctx = False
    return _roctx


def enable(on: bool = True) -> None:
    global _enabled
    _enabled = on


def range_push(name: str) -> None:
    if _enabled and _lib():
        _lib().roctxRangePushA(name.encode())


def range_pop() -> None:
    if _enabled and _lib():
        _lib().roctxRangePop()


@contextlib.contextmanager
def annotate(name: str):
    range_push(name)
    try:
        yield
    finally:
        range_pop()


class ThroughputMeter:
    def __init__(self, batch_size: int, replicas_per_update: int = 1):
        self.batch_size, self.replicas = batch_size, replicas_per_update
        self.t0 = time.perf_counter()
        self.updates0: Optional[int] = None
        self.last = None

    def start(self, applied_updates: int) -> None:
        self.t0, self.updates0 = time.perf_counter(), applied_updates

    def rate(self, applied_updates: int) -> float:
        """examples/sec since ``start``."""
        dt = max(time.perf_counter() - self.t0, 1e-12)
        self.last = self.batch_size * (applied_updates - (self.updates0 or 0)) * self.replicas / dt
        return self.last


def steps_to_target(losses: Iterable[float], target: float, window: int = 1) -> Optional[int]:
    """1-based index of the first step whose trailing-``window`` mean loss is <= target, else None."""
    buf = []
    for i, l in enumerate(losses, 1):
        buf.append(l)
        if len(buf) > window:
            buf.pop(0)
        if len(buf) == window and sum(buf) / window <= target:
            return i
    return None

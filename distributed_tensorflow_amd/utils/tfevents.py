"""TensorBoard event files (``events.out.tfevents.*``) in the Supervisor's log directory.

The reference's ``tf.train.Supervisor(..., global_step=global_step)`` (/root/reference/distributed.py:110-111)
writes a ``global_step/sec`` summary into an events file of its logdir.  The writer is native
(``csrc/runtime/events_writer.cpp``: TFRecord framing with masked crc32c, hand-encoded Event/Summary protobufs);
the reader below is independent Python, used by the tests and handy for inspecting a run without TensorBoard.
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import time
from typing import Iterator, List, Optional, Tuple

from ..checkpoint.pyformat import crc32c as _py_crc32c
from . import native


def _lib():
    lib = native._rt()
    if not getattr(lib, "_events_declared", False):
        lib.dtf_events_open.restype = C.c_void_p
        lib.dtf_events_open.argtypes = [C.c_char_p, C.c_double]
        lib.dtf_events_scalar.restype = C.c_int
        lib.dtf_events_scalar.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_char_p, C.c_float]
        lib.dtf_events_close.restype = C.c_int
        lib.dtf_events_close.argtypes = [C.c_void_p]
        lib._events_declared = True
    return lib


class EventFileWriter:
    def __init__(self, logdir: str, now: Optional[float] = None):
        now = time.time() if now is None else now
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(now), socket.gethostname()))
        self._h = _lib().dtf_events_open(self.path.encode(), float(now))
        if not self._h:
            raise OSError(f"cannot open event file {self.path}")

    def add_scalar(self, tag: str, value: float, step: int, wall_time: Optional[float] = None) -> None:
        rc = _lib().dtf_events_scalar(self._h, time.time() if wall_time is None else wall_time, int(step), tag.encode(),
                                      float(value))
        if rc:
            raise OSError(f"event file write failed ({rc})")

    def close(self) -> None:
        if self._h:
            _lib().dtf_events_close(self._h)
            self._h = None


# ---------------------------------------------------------------------------------- reader (tests, inspection)
def _mask(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    v = s = 0
    while True:
        x = b[i]
        i += 1
        v |= (x & 0x7F) << s
        if x < 0x80:
            return v, i
        s += 7


def _fields(b: bytes) -> Iterator[Tuple[int, int, object]]:
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        elif wt == 2:
            n, i = _varint(b, i)
            v, i = b[i:i + n], i + n
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def read_events(path: str) -> List[dict]:
    """Every record of an event file: ``{"wall_time", "step", "file_version" | "scalars": {tag: value}}``;
    raises ``ValueError`` on a framing or checksum error."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 12]
        if len(head) < 12:
            raise ValueError("truncated record header")
        (n,), (hc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
        if _mask(_py_crc32c(head[:8])) != hc:
            raise ValueError("length checksum mismatch")
        body = data[i + 12:i + 12 + n]
        (dc,) = struct.unpack("<I", data[i + 12 + n:i + 16 + n])
        if len(body) != n or _mask(_py_crc32c(body)) != dc:
            raise ValueError("data checksum mismatch")
        i += 16 + n
        ev = {"wall_time": 0.0, "step": 0}
        for fld, _, v in _fields(body):
            if fld == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif fld == 2:
                ev["step"] = v
            elif fld == 3:
                ev["file_version"] = v.decode()
            elif fld == 5:
                scalars = {}
                for f2, _, val in _fields(v):
                    if f2 == 1:
                        tag, num = None, None
                        for f3, _, x in _fields(val):
                            if f3 == 1:
                                tag = x.decode()
                            elif f3 == 2:
                                num = struct.unpack("<f", x)[0]
                        scalars[tag] = num
                ev["scalars"] = scalars
        out.append(ev)
    return out

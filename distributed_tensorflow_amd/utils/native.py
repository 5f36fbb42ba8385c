"""ctypes front end of the host data-pipeline functions (``csrc/runtime/idx_reader.cpp``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ..ops import _lib

_declared = False


def available() -> bool:
    return _lib.runtime_available()


def _rt():
    global _declared
    lib = _lib.runtime()
    if not _declared:
        lib.dtf_idx_open.restype = C.c_void_p
        lib.dtf_idx_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.dtf_idx_ndim.restype = C.c_int
        lib.dtf_idx_ndim.argtypes = [C.c_void_p]
        lib.dtf_idx_dim.restype = C.c_int64
        lib.dtf_idx_dim.argtypes = [C.c_void_p, C.c_int]
        lib.dtf_idx_copy_u8.restype = None
        lib.dtf_idx_copy_u8.argtypes = [C.c_void_p, C.c_void_p]
        lib.dtf_idx_copy_scaled_f32.restype = None
        lib.dtf_idx_copy_scaled_f32.argtypes = [C.c_void_p, C.c_void_p]
        lib.dtf_idx_copy_one_hot.restype = C.c_int64
        lib.dtf_idx_copy_one_hot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.dtf_idx_close.restype = None
        lib.dtf_idx_close.argtypes = [C.c_void_p]
        lib.dtf_gather_rows.restype = C.c_int
        lib.dtf_gather_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p]
        _declared = True
    return lib


class _Idx:
    def __init__(self, path: str):
        lib = _rt()
        err = C.create_string_buffer(512)
        self.h = lib.dtf_idx_open(path.encode(), err, 512)
        if not self.h:
            msg = err.value.decode()
            if "cannot read file" in msg:
                raise FileNotFoundError(msg)
            raise ValueError(msg)
        self.shape = tuple(lib.dtf_idx_dim(self.h, i) for i in range(lib.dtf_idx_ndim(self.h)))

    def close(self):
        if self.h:
            _rt().dtf_idx_close(self.h)
            self.h = None


def read_idx(path: str) -> np.ndarray:
    f = _Idx(path)
    try:
        out = np.empty(f.shape, dtype=np.uint8)
        _rt().dtf_idx_copy_u8(f.h, out.ctypes.data_as(C.c_void_p))
        return out
    finally:
        f.close()


def read_images_f32(path: str) -> np.ndarray:
    """[n, rows*cols] float32 in [0, 1]."""
    f = _Idx(path)
    try:
        n = f.shape[0]
        out = np.empty((n, int(np.prod(f.shape[1:]))), dtype=np.float32)
        _rt().dtf_idx_copy_scaled_f32(f.h, out.ctypes.data_as(C.c_void_p))
        return out
    finally:
        f.close()


def read_labels_one_hot(path: str, classes: int = 10) -> np.ndarray:
    f = _Idx(path)
    try:
        out = np.empty((f.shape[0], classes), dtype=np.float32)
        bad = _rt().dtf_idx_copy_one_hot(f.h, out.ctypes.data_as(C.c_void_p), classes)
        if bad:
            raise ValueError(f"{path}: {bad} labels are outside [0, {classes})")
        return out
    finally:
        f.close()


def gather_rows(src: np.ndarray, rows: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.empty((rows.shape[0],) + src.shape[1:], dtype=src.dtype)
    rc = _rt().dtf_gather_rows(src.ctypes.data_as(C.c_void_p), src.shape[0], rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                               src.strides[0], out.ctypes.data_as(C.c_void_p))
    if rc:
        raise IndexError("row index out of range")
    return out

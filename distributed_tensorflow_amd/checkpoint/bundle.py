"""ctypes front end of the native tensor-bundle reader/writer (``csrc/runtime/bundle.cpp``)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from ..ops import _lib

# TensorFlow DataType enum values used in BundleEntryProto.dtype
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT64 = 1, 2, 3, 4, 9
_NP_TO_DT = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
             np.dtype("uint8"): DT_UINT8, np.dtype("int64"): DT_INT64}
_DT_TO_NP = {v: k for k, v in _NP_TO_DT.items()}

_declared = False


def _rt():
    global _declared
    lib = _lib.runtime()
    if not _declared:
        lib.dtf_bundle_writer_open.restype = C.c_void_p
        lib.dtf_bundle_writer_open.argtypes = [C.c_char_p]
        lib.dtf_bundle_writer_add.restype = C.c_int
        lib.dtf_bundle_writer_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_void_p,
                                              C.c_size_t]
        lib.dtf_bundle_writer_finish.restype = C.c_int
        lib.dtf_bundle_writer_finish.argtypes = [C.c_void_p]
        lib.dtf_bundle_writer_abort.restype = None
        lib.dtf_bundle_writer_abort.argtypes = [C.c_void_p]
        lib.dtf_bundle_reader_open.restype = C.c_void_p
        lib.dtf_bundle_reader_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.dtf_bundle_reader_num_entries.restype = C.c_int
        lib.dtf_bundle_reader_num_entries.argtypes = [C.c_void_p]
        lib.dtf_bundle_reader_entry.restype = C.c_int
        lib.dtf_bundle_reader_entry.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        lib.dtf_bundle_reader_read.restype = C.c_int
        lib.dtf_bundle_reader_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        lib.dtf_bundle_reader_close.restype = None
        lib.dtf_bundle_reader_close.argtypes = [C.c_void_p]
        lib.dtf_crc32c.restype = C.c_uint32
        lib.dtf_crc32c.argtypes = [C.c_void_p, C.c_size_t]
        lib.dtf_crc32c_masked.restype = C.c_uint32
        lib.dtf_crc32c_masked.argtypes = [C.c_void_p, C.c_size_t]
        _declared = True
    return lib


def crc32c(data: bytes) -> int:
    return _rt().dtf_crc32c(data, len(data))


def crc32c_masked(data: bytes) -> int:
    return _rt().dtf_crc32c_masked(data, len(data))


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``<prefix>.index`` and ``<prefix>.data-00000-of-00001``."""
    lib = _rt()
    w = lib.dtf_bundle_writer_open(prefix.encode())
    try:
        for name, arr in tensors.items():
            arr = np.asarray(arr)
            if not arr.flags.c_contiguous:
                arr = arr.copy(order="C")
            if arr.dtype not in _NP_TO_DT:
                raise TypeError(f"{name}: dtype {arr.dtype} has no bundle encoding here")
            if arr.dtype.byteorder == ">":
                arr = arr.astype(arr.dtype.newbyteorder("<"))
            dims = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            rc = lib.dtf_bundle_writer_add(w, name.encode(), _NP_TO_DT[arr.dtype], dims, arr.ndim,
                                           arr.ctypes.data_as(C.c_void_p), arr.nbytes)
            if rc:
                raise RuntimeError(f"bundle writer refused {name!r} (code {rc})")
    except Exception:
        lib.dtf_bundle_writer_abort(w)
        raise
    rc = lib.dtf_bundle_writer_finish(w)
    if rc:
        raise OSError(f"writing bundle {prefix!r} failed (code {rc})")


def read_bundle(prefix: str) -> Dict[str, np.ndarray]:
    """Read every tensor of a bundle, verifying block and tensor checksums."""
    lib = _rt()
    err = C.create_string_buffer(256)
    r = lib.dtf_bundle_reader_open(prefix.encode(), err, 256)
    if not r:
        raise OSError(f"cannot open bundle {prefix!r}: {err.value.decode()}")
    try:
        out = {}
        name = C.create_string_buffer(1024)
        dims = (C.c_int64 * 16)()
        dtype, ndims, nbytes = C.c_int(), C.c_int(), C.c_uint64()
        for i in range(lib.dtf_bundle_reader_num_entries(r)):
            rc = lib.dtf_bundle_reader_entry(r, i, name, 1024, C.byref(dtype), dims, 16, C.byref(ndims), C.byref(nbytes))
            if rc:
                raise RuntimeError(f"bundle entry {i}: code {rc}")
            if dtype.value not in _DT_TO_NP:
                raise TypeError(f"{name.value.decode()}: unsupported bundle dtype {dtype.value}")
            shape = tuple(dims[d] for d in range(ndims.value))
            arr = np.empty(shape, dtype=_DT_TO_NP[dtype.value])
            if arr.nbytes != nbytes.value:
                raise ValueError(f"{name.value.decode()}: size {nbytes.value} does not match shape {shape}")
            rc = lib.dtf_bundle_reader_read(r, name.value, arr.ctypes.data_as(C.c_void_p), arr.nbytes)
            if rc:
                raise OSError(f"{name.value.decode()}: read failed (code {rc}; 5 = checksum mismatch)")
            out[name.value.decode()] = arr
        return out
    finally:
        lib.dtf_bundle_reader_close(r)

"""Native RCCL communicator (``csrc/kernels/rccl_comm.hip``) and the fused update entry of the
step schedule: gradient all-reduce / reduce-scatter, optimizer on the owned slice and all-gather are
enqueued by the same native call as the compute kernels, on the same HIP stream."""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import torch
import torch.distributed as dist

from ..ops import _lib

UPDATE_LOCAL, UPDATE_ALLREDUCE, UPDATE_SHARDED = 0, 1, 2
OP_UPDATE = 6


class UpdateArgs(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p), ("rs_out", C.c_void_p),
                ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("total", C.c_int64),
                ("shard_offset", C.c_int64), ("shard_elems", C.c_int64), ("mode", C.c_int), ("optimizer", C.c_int),
                ("lr", C.c_float), ("grad_scale", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("pad_", C.c_int)]


_declared = False


def _lib_k():
    global _declared
    lib = _lib.kernels()
    if not _declared:
        lib.dtf_comm_load.restype = C.c_int
        lib.dtf_comm_load.argtypes = [C.c_char_p]
        lib.dtf_comm_unique_id.restype = C.c_int
        lib.dtf_comm_unique_id.argtypes = [C.c_char_p]
        lib.dtf_comm_init.restype = C.c_void_p
        lib.dtf_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.dtf_comm_destroy.restype = C.c_int
        lib.dtf_comm_destroy.argtypes = [C.c_void_p]
        lib.dtf_comm_count.restype = C.c_int
        lib.dtf_comm_count.argtypes = [C.c_void_p]
        lib.dtf_comm_error_string.restype = C.c_char_p
        lib.dtf_comm_error_string.argtypes = [C.c_int]
        for name in ("dtf_comm_all_reduce_f32",):
            getattr(lib, name).restype = C.c_int
        lib.dtf_comm_all_reduce_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dtf_comm_reduce_scatter_f32.restype = C.c_int
        lib.dtf_comm_reduce_scatter_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dtf_comm_all_gather_f32.restype = C.c_int
        lib.dtf_comm_all_gather_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dtf_comm_broadcast_bytes.restype = C.c_int
        lib.dtf_comm_broadcast_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        lib.dtf_run_update.restype = C.c_int
        lib.dtf_run_update.argtypes = [C.POINTER(UpdateArgs), C.c_void_p]
        _declared = True
    return lib


def rccl_library_path() -> Optional[str]:
    """The RCCL build PyTorch itself uses (one RCCL per process), else the ROCm one."""
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"]
    for c in cands:
        if os.path.exists(c):
            return c
    return None


def enabled() -> bool:
    return os.environ.get("DTF_NATIVE_COMM", "1") != "0" and _lib.kernels_available() and rccl_library_path() is not None


class NativeComm:
    """One RCCL communicator over all worker ranks, created from a unique id that rank 0 generates
    and ``exchange`` distributes (a broadcast over an existing torch.distributed group or a store)."""

    def __init__(self, rank: int, world: int, device: torch.device, exchange: Optional[Callable[[Optional[bytes]], bytes]] = None):
        lib = _lib_k()
        path = rccl_library_path()
        rc = lib.dtf_comm_load(path.encode())
        if rc:
            raise RuntimeError(f"cannot load RCCL from {path} (code {rc})")
        if device.index is not None:
            torch.cuda.set_device(device)
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            rc = lib.dtf_comm_unique_id(buf)
            if rc:
                raise RuntimeError(f"ncclGetUniqueId failed: {lib.dtf_comm_error_string(rc).decode()}")
            uid = buf.raw
        if world > 1:
            exchange = exchange or exchange_via_dist
            uid = exchange(uid)
        rc_out = C.c_int(0)
        self.handle = lib.dtf_comm_init(uid, world, rank, C.byref(rc_out))
        if not self.handle:
            raise RuntimeError(f"ncclCommInitRank failed: {lib.dtf_comm_error_string(rc_out.value).decode()}")
        self.rank, self.world = rank, world
        self._lib = lib

    def _check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what}: {self._lib.dtf_comm_error_string(rc).decode()}")

    def count(self) -> int:
        """Ranks of the communicator as RCCL reports them (``ncclCommCount``)."""
        return int(self._lib.dtf_comm_count(self.handle))

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        self._check(self._lib.dtf_comm_all_reduce_f32(self.handle, t.data_ptr(), t.numel(), _lib.stream_handle()), "all_reduce")

    def reduce_scatter_sum(self, out_shard: torch.Tensor, full: torch.Tensor) -> None:
        self._check(self._lib.dtf_comm_reduce_scatter_f32(self.handle, full.data_ptr(), out_shard.data_ptr(),
                                                          out_shard.numel(), _lib.stream_handle()), "reduce_scatter")

    def all_gather(self, full: torch.Tensor, shard: torch.Tensor) -> None:
        self._check(self._lib.dtf_comm_all_gather_f32(self.handle, shard.data_ptr(), full.data_ptr(), shard.numel(),
                                                      _lib.stream_handle()), "all_gather")

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        self._check(self._lib.dtf_comm_broadcast_bytes(self.handle, t.data_ptr(), t.numel() * t.element_size(), src,
                                                       _lib.stream_handle()), "broadcast")

    def destroy(self) -> None:
        if self.handle:
            self._lib.dtf_comm_destroy(self.handle)
            self.handle = None


def exchange_via_dist(uid: Optional[bytes]) -> bytes:
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def exchange_via_store(store, key: str = "dtf/rccl_uid"):
    def ex(uid: Optional[bytes]) -> bytes:
        if uid is not None:
            store.set(key, uid)
            return uid
        return bytes(store.get(key))
    return ex


def make_update_args(state, mode: int, grad_scale: float, comm: Optional[NativeComm] = None,
                     rs_out: Optional[torch.Tensor] = None) -> UpdateArgs:
    lo, hi = state.shard
    return UpdateArgs(comm=None if comm is None else comm.handle, params=state.params.data_ptr(), grads=state.grads.data_ptr(),
                      rs_out=_lib.ptr(rs_out), m=_lib.ptr(state.m), v=_lib.ptr(state.v), step=state.state[2:3].data_ptr(),
                      total=state.layout.total, shard_offset=lo, shard_elems=hi - lo, mode=mode,
                      optimizer=0 if state.optimizer == "sgd" else 1, lr=state.lr, grad_scale=grad_scale, beta1=state.beta1,
                      beta2=state.beta2, eps=state.eps, pad_=0)

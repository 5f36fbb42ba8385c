"""ctypes bindings of the in-tree native libraries.

``libdtf_kernels.so`` holds the gfx950 HIP kernels, ``libdtf_runtime.so`` the
host runtime (checkpoint bundle I/O, data pipeline, parameter-server service).
Both are built by ``csrc/Makefile`` into ``distributed_tensorflow_amd/lib``.

The kernel library is only loaded when a GPU is visible.  On a GPU box a
missing library is an error (never a silent fallback to eager PyTorch): see
:func:`kernels`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
_lock = threading.Lock()
_kernels = None
_runtime = None

ERR_NAMES = {1: "operand shapes do not match the kernel's grid assumptions", 2: "pointer/leading dimension not 16-byte aligned",
             3: "hip launch failed", 4: "null pointer or unsupported option"}


class NativeLibraryMissing(RuntimeError):
    pass


class TailArgs(C.Structure):
    _fields_ = [("state", C.c_void_p), ("log", C.c_void_p), ("loss_part", C.c_void_p), ("correct_part", C.c_void_p),
                ("post_part", C.c_void_p), ("ridx_cur", C.c_void_p), ("ridx_next", C.c_void_p), ("perm", C.c_void_p),
                ("nparts", C.c_int), ("log_cap", C.c_int), ("B", C.c_int), ("n_train", C.c_int), ("gs_inc", C.c_int),
                ("enable", C.c_int), ("inv_b", C.c_float), ("pad_", C.c_int)]


class SmallGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("a_rows", C.c_void_p), ("B", C.c_void_p), ("bias", C.c_void_p), ("mask", C.c_void_p),
                ("C", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int),
                ("ldc", C.c_int), ("ldmask", C.c_int), ("relu", C.c_int), ("b_transposed", C.c_int),
                ("a_bytes", C.c_uint), ("b_bytes", C.c_uint)]


MAX_WGRAD_PROBLEMS = 8


class WgradProblem(C.Structure):
    _fields_ = [("X", C.c_void_p), ("x_rows", C.c_void_p), ("dY", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("W", C.c_void_p), ("b", C.c_void_p), ("Mo", C.c_int), ("N", C.c_int), ("Kb", C.c_int), ("ldx", C.c_int),
                ("ldy", C.c_int), ("ldw", C.c_int), ("tile_start", C.c_int), ("tiles_n", C.c_int),
                ("x_bytes", C.c_uint), ("y_bytes", C.c_uint)]


class WgradArgs(C.Structure):
    _fields_ = [("prob", WgradProblem * MAX_WGRAD_PROBLEMS), ("nprob", C.c_int), ("total_tiles", C.c_int),
                ("mode", C.c_int), ("lr", C.c_float), ("tail", TailArgs)]


class HeadArgs(C.Structure):
    _fields_ = [("Hact", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("labels", C.c_void_p), ("rows", C.c_void_p),
                ("dz2", C.c_void_p), ("dz1", C.c_void_p), ("probs", C.c_void_p), ("loss_part", C.c_void_p),
                ("correct_part", C.c_void_p), ("post_part", C.c_void_p), ("M_total", C.c_int), ("M_train", C.c_int),
                ("Hd", C.c_int), ("C", C.c_int), ("ldh", C.c_int), ("ld_dz1", C.c_int), ("loss_mode", C.c_int),
                ("inv_b", C.c_float), ("h_bytes", C.c_uint), ("pad_", C.c_int), ("dz1_bf", C.c_void_p),
                ("dz1_bf_t", C.c_void_p), ("ld_bf", C.c_int), ("ld_bf_t", C.c_int)]


class HeadWideArgs(C.Structure):
    _fields_ = [("H", C.c_void_p), ("Ht", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("labels", C.c_void_p),
                ("rows", C.c_void_p), ("dz2", C.c_void_p), ("dz1_bf", C.c_void_p), ("dz1_bf_t", C.c_void_p),
                ("loss_part", C.c_void_p), ("correct_part", C.c_void_p), ("post_part", C.c_void_p), ("M", C.c_int),
                ("Hd", C.c_int), ("C", C.c_int), ("ldh", C.c_int), ("ldht", C.c_int), ("ld_bf", C.c_int),
                ("ld_bf_t", C.c_int), ("loss_mode", C.c_int), ("inv_b", C.c_float), ("pad_", C.c_int)]


class OutWgradWideArgs(C.Structure):
    _fields_ = [("Ht", C.c_void_p), ("dz2", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("Hd", C.c_int),
                ("C", C.c_int), ("B", C.c_int), ("ldht", C.c_int), ("ldw", C.c_int), ("pad_", C.c_int),
                ("tail", TailArgs)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("Ct", C.c_void_p), ("C32", C.c_void_p),
                ("bias", C.c_void_p), ("mask", C.c_void_p), ("bias_out", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("ldct", C.c_int), ("ldc32", C.c_int),
                ("ldmask", C.c_int), ("relu", C.c_int), ("epilogue", C.c_int), ("dtype", C.c_int), ("a_bytes", C.c_uint),
                ("b_bytes", C.c_uint), ("maskT", C.c_void_p), ("ldmaskt", C.c_int), ("pad_", C.c_int),
                ("opt_p", C.c_void_p), ("opt_m", C.c_void_p), ("opt_v", C.c_void_p), ("opt_w", C.c_void_p),
                ("opt_wt", C.c_void_p), ("opt_step", C.c_void_p), ("opt_step_offset", C.c_int), ("ldwt", C.c_int),
                ("opt_lr", C.c_float), ("opt_b1", C.c_float), ("opt_b2", C.c_float), ("opt_eps", C.c_float),
                ("ksplit", C.c_int), ("pad2_", C.c_int), ("split_ws", C.c_void_p), ("split_bias_ws", C.c_void_p)]


WLOG_INTS = 64


MAX_OPT_RANGES = 16


class OptRangesArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step_ptr", C.c_void_p),
                ("lo", C.c_longlong * MAX_OPT_RANGES), ("hi", C.c_longlong * MAX_OPT_RANGES), ("nranges", C.c_int),
                ("step_offset", C.c_int), ("lr", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float),
                ("gscale", C.c_float), ("pad_", C.c_int)]


class XentArgs(C.Structure):
    """Mirror of ``DtfXentArgs`` (softmax_xent.hip)."""
    _fields_ = [("logits", C.c_void_p), ("labels_f", C.c_void_p), ("labels_i", C.c_void_p), ("dlogits", C.c_void_p),
                ("dlogits_bf", C.c_void_p), ("loss_rows", C.c_void_p), ("loss_part", C.c_void_p), ("correct_part", C.c_void_p),
                ("B", C.c_int), ("C", C.c_int), ("ld_logits", C.c_int), ("ld_labels", C.c_int), ("ld_dlogits", C.c_int),
                ("ld_dlogits_bf", C.c_int), ("loss_mode", C.c_int), ("grad_scale", C.c_float), ("label_rows", C.c_void_p)]


class Mlp2Peer(C.Structure):
    """Mirror of ``DtfMlp2Peer``: peers' gradient slots and arrival flags for the in-launch exchange."""
    _fields_ = [("slot", C.c_void_p * 8), ("wgflags", C.c_void_p * 8), ("err", C.c_void_p), ("rank", C.c_int),
                ("world", C.c_int), ("timeout_ticks", C.c_uint), ("pad_", C.c_int)]


class Mlp2Async(C.Structure):
    """Mirror of ``DtfMlp2Async``: mapped master slices and counters for the in-launch push/pull."""
    _fields_ = [("master", C.c_void_p * 8), ("counters", C.c_void_p * 8), ("m", C.c_void_p * 8), ("v", C.c_void_p * 8),
                ("params_base", C.c_void_p), ("slice", C.c_longlong), ("shards", C.c_int), ("optimizer", C.c_int),
                ("pull_every", C.c_int), ("bound", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("timeout_ticks", C.c_uint), ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float),
                ("astate", C.c_void_p), ("alog", C.c_void_p), ("applied_snap", C.c_void_p), ("alog_cap", C.c_int),
                ("pad_", C.c_int)]


class Mlp2Shard(C.Structure):
    """Mirror of ``DtfMlp2Shard``: regions of all ranks and the box / flag offsets of the sharded exchange."""
    _fields_ = [("region", C.c_void_p * 8), ("params_base", C.c_void_p), ("err", C.c_void_p), ("inbox_off", C.c_longlong),
                ("outbox_off", C.c_longlong), ("inflag_off", C.c_longlong), ("outflag_off", C.c_longlong), ("n", C.c_longlong),
                ("rank", C.c_int), ("world", C.c_int), ("timeout_ticks", C.c_uint), ("adam", C.c_int),
                ("m", C.c_void_p), ("v", C.c_void_p), ("adam_lr", C.c_float), ("grad_scale", C.c_float), ("b1", C.c_float),
                ("b2", C.c_float), ("eps", C.c_float), ("adam_t_offset", C.c_int)]


class Mlp2Args(C.Structure):
    _fields_ = [("X", C.c_void_p), ("labels", C.c_void_p), ("Xg", C.c_void_p), ("labg", C.c_void_p), ("ridx", C.c_void_p),
                ("H", C.c_void_p), ("zpart", C.c_void_p), ("W2s", C.c_void_p), ("ctr_snap", C.c_void_p), ("W1", C.c_void_p),
                ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("dW1", C.c_void_p), ("db1", C.c_void_p),
                ("dW2", C.c_void_p), ("db2", C.c_void_p), ("wlog", C.c_void_p), ("In", C.c_int), ("Hd", C.c_int),
                ("C", C.c_int), ("B", C.c_int), ("R", C.c_int), ("ldx", C.c_int), ("ldh", C.c_int), ("ridx_len", C.c_int),
                ("h_rows", C.c_int), ("q", C.c_int), ("loss_mode", C.c_int), ("mode", C.c_int), ("log_cap", C.c_int), ("lr", C.c_float),
                ("inv_b", C.c_float), ("x_bytes", C.c_uint), ("tail", TailArgs), ("peer", Mlp2Peer), ("async_", Mlp2Async),
                ("shard", Mlp2Shard)]


class PipeArgs(C.Structure):
    """Mirror of ``DtfPipeArgs``: the one-launch pipelined step (mlp_pipe.hip)."""
    _fields_ = [("X", C.c_void_p), ("labels", C.c_void_p), ("Xa", C.c_void_p), ("labg", C.c_void_p), ("ridx", C.c_void_p),
                ("Wm", C.c_void_p), ("W2m", C.c_void_p), ("Q", C.c_void_p), ("Z1", C.c_void_p), ("labi", C.c_void_p),
                ("GG", C.c_void_p), ("D1", C.c_void_p), ("P2", C.c_void_p), ("ctr", C.c_void_p), ("W1", C.c_void_p),
                ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("wlog", C.c_void_p), ("In", C.c_int),
                ("Hd", C.c_int), ("C", C.c_int), ("B", C.c_int), ("R", C.c_int), ("HP", C.c_int), ("KAp", C.c_int),
                ("ldxa", C.c_int), ("ldx", C.c_int), ("ridx_len", C.c_int), ("q", C.c_int), ("loss_mode", C.c_int),
                ("log_cap", C.c_int), ("roles", C.c_int), ("flush", C.c_int), ("prime", C.c_int), ("td", C.c_int), ("lr", C.c_float),
                ("inv_b", C.c_float), ("tail", TailArgs)]


class PipeMultiArgs(C.Structure):
    """Mirror of ``DtfPipeMultiArgs``: the one-launch pipelined step of N synchronous workers (mlp_pipe_multi.hip)."""
    _fields_ = [("b", PipeArgs), ("region", C.c_void_p * 8), ("perm", C.c_void_p * 8), ("err", C.c_void_p),
                ("rank", C.c_int), ("world", C.c_int), ("timeout_ticks", C.c_uint), ("plain_order", C.c_int)]


PIPE_ROLE_L, PIPE_ROLE_Q, PIPE_ROLE_G, PIPE_ROLE_W, PIPE_ROLE_T, PIPE_ROLE_F, PIPE_ROLE_C = 1, 2, 4, 8, 16, 32, 64
PIPE_SLOTS = 8

OP_SMALL_GEMM, OP_HEAD, OP_WGRAD, OP_TAIL, OP_GEMM_TILED, OP_UPDATE, OP_MLP2_FWD, OP_MLP2_BWD = 1, 2, 3, 4, 5, 6, 7, 8
OP_SOFTMAX_XENT = 12


class ScheduleEntry(C.Structure):
    _fields_ = [("op", C.c_int), ("flag", C.c_int), ("args", C.c_void_p)]


class Schedule:
    """A launch sequence handed to the native executor in one call.  Keeps the argument structs alive."""

    def __init__(self, entries):
        self._keep = [a for _, a, _ in entries]
        self.n = len(entries)
        self.array = (ScheduleEntry * self.n)()
        for i, (op, arg, flag) in enumerate(entries):
            self.array[i].op, self.array[i].flag, self.array[i].args = op, flag, C.addressof(arg)

    def run(self):
        rc = kernels().dtf_run_schedule(self.array, self.n, stream_handle())
        if rc:
            raise RuntimeError(f"schedule entry {rc >> 8} refused its launch ({ERR_NAMES.get(rc & 0xff, rc & 0xff)})")


def run_schedule_steps(schedules, first_slot: int, n_steps: int) -> None:
    """``n_steps`` steps in ONE native call: step i runs ``schedules[(first_slot + i) % len(schedules)]``."""
    n = len(schedules)
    arr = (C.POINTER(ScheduleEntry) * n)(*[C.cast(s.array, C.POINTER(ScheduleEntry)) for s in schedules])
    counts = (C.c_int * n)(*[s.n for s in schedules])
    rc = kernels().dtf_run_schedule_steps(arr, counts, n, first_slot, n_steps, stream_handle())
    if rc:
        raise RuntimeError(f"schedule entry {rc >> 8} refused its launch ({ERR_NAMES.get(rc & 0xff, rc & 0xff)})")


def span_bytes(t) -> int:
    """Bytes addressable from ``t.data_ptr()`` for a row-major 2-D (or 1-D) tensor view."""
    if t.dim() == 1:
        n = t.shape[0]
    else:
        n = (t.shape[0] - 1) * t.stride(0) + t.shape[1]
    b = n * t.element_size()
    if b >= 0xFFFFFFF0:
        raise ValueError("operand of a small-MLP kernel must stay below 4 GiB (32-bit buffer offsets)")
    return b


def lib_path(name: str) -> str:
    return os.path.join(_LIB_DIR, name)


def _declare_kernels(lib):
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
    lib.dtf_small_gemm.argtypes = [C.POINTER(SmallGemmArgs), vp]
    lib.dtf_wgrad_grouped.argtypes = [C.POINTER(WgradArgs), vp]
    lib.dtf_head.argtypes = [C.POINTER(HeadArgs), vp]
    lib.dtf_step_tail.argtypes = [C.POINTER(TailArgs), i32, vp]
    lib.dtf_sgd_apply.argtypes = [vp, vp, i64, f32, f32, vp]
    lib.dtf_adam_apply.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, i32, vp]
    lib.dtf_truncated_normal.argtypes = [vp, i64, f32, u64, u64, vp]
    lib.dtf_accumulate.argtypes = [vp, vp, i64, f32, f32, vp]
    lib.dtf_gemm_tiled.argtypes = [C.POINTER(GemmArgs), vp]
    lib.dtf_mlp2_fwd.argtypes = [C.POINTER(Mlp2Args), vp]
    lib.dtf_mlp2_bwd.argtypes = [C.POINTER(Mlp2Args), vp]
    for name in ("dtf_pipe_step", "dtf_pipe_prime", "dtf_pipe_flush", "dtf_pipe_launch_roles"):
        getattr(lib, name).argtypes = [C.POINTER(PipeArgs), vp]
        getattr(lib, name).restype = i32
    for name in ("dtf_pipem_step", "dtf_pipem_prime", "dtf_pipem_flush", "dtf_pipem_launch_roles"):
        getattr(lib, name).argtypes = [C.POINTER(PipeMultiArgs), vp]
        getattr(lib, name).restype = i32
    lib.dtf_pipem_region_bytes.argtypes = []
    lib.dtf_pipem_region_bytes.restype = C.c_longlong
    lib.dtf_run_schedule.argtypes = [C.POINTER(ScheduleEntry), i32, vp]
    lib.dtf_run_schedule_steps.argtypes = [C.POINTER(C.POINTER(ScheduleEntry)), C.POINTER(i32), i32, i32, i32, vp]
    lib.dtf_run_schedule_steps.restype = i32
    lib.dtf_softmax_xent.argtypes = [C.POINTER(XentArgs), vp]
    lib.dtf_softmax_xent.restype = i32
    lib.dtf_softmax_xent_blocks.argtypes = [i32]
    lib.dtf_softmax_xent_blocks.restype = i32
    lib.dtf_gather_cast_transpose.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.dtf_upload_words.argtypes = [vp, vp, i64, vp]
    lib.dtf_upload_words.restype = i32
    lib.dtf_head_wide.argtypes = [C.POINTER(HeadWideArgs), vp]
    lib.dtf_head_wide.restype = i32
    lib.dtf_out_wgrad_wide.argtypes = [C.POINTER(OutWgradWideArgs), vp]
    lib.dtf_out_wgrad_wide.restype = i32
    for name in ("dtf_small_gemm", "dtf_wgrad_grouped", "dtf_head", "dtf_step_tail", "dtf_sgd_apply", "dtf_adam_apply",
                 "dtf_truncated_normal", "dtf_accumulate", "dtf_kernels_abi_version", "dtf_gemm_tiled",
                 "dtf_gather_cast_transpose", "dtf_run_schedule", "dtf_mlp2_fwd", "dtf_mlp2_bwd"):
        getattr(lib, name).restype = i32


def kernels():
    """Return the HIP kernel library, loading it on first use.

    Raises :class:`NativeLibraryMissing` if the ``.so`` has not been built; the
    callers in :mod:`distributed_tensorflow_amd.ops` never substitute eager
    PyTorch for a missing kernel on a GPU tensor.
    """
    global _kernels
    if _kernels is None:
        with _lock:
            if _kernels is None:
                # DTF_KERNELS_LIB selects another build of the same ABI (the stamped diagnostic library)
                path = os.environ.get("DTF_KERNELS_LIB") or lib_path("libdtf_kernels.so")
                if not os.path.exists(path):
                    raise NativeLibraryMissing(
                        f"{path} not found: build it with `make -C distributed_tensorflow_amd/csrc` "
                        "before using the GPU path")
                lib = C.CDLL(path)
                _declare_kernels(lib)
                _kernels = lib
    return _kernels


def kernels_available() -> bool:
    return os.path.exists(lib_path("libdtf_kernels.so"))


def runtime():
    """Return the host runtime library (pure C++, no GPU needed)."""
    global _runtime
    if _runtime is None:
        with _lock:
            if _runtime is None:
                path = lib_path("libdtf_runtime.so")
                if not os.path.exists(path):
                    raise NativeLibraryMissing(f"{path} not found: build it with `make -C distributed_tensorflow_amd/csrc`")
                _runtime = C.CDLL(path)
    return _runtime


def runtime_available() -> bool:
    return os.path.exists(lib_path("libdtf_runtime.so"))


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what}: native launcher refused the call ({ERR_NAMES.get(code, code)})")


def stream_handle() -> int:
    """hipStream_t of PyTorch's current stream (also valid while it is being captured)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()

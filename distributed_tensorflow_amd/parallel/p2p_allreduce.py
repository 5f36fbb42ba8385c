"""One-shot all-reduce + SGD over peer-mapped gradient buffers (``csrc/kernels/p2p_allreduce.hip``).

For the reference model one sync step exchanges 318 KB per worker (/root/reference/distributed.py:97-102):
latency, not bandwidth.  On the xGMI full mesh of one node every GPU reads each peer's gradient slot
directly, all links at once, and applies the averaged update in the same kernel - one hop instead of the
2(N-1) dependent hops of a ring.

Used for the small-model sync step when every rank sits on one node (SGD, all-reduce collective, RCCL
process group); ``DTF_P2P_ALLREDUCE=0`` keeps RCCL, ``=1`` also allows it on gloo rehearsal groups.  Before
the engine trusts it, :meth:`P2PAllReduce.self_check` runs the same kernel on throw-away buffers and
compares with the process group's all-reduce; any mismatch, mapping failure or timed-out wait makes the
engine keep RCCL.  Waits are bounded by a wall-clock limit (``DTF_P2P_TIMEOUT_S``, default 30 s; the kernels clip it
to 40 s); once a wait has hit it, every later wait of the run gives up at once (the error word is checked first), so
a dead link costs one limit, not one per step of a replayed graph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops import _lib

OP_P2P_UPDATE = 9
OP_P2P_SYNC_GENERAL = 11
SLOTS = 3
HANDLE_BYTES = 64


class P2PUpdateArgs(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("params", C.c_void_p), ("sum_out", C.c_void_p), ("seq_ptr", C.c_void_p),
                ("timeout_s", C.c_double), ("scale", C.c_float), ("slot", C.c_int)]


class P2PGeneralArgs(C.Structure):
    """Mirror of ``DtfP2PGeneralArgs``: one gradient contribution of the general SyncReplicas path."""
    _fields_ = [("comm", C.c_void_p), ("params", C.c_void_p), ("acc", C.c_void_p), ("state", C.c_void_p), ("wlog", C.c_void_p),
                ("seq_ptr", C.c_void_p), ("timeout_s", C.c_double), ("delay_s", C.c_double), ("scale", C.c_float),
                ("slot", C.c_int), ("need", C.c_int), ("first_round", C.c_int), ("last_round", C.c_int),
                ("stamp_offset", C.c_int), ("log_cap", C.c_int), ("pad_", C.c_int)]


_declared = False


def _lib_k():
    global _declared
    lib = _lib.kernels()
    if not _declared:
        lib.dtf_p2p_create.restype = C.c_void_p
        lib.dtf_p2p_create.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_char_p]
        lib.dtf_p2p_connect.restype = C.c_int
        lib.dtf_p2p_connect.argtypes = [C.c_void_p, C.c_char_p]
        lib.dtf_p2p_slot_ptr.restype = C.c_void_p
        lib.dtf_p2p_slot_ptr.argtypes = [C.c_void_p, C.c_int]
        lib.dtf_p2p_allreduce_sgd.restype = C.c_int
        lib.dtf_p2p_allreduce_sgd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_double,
                                              C.c_void_p]
        lib.dtf_p2p_mlp2_peer.restype = C.c_int
        lib.dtf_p2p_mlp2_peer.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(_lib.Mlp2Peer)]
        lib.dtf_p2p_mlp2_shard.restype = C.c_int
        lib.dtf_p2p_mlp2_shard.argtypes = [C.c_void_p, C.c_double, C.POINTER(_lib.Mlp2Shard)]
        lib.dtf_p2p_error.restype = C.c_int
        lib.dtf_p2p_error.argtypes = [C.c_void_p]
        lib.dtf_p2p_destroy.restype = None
        lib.dtf_p2p_destroy.argtypes = [C.c_void_p]
        _declared = True
    return lib


def requested(backend: str) -> bool:
    mode = os.environ.get("DTF_P2P_ALLREDUCE", "auto")
    if mode == "1":
        return True
    return mode == "auto" and backend == "nccl"


def default_timeout_s() -> float:
    return float(os.environ.get("DTF_P2P_TIMEOUT_S", "30"))


class _DevicePtr:
    """Minimal ``__cuda_array_interface__`` carrier so that torch can view memory it did not allocate."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class P2PAllReduce:
    """Peer-mapped gradient slots of ``n`` floats (multiple of 4) on every rank of ``group`` (one node)."""

    def __init__(self, n: int, device: torch.device, group=None, timeout_s: Optional[float] = None):
        if n % 4:
            raise ValueError("p2p all-reduce: element count must be a multiple of 4")
        self.n, self.device, self.group = n, device, group
        self.timeout_s = default_timeout_s() if timeout_s is None else timeout_s
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise ValueError("p2p all-reduce: at most 8 ranks (one node)")
        lib = _lib_k()
        handle = C.create_string_buffer(HANDLE_BYTES)
        with torch.cuda.device(device):
            self._comm = lib.dtf_p2p_create(self.rank, self.world, n, handle)
        # A failure on one rank must not leave the others waiting in a collective: every rank goes through the
        # same exchanges and all of them learn the outcome together.
        handles: List[Optional[bytes]] = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw) if self._comm else b"", group=group)
        good = bool(self._comm) and all(len(h) == HANDLE_BYTES for h in handles)
        if good:
            with torch.cuda.device(device):
                good = lib.dtf_p2p_connect(self._comm, b"".join(handles)) == 0
        ok = torch.tensor([1 if good else 0], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            self.close()
            raise RuntimeError("p2p all-reduce: a rank could not allocate, export or map the gradient regions")
        self.slots = [torch.as_tensor(_DevicePtr(lib.dtf_p2p_slot_ptr(self._comm, s), n), device=device) for s in range(SLOTS)]

    def update_args(self, slot: int, params: torch.Tensor, seq: torch.Tensor, scale: float,
                    sum_out: Optional[torch.Tensor] = None) -> P2PUpdateArgs:
        """Schedule entry arguments: ``params -= scale * sum_ranks(slot)``; ``seq`` is the int32 device counter that
        holds the step number when the entry runs (equal on every rank, growing by one per step)."""
        if params.numel() != self.n or params.dtype != torch.float32 or not params.is_contiguous():
            raise ValueError("p2p all-reduce: params must be the flat float32 buffer the slots were sized for")
        return P2PUpdateArgs(comm=self._comm, params=params.data_ptr(), sum_out=_lib.ptr(sum_out), seq_ptr=seq.data_ptr(),
                             timeout_s=self.timeout_s, scale=float(scale), slot=slot)

    def mlp2_peer(self, slot: int) -> "_lib.Mlp2Peer":
        """Peers' slot pointers and per-workgroup arrival flags for the two-launch step's own exchange
        (``csrc/kernels/mlp2_fused.hip``, mode 2): no separate update launch."""
        out = _lib.Mlp2Peer()
        _lib.check(_lib_k().dtf_p2p_mlp2_peer(self._comm, slot, self.timeout_s, C.byref(out)), "p2p_mlp2_peer")
        return out

    def mlp2_shard(self, params: torch.Tensor, adam: Optional[dict] = None) -> "_lib.Mlp2Shard":
        """Regions and box offsets for the two-launch step's sharded exchange (``mlp2_fused.hip``, mode 4): workgroup w
        belongs to rank ``w % world``, which sums the contributions, updates and redistributes the tiles.  ``adam``:
        ``{m, v (flat float32 buffers of this rank, laid out like params), lr, grad_scale, beta1, beta2, eps,
        t_offset}`` makes the owner apply Adam with its own moments instead of SGD."""
        if params.numel() != self.n or params.dtype != torch.float32 or not params.is_contiguous():
            raise ValueError("p2p sharded exchange: params must be the flat float32 buffer the boxes were sized for")
        out = _lib.Mlp2Shard()
        _lib.check(_lib_k().dtf_p2p_mlp2_shard(self._comm, self.timeout_s, C.byref(out)), "p2p_mlp2_shard")
        out.params_base = params.data_ptr()
        if adam is not None:
            for name in ("m", "v"):
                t = adam[name]
                if t.numel() != self.n or t.dtype != torch.float32 or not t.is_contiguous() or t.device != params.device:
                    raise ValueError(f"p2p sharded exchange: Adam {name} must be a flat float32 buffer like params")
            out.adam, out.m, out.v = 1, adam["m"].data_ptr(), adam["v"].data_ptr()
            out.adam_lr, out.grad_scale = float(adam["lr"]), float(adam["grad_scale"])
            out.b1, out.b2, out.eps = float(adam["beta1"]), float(adam["beta2"]), float(adam["eps"])
            out.adam_t_offset = int(adam["t_offset"])
        return out

    def general_args(self, slot: int, params: torch.Tensor, acc: torch.Tensor, state: torch.Tensor, seq: torch.Tensor,
                     scale: float, need: int, first_round: bool, last_round: bool, wlog: Optional[torch.Tensor] = None,
                     delay_s: float = 0.0, stamp_offset: int = 0) -> P2PGeneralArgs:
        """Schedule entry arguments of one contribution round of the general SyncReplicas path (``need`` of the ranks'
        gradients are taken by the chief; the update is applied in the last round of an update)."""
        for name, t in (("params", params), ("acc", acc)):
            if t.numel() != self.n or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"p2p general sync: {name} must be a flat float32 buffer of the slots' size")
        if state.dtype != torch.int32 or state.numel() < 4:
            raise ValueError("p2p general sync: state must hold at least 4 int32 counters")
        if not 1 <= need <= self.world:
            raise ValueError("p2p general sync: need must be within 1..world")
        return P2PGeneralArgs(comm=self._comm, params=params.data_ptr(), acc=acc.data_ptr(), state=state.data_ptr(),
                              wlog=_lib.ptr(wlog), seq_ptr=seq.data_ptr(), timeout_s=self.timeout_s, delay_s=float(delay_s),
                              scale=float(scale), slot=slot, need=need, first_round=int(first_round),
                              last_round=int(last_round), stamp_offset=int(stamp_offset),
                              log_cap=0 if wlog is None else wlog.shape[0], pad_=0)

    def allreduce_sgd(self, slot: int, params: torch.Tensor, seq: torch.Tensor, scale: float,
                      sum_out: Optional[torch.Tensor] = None) -> None:
        rc = _lib_k().dtf_p2p_allreduce_sgd(self._comm, slot, seq.data_ptr(), params.data_ptr(), _lib.ptr(sum_out), float(scale),
                                            self.timeout_s, _lib.stream_handle())
        _lib.check(rc, "p2p_allreduce_sgd")

    def timed_out(self) -> bool:
        return _lib_k().dtf_p2p_error(self._comm) != 0

    def close(self) -> None:
        if getattr(self, "_comm", None):
            self.slots = []
            _lib_k().dtf_p2p_destroy(self._comm)
            self._comm = None

    @staticmethod
    def self_check(n: int, device: torch.device, group=None, rounds: int = 12) -> bool:
        """The kernel on throw-away buffers against the process group's all-reduce, several rounds over all slots.
        Collective: every rank calls it; every rank gets the same answer."""
        good = 1
        comm = None
        rank = dist.get_rank(group)
        nccl = dist.get_backend(group) == "nccl"
        gen = torch.Generator(device="cpu")

        def contribution(r: int) -> torch.Tensor:
            gen.manual_seed(1000 * r + rank)
            return torch.rand(n, generator=gen).to(device)

        params = torch.zeros(n, dtype=torch.float32, device=device)
        try:
            comm = P2PAllReduce(n, device, group, timeout_s=10.0)     # raises on every rank or on none
        except (RuntimeError, OSError, ValueError):
            good = 0
        if comm is not None:
            try:
                # all rounds back to back, with nothing but the kernel's own flags keeping the ranks in step
                # (slots are reused every third round)
                seq = torch.zeros(1, dtype=torch.int32, device=device)
                for r in range(rounds):
                    slot = r % SLOTS
                    comm.slots[slot].copy_(contribution(r))
                    seq.fill_(r + 1)
                    comm.allreduce_sgd(slot, params, seq, 0.5)
                torch.cuda.synchronize(device)
                if comm.timed_out():
                    good = 0
            except (RuntimeError, OSError, ValueError):
                good = 0
        # the reference result through the process group; every rank runs these collectives whatever happened above
        want = torch.zeros(n, dtype=torch.float32, device=device)
        for r in range(rounds):
            ref = contribution(r)
            if nccl:
                dist.all_reduce(ref, group=group)
            else:
                h = ref.cpu()
                dist.all_reduce(h, group=group)
                ref = h.to(device)
            want -= 0.5 * ref
        if good and not torch.allclose(params, want, rtol=1e-5, atol=1e-5):
            good = 0
        torch.cuda.synchronize(device)
        ok = torch.tensor([good], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if comm is not None:
            comm.close()
        return int(ok.item()) == 1

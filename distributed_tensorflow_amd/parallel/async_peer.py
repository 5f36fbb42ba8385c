"""Asynchronous (Hogwild) training over peer-mapped parameter slices: the device-side async engine.

Reference behaviour (/root/reference/distributed.py:89, :102 with ``--sync_replicas`` unset): every worker
applies its own gradient to the ps-resident variables without locks and reads them back; ``global_step``
counts the updates of all workers.

Here the ps-resident variables are ``num_shards`` slices of the flat parameter buffer, each living in an
uncached region of its owner's HBM that every rank of the node maps through an IPC handle.  A worker step is
the usual compute launches followed by ONE kernel (``dtf_async_push_pull``) that applies ``lr * gradient`` to
every slice in place through the mapped pointers and keeps the fresh values as its local replica; the
global-step counter sits next to shard 0 and is bumped with a system-scope atomic.  There is no service
thread and no receive loop: owners are not involved in their peers' updates at all, which is exactly the
lock-free semantics of the reference's ``use_locking=False``.  A pull is never older than the worker's own
last step (staleness bound 0).

``AsyncParameterServerEngine`` (send/recv, any backend, Adam) remains the general engine; this one is picked
for SGD on one node with an RCCL process group after a start-up check on the actual machine
(``DTF_PEER_ASYNC=0`` disables, ``=1`` also allows gloo rehearsal groups).
"""
from __future__ import annotations

import ctypes as C
import os
import socket
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops import _lib
from .sharded_state import ShardedState

OP_ASYNC_PUSH_PULL = 10
HANDLE_BYTES = 64
COUNTER_BYTES = 64


class AsyncPushPullArgs(C.Structure):
    _fields_ = [("master", C.c_void_p * 8), ("counters", C.c_void_p * 8), ("grads", C.c_void_p), ("params", C.c_void_p),
                ("state", C.c_void_p), ("slice", C.c_int64), ("shards", C.c_int), ("mode", C.c_int), ("lr", C.c_float),
                ("pad_", C.c_int)]


_declared = False


def _lib_k():
    global _declared
    lib = _lib.kernels()
    if not _declared:
        lib.dtf_ipc_alloc.restype = C.c_void_p
        lib.dtf_ipc_alloc.argtypes = [C.c_int64, C.c_char_p]
        lib.dtf_ipc_open.restype = C.c_void_p
        lib.dtf_ipc_open.argtypes = [C.c_char_p]
        lib.dtf_ipc_close.restype = C.c_int
        lib.dtf_ipc_close.argtypes = [C.c_void_p]
        lib.dtf_ipc_free.restype = C.c_int
        lib.dtf_ipc_free.argtypes = [C.c_void_p]
        lib.dtf_async_push_pull.restype = C.c_int
        lib.dtf_async_push_pull.argtypes = [C.POINTER(AsyncPushPullArgs), C.c_void_p]
        _declared = True
    return lib


def wanted(backend: str, world: int) -> bool:
    mode = os.environ.get("DTF_PEER_ASYNC", "auto")
    if mode == "1":
        return True
    return mode == "auto" and backend == "nccl" and world > 1


class _DevicePtr:
    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class PeerMappedUnavailable(RuntimeError):
    """Raised on EVERY rank together when the peer-mapped path cannot be used on this machine."""


class PeerMappedAsyncEngine:
    def __init__(self, program, state: ShardedState, num_shards: int = 1, group=None):
        if program.fused or program.gs_inc != 0:
            raise ValueError("async mode needs a program that stores gradients and leaves global_step to the ps")
        if state.optimizer != "sgd" or state.device.type != "cuda":
            raise ValueError("the peer-mapped async engine implements plain SGD on GPUs")
        self.program, self.state, self.group = program, state, group
        self.world, self.rank = state.world_size, state.rank
        self.device = state.device
        self.S = max(1, min(num_shards, self.world, 8))
        n = state.layout.total
        if n % (4 * self.S):
            raise ValueError(f"{self.S} shards do not divide the flat buffer of {n} elements into 16-byte multiples")
        self.slice = n // self.S
        self.is_owner = self.rank < self.S
        self.staleness = 0
        self.updates_pushed = 0
        self.last_global_step = 1
        self._applied_final: Optional[int] = None
        self._local = None
        self._mapped: List[int] = []
        lib = _lib_k()
        cpu_flags = dist.get_backend(group) != "nccl"

        def all_ok(flag: bool) -> bool:
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if cpu_flags else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        # ---- every rank walks through the same collectives, whatever fails locally ----
        hosts = [None] * self.world
        dist.all_gather_object(hosts, socket.gethostname(), group=group)
        handle = C.create_string_buffer(HANDLE_BYTES)
        good = len(set(hosts)) == 1 and self.world <= 8
        if good and self.is_owner:
            with torch.cuda.device(self.device):
                self._local = lib.dtf_ipc_alloc(self.slice * 4 + COUNTER_BYTES, handle)
            good = bool(self._local)
        handles: List[Optional[bytes]] = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw) if (self.is_owner and self._local) else b"", group=group)
        base = [0] * self.S
        if good:
            for s in range(self.S):
                if s == self.rank:
                    base[s] = self._local
                elif len(handles[s]) == HANDLE_BYTES:
                    with torch.cuda.device(self.device):
                        p = lib.dtf_ipc_open(handles[s])
                    if p:
                        self._mapped.append(p)
                        base[s] = p
                    else:
                        good = False
                else:
                    good = False
        if not all_ok(good):
            self._release()
            raise PeerMappedUnavailable("a rank could not allocate, export or map the parameter slices")
        self._base = base
        self._push = self._make_args(0, state.lr)
        self._pull = self._make_args(1, 0.0)
        # ---- start-up check on the actual machine: remote read-modify-write, counters, visibility ----
        if not all_ok(self._startup_check()):
            self._release()
            raise PeerMappedUnavailable("the peer-mapped push/pull failed its start-up check on this machine")
        self.publish_state()
        entry = (OP_ASYNC_PUSH_PULL, self._push, 0)
        # Opt-in (DTF_PEER_ASYNC_FUSED=1): the backward launch of the two-launch step pushes and pulls tile by tile,
        # two launches per step.  Measured on one rank: the launch then ends 2.9 us later (it has to see its stores
        # to the uncached master acknowledged), the separate push/pull kernel costs 3.9 us under the profiler and
        # 3.0 us in the step: 14.8 us/step either way, so the simpler sequence stays the default.
        self.fused = bool(getattr(program, "fused2", False)) and os.environ.get("DTF_PEER_ASYNC_FUSED", "0") == "1"
        if self.fused:
            a = _lib.Mlp2Async()
            for s in range(self.S):
                a.master[s] = self._base[s]
                a.counters[s] = self._base[s] + self.slice * 4
            a.params_base, a.slice, a.shards, a.pad_ = state.params.data_ptr(), self.slice, self.S, 0
            self._mlp2_async = a
            program.use_async_push(a, state.lr)
            program.after_compute = None
        elif hasattr(program, "schedule_tail_entries"):
            program.schedule_tail_entries.append(entry)
            program._schedules.clear()
            program.after_compute = None
        else:
            self._sched = _lib.Schedule([entry])
            program.after_compute = self._sched.run

    # ------------------------------------------------------------------ helpers
    def _make_args(self, mode: int, lr: float) -> AsyncPushPullArgs:
        st = self.state
        a = AsyncPushPullArgs()
        for s in range(self.S):
            a.master[s] = self._base[s]
            a.counters[s] = self._base[s] + self.slice * 4
        a.grads, a.params, a.state = st.grads.data_ptr(), st.params.data_ptr(), st.state.data_ptr()
        a.slice, a.shards, a.mode, a.lr, a.pad_ = self.slice, self.S, mode, float(lr), 0
        return a

    def _master_view(self) -> torch.Tensor:
        return torch.as_tensor(_DevicePtr(self._local, self.slice, "<f4"), device=self.device)

    def _counter_view(self) -> torch.Tensor:
        return torch.as_tensor(_DevicePtr(self._local + self.slice * 4, COUNTER_BYTES // 4, "<i4"), device=self.device)

    def _barrier(self) -> None:
        if dist.get_backend(self.group) == "nccl":
            dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            dist.barrier(group=self.group)

    def _launch(self, args: AsyncPushPullArgs) -> None:
        _lib.check(_lib_k().dtf_async_push_pull(C.byref(args), _lib.stream_handle()), "async_push_pull")

    def _startup_check(self) -> bool:
        """One rank at a time pushes a known gradient (so that nothing races), then everybody pulls: every rank
        must see the sum of all pushes in every slice and a step counter equal to the number of pushes."""
        st = self.state
        keep_p, keep_g, keep_s = st.params.clone(), st.grads.clone(), st.state.clone()
        ok = True
        try:
            chk = self._make_args(0, 1.0)
            for r in range(self.world):
                if r == self.rank:
                    st.grads.fill_(float(r + 1))
                    self._launch(chk)
                    torch.cuda.synchronize(self.device)
                self._barrier()
            self._launch(self._pull)
            torch.cuda.synchronize(self.device)
            want = -float(self.world * (self.world + 1) // 2)
            ok = bool((st.params == want).all().item()) and int(st.state[0].item()) == self.world
        except RuntimeError:
            ok = False
        st.params.copy_(keep_p)
        st.grads.copy_(keep_g)
        st.state.copy_(keep_s)
        return ok

    def publish_state(self) -> None:
        """The owners (re)publish their slices and the step counter from their local state.  Collective; called at
        construction and again by the training script once the chief has initialised or restored the variables."""
        st = self.state
        torch.cuda.synchronize(self.device)
        self._barrier()                                   # nobody is pushing any more
        if self.is_owner:
            lo = self.rank * self.slice
            self._master_view().copy_(st.params[lo:lo + self.slice])
            self._counter_view()[0:2] = torch.tensor([st.global_step() if self.rank == 0 else 0, 0], dtype=torch.int32,
                                                      device=self.device)
        torch.cuda.synchronize(self.device)
        self._barrier()

    # ------------------------------------------------------------------ engine interface
    def step(self) -> None:
        prog = self.program
        saved, prog.use_graphs = prog.use_graphs, False
        prog.run(1)
        prog.use_graphs = saved
        self.updates_pushed += 1

    def run(self, n_steps: int, graph_steps: int = 64) -> None:
        """``n_steps`` worker steps; the push/pull kernel is part of every step, so whole launches of steps can be
        replayed from a hipGraph."""
        self.program.run(n_steps, graph_steps if graph_steps else 64)
        self.updates_pushed += n_steps

    def sync_params(self) -> int:
        self._launch(self._pull)
        self.last_global_step = self.state.global_step()
        return self.last_global_step

    def finish(self) -> None:
        torch.cuda.synchronize(self.device)
        self._barrier()                                   # nobody unmaps while a peer may still push
        self._launch(self._pull)
        torch.cuda.synchronize(self.device)
        if self.is_owner:
            self._applied_final = int(self._counter_view()[1].item())
        self._barrier()
        self._release()

    def _release(self) -> None:
        lib = _lib_k()
        for p in self._mapped:
            lib.dtf_ipc_close(p)
        self._mapped = []
        if self._local:
            lib.dtf_ipc_free(self._local)
            self._local = None

    @property
    def updates_applied_here(self) -> int:
        if self._applied_final is not None:
            return self._applied_final
        return int(self._counter_view()[1].item()) if self.is_owner and self._local else 0


def make_async_engine(program, state: ShardedState, num_shards: int, staleness: int, group=None):
    """The device-side engine where it applies (one node, SGD, GPUs, RCCL group), else the send/recv engine."""
    from .async_ps import AsyncParameterServerEngine
    import warnings
    if (state.device.type == "cuda" and state.optimizer == "sgd" and state.world_size > 1 and state.world_size <= 8
            and state.layout.total % (4 * max(1, min(num_shards, state.world_size))) == 0
            and wanted(dist.get_backend(group), state.world_size) and _lib.kernels_available()):
        try:
            return PeerMappedAsyncEngine(program, state, num_shards, group)
        except PeerMappedUnavailable as e:                 # raised on every rank together
            warnings.warn(f"peer-mapped async engine unavailable ({e}); using send/recv")
    return AsyncParameterServerEngine(program, state, num_shards=num_shards, staleness=staleness)

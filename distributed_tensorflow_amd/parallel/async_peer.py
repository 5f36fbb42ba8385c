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
lock-free semantics of the reference's ``use_locking=False``.

Staleness: with ``staleness = k`` a worker pushes on every step and pulls on every (k+1)-th; in between it applies
its own update to its replica, which therefore misses only the other workers' updates since its last pull.  The
kernel logs, per step, how many such updates the master had received when the gradient was pushed
(``staleness_stats``): the bound is measured, not assumed.  Adam keeps its moments next to the master slice in
the owner's region (slot variables of a ps-resident optimizer), updated with the same lock-free read-modify-write.

Steady state is free of collectives: ``run``, ``step`` and ``sync_params`` only launch kernels, so a worker that
dies or leaves does not stall the others (workers talk to the ps shards, never to each other:
``device_filters`` at distributed.py:116-117); ``finish`` waits for the others through the rendezvous store, with
a time limit.

``AsyncParameterServerEngine`` (send/recv, any backend, several nodes) remains the general engine; this one is
picked on one node with an RCCL process group after a start-up check on the actual machine
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
    """Mirror of ``DtfAsyncPushPullArgs``."""
    _fields_ = [("master", C.c_void_p * 8), ("counters", C.c_void_p * 8), ("grads", C.c_void_p), ("params", C.c_void_p),
                ("state", C.c_void_p), ("slice", C.c_int64), ("shards", C.c_int), ("mode", C.c_int), ("lr", C.c_float),
                ("optimizer", C.c_int), ("m", C.c_void_p * 8), ("v", C.c_void_p * 8), ("astate", C.c_void_p),
                ("alog", C.c_void_p), ("alog_cap", C.c_int), ("pull_every", C.c_int), ("b1", C.c_float), ("b2", C.c_float),
                ("eps", C.c_float), ("bound", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("timeout_ticks", C.c_uint)]


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
    _instances = 0          # engines built so far in this process (the same on every rank): names the store keys

    def __init__(self, program, state: ShardedState, num_shards: int = 1, group=None, staleness: int = 0,
                 log_capacity: int = 4096, staleness_bound: Optional[int] = None):
        if program.fused or program.gs_inc != 0:
            raise ValueError("async mode needs a program that stores gradients and leaves global_step to the ps")
        if state.device.type != "cuda":
            raise ValueError("the peer-mapped async engine runs on GPUs")
        self.program, self.state, self.group = program, state, group
        self.world, self.rank = state.world_size, state.rank
        self.device = state.device
        self.S = max(1, min(num_shards, self.world, 8))
        n = state.layout.total
        if n % (4 * self.S):
            raise ValueError(f"{self.S} shards do not divide the flat buffer of {n} elements into 16-byte multiples")
        self.slice = n // self.S
        self.is_owner = self.rank < self.S
        self.adam = state.optimizer == "adam"
        # A worker pulls every (staleness + 1)-th local step and pushes on every step; in between it applies its own
        # update to its replica.  0 = pull with every push.
        self.staleness = max(0, int(staleness))
        # Bounded staleness (None: the default below; negative: none, plain Hogwild as the reference runs it).  A replica
        # that misses more than the bound of OTHER workers' updates is refreshed with the next push, and a worker more
        # than the bound in steps ahead of the slowest waits (stale-synchronous clocks next to global_step): a gradient
        # is never applied more than bound + 2 x bound x (workers - 1) updates late.
        if staleness_bound is None:
            env = os.environ.get("DTF_STALENESS_BOUND")
            staleness_bound = int(env) if env is not None else self.staleness + 2 * (self.world - 1)
        self.staleness_bound = int(staleness_bound) if int(staleness_bound) >= 0 else None
        # per local step {global_step after the push, updates of other workers the replica had not seen}: device ring
        self.alog = torch.zeros((log_capacity, 2), dtype=torch.int32, device=self.device)
        self._astate = torch.zeros(16, dtype=torch.int32, device=self.device)
        PeerMappedAsyncEngine._instances += 1
        self._key = f"dtf/async_peer/{PeerMappedAsyncEngine._instances}"
        self.updates_pushed = 0
        self.last_global_step = 1
        self._applied_final: Optional[int] = None
        self._finished = False
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
                # [master slice | Adam m | Adam v | counters]: the moments are ps-resident like the variables
                self._local = lib.dtf_ipc_alloc(self.slice * 4 * (3 if self.adam else 1) + COUNTER_BYTES, handle)
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
        # The backward launch of the two-launch step pushes and pulls tile by tile (mode 3 of mlp2_fused.hip): two
        # launches per step and no third kernel.  Every variant of the separate kernel is covered: a pull period,
        # Adam with the owner-resident moments, the staleness log and bound.  DTF_PEER_ASYNC_FUSED=0 keeps the
        # separate push/pull kernel behind the step's two launches.
        self.fused = bool(getattr(program, "fused2", False)) and os.environ.get("DTF_PEER_ASYNC_FUSED", "1") != "0"
        if self.fused:
            a = _lib.Mlp2Async()
            p = self._push
            for s in range(self.S):
                a.master[s] = self._base[s]
                a.counters[s] = self._base[s] + self._counter_off
                if self.adam:
                    a.m[s], a.v[s] = p.m[s], p.v[s]
            a.params_base, a.slice, a.shards, a.pad_ = state.params.data_ptr(), self.slice, self.S, 0
            a.optimizer, a.pull_every, a.bound, a.rank, a.world = p.optimizer, p.pull_every, p.bound, p.rank, p.world
            a.timeout_ticks, a.b1, a.b2, a.eps = p.timeout_ticks, p.b1, p.b2, p.eps
            a.astate, a.alog, a.alog_cap = p.astate, p.alog, p.alog_cap
            self._applied_snap = torch.zeros(8, dtype=torch.int32, device=self.device)
            a.applied_snap = self._applied_snap.data_ptr()
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
    @property
    def _counter_off(self) -> int:
        return self.slice * 4 * (3 if self.adam else 1)

    def _make_args(self, mode: int, lr: float, check: bool = False) -> AsyncPushPullArgs:
        """``check``: the start-up check's plain SGD push without staleness bookkeeping."""
        st = self.state
        a = AsyncPushPullArgs()
        for s in range(self.S):
            a.master[s] = self._base[s]
            a.counters[s] = self._base[s] + self._counter_off
            if self.adam:
                a.m[s] = self._base[s] + self.slice * 4
                a.v[s] = self._base[s] + self.slice * 8
        a.grads, a.params, a.state = st.grads.data_ptr(), st.params.data_ptr(), st.state.data_ptr()
        a.slice, a.shards, a.mode, a.lr = self.slice, self.S, mode, float(lr)
        a.bound = -1 if (check or self.staleness_bound is None) else self.staleness_bound
        a.rank, a.world = self.rank, self.world
        # how long a worker that is too far ahead waits for the slowest before it gives the bound up (a worker that
        # died never advances its clock; one that finished parks it at the maximum)
        a.timeout_ticks = int(min(4.0e9, float(os.environ.get("DTF_STALENESS_WAIT_S", "5")) * 1e8))
        a.optimizer = 1 if (self.adam and not check) else 0
        a.b1, a.b2, a.eps = st.beta1, st.beta2, st.eps
        if check:
            a.pull_every, a.astate, a.alog, a.alog_cap = 1, None, None, 0
        else:
            a.pull_every = self.staleness + 1
            a.astate, a.alog, a.alog_cap = self._astate.data_ptr(), self.alog.data_ptr(), self.alog.shape[0]
        return a

    def _master_view(self) -> torch.Tensor:
        return torch.as_tensor(_DevicePtr(self._local, self.slice, "<f4"), device=self.device)

    def moment_views(self):
        """The owner-resident Adam moments (m, v) of this rank's slice, or None."""
        if not (self.adam and self.is_owner and self._local):
            return None
        return (torch.as_tensor(_DevicePtr(self._local + self.slice * 4, self.slice, "<f4"), device=self.device),
                torch.as_tensor(_DevicePtr(self._local + self.slice * 8, self.slice, "<f4"), device=self.device))

    def gather_moments(self):
        """Full flat Adam moments read straight out of the owners' mapped regions (one-sided: no other rank takes
        part, so the chief can checkpoint while everybody keeps training), or None for SGD."""
        if not self.adam:
            return None
        if self._finished:
            # finish() has unmapped the peers' regions (reading them now would fault the GPU); what this rank owned
            # came home into state.m / state.v there, the other owners' slices are out of reach
            raise RuntimeError("gather_moments after finish(): take the final checkpoint before the engine is torn down")
        n = self.state.layout.total
        full_m = torch.empty(n, dtype=torch.float32, device=self.device)
        full_v = torch.empty(n, dtype=torch.float32, device=self.device)
        for s in range(self.S):
            lo = s * self.slice
            full_m[lo:lo + self.slice].copy_(torch.as_tensor(_DevicePtr(self._base[s] + self.slice * 4, self.slice, "<f4"),
                                                             device=self.device))
            full_v[lo:lo + self.slice].copy_(torch.as_tensor(_DevicePtr(self._base[s] + self.slice * 8, self.slice, "<f4"),
                                                             device=self.device))
        # the update count the bias correction uses is the owners' (shard 0 speaks for the job)
        cnt = torch.as_tensor(_DevicePtr(self._base[0] + self._counter_off, COUNTER_BYTES // 4, "<i4"), device=self.device)
        self.state.state[2] = int(cnt[1].item())
        return full_m, full_v

    def _counter_view(self) -> torch.Tensor:
        return torch.as_tensor(_DevicePtr(self._local + self._counter_off, COUNTER_BYTES // 4, "<i4"), device=self.device)

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
            chk = self._make_args(0, 1.0, check=True)
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
            applied = int(st.state[2].item()) if self.adam else 0      # Adam's update count (restored from a checkpoint)
            self._counter_view()[0:2] = torch.tensor([st.global_step() if self.rank == 0 else 0, applied], dtype=torch.int32,
                                                      device=self.device)
            self._counter_view()[4:12] = st.local_step()      # every worker's step clock starts where the counters stand
            mv = self.moment_views()
            if mv is not None:
                lo = self.rank * self.slice
                own = (lo, lo + self.slice) == tuple(st.shard) and st.m is not None
                mv[0].copy_(st.m) if own else mv[0].zero_()
                mv[1].copy_(st.v) if own else mv[1].zero_()
        self._astate.copy_(torch.tensor([st.global_step()] + [0] * 15, dtype=torch.int32))
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

    def _store(self):
        try:
            return dist.distributed_c10d._get_default_store()
        except (RuntimeError, AttributeError):
            return None

    def finish(self, timeout_s: float = 120.0) -> bool:
        """This worker has pushed its last gradient.  Waits (bounded) until every worker has said the same through the
        rendezvous store - not a collective: a worker that died never blocks the others' steps, and holds up their
        ``finish`` for ``timeout_s`` at most.  Returns whether everybody reported.  An owner whose peers did not all
        report keeps its slice allocated until the process ends (somebody may still be writing to it)."""
        import time
        torch.cuda.synchronize(self.device)
        # this worker's step clock parks at the maximum: nobody waits for a worker that has left
        if self.staleness_bound is not None and self._base and self._base[0]:
            clocks = torch.as_tensor(_DevicePtr(self._base[0] + self._counter_off, COUNTER_BYTES // 4, "<i4"), device=self.device)
            clocks[4 + self.rank] = 0x7fffffff
            torch.cuda.synchronize(self.device)
        store = self._store()
        everyone = store is not None
        if store is not None:
            store.set(f"{self._key}/done/{self.rank}", "1")
            deadline = time.time() + timeout_s
            for r in range(self.world):
                while not store.check([f"{self._key}/done/{r}"]):
                    if time.time() > deadline:
                        everyone = False
                        break
                    time.sleep(0.01)
        self._launch(self._pull)
        torch.cuda.synchronize(self.device)
        if self.is_owner:
            self._applied_final = int(self._counter_view()[1].item())
            mv = self.moment_views()
            lo = self.rank * self.slice
            if mv is not None and self.state.m is not None and (lo, lo + self.slice) == tuple(self.state.shard):
                self.state.m.copy_(mv[0])                # the slot variables come home for the final checkpoint
                self.state.v.copy_(mv[1])
                self.state.state[2] = self._applied_final
        self._finished = True
        self._release(free_local=everyone)
        return everyone

    def staleness_stats(self, first_local_step: int, count: int):
        """(max, mean) over local steps ``[first, first+count)`` of the number of other workers' updates that the replica
        a gradient was computed on had not seen when the gradient was pushed.  Synchronises with the device."""
        cap = self.alog.shape[0]
        count = min(count, cap)
        if count <= 0:
            return 0, 0.0
        log = self.alog.cpu()
        idx = [(first_local_step + i) % cap for i in range(count)]
        vals = log[idx, 1].to(torch.float64)
        return int(vals.max().item()), float(vals.mean().item())

    def forced_pulls(self) -> int:
        """Pushes of this worker that refreshed its replica because it had fallen behind the staleness bound."""
        return int(self._astate[2].item())

    def bound_waits(self) -> int:
        """Pushes after which this worker waited for a slower one (it was more than the bound in steps ahead)."""
        return int(self._astate[3].item())

    def bound_given_up(self) -> bool:
        """True once a wait for the slowest worker ran into the time limit (the run continues unbounded)."""
        return int(self._astate[5].item()) != 0

    def _release(self, free_local: bool = True) -> None:
        lib = _lib_k()
        for p in self._mapped:
            lib.dtf_ipc_close(p)
        self._mapped = []
        if self._local and free_local:
            lib.dtf_ipc_free(self._local)
            self._local = None

    @property
    def updates_applied_here(self) -> int:
        if self._applied_final is not None:
            return self._applied_final
        return int(self._counter_view()[1].item()) if self.is_owner and self._local else 0


def make_async_engine(program, state: ShardedState, num_shards: int, staleness: int, group=None, staleness_bound=None):
    """The device-side engine where it applies (one node, SGD, GPUs, RCCL group), else the send/recv engine."""
    from .async_ps import AsyncParameterServerEngine
    import warnings
    if (state.device.type == "cuda" and state.world_size > 1 and state.world_size <= 8
            and state.layout.total % (4 * max(1, min(num_shards, state.world_size))) == 0
            and wanted(dist.get_backend(group), state.world_size) and _lib.kernels_available()):
        try:
            return PeerMappedAsyncEngine(program, state, num_shards, group, staleness=staleness, staleness_bound=staleness_bound)
        except PeerMappedUnavailable as e:                 # raised on every rank together
            warnings.warn(f"peer-mapped async engine unavailable ({e}); using send/recv")
    return AsyncParameterServerEngine(program, state, num_shards=num_shards, staleness=staleness)

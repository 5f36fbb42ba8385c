"""Asynchronous (Hogwild-style) parameter-server training over point-to-point transfers.

Reference behaviour (/root/reference/distributed.py:89, :102 with ``--sync_replicas``
unset, the default): every worker applies its own gradient to the shared variables the
moment it is computed — no aggregation, no barrier, no locking — and ``global_step``
counts the updates of all workers together.

Here the shared variables are ``num_shards`` contiguous slices of the flat parameter
buffer, each resident in the HBM of its owner rank (ranks ``0 .. num_shards-1`` own one
slice each and are workers as well).  A worker step is

    compute gradients on the local replica
    for every shard:   push = send(grad slice) to the owner        (side stream)
                       pull = recv(updated slice) from the owner   (side stream)
    continue with the next step; a pull may stay in flight for up to ``staleness`` steps

and an owner runs one service thread per remote worker.  Every request is announced by a
two-word control message on a host-side (gloo) channel of the pair - the doorbell - and only
then does the owner post the receive of the payload on the pair's data communicator: no
receive kernel ever sits on the GPU waiting for a worker that has nothing to say (on RCCL a
parked receive occupies a hardware queue and can stall the compute stream that shares it),
and the request kind reaches the service thread without a device synchronisation.  The owner
applies each arriving gradient to its slice with the fused SGD (or Adam: moments live with
the owner, like the slot variables of a ps-resident optimizer) kernel, bumps ``global_step``
(owner of shard 0) and sends the fresh slice back.  ``backend='nccl'`` is RCCL over xGMI;
``gloo`` runs the same code on CPU.  Races between an owner's compute reading its slice and
the service thread updating it are intended (lock-free updates, as the reference's
use_locking=False).  Steps are launched one by one: every gradient has to pass through a
host-issued send.  On one node the device-side engine (``async_peer.py``) is used instead.

Control message: int64 [kind, worker_local_step] with kind 0 = gradient follows, 1 = worker
finished, 2 = pull only, 3 = send me your Adam moments.  Payloads are float32 buffers of
``shard_size + 2`` elements (the two extra slots of a reply carry int32 bit patterns:
[params..., global_step, 0]).
"""
from __future__ import annotations

import threading
import time
from collections import deque
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import ops
from .sharded_state import ShardedState

KIND_GRAD, KIND_DONE, KIND_PULL, KIND_MOMENTS = 0, 1, 2, 3


def build_pair_groups(world: int, num_shards: int) -> Dict[Tuple[int, int], object]:
    """Per (owner, worker) pair a data group (the job's backend) under key ``(o, w)`` and a host-side control
    group (gloo) under ``(o, w, "ctl")``.  Collective: every rank must call this with the same arguments, in the
    same order."""
    groups = {}
    for o in range(num_shards):
        for w in range(world):
            if w != o:
                groups[(o, w)] = dist.new_group([o, w])
                groups[(o, w, "ctl")] = dist.new_group([o, w], backend="gloo")
    return groups


class AsyncParameterServerEngine:
    def __init__(self, program, state: ShardedState, num_shards: int = 1, staleness: int = 4,
                 pair_groups: Optional[Dict] = None):
        self.program, self.state = program, state
        self.world, self.rank = state.world_size, state.rank
        if program.fused or program.gs_inc != 0:
            raise ValueError("async mode needs a program that stores gradients and leaves global_step to the owner")
        self.S = max(1, min(num_shards, self.world))
        self.staleness = max(0, staleness)
        self.device = state.device
        n = state.layout.total
        if n % self.S:
            # the flat buffer is padded to world*ALIGN, so any S that divides world divides it
            raise ValueError(f"{self.S} shards do not divide the flat buffer of {n} elements")
        self.slice = n // self.S
        self.groups = pair_groups if pair_groups is not None else (
            build_pair_groups(self.world, self.S) if self.world > 1 else {})
        f32 = torch.float32
        self.is_owner = self.rank < self.S
        # worker side: per remote shard a ring of staleness+1 push/pull staging buffers — at most
        # `staleness` exchanges are still in flight when a new one starts, so a free slot always exists
        self._ring = self.staleness + 1
        remote = [s for s in range(self.S) if s != self.rank]
        self._push_buf = {s: torch.zeros((self._ring, self.slice + 2), dtype=f32, device=self.device) for s in remote}
        self._pull_buf = {s: torch.zeros((self._ring, self.slice + 2), dtype=f32, device=self.device) for s in remote}
        self._exchanges = 0
        self._pending = deque()          # (slot, [(shard, send_work, recv_work)])
        self.updates_pushed = 0
        self.last_global_step = 1
        # owner side: one service thread per remote worker, each blocking on its own two-rank group
        # (and, on GPU, issuing on its own stream), so arrivals are served in any order
        self._threads: List[threading.Thread] = []
        self._service_error: Optional[BaseException] = None
        self._applied = 0
        # global_step lives with the owner of shard 0; host-side under a lock so that concurrent
        # service threads and the owner's own pushes never lose an increment
        self._gs_lock = threading.Lock()
        self._gs = 1
        # Adam: the owner keeps the moments of its slice and counts the updates it applied (bias correction)
        self._m = self._v = None
        self._adam_t = 0
        self._finished = False
        if state.optimizer == "adam" and self.is_owner:
            self._m = torch.zeros(self.slice, dtype=f32, device=self.device)
            self._v = torch.zeros(self.slice, dtype=f32, device=self.device)
        if self.is_owner and self.world > 1:
            self._in = {w: torch.zeros(self.slice + 2, dtype=f32, device=self.device) for w in range(self.world) if w != self.rank}
            self._out = {w: torch.zeros(self.slice + 2, dtype=f32, device=self.device) for w in range(self.world) if w != self.rank}
            for w in self._in:
                t = threading.Thread(target=self._serve, args=(w,), name=f"dtf-ps-service-{w}", daemon=True)
                t.start()
                self._threads.append(t)

    # ------------------------------------------------------------------ helpers
    def _shard_range(self, s: int) -> Tuple[int, int]:
        return s * self.slice, (s + 1) * self.slice

    @staticmethod
    def _ints(buf: torch.Tensor) -> torch.Tensor:
        return buf[-2:].view(torch.int32)

    # ------------------------------------------------------------------ owner service threads
    def _bump_global_step(self) -> int:
        with self._gs_lock:
            self._gs += 1
            self._applied += 1
            return self._gs

    def _apply(self, master: torch.Tensor, grad: torch.Tensor) -> None:
        """One optimizer application on the owned slice.  SGD is lock-free (Hogwild: the reference's optimizer without
        ``use_locking``, /root/reference/distributed.py:89).  Adam is applied under a lock: its update reads and
        writes three buffers (parameters, m, v), and two service threads interleaving inside it pair one gradient's
        first moment with another's second moment, which was seen to throw a run onto the loss plateau about once in
        ten runs of three workers."""
        st = self.state
        if st.optimizer == "sgd":
            ops.sgd_apply(master, grad, st.lr, 1.0)
            return
        with self._gs_lock:
            self._adam_t += 1
            ops.adam_apply(master, grad, self._m, self._v, st.lr, st.beta1, st.beta2, st.eps, 1.0, step=self._adam_t)

    def _serve(self, w: int) -> None:
        try:
            if self.device.type == "cuda":
                with torch.cuda.stream(torch.cuda.Stream(self.device)):
                    self._serve_loop(w)
            else:
                self._serve_loop(w)
        except BaseException as e:  # surfaced by the training thread
            self._service_error = e

    def _serve_loop(self, w: int) -> None:
        st = self.state
        lo, hi = self._shard_range(self.rank)
        master = st.params[lo:hi]
        group, ctl_group = self.groups[(self.rank, w)], self.groups[(self.rank, w, "ctl")]
        msg, out = self._in[w], self._out[w]
        ctl = torch.zeros(2, dtype=torch.int64)
        while True:
            dist.recv(ctl, src=w, group=ctl_group)      # the doorbell: blocks on the host, nothing is parked on the GPU
            kind = int(ctl[0])
            if kind == KIND_DONE:
                return
            if kind == KIND_MOMENTS:
                # m, v and the update count of ONE update: snapshot under the lock the Adam applications hold
                with self._gs_lock:
                    snap = [t.clone() if t is not None else None for t in (self._m, self._v)]
                    count = self._adam_t
                for t in snap:
                    out[:self.slice].copy_(t) if t is not None else out[:self.slice].zero_()
                    self._ints(out)[0] = count
                    dist.send(out, dst=w, group=group)
                continue
            gs = None
            if kind == KIND_GRAD:
                dist.recv(msg, src=w, group=group)       # posted only now that the worker has announced its push
                # lock-free apply on the owned slice (ApplyGradientDescent on the ps, use_locking=False)
                self._apply(master, msg[:self.slice])
                if self.rank == 0:
                    gs = self._bump_global_step()   # global_step counts every worker's update
                else:
                    with self._gs_lock:
                        self._applied += 1
            if gs is None:
                with self._gs_lock:
                    gs = self._gs
            out[:self.slice].copy_(master)
            self._ints(out)[0] = gs
            dist.send(out, dst=w, group=group)

    # ------------------------------------------------------------------ worker side
    def _complete_oldest(self) -> None:
        slot, works = self._pending.popleft()
        st = self.state
        for s, send_w, recv_w in works:
            if send_w is not None:
                send_w.wait()
            recv_w.wait()
            lo, hi = self._shard_range(s)
            buf = self._pull_buf[s][slot]
            st.params[lo:hi].copy_(buf[:self.slice])
            if s == 0:
                st.state[0:1].copy_(self._ints(buf)[0:1])

    def _exchange(self, kind: int) -> None:
        """Push (gradient / control) to every remote owner and post the matching pulls."""
        st = self.state
        works = []
        while len(self._pending) >= self._ring:
            self._complete_oldest()
        slot = self._exchanges % self._ring
        self._exchanges += 1
        for s in range(self.S):
            lo, hi = self._shard_range(s)
            if s == self.rank:
                if kind == KIND_GRAD:
                    self._apply(st.params[lo:hi], st.grads[lo:hi])
                    if s == 0:
                        st.state[0] = self._bump_global_step()
                    else:
                        with self._gs_lock:
                            self._applied += 1
                elif s == 0:
                    with self._gs_lock:         # a pull by the owner of shard 0: the counter its service threads advanced
                        st.state[0] = self._gs
                continue
            g = self.groups[(s, self.rank)]
            dist.send(torch.tensor([kind, self.updates_pushed], dtype=torch.int64), dst=s, group=self.groups[(s, self.rank, "ctl")])
            if kind == KIND_DONE:
                continue
            send_w = None
            if kind == KIND_GRAD:
                buf = self._push_buf[s][slot]
                buf[:self.slice].copy_(st.grads[lo:hi])
                send_w = dist.isend(buf, dst=s, group=g)
            recv_w = dist.irecv(self._pull_buf[s][slot], src=s, group=g)
            works.append((s, send_w, recv_w))
        if works:
            self._pending.append((slot, works))

    def _check_service(self) -> None:
        if self._service_error is not None:
            raise RuntimeError("parameter-service thread failed") from self._service_error

    def step(self) -> None:
        """One asynchronous worker step."""
        self._check_service()
        # staleness bound: at most `staleness` of my own pushes may be missing from the replica I compute on
        while len(self._pending) > self.staleness:
            self._complete_oldest()
        prog = self.program
        saved, prog.use_graphs = prog.use_graphs, False
        prog.run(1)
        prog.use_graphs = saved
        self._exchange(KIND_GRAD)
        self.updates_pushed += 1

    def run(self, n_steps: int, graph_steps: int = 0) -> None:
        for _ in range(n_steps):
            self.step()

    def sync_params(self) -> int:
        """Drain outstanding pulls and fetch the owners' current values (used before evaluation).
        Returns the freshest global_step this worker knows."""
        while self._pending:
            self._complete_oldest()
        self._exchange(KIND_PULL)
        while self._pending:
            self._complete_oldest()
        self.last_global_step = self.state.global_step()
        return self.last_global_step

    def publish_state(self) -> None:
        """After the chief has initialised or restored the variables: the owners take their counters (and Adam
        moments) from the local state instead of starting from scratch.  Collective: nobody returns - and so nobody
        pushes - before every owner has taken its counters (a push that arrived earlier would be counted and then
        overwritten)."""
        st = self.state
        with self._gs_lock:
            self._gs = st.global_step()
            self._adam_t = int(st.state[2].item())
        lo, hi = self._shard_range(self.rank) if self.is_owner else (0, 0)
        if self._m is not None and st.m is not None and (lo, hi) == tuple(st.shard):
            self._m.copy_(st.m)
            self._v.copy_(st.v)
        if self.world > 1 and dist.is_initialized():
            dist.barrier()

    def gather_moments(self):
        """Full flat Adam moments: this rank's own slice directly, the other owners' through a request on the pair
        channels (the requester is the only other party involved), or None for SGD."""
        st = self.state
        if st.optimizer != "adam":
            return None
        if self._finished:
            # the owners' service threads have returned: a request for their moments would never be answered
            raise RuntimeError("gather_moments after finish(): take the final checkpoint before the engine is torn down")
        while self._pending:
            self._complete_oldest()
        n = st.layout.total
        full = [torch.zeros(n, dtype=torch.float32, device=self.device) for _ in range(2)]
        scratch = torch.zeros(self.slice + 2, dtype=torch.float32, device=self.device)
        for s in range(self.S):
            lo, hi = self._shard_range(s)
            if s == self.rank:
                with self._gs_lock:
                    full[0][lo:hi].copy_(self._m)
                    full[1][lo:hi].copy_(self._v)
                    st.state[2] = self._adam_t
                continue
            dist.send(torch.tensor([KIND_MOMENTS, 0], dtype=torch.int64), dst=s, group=self.groups[(s, self.rank, "ctl")])
            for k in range(2):
                dist.recv(scratch, src=s, group=self.groups[(s, self.rank)])
                full[k][lo:hi].copy_(scratch[:self.slice])
            if s == 0:
                st.state[2:3].copy_(self._ints(scratch)[0:1])
        return full[0], full[1]

    def finish(self, timeout_s: float = 120.0) -> bool:
        """This worker is done: tell the owners, then (owners) keep serving until everyone is done - or until the time
        limit: a worker that died never says so, and must not keep the others from leaving.  Returns whether every
        service thread saw its worker finish."""
        import time
        while self._pending:
            self._complete_oldest()
        self._finished = True
        if self.world > 1:
            self._exchange(KIND_DONE)
        deadline = time.time() + timeout_s
        for t in self._threads:
            t.join(max(0.0, deadline - time.time()))
        everyone = not any(t.is_alive() for t in self._threads)      # daemon threads: a stuck one ends with the process
        self._check_service()
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        if self.rank == 0:
            self.state.state[0] = self._gs
        if self._m is not None and self.state.m is not None and self._shard_range(self.rank) == tuple(self.state.shard):
            self.state.m.copy_(self._m)
            self.state.v.copy_(self._v)
            self.state.state[2] = self._adam_t
        return everyone

    @property
    def updates_applied_here(self) -> int:
        return self._applied

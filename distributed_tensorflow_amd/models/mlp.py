"""The reference network as a device-side step program.

Model: ``x[B,784] -> relu(x W1 + b1) -> softmax(h W2 + b2)`` with the loss and
accuracy of /root/reference/distributed.py:78-87; any number of hidden layers
is supported (the wide-MLP configuration), one is the reference.

A *step program* owns every buffer of the training step and enqueues the whole
step — forward, loss/accuracy, backward, weight gradients, optionally the SGD
update, the counters and the selection of the next batch — on the current HIP
stream without touching the host.  ``run(n)`` replays hipGraphs of such steps.

Row layout of one step (``R = 2B`` rows when post-update accuracy is on):
rows ``[0, B)`` are the training batch; rows ``[B, 2B)`` are the PREVIOUS
batch, pushed through the forward pass again with the now-updated parameters
only to count correct predictions.  That reproduces the reference's second
``sess.run`` per iteration (train accuracy after the update, distributed.py:
148-149) without a second set of launches.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from .. import ops
from ..data.device_dataset import DeviceDataset
from ..parallel.sharded_state import ShardedState, mlp_variables


@dataclass
class StepRecord:
    local_step: int          # 1-based, value of local_step after the increment (distributed.py:146)
    global_step: int         # value fetched with the train op (before this step's own increment)
    loss: float
    train_correct: int       # correct predictions on the batch with the parameters it was trained from
    post_correct: Optional[int] = None   # ... with the parameters after the update (None until known)
    batch: int = 0

    @property
    def train_accuracy(self) -> float:
        c = self.post_correct if self.post_correct is not None else self.train_correct
        return c / float(self.batch)


class _no_gc:
    """Keep the cyclic garbage collector off for the duration of the block (a stream capture).  Garbage that exists
    already stays where it is until the collector runs again after the block; nothing is collected first (a full
    collection costs 30 ms in a process that has imported torch)."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def hidden_names(n_hidden: int) -> List[str]:
    return ["hid" if i == 0 else f"hid{i + 1}" for i in range(n_hidden)]


class MLPProgram:
    _supports_mlp2 = True

    def __init__(self, state: ShardedState, dataset: DeviceDataset, batch: int, hidden: Sequence[int], classes: int = 10,
                 loss_mode: int = ops.LOSS_REFERENCE, fused_update: bool = True, post_update_accuracy: bool = True,
                 gs_inc: int = 1, log_capacity: int = 2048, use_graphs: Optional[bool] = None):
        self.state, self.data = state, dataset
        self.device = state.device
        self.B, self.hidden, self.C = batch, list(hidden), classes
        self.loss_mode = loss_mode
        self.post = post_update_accuracy
        self.fused = fused_update
        if fused_update and (state.optimizer != "sgd" or state.world_size != 1):
            raise ValueError("the update can only be fused into the weight-gradient epilogue for SGD on one rank")
        self.gs_inc = gs_inc
        self.R = 2 * batch if self.post else batch
        self.is_cuda = self.device.type == "cuda"
        self.use_graphs = self.is_cuda if use_graphs is None else (use_graphs and self.is_cuda)
        dev, f32, i32 = self.device, torch.float32, torch.int32
        self.names = hidden_names(len(self.hidden))
        self.hact = [torch.zeros((max(self.R, 128), h), dtype=f32, device=dev) for h in self.hidden]
        self.dz = [torch.zeros((batch, h), dtype=f32, device=dev) for h in self.hidden]
        self.dz2 = torch.zeros((batch, 16 if classes <= 16 else classes), dtype=f32, device=dev)
        # more than 16 classes: the last layer runs as GEMM + general softmax cross-entropy kernel + masked GEMM
        # (ops.head dispatches), enqueued op by op instead of through the prebuilt native schedule
        self.wide_head = classes > 16
        self.nblk = (self.R + 15) // 16
        self.loss_part = torch.zeros(self.nblk, dtype=f32, device=dev)
        self.correct_part = torch.zeros(self.nblk, dtype=i32, device=dev)
        self.post_part = torch.zeros(self.nblk, dtype=i32, device=dev)
        self.log_cap = log_capacity
        self.log = torch.zeros((log_capacity, 4), dtype=i32, device=dev)
        # Two-launch step (csrc/kernels/mlp2_fused.hip) for the reference shape; DTF_FUSED_STEP=0 selects the
        # general three-launch sequence for A/B measurements.
        in_dim = state.var(f"{self.names[0]}_w").shape[0]
        self.fused2 = (self.is_cuda and type(self)._supports_mlp2 and os.environ.get("DTF_FUSED_STEP", "1") != "0"
                       and ops.mlp2_eligible(in_dim, self.hidden, classes, batch))
        # One worker with SGD in the step: the one-launch pipelined step (csrc/kernels/mlp_pipe.hip), whose dependent
        # path works on products with the weights of two steps ago plus Gram-matrix corrections; DTF_PIPELINED_STEP=0
        # keeps the two-launch step.
        self.pipe = (self.is_cuda and type(self)._supports_mlp2 and os.environ.get("DTF_FUSED_STEP", "1") != "0"
                     and self.fused and os.environ.get("DTF_PIPELINED_STEP", "1") != "0"
                     and ops.pipe_eligible(in_dim, self.hidden, classes, batch)
                     and dataset.n >= 8 * batch)        # a launch looks three batches ahead within two epochs
        if (self.is_cuda and type(self)._supports_mlp2 and not self.fused2 and not self.pipe
                and os.environ.get("DTF_FUSED_STEP", "1") != "0"):
            # a different code path, not just a different speed: say so once per program
            import sys
            sys.stderr.write(f"step program: {len(self.hidden)} hidden layer(s) of {self.hidden}, batch {batch}, {classes} classes "
                             "is outside the range of the two-launch step (one hidden layer <= 128 wide, batch <= 128, <= 16 "
                             "classes) and of the one-launch step (one worker, one hidden layer <= 256 wide, batch <= 256): "
                             "using the general three-launch sequence; multi-worker sync then aggregates through "
                             "RCCL instead of the peer exchange inside the backward launch\n")
        if self.pipe:
            self.fused2 = False
        # Per-step buffers rotate over P slots (slot of a step = local step % P).  P = 2: the tail of step s
        # writes the rows of step s+1 while step s is still being read.  The two-launch step also keeps the
        # gathered batch of step s-1 (its accuracy-only rows), hence P = 3.  The pipelined step reads the batches
        # of steps s-1 .. s+2 while it gathers the one of step s+3: P = 8.
        self.P = ops._lib.PIPE_SLOTS if self.pipe else ops.MLP2_SLOTS if self.fused2 else 2
        self.ridx = torch.zeros((self.P, max(128, (2 * batch + 15) // 16 * 16)), dtype=i32, device=dev)
        if self.pipe:
            self.pbuf = ops.PipeBuffers(in_dim, self.hidden[0], dev, batch)
            self.wlog = torch.zeros((log_capacity, ops._lib.WLOG_INTS), dtype=i32, device=dev)
        if self.fused2:
            r16 = (self.R + 15) // 16
            self.xg = torch.zeros((self.P, ops.MLP2_ROWS, in_dim), dtype=f32, device=dev)
            self.labg = torch.zeros((self.P, 16, ops.MLP2_LABG_ROWS), dtype=f32, device=dev)
            self.zpart = torch.zeros(16 * r16 * 16 * 8, dtype=f32, device=dev)
            self.w2_snap = torch.zeros(self.hidden[0] * classes, dtype=f32, device=dev)
            self.ctr_snap = torch.zeros(2, dtype=i32, device=dev)
            self.wlog = torch.zeros((log_capacity, ops._lib.WLOG_INTS), dtype=i32, device=dev)
        self.steps_enqueued = 0           # host mirror of state[1] (counts enqueued, not necessarily finished, steps)
        self._graphs: Dict = {}
        self._call_cache: Dict = {}
        self._schedules: Dict = {}
        self.schedule_tail_entries = []   # appended by the engine: the native update entry of each step
        self.schedule_tail_fn = None      # optional: (slot, phase) -> extra entries (an update entry that depends on them)
        self.phase_period = 1             # steps whose schedules differ beyond the buffer slot repeat with this period
                                          # (phase of a step = local step % phase_period: rounds of a multi-round update)
        self.grad_slots = None            # optional: one flat gradient buffer per slot (peer-mapped all-reduce)
        self.peer_exchange = None         # optional: (per-slot Mlp2Peer structs, lr / R): launch 2 exchanges and updates
        self.async_push = None            # optional: (Mlp2Async, lr): launch 2 applies its gradients to the mapped masters
        self.shard_exchange = None        # optional: (Mlp2Shard, lr / R): launch 2 sends tiles to their owner rank, which updates
        self._eval_buf: Dict = {}
        self.pipe_multi = None            # PipeExchange: the pipelined step of N sync workers (use_pipe_exchange)
        self._args = [self._build_args(q) for q in range(self.P)] if self.is_cuda else None
        self._started = False
        self._log_stream = None                 # stream of log reads that must not wait for later launches
        self.pipe_primes = 0                    # prime launches so far (diagnostic)
        self._pipe_live = None                  # _pipe_token() of the state the pipeline's working copies continue from
        # Called at the end of every step, inside the captured region: the engine's gradient
        # aggregation + shard update + parameter redistribution when the update is not fused.
        self.after_compute = None

    # ------------------------------------------------------------------ argument packs (one per parity)
    def _tail(self, q: int):
        return ops.make_tail(self.state.state, self.log, self.loss_part, self.correct_part, self.post_part,
                             ridx_cur=self.ridx[q], ridx_next=self.ridx[(q + 1) % self.P], perm=self.data.perm,
                             batch=self.B, n_train=self.data.n, gs_inc=self.gs_inc)

    def _wgrad_items(self, q: int):
        st, items = self.state, []
        for l, name in enumerate(self.names):
            x = self.data.images if l == 0 else self.hact[l - 1]
            rows = self.ridx[q] if l == 0 else None
            items.append(ops.WgradItem(x, self.dz[l], self.B, rows=rows, dw=st.grad(f"{name}_w"), db=st.grad(f"{name}_b"),
                                       w=st.var(f"{name}_w"), b=st.var(f"{name}_b")))
        items.append(ops.WgradItem(self.hact[-1], self.dz2, self.B, n=self.C, dw=st.grad("sm_w"), db=st.grad("sm_b"),
                                   w=st.var("sm_w"), b=st.var("sm_b")))
        return items

    def _build_args(self, q: int):
        st = self.state
        if self.pipe_multi is not None:
            return self._build_pipem_args(q)
        head = None if self.wide_head else ops.build_head_args(
            self.hact[-1], st.var("sm_w"), st.var("sm_b"), self.data.labels, self.R, self.B, self.loss_mode,
            rows=self.ridx[q], dz2=self.dz2, dz1=self.dz[-1], loss_part=self.loss_part, correct_part=self.correct_part,
            post_part=self.post_part)
        items = self._wgrad_items(q)
        wgrad = ops.build_wgrad_args(items, lr=st.lr if self.fused else None, tail=self._tail(q))
        out = {"head": head, "wgrad": wgrad, "items": items}
        if self.pipe:
            L = ops._lib
            tail = ops.make_tail(st.state, perm=self.data.perm, batch=self.B, n_train=self.data.n, gs_inc=self.gs_inc)

            def pack(slot, roles, flush=False, prime=False):
                return ops.build_pipe_args(self.data.images, self.data.labels, self.pbuf, self.ridx, st.var("hid_w"),
                                           st.var("hid_b"), st.var("sm_w"), st.var("sm_b"), self.wlog, self.B, self.R,
                                           self.loss_mode, slot % self.P, tail, st.lr, roles, flush, prime)
            # a run whose first step has slot q starts with pipe_prime[q]; one whose last step had slot q ends with pipe_flush[q]
            out["pipe_step"] = pack(q, L.PIPE_ROLE_L | L.PIPE_ROLE_Q | L.PIPE_ROLE_G | L.PIPE_ROLE_W | L.PIPE_ROLE_T)
            out["pipe_prime"] = pack(q - 1, L.PIPE_ROLE_Q | L.PIPE_ROLE_C, prime=True)
            out["pipe_flush"] = pack(q + 1, L.PIPE_ROLE_W | L.PIPE_ROLE_F, flush=True)
        if self.fused2:
            tail = ops.make_tail(st.state, perm=self.data.perm, batch=self.B, n_train=self.data.n, gs_inc=self.gs_inc)
            if self.fused or self.async_push is not None or self.shard_exchange is not None:
                g = lambda n: None
            elif self.grad_slots is not None:
                g = lambda n: st._view(self.grad_slots[q], n)      # gradients land in this slot's exported buffer
            else:
                g = st.grad
            out["mlp2"] = ops.build_mlp2_args(
                self.data.images, self.data.labels, self.xg, self.labg, self.ridx, self.hact[0], self.zpart, self.w2_snap,
                self.ctr_snap, st.var("hid_w"), st.var("hid_b"), st.var("sm_w"), st.var("sm_b"), self.wlog, self.B, self.R,
                self.loss_mode, q, tail, dw1=g("hid_w"), db1=g("hid_b"), dw2=g("sm_w"), db2=g("sm_b"),
                lr=(st.lr if self.fused else self.async_push[1] if self.async_push is not None
                    else self.shard_exchange[1] if self.shard_exchange is not None
                    else self.peer_exchange[1] if self.peer_exchange is not None else None),
                peer=self.peer_exchange[0][q] if self.peer_exchange is not None else None,
                shard=self.shard_exchange[0] if self.shard_exchange is not None else None,
                async_push=self.async_push[0] if self.async_push is not None else None)
        return out

    def _build_pipem_args(self, q: int):
        """Launch packs of the N-worker pipelined step for slot q (step, the prime launch in front of it, the flush
        launch behind it)."""
        st, L, x = self.state, ops._lib, self.pipe_multi
        tail = ops.make_tail(st.state, perm=None, batch=self.B, n_train=self.data.n, gs_inc=self.gs_inc)

        def pack(slot, roles, flush=False, prime=False):
            return ops.build_pipem_args(self.data.images, self.data.labels, self.pbuf, self.ridx, st.var("hid_w"),
                                        st.var("hid_b"), st.var("sm_w"), st.var("sm_b"), self.wlog, self.B, self.R,
                                        self.loss_mode, slot % self.P, tail, self._pipem_lr, roles, x.regions,
                                        self.data.peer_perm, x.err, x.rank, x.world, x.timeout_s, flush, prime)
        return {"pipe_step": pack(q, L.PIPE_ROLE_L | L.PIPE_ROLE_Q | L.PIPE_ROLE_G | L.PIPE_ROLE_W | L.PIPE_ROLE_T),
                "pipe_prime": pack(q - 1, L.PIPE_ROLE_Q | L.PIPE_ROLE_C, prime=True),
                "pipe_flush": pack(q + 1, L.PIPE_ROLE_W | L.PIPE_ROLE_F, flush=True)}

    def pipe_exchange_eligible(self, world: int) -> bool:
        """Can this program run the N-worker pipelined step (``use_pipe_exchange``)?"""
        in_dim = self.state.var(f"{self.names[0]}_w").shape[0]
        return (self.is_cuda and type(self)._supports_mlp2 and not self.fused and self.state.optimizer == "sgd"
                and ops.pipem_eligible(in_dim, self.hidden, self.C, self.B, world) and self.data.n >= 8 * self.B
                and os.environ.get("DTF_PIPELINED_STEP", "1") != "0")

    def use_pipe_exchange(self, exchange, lr_over_r: float) -> None:
        """Sync mode, every update the mean of one gradient per worker: the one-launch pipelined step of N workers
        (csrc/kernels/mlp_pipe_multi.hip).  ``exchange``: the ``PipeExchange`` with every rank's region mapped; the data
        set must know every worker's shuffle (``DeviceDataset.enable_peer_shuffles``).  Parameters and counters carry
        over; the next run starts with a prime launch."""
        if not self.pipe_exchange_eligible(exchange.world):
            raise ValueError("the N-worker pipelined step does not cover this program (shape, optimizer or world)")
        if self.data.peer_perm is None or len(self.data.peer_perm) != exchange.world:
            raise ValueError("the N-worker pipelined step needs every worker's shuffle (enable_peer_shuffles)")
        in_dim = self.state.var(f"{self.names[0]}_w").shape[0]
        dev, i32 = self.device, torch.int32
        self.pipe, self.pipe_multi, self._pipem_lr = True, exchange, float(lr_over_r)
        self._fused2_before_pipe = self.fused2 or getattr(self, "_fused2_before_pipe", False)
        self.fused2 = False
        self.peer_exchange = self.shard_exchange = self.grad_slots = self.async_push = None
        self.schedule_tail_fn = None
        self.P = ops._lib.PIPE_SLOTS
        self.ridx = torch.zeros((self.P, exchange.world, max(128, (2 * self.B + 15) // 16 * 16)), dtype=i32, device=dev)
        self.pbuf = ops.PipeMultiBuffers(in_dim, self.hidden[0], exchange.world, dev)
        if not hasattr(self, "wlog"):
            self.wlog = torch.zeros((self.log_cap, ops._lib.WLOG_INTS), dtype=i32, device=dev)
        self._args = [self._build_args(q) for q in range(self.P)]
        self._schedules.clear()
        self._graphs.clear()
        self._started = False
        self._pipe_live = None

    def use_two_launch(self) -> None:
        """Back from the N-worker pipelined step to the two-launch step (the transports that exchange weight-gradient
        tiles work on it); parameters and counters carry over."""
        if self.pipe_multi is None:
            return
        dev, i32 = self.device, torch.int32
        self.pipe, self.pipe_multi = False, None
        self.fused2 = bool(getattr(self, "_fused2_before_pipe", False))
        self.P = ops.MLP2_SLOTS if self.fused2 else 2
        self.ridx = torch.zeros((self.P, max(128, (2 * self.B + 15) // 16 * 16)), dtype=i32, device=dev)
        self._args = [self._build_args(q) for q in range(self.P)]
        self._schedules.clear()
        self._graphs.clear()
        self._started = False
        self._pipe_live = None

    def use_gradient_slots(self, slots) -> None:
        """Make the two-launch step write its gradients into ``slots[slot of the step]`` (flat buffers laid out
        like the state's gradient buffer) instead of the state's own gradient buffer."""
        if not self.fused2 or self.fused or len(slots) != self.P:
            raise ValueError("gradient slots need the two-launch step with stored gradients and one buffer per slot")
        if any(t.numel() != self.state.grads.numel() or t.dtype != torch.float32 for t in slots):
            raise ValueError("gradient slots must match the flat gradient buffer")
        self.grad_slots = list(slots)
        self.peer_exchange = self.shard_exchange = None
        self._args = [self._build_args(q) for q in range(self.P)]
        self._schedules.clear()
        self._graphs.clear()

    def use_peer_exchange(self, slots, peers, lr_over_r: float) -> None:
        """Gradient slots as in :meth:`use_gradient_slots`, and the backward launch itself exchanges the tiles with
        the peers (``peers[q]``: ``Mlp2Peer`` of slot q) and applies ``theta -= lr_over_r * sum over ranks``."""
        if len(peers) != self.P:
            raise ValueError("one peer struct per slot")
        self.use_gradient_slots(slots)
        self.peer_exchange = (list(peers), float(lr_over_r))
        self._args = [self._build_args(q) for q in range(self.P)]

    def use_sharded_exchange(self, shard, lr_over_r: float) -> None:
        """Sync mode with the parameter-server state sharded over the ranks, inside the two-launch step: every training
        workgroup of the backward launch sends its gradient tiles to the rank that owns them (``shard``: ``Mlp2Shard``),
        the owner applies ``theta -= lr_over_r * sum over ranks`` and stores the new tiles into every replica."""
        if not self.fused2 or self.fused:
            raise ValueError("the sharded exchange needs the two-launch step without the fused local update")
        self.shard_exchange = (shard, float(lr_over_r))
        self.peer_exchange = self.grad_slots = self.async_push = None
        self._args = [self._build_args(q) for q in range(self.P)]
        self._schedules.clear()
        self._graphs.clear()

    def use_async_push(self, async_struct, lr: float) -> None:
        """Async mode on the two-launch step: the backward launch applies ``lr * gradient`` to the mapped master
        slices (``async_struct``: ``Mlp2Async``) and keeps the result as the local replica; no gradients are stored."""
        if not self.fused2 or self.fused or self.gs_inc != 0:
            raise ValueError("the in-launch async push needs the two-launch step with global_step left to the ps")
        self.async_push = (async_struct, float(lr))
        self.peer_exchange = self.grad_slots = self.shard_exchange = None
        self._args = [self._build_args(q) for q in range(self.P)]
        self._schedules.clear()
        self._graphs.clear()

    def _launch_cached(self, key, build):
        """Argument structs of a launch are built once per (step parity, call site) and reused: the
        eager path then costs one ctypes call per kernel."""
        ent = self._call_cache.get(key)
        if ent is None:
            ent = build()
            self._call_cache[key] = ent
        fn, arg = ent
        ops._lib.check(fn(arg, ops._lib.stream_handle()), key[0])

    # ------------------------------------------------------------------ one step
    def start(self) -> None:
        """Select the rows of the first batch (runs once, before the first step)."""
        self.steps_enqueued = self.state.local_step()
        q = self.steps_enqueued % self.P
        if self.pipe:
            self._start_pipe(q)
            self._started = True
            return
        self.data.prepare(self.steps_enqueued, 0, self.B)
        if self.is_cuda:
            t = ops.make_tail(self.state.state, ridx_cur=None, ridx_next=self.ridx[q], perm=self.data.perm, batch=self.B,
                              n_train=self.data.n, gs_inc=0)
            ops.step_tail(t, init_only=True)
            if self.fused2:
                # the first step finds its batch (and, as its accuracy-only rows, the same batch) already gathered
                rows = self.ridx[q][:self.B].long()
                for slot in (q, (q - 1) % self.P):
                    self.xg[slot][:self.B].copy_(self.data.images.index_select(0, rows))
                    self.labg[slot].zero_()
                    self.labg[slot][:self.C, :self.B].copy_(self.data.labels.index_select(0, rows).t())
        else:
            self._tail_cpu(q, init_only=True)
        self._started = True

    def _rows_of_step(self, step: int, seed=None) -> torch.Tensor:
        """Dataset rows of the batch of 0-based local step ``step``, computed on the host (the same permutation the
        device holds) and uploaded: no device arithmetic, whose first use costs tens of milliseconds."""
        import numpy as np
        from ..data.mnist import epoch_permutation
        pos = np.arange(step * self.B, (step + 1) * self.B, dtype=np.int64)
        epoch, off = pos // self.data.n, pos % self.data.n
        if self.data.perm is None:
            rows = off
        else:
            rows = np.empty(self.B, dtype=np.int64)
            for e in np.unique(epoch):
                m = epoch == e
                rows[m] = epoch_permutation(self.data.seed if seed is None else seed, int(e), self.data.n)[off[m]]
        return torch.from_numpy(rows).to(self.device)

    def _start_pipe(self, q: int) -> None:
        """The pipelined step finds the batches of its first three steps gathered (a step gathers the batch three
        steps ahead, through the row indices selected one step earlier still) and, as the accuracy-only rows of the
        first step, the first batch again."""
        ls, B, C, b = self.steps_enqueued, self.B, self.C, self.pbuf
        in_dim = self.data.images.shape[1]
        self.data.prepare(ls, 3, B)
        if self.pipe_multi is not None:
            return self._start_pipe_multi(q)
        prev = None
        for j in range(4):
            slot = (q + j) % self.P
            rows = self._rows_of_step(ls + j)
            self.ridx[slot][:B] = rows.to(torch.int32)
            self.ridx[slot][B:2 * B] = (rows if prev is None else prev).to(torch.int32)
            prev = rows
            if j == 3:
                break                                   # the first step gathers this batch itself
            lab = self.data.labels.index_select(0, rows)
            b.xa[slot].zero_()
            b.xa[slot][:B, :in_dim].copy_(self.data.images.index_select(0, rows))
            b.xa[slot][:B, in_dim] = 1.0
            b.labg[slot].zero_()
            b.labg[slot][:C, :B].copy_(lab.t())
            b.labi[slot].zero_()
            b.labi[slot][:B] = lab.argmax(1).to(torch.int32)     # first maximal entry, as tf.argmax
        before = (q - 1) % self.P
        b.xa[before].copy_(b.xa[q])
        b.labg[before].copy_(b.labg[q])
        b.labi[before].copy_(b.labi[q])

    def _start_pipe_multi(self, q: int) -> None:
        """As ``_start_pipe``, for every worker's batches (labels: this worker's only)."""
        ls, B, C, b, x = self.steps_enqueued, self.B, self.C, self.pbuf, self.pipe_multi
        in_dim = self.data.images.shape[1]
        for src in range(x.world):
            seed = self.data.peer_seeds[src]
            own = src == x.rank
            prev = None
            for j in range(4):
                slot = (q + j) % self.P
                rows = self._rows_of_step(ls + j, seed)
                self.ridx[slot, src, :B] = rows.to(torch.int32)
                self.ridx[slot, src, B:2 * B] = (rows if prev is None else prev).to(torch.int32)
                prev = rows
                if j == 3:
                    break                               # the first step gathers this batch itself
                b.xa[slot, src].zero_()
                b.xa[slot, src, :B, :in_dim].copy_(self.data.images.index_select(0, rows))
                b.xa[slot, src, :B, in_dim] = 1.0
                if own:
                    lab = self.data.labels.index_select(0, rows)
                    b.labg[slot].zero_()
                    b.labg[slot][:C, :B].copy_(lab.t())
                    b.labi[slot].zero_()
                    b.labi[slot][:B] = lab.argmax(1).to(torch.int32)
            before = (q - 1) % self.P
            b.xa[before, src].copy_(b.xa[q, src])
            # (steps -2 of the first launches multiply gathered rows with zero gradients: any finite rows do)
            b.xa[(q - 2) % self.P, src].copy_(b.xa[q, src])
        before = (q - 1) % self.P
        b.labg[before].copy_(b.labg[q])
        b.labi[before].copy_(b.labi[q])

    def _pipe_token(self):
        """What the pipeline's working copies were built from: a change of any of these between two runs (parameters
        or counters rewritten through the state's methods or by an in-place tensor operation, steps enqueued
        elsewhere) means the next run starts with a prime launch.  Kernels write through raw pointers and leave the
        tensors' version counters alone."""
        st = self.state
        if self.pipe_multi is not None and self.pipe_multi.world > 1:
            # Several workers prime together or not at all (the prime launch is bracketed by barriers), so the decision
            # may only rest on what every rank sees alike: parameters rewritten through the state's methods (initialise,
            # restore, broadcast - every rank goes through them) and the number of steps.  An in-place edit of the flat
            # tensors on one rank is not noticed here: the pipeline carries on from its working copies and the next
            # flush overwrites the edit.
            return (st.values_version, self.steps_enqueued)
        return (st.values_version, st.params._version, st.state._version, self.steps_enqueued)

    def _pipe_open(self) -> None:
        """Start of a run of pipelined steps.  The flush launch at the end of a run only materialises the flat
        parameters: working copies, pending gradients and the products for the next step stay as the last step left
        them, so a run that follows with nothing rewritten in between continues the pipeline as if it were one longer
        run.  Otherwise the prime launch: working copies and Q of the next step from the flat parameters."""
        if self._pipe_live == self._pipe_token():
            return
        if self.pipe_multi is not None:
            # every rank's boxes are zeroed by its own prime launch: nobody steps before everybody has primed (and
            # nobody primes while a peer's steps of an earlier run could still be writing: its flush has waited for them)
            self.pipe_multi.barrier()
            ops.pipem_prime(self._args[self.steps_enqueued % self.P]["pipe_prime"])
            self.pipe_multi.barrier()
        else:
            ops.pipe_prime(self._args[self.steps_enqueued % self.P]["pipe_prime"])
        self.pipe_primes += 1

    def _pipe_close(self, launched: bool = False) -> None:
        """Flush launch: the parameters after the last enqueued step into the flat buffer (``launched``: the last
        graph of the run ended with it)."""
        if not launched:
            self._pipe_flush_launch()
        self._pipe_live = self._pipe_token()

    def _pipe_flush_launch(self) -> None:
        args = self._args[(self.steps_enqueued - 1) % self.P]["pipe_flush"]
        ops.pipem_flush(args) if self.pipe_multi is not None else ops.pipe_flush(args)

    def _pipe_step(self) -> None:
        args = self._args[self.steps_enqueued % self.P]["pipe_step"]
        ops.pipem_step(args) if self.pipe_multi is not None else ops.pipe_step(args)
        self.steps_enqueued += 1

    def _compute_schedule(self, q: int):
        """The whole compute part of a step (forward, head, backward, weight gradients + tail) as one
        native launch sequence; built once per step parity."""
        phase = self.steps_enqueued % self.phase_period
        sch = self._schedules.get((q, phase))
        if sch is None:
            st, rows, L = self.state, self.ridx[q], ops._lib
            entries = []
            x = self.data.images
            if self.fused2:
                entries = [(L.OP_MLP2_FWD, self._args[q]["mlp2"], 0), (L.OP_MLP2_BWD, self._args[q]["mlp2"], 0)]
                entries.extend(self.schedule_tail_entries)
                if self.schedule_tail_fn is not None:
                    entries.extend(self.schedule_tail_fn(q, phase))
                sch = self._schedules[(q, phase)] = L.Schedule(entries)
                return sch
            for l, name in enumerate(self.names):
                entries.append((L.OP_SMALL_GEMM, ops.linear_fwd(x, st.var(f"{name}_w"), st.var(f"{name}_b"), True,
                                                                rows if l == 0 else None, self.hact[l], self.R,
                                                                _build_only=True), 0))
                x = self.hact[l]
            entries.append((L.OP_HEAD, self._args[q]["head"], 0))
            for l in range(len(self.names) - 1, 0, -1):
                entries.append((L.OP_SMALL_GEMM, ops.linear_bwd_data(self.dz[l], st.var(f"{self.names[l]}_w"),
                                                                     self.hact[l - 1], self.dz[l - 1], self.B,
                                                                     _build_only=True), 0))
            entries.append((L.OP_WGRAD, self._args[q]["wgrad"], 0))
            entries.extend(self.schedule_tail_entries)
            if self.schedule_tail_fn is not None:
                entries.extend(self.schedule_tail_fn(q, phase))
            sch = L.Schedule(entries)
            self._schedules[(q, phase)] = sch
        return sch

    def enqueue_step(self) -> None:
        """Enqueue one full training step on the current stream (no host synchronisation)."""
        q = self.steps_enqueued % self.P
        st, rows = self.state, self.ridx[q]
        if self.pipe:
            # one step on its own: between prime and flush, so that the flat parameters are current before and after
            self._pipe_open()
            self._pipe_step()
            self._pipe_close()
            return
        if self.is_cuda and not self.wide_head:
            self._compute_schedule(q).run()
        else:
            x = self.data.images
            for l, name in enumerate(self.names):
                ops.linear_fwd(x, st.var(f"{name}_w"), st.var(f"{name}_b"), relu=True, rows=rows if l == 0 else None,
                               out=self.hact[l], m=self.R)
                x = self.hact[l]
            ops.head(self.hact[-1], st.var("sm_w"), st.var("sm_b"), self.data.labels, self.R, self.B, self.loss_mode,
                     rows=rows, dz2=self.dz2, dz1=self.dz[-1], loss_part=self.loss_part, correct_part=self.correct_part,
                     post_part=self.post_part)
            for l in range(len(self.names) - 1, 0, -1):
                ops.linear_bwd_data(self.dz[l], st.var(f"{self.names[l]}_w"), act=self.hact[l - 1], out=self.dz[l - 1],
                                    m=self.B)
            if self.is_cuda:
                ops.wgrad_grouped(self._args[q]["items"], args=self._args[q]["wgrad"])      # with the step tail
            else:
                ops.wgrad_grouped(self._wgrad_items(q), lr=st.lr if self.fused else None)
                self._tail_cpu(q, init_only=False)
        if self.after_compute is not None:
            self.after_compute()
        self.steps_enqueued += 1

    def _tail_cpu(self, q: int, init_only: bool) -> None:
        s = self.state.state
        gs, ls = int(s[0]), int(s[1])
        if not init_only:
            rec = self.log[ls % self.log_cap]
            loss = float(self.loss_part.sum()) * (1.0 / self.B)
            rec[0] = struct.unpack("<i", struct.pack("<f", loss))[0]
            rec[1] = int(self.correct_part.sum())
            rec[2] = int(self.post_part.sum())
            rec[3] = gs
            s[0] = gs + self.gs_inc
            s[1] = ls + 1
        nxt = ls if init_only else ls + 1
        pos = torch.arange(nxt * self.B, (nxt + 1) * self.B, dtype=torch.int64)
        epoch, off = pos // self.data.n, pos % self.data.n
        if self.data.perm is not None:
            rows = self.data.perm[(epoch & 1), off].to(torch.int32)
        else:
            rows = off.to(torch.int32)
        dst = self.ridx[q] if init_only else self.ridx[(q + 1) % self.P]
        prev = rows if init_only else self.ridx[q][:self.B].clone()
        dst[self.B:2 * self.B] = prev
        dst[:self.B] = rows

    # ------------------------------------------------------------------ multi-step execution
    def _graph_key(self, n: int, at: int, last: bool = False):
        key = (n, at % self.P) if self.phase_period == 1 else (n, at % self.P, at % self.phase_period)
        return key + ("flush",) if last and self.pipe else key

    def _graph_for(self, n: int, last: bool = False):
        """The graph of ``n`` steps from ``steps_enqueued`` on.  ``last`` (pipelined step): the graph of the last
        launch of a run ends with the flush launch, so that a short run is ONE graph launch."""
        key = self._graph_key(n, self.steps_enqueued, last)
        g = self._graphs.get(key)
        if g is None:
            saved = self.steps_enqueued
            # warm the kernels once outside capture is not needed: nothing here allocates or JIT-compiles.
            # The cyclic garbage collector must not run inside the capture: if it finds an older program there, that
            # program's hipGraphs are destroyed while this stream is capturing, which HIP refuses (an abort from the
            # destructor) or, worse, lets through with a damaged capture (observed: a replayed graph whose first steps
            # computed on stale buffers).  Collect first, then keep the collector off until the capture has ended.
            g = torch.cuda.CUDAGraph()
            with _no_gc():
                with torch.cuda.graph(g):
                    for _ in range(n):
                        # (a graph of pipelined steps holds the step launches, the last graph of a run the flush
                        # launch as well; the pipeline state carries over between graphs and between runs)
                        self._pipe_step() if self.pipe else self.enqueue_step()
                    if last and self.pipe:
                        self._pipe_flush_launch()
            self.steps_enqueued = saved   # capture enqueues nothing
            self._graphs[key] = g
        return g

    def _launch_sizes(self, n_steps: int, graph_steps: int, exact_tail: bool = False):
        """Steps per launch of a run of ``n_steps`` that starts at ``steps_enqueued``: whole graphs of ``graph_steps``,
        then the rest.  The rest is ONE graph of exactly that many steps when such a graph exists already (or
        ``exact_tail`` asks for it: ``prepare_graphs``), so that a prepared run of 20 steps is one replay; otherwise
        powers of two, which bounds the number of different graphs an unprepared caller can make us capture."""
        limit = max(1, self.data.max_steps_per_launch(self.B) - self._lookahead())
        graph_steps = max(2, min(graph_steps, limit) & ~1)
        remaining, at = n_steps, self.steps_enqueued
        while remaining > 0:
            if self.use_graphs and remaining >= 2:
                if remaining >= graph_steps:
                    n = graph_steps
                elif exact_tail or self._graph_key(remaining, at, True) in self._graphs:
                    n = remaining
                else:
                    n = 1 << (remaining.bit_length() - 1)
            else:
                n = 1
            yield n
            remaining -= n
            at += n

    def _lookahead(self) -> int:
        """Steps beyond the next one whose rows a step already selects (the pipelined step selects four ahead)."""
        return 3 if self.pipe else 0

    def prepare_graphs(self, n_steps: int, graph_steps: int = 64, start_offset: int = 0) -> None:
        """Capture and instantiate, without launching anything, the hipGraphs a later ``run(n_steps, graph_steps)``
        will replay (a graph is keyed by its length and the buffer slot it starts on): set-up cost kept out of a
        timed run.  The steps beyond the last whole graph become one graph of exactly that length.  ``start_offset``:
        that run starts after this many further steps (a warm-up still to come), so that nothing but the warm-up
        itself has to happen between preparing and timing."""
        if not self.use_graphs:
            return
        if not self._started:
            self.start()
        saved = self.steps_enqueued
        self.steps_enqueued += start_offset
        try:
            sizes = list(self._launch_sizes(n_steps, graph_steps, exact_tail=True))
            for i, n in enumerate(sizes):
                if n >= 2:
                    self._graph_for(n, last=i == len(sizes) - 1)
                self.steps_enqueued += n
        finally:
            self.steps_enqueued = saved

    def native_launch_ok(self) -> bool:
        """Steps of this program can be launched many at a time by the native executor (no Python between them)."""
        return (self.is_cuda and not self.wide_head and not self.pipe and self.after_compute is None
                and self.phase_period == 1 and hasattr(self, "schedule_tail_entries"))

    def run_native(self, n_steps: int) -> None:
        """``n_steps`` steps enqueued by ONE native call (``dtf_run_schedule_steps``): the device starts a few
        microseconds after the call, where a graph replay needs 10-16 us to start; the host stays ahead as long as a
        kernel runs longer than its launch call takes."""
        if not self._started:
            self.start()
        if not self.native_launch_ok():
            raise RuntimeError("this program's steps need Python between them")
        saved = self.steps_enqueued
        scheds = []
        for q in range(self.P):                       # one schedule per buffer slot, built on first use
            self.steps_enqueued = q
            scheds.append(self._compute_schedule(q))
        self.steps_enqueued = saved
        limit = self.data.max_steps_per_launch(self.B)         # a call may touch two epochs' permutations at most
        while n_steps > 0:
            n = min(n_steps, limit)
            self.data.prepare(self.steps_enqueued, n, self.B)
            ops._lib.run_schedule_steps(scheds, self.steps_enqueued % self.P, n)
            self.steps_enqueued += n
            n_steps -= n

    def run(self, n_steps: int, graph_steps: int = 64) -> None:
        """Enqueue ``n_steps`` steps; returns without waiting for the device."""
        if not self._started:
            self.start()
        if n_steps <= 0:
            return
        if self.pipe:
            self._pipe_open()
        sizes = list(self._launch_sizes(n_steps, graph_steps))
        flushed = False
        for i, n in enumerate(sizes):
            self.data.prepare(self.steps_enqueued, n + self._lookahead(), self.B)
            if n >= 2:
                flushed = self.pipe and i == len(sizes) - 1
                self._graph_for(n, last=flushed).replay()
                self.steps_enqueued += n
            elif self.pipe:
                self._pipe_step()
            else:
                self.enqueue_step()
        if self.pipe:
            self._pipe_close(launched=flushed)

    def read_log(self, first_local_step: int, count: int, after=None) -> List[StepRecord]:
        """Records of local steps ``[first, first+count)`` (0-based).  Synchronises with the device; with ``after`` (an
        event recorded behind the launch that ran those steps) only with that event: the records cross the host link
        on a stream of their own while later launches keep the device busy."""
        if count > self.log_cap:
            raise ValueError("asked for more records than the log ring holds")
        log = self._log_host(first_local_step, count, after).numpy()
        loss = log[:, 0].copy().view("<f4").tolist()
        correct, post, gs = log[:, 1].tolist(), log[:, 2].tolist(), log[:, 3].tolist()
        self.first_post_of_last_read = post[0]       # accuracy-only rows of the first step read (= the step before it)
        out = [StepRecord(local_step=first_local_step + i + 1, global_step=gs[i], loss=loss[i], train_correct=correct[i],
                          batch=self.B) for i in range(count)]
        if self.post:
            # the post-update count of step s is produced by the accuracy-only rows of step s+1
            for i in range(count - 1):
                out[i].post_correct = post[i + 1]
        return out

    def _log_host(self, first_local_step: int = 0, count: Optional[int] = None, after=None) -> torch.Tensor:
        """Records of local steps ``[first, first + count)`` (default: the whole ring, indexed by step % capacity) on
        the host as ``[count, 4]`` int32 {loss bits, correct, post, global_step}.  The two-launch and the one-launch
        step log per-row-block partial sums; they are folded here in a fixed order.  Only the rows asked for cross
        the host link."""
        src = self.wlog if (self.fused2 or self.pipe) else self.log

        def fetch():
            if count is None:
                return src.cpu()
            lo = first_local_step % self.log_cap
            if lo + count <= self.log_cap:
                return src[lo:lo + count].cpu()
            return torch.cat([src[lo:], src[:lo + count - self.log_cap]]).cpu()     # the range wraps around the ring
        if after is not None and self.is_cuda:
            if self._log_stream is None:
                self._log_stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(self._log_stream):
                self._log_stream.wait_event(after)
                w = fetch()                             # waits for this stream only
        else:
            w = fetch()
        if not (self.fused2 or self.pipe):
            return w
        nb = min(16, (self.R + 15) // 16)           # 16 words per field of the record (loss parts, correct, post)
        loss = torch.zeros(w.shape[0], dtype=torch.float32)
        parts = w[:, 4:4 + nb].contiguous().view(torch.float32)
        for rb in range(nb):
            loss = loss + parts[:, rb]
        loss = loss * torch.tensor(1.0 / self.B, dtype=torch.float32)
        out = torch.empty((w.shape[0], 4), dtype=torch.int32)
        out[:, 0] = loss.view(torch.int32)
        out[:, 1] = w[:, 20:20 + nb].sum(1).to(torch.int32)
        out[:, 2] = w[:, 36:36 + nb].sum(1).to(torch.int32)
        out[:, 3] = w[:, 0]
        return out

    def post_count_logged_at(self, local_step: int) -> int:
        """Correct predictions on the accuracy-only rows of 0-based step ``local_step`` (= the batch of the
        step before it, scored with the parameters that step produced)."""
        return int(self._log_host(local_step, 1)[0][2])

    def post_correct_of_last_step(self) -> int:
        """Post-update accuracy count of the most recent step (needs one extra forward)."""
        q = self.steps_enqueued % self.P
        own = self.ridx[q] if self.pipe_multi is None else self.ridx[q, self.pipe_multi.rank]
        rows = own[self.B:2 * self.B].contiguous()
        return self._count_correct(self.data.images, self.data.labels, rows=rows, n=self.B)

    # ------------------------------------------------------------------ evaluation
    def _count_correct(self, images, labels, rows=None, n=None, chunk: int = 16384) -> int:
        n = n if n is not None else images.shape[0]
        st = self.state
        total = 0
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            key = m
            buf = self._eval_buf.get(key)
            if buf is None:
                nb = (m + 15) // 16
                buf = {"h": [torch.empty((m, h), dtype=torch.float32, device=self.device) for h in self.hidden],
                       "lp": torch.zeros(nb, dtype=torch.float32, device=self.device),
                       "cp": torch.zeros(nb, dtype=torch.int32, device=self.device),
                       "pp": torch.zeros(nb, dtype=torch.int32, device=self.device)}
                self._eval_buf = {key: buf}
            if rows is not None:
                r, xi, li = rows[lo:lo + m].contiguous(), images, labels
            else:
                r, xi, li = None, images[lo:lo + m], labels[lo:lo + m]
            x = xi
            for l, name in enumerate(self.names):
                ops.linear_fwd(x, st.var(f"{name}_w"), st.var(f"{name}_b"), relu=True, rows=r if l == 0 else None,
                               out=buf["h"][l], m=m)
                x = buf["h"][l]
            ops.head(x, st.var("sm_w"), st.var("sm_b"), li, m, 0, self.loss_mode, rows=r, loss_part=buf["lp"],
                     correct_part=buf["cp"], post_part=buf["pp"])
            total += int(buf["pp"].sum().item())
        return total

    def accuracy(self, images: torch.Tensor, labels: torch.Tensor) -> float:
        """``accuracy.eval`` over a whole split in one go (distributed.py:141-142, :163-164)."""
        return self._count_correct(images, labels) / float(images.shape[0])


def build_mlp_state(input_dim: int, hidden: Sequence[int], classes: int, device, world_size=1, rank=0, optimizer="sgd",
                    lr=0.01, **kw) -> ShardedState:
    return ShardedState(mlp_variables(input_dim, hidden, classes), device, world_size, rank, optimizer, lr, **kw)

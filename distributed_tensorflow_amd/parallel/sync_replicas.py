"""Synchronous aggregate-then-apply training (``tf.train.SyncReplicasOptimizer``
as used at /root/reference/distributed.py:91-106, :128-131).

Semantics kept from the reference:

* an update applies the MEAN of ``replicas_to_aggregate`` (R) gradients, all
  computed on the same parameter version;
* ``global_step`` advances once per aggregated update;
* R < num_workers: the slowest ``num_workers - R`` gradients of a step are
  dropped (backup workers); R > num_workers: workers contribute more than once
  per update;
* workers leave a step together (the token queue is a barrier).

What replaces the machinery: the per-variable ConditionalAccumulators on the
ps and the chief's queue-runner thread become ONE reduce-scatter over the flat
gradient buffer, the fused optimizer kernel on the slice this rank owns (scale
1/R folded in), and ONE all-gather of the updated slices.  The token queue is
the collective itself.  For R == num_workers the whole step — compute kernels,
RCCL collectives, shard update — is captured in a hipGraph and the host is out
of the loop.  For R != num_workers the set of contributors is decided per step
from arrival times exchanged in a small all-gather, which needs the host.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional

import torch
import torch.distributed as dist

from .. import ops
from .comm import Collectives
from .sharded_state import ShardedState

# Below this many parameter bytes one all-reduce + replicated SGD beats reduce-scatter + all-gather
# (two latency-bound collectives instead of one).  Only valid for stateless SGD.
ALLREDUCE_BELOW_BYTES = 1 << 20
# From this many parameter bytes on, programs that can signal per-bucket gradient completion overlap the
# aggregation with their backward pass (parallel/bucketed.py).
BUCKETED_ABOVE_BYTES = 16 << 20


class SyncReplicasEngine:
    def __init__(self, program, state: ShardedState, group=None, replicas_to_aggregate: Optional[int] = None,
                 total_num_replicas: Optional[int] = None, collective: str = "auto",
                 arrival_delay: Optional[Callable[[int], float]] = None, force_collectives: bool = False,
                 capture_collectives: Optional[bool] = None, device_delay_s: float = 0.0, stamp_offset: int = 0):
        self.program, self.state = program, state
        self.world = state.world_size
        self.total = total_num_replicas if total_num_replicas is not None else self.world
        if self.total != self.world:
            raise ValueError("total_num_replicas must equal the number of worker ranks")
        self.R = replicas_to_aggregate if replicas_to_aggregate is not None else self.world
        if self.R <= 0:
            raise ValueError("replicas_to_aggregate must be positive")
        # force_collectives: issue the collectives even on a one-rank group (exercises the RCCL calls
        # and their hipGraph capture on a single-GPU box)
        self.use_coll = self.world > 1 or force_collectives
        # Steps that contain collectives are launched eagerly unless asked otherwise: with several ranks
        # the collective's latency, not the host, bounds the step, and an eager launch cannot wedge a
        # graph replay if one rank falls over.  DTF_CAPTURE_COLLECTIVES=1 (or the argument) captures them.
        if capture_collectives is None:
            capture_collectives = os.environ.get("DTF_CAPTURE_COLLECTIVES", "0") == "1" or force_collectives
        self.capture_collectives = capture_collectives
        self.coll = Collectives(group) if self.use_coll else None
        # Peer-mapped transports of the small-model step (one node): "replicated" = every rank reads every peer's
        # gradient tiles and updates its own replica (one hop, (N-1) x all gradient bytes per rank), "sharded" = every
        # tile is summed and updated by its owner rank and stored into every replica (two hops, 2 (N-1)/N of the
        # bytes).  p2p_* name the wish; the library collective named next to it is what runs where peers cannot be mapped.
        self._want_p2p = None
        self._p2p_pinned = collective in ("p2p_sharded", "p2p_replicated", "p2p_pipelined")
        if collective == "p2p_sharded":
            self._want_p2p, collective = "sharded", "reduce_scatter"
        elif collective == "p2p_replicated":
            self._want_p2p, collective = "replicated", "all_reduce"
        elif collective == "p2p_pipelined":
            self._want_p2p, collective = "pipelined", "all_reduce"
        if collective == "auto":
            small = state.nbytes() < ALLREDUCE_BELOW_BYTES and state.optimizer == "sgd"
            collective = "all_reduce" if small else "reduce_scatter"
            if state.nbytes() < ALLREDUCE_BELOW_BYTES and state.optimizer == "adam":
                self._want_p2p = "sharded"      # small model with Adam: owner-resident moments on the sharded exchange
            # large state and a program that reports when parts of its gradient are complete: aggregate bucket by
            # bucket behind the backward pass (DTF_BUCKETED=0 keeps the single reduce-scatter after it)
            if (state.nbytes() >= BUCKETED_ABOVE_BYTES and self.use_coll and hasattr(program, "grad_groups")
                    and os.environ.get("DTF_BUCKETED", "1") != "0"):
                collective = "bucketed"
        if collective == "bucketed" and not hasattr(program, "grad_groups"):
            raise ValueError("collective='bucketed' needs a program that reports gradient buckets (grad_groups)")
        if collective == "all_reduce" and state.optimizer != "sgd":
            raise ValueError("the replicated all-reduce path is only defined for SGD")
        self.collective = collective
        if self._want_p2p is None and collective == "all_reduce":
            # No measurement to go by (bench.py and the training script time the candidates on the machine and call
            # set_p2p_mode): the one-launch pipelined step where it applies and few workers keep its Gram work small,
            # else the sharded exchange from four workers on (two hops, 1/4 of the bytes per link), else replicated.
            default = "pipelined" if self.world <= 4 else "sharded"
            self._want_p2p = os.environ.get("DTF_P2P_MODE", default)
            if self._want_p2p not in ("replicated", "sharded", "pipelined"):
                raise ValueError("DTF_P2P_MODE must be 'replicated', 'sharded' or 'pipelined'")
        self.p2p_mode = None                    # the peer-mapped transport in use, else None
        self.pipe_x = None                      # exchange regions of the N-worker pipelined step, once opened
        self._pipe_refused = False
        # DTF_FORCE_GENERAL_SYNC=1: R == workers through the general (stamped, chief-selected) path as well
        self.fast = self.R == self.world and os.environ.get("DTF_FORCE_GENERAL_SYNC", "0") != "1"
        self.arrival_delay = arrival_delay      # test hook: artificial per-step compute delay of this rank (host path)
        self.device_delay_s = device_delay_s    # test hooks of the device path: this rank's contributions arrive late by
        self.stamp_offset = stamp_offset        # this much / carry the stamp of an older round (stale)
        self.rounds = max(1, -(-self.R // self.world))      # gradient contributions of a rank per aggregated update
        self.general_on_device = False          # R != workers handled by the aggregation kernel, no host in the loop
        self._rs_out = torch.empty(state.layout.shard_size, dtype=torch.float32, device=state.device) \
            if self.use_coll else None
        self._acc = None
        self.updates_applied = 0
        self.dropped = 0                        # gradients of THIS rank that were discarded as late
        if self.world == 1:
            # one rank: R gradients can only come from R successive batches on the same parameters
            self.fast = self.R == 1 and self.fast
        self.bucketed = None
        if self.collective == "bucketed" and not (self.fast and not program.fused):
            self.collective = "reduce_scatter"      # the multi-round path aggregates whole gradients
        if program.gs_inc != (1 if self.fast else 0):
            raise ValueError("program.gs_inc must be 1 on the fast path and 0 when the engine counts updates itself")
        if program.fused and not (self.fast and self.world == 1):
            raise ValueError("a program with the update fused into its epilogue only fits one rank with R == 1")
        self.native_comm = None
        self.p2p = None                         # peer-mapped one-shot all-reduce, else None
        self.p2p_fused = False                  # ... done by the backward launch itself (no update launch)
        self._update_schedule = None
        if not self.fast:
            self._try_install_general_p2p(group)
        if self.fast and not program.fused:
            program.after_compute = self._aggregate_and_apply if self.use_coll else (lambda: state.apply_shard(None, 1.0))
            # one rank: the optimizer of layer l beside the products of the layers below.  Opt-in: measured 2.83 ms
            # against 2.73 ms per wide-MLP step, the 8-wave GEMM leaves no registers for a second kernel's waves
            overlap_local = (not self.use_coll and state.device.type == "cuda" and hasattr(program, "grad_groups")
                             and state.nbytes() >= BUCKETED_ABOVE_BYTES and os.environ.get("DTF_OVERLAP_UPDATE", "0") == "1")
            fold = (not self.use_coll and self.world == 1 and state.device.type == "cuda"
                    and hasattr(program, "enable_folded_optimizer") and getattr(program, "cdtype", None) == torch.bfloat16
                    and os.environ.get("DTF_FOLD_OPTIMIZER", "1") != "0" and not overlap_local)
            if fold:
                # one worker: the big weight-gradient products update their weights in the epilogue; the rest here
                rest = program.enable_folded_optimizer()
                self.folded_ranges = rest
                program.after_compute = lambda: state.apply_ranges(rest)
            elif self.collective == "bucketed" or overlap_local:
                self._install_bucketed(group)
            elif state.device.type == "cuda":
                self._install_native_update(group)

    def _install_bucketed(self, group) -> None:
        """Per-bucket reduce-scatter -> optimizer -> all-gather on a side stream, started by the program as each
        bucket's gradient kernels are enqueued.  On RCCL groups the collectives are issued from the native
        communicator (C++ on the side stream); otherwise through torch.distributed."""
        from . import native_comm as nc
        from .bucketed import BucketedUpdate
        st, prog = self.state, self.program
        native = None
        if (st.device.type == "cuda" and self.use_coll and self.coll is not None and nc.enabled() and self.coll.backend == "nccl"
                and group is None):
            try:
                native = self.native_comm = nc.NativeComm(self.coll.rank, self.coll.world, st.device)
            except (RuntimeError, OSError) as e:
                import warnings
                warnings.warn(f"native RCCL communicator unavailable ({e}); using torch.distributed collectives")
        self.bucketed = BucketedUpdate(st, prog.grad_groups(), self.coll, 1.0 / self.R, native)
        prog.grad_ready = self.bucketed.ready
        prog.after_compute = self.bucketed.finish
        if hasattr(prog, "use_side_refresh"):
            prog.use_side_refresh(self.bucketed.run_behind)     # compute copies of the weights rebuilt behind each bucket

    def _install_native_update(self, group) -> None:
        """GPU fast path: aggregation + optimizer + redistribution become ONE entry of the native step
        schedule (RCCL called from C++ on the compute stream).  Falls back to the torch.distributed
        collectives above if the native communicator cannot be created."""
        from . import native_comm as nc
        from ..ops import _lib
        st, prog = self.state, self.program
        if not nc.enabled():
            return
        if self.use_coll and self._try_install_p2p(group):
            return
        if self.use_coll:
            if self.coll.backend != "nccl" or group is not None:
                return                      # gloo rehearsals and sub-groups keep the torch path
            try:
                self.native_comm = nc.NativeComm(self.coll.rank, self.coll.world, st.device)
            except (RuntimeError, OSError) as e:
                import warnings
                warnings.warn(f"native RCCL communicator unavailable ({e}); using torch.distributed collectives")
                return
            mode = nc.UPDATE_ALLREDUCE if self.collective == "all_reduce" else nc.UPDATE_SHARDED
        else:
            mode = nc.UPDATE_LOCAL
        self._update_args = nc.make_update_args(st, mode, 1.0 / self.R, self.native_comm, self._rs_out)
        entry = (nc.OP_UPDATE, self._update_args, 0)
        if hasattr(prog, "schedule_tail_entries"):
            prog.schedule_tail_entries.append(entry)      # same native call as the compute kernels
            prog._schedules.clear()
            prog.after_compute = None
        else:
            self._update_schedule = _lib.Schedule([entry])
            prog.after_compute = self._update_schedule.run

    def _try_install_p2p(self, group) -> bool:
        """Single-node small-model path (DTF_P2P_ALLREDUCE=0 disables): gradients are written into peer-mapped
        slots and one kernel per step reads every peer's slot and applies the averaged SGD update.  Every rank takes the same decision: the
        conditions are the same everywhere and the self-check ends in a MIN all-reduce."""
        from . import p2p_allreduce as p2p
        import socket
        st, prog = self.state, self.program
        # "auto": real multi-rank RCCL groups only; "1": also one-rank groups and gloo rehearsals
        wanted = (p2p.requested("") or (p2p.requested(self.coll.backend) and self.world > 1)
                  or (self._p2p_pinned and os.environ.get("DTF_P2P_ALLREDUCE", "auto") != "0"))
        eligible = (self._want_p2p is not None and st.optimizer in ("sgd", "adam") and self.world <= 8
                    and getattr(prog, "fused2", False) and not prog.fused and prog.P == p2p.SLOTS)
        fused = os.environ.get("DTF_P2P_FUSED", "1") != "0" and not (os.environ.get("DTF_SHARE_GPU") == "1" and self.world > 3)
        # Adam exists on the sharded exchange inside the backward launch only (the owner of a tile keeps its moments)
        if not (wanted and eligible) or (st.optimizer == "adam" and not fused):
            return False
        if not self._open_p2p(group):
            return False
        self.p2p_fused = os.environ.get("DTF_P2P_FUSED", "1") != "0"
        self.p2p_mode = "sharded" if st.optimizer == "adam" else self._want_p2p
        if self.p2p_mode == "pipelined" and not self._open_pipe(group):
            # (the same answer on every rank) shape, optimizer, data set or mapping rule the pipelined step out
            self.p2p_mode = "sharded" if self.world >= 4 else "replicated"
        self._adam = None
        if st.optimizer == "adam":
            # ps-resident slot variables: flat moment buffers on every rank, of which a rank only ever touches the
            # tiles it owns; `loaded` = what they held when last published (zeros, or a restored checkpoint)
            z = lambda: torch.zeros(st.layout.total, dtype=torch.float32, device=st.device)
            self._adam = {"m": z(), "v": z(), "loaded_m": z(), "loaded_v": z(),
                          "t_offset": int(st.state[2].item()) - int(st.state[1].item())}
        # replicas are compared after the first run() (a warm-up, as a rule) and then every so many run() calls
        self._replica_check_every = max(1, int(os.environ.get("DTF_REPLICA_CHECK_EVERY", "256")))
        self._runs_since_check = 0
        self._replica_checks_left = 1 if self.world > 1 else 0
        if os.environ.get("DTF_SHARE_GPU") == "1" and self.world > 3:
            # rehearsal with several ranks on ONE GPU: the waiting backward launches of the other ranks (108 workgroups
            # each) would occupy every CU, and a forward workgroup (1024 threads) needs a whole one - a deadlock that
            # cannot occur with one rank per GPU.  The separate update launch waits with fewer workgroups.
            self.p2p_fused = False
        self._wire_p2p()
        return True

    def _open_p2p(self, group) -> bool:
        """Map the peers' regions (collective; every rank gets the same answer): one node, self-check passed."""
        from . import p2p_allreduce as p2p
        import socket
        import warnings
        st = self.state
        hosts = [None] * self.world
        dist.all_gather_object(hosts, socket.gethostname(), group=group)
        if len(set(hosts)) != 1:
            return False                            # peer mapping needs every rank on one node
        n = st.grads.numel()
        if not p2p.P2PAllReduce.self_check(n, st.device, group):
            warnings.warn("peer-mapped all-reduce failed its self-check on this machine; using RCCL")
            return False
        try:
            self.p2p = p2p.P2PAllReduce(n, st.device, group)
        except RuntimeError as e:                   # collective failure: raised on every rank
            warnings.warn(f"peer-mapped all-reduce unavailable ({e}); using RCCL")
            return False
        return True

    def _try_install_general_p2p(self, group) -> bool:
        """replicas_to_aggregate != workers without the host: one aggregation kernel per gradient contribution behind
        the backward launch (stamps, first-R selection by the chief, accepted-rank mask, multi-round accumulation:
        ``sync_general_kernel`` in ``csrc/kernels/p2p_allreduce.hip``).  Needs the peer-mapped regions (one node, GPUs,
        SGD, the two-launch step); otherwise the host path below remains."""
        from . import native_comm as nc
        from . import p2p_allreduce as p2p
        st, prog = self.state, self.program
        if not (self.coll is not None or dist.is_initialized()):
            return False
        if self.coll is None:
            self.coll = Collectives(group)
        wanted = p2p.requested("") or (p2p.requested(self.coll.backend) and self.world > 1)
        eligible = (st.device.type == "cuda" and st.optimizer == "sgd" and self.world <= 8 and nc.enabled()
                    and getattr(prog, "fused2", False) and not prog.fused and prog.P == p2p.SLOTS and prog.gs_inc == 0)
        if not (wanted and eligible) or not self._open_p2p(group):
            return False
        W, R = self.world, self.R
        self._acc = torch.zeros_like(st.grads)
        prog.use_gradient_slots(self.p2p.slots)
        prog.phase_period = self.rounds
        seq = st.state[1:2]                         # local step after the tail's increment = number of the round
        self._general_args = {}
        for q in range(prog.P):
            for ph in range(self.rounds):
                self._general_args[(q, ph)] = self.p2p.general_args(
                    q, st.params, self._acc, st.state, seq, st.lr / R, need=min(W, R - ph * W), first_round=ph == 0,
                    last_round=ph == self.rounds - 1, wlog=getattr(prog, "wlog", None), delay_s=self.device_delay_s,
                    stamp_offset=self.stamp_offset)
        prog.schedule_tail_fn = lambda q, ph: [(p2p.OP_P2P_SYNC_GENERAL, self._general_args[(q, ph)], 0)]
        prog._schedules.clear()
        prog._graphs.clear()
        prog.after_compute = None
        self.general_on_device = True
        self.p2p_mode = "general"
        return True

    def _open_pipe(self, group) -> bool:
        """Exchange regions and data-set knowledge of the N-worker pipelined step (collective; every rank gets the same
        answer): the update must be the mean of one gradient per worker with SGD, the program's shape in range, every
        rank on one node with the SAME data set (every worker holds its own full copy, /root/reference/distributed.py:38)
        - a rank gathers its peers' batches itself, from their shuffle seeds."""
        from .pipe_exchange import PipeExchange, PipeExchangeUnavailable
        import warnings
        if self.pipe_x is not None:
            return True
        if self._pipe_refused:
            return False
        st, prog = self.state, self.program
        ok = (self.fast and st.optimizer == "sgd" and hasattr(prog, "pipe_exchange_eligible")
              and prog.pipe_exchange_eligible(self.world))
        seeds = [prog.data.seed]
        if self.world > 1:
            digest = None
            if ok:
                d = prog.data
                digest = (int(d.n), float(d.images[:: max(1, d.n // 64)].double().sum().item()),
                          float(d.labels[:: max(1, d.n // 64)].double().sum().item()), bool(d.shuffle))
            seen = [None] * self.world
            dist.all_gather_object(seen, (ok, digest, int(prog.data.seed)), group=group)
            ok = all(x[0] for x in seen) and all(x[1] == seen[0][1] for x in seen)
            seeds = [x[2] for x in seen]
        if ok:
            try:
                self.pipe_x = PipeExchange(st.device, self.coll.rank if self.coll is not None else 0, self.world, group)
            except PipeExchangeUnavailable as e:            # raised on every rank together
                warnings.warn(f"pipelined exchange unavailable ({e})")
                ok = False
        if ok:
            prog.data.enable_peer_shuffles(seeds, self.pipe_x.rank)
        self._pipe_refused = not ok
        return ok

    def _wire_p2p(self) -> None:
        from . import p2p_allreduce as p2p
        st, prog = self.state, self.program
        if self.p2p_mode == "pipelined":
            # one launch per step: the row workgroups exchange dz1 and the output layer's shares with every rank and
            # every rank applies the same sums to its own replica (mlp_pipe_multi.hip)
            prog.schedule_tail_fn = None
            prog.use_pipe_exchange(self.pipe_x, st.lr / self.R)
            prog.after_compute = None
            return
        if getattr(prog, "pipe_multi", None) is not None:
            prog.use_two_launch()
        if self.p2p_fused and self.p2p_mode == "sharded":
            # the backward launch sends every tile to its owner rank, which updates it and stores it into every replica
            prog.schedule_tail_fn = None
            adam = None
            if self._adam is not None:
                adam = {"m": self._adam["m"], "v": self._adam["v"], "lr": st.lr, "grad_scale": 1.0 / self.R,
                        "beta1": st.beta1, "beta2": st.beta2, "eps": st.eps, "t_offset": self._adam["t_offset"]}
            self._shard_struct = self.p2p.mlp2_shard(st.params, adam=adam)
            prog.use_sharded_exchange(self._shard_struct, st.lr / self.R)
        elif self.p2p_fused:
            # the backward launch exchanges gradient tiles with the peers and applies the update: two launches per step
            prog.schedule_tail_fn = None
            prog.use_peer_exchange(self.p2p.slots, [self.p2p.mlp2_peer(q) for q in range(prog.P)], st.lr / self.R)
        else:
            prog.use_gradient_slots(self.p2p.slots)
            # the step number the update kernel publishes: local_step after the tail's increment (equal on all ranks)
            seq = st.state[1:2]
            self._p2p_args = [self.p2p.update_args(q, st.params, seq, st.lr / self.R) for q in range(prog.P)]
            prog.schedule_tail_fn = lambda q, phase: [(p2p.OP_P2P_UPDATE, self._p2p_args[q], 0)]
        prog._schedules.clear()
        prog.after_compute = None

    def set_p2p_mode(self, mode: str) -> None:
        """Switch the peer-mapped transport between "replicated" and "sharded" (collective: every rank, between runs,
        with the device idle).  Both work on the same mapped regions; parameters and counters carry over."""
        if self.p2p is None:
            raise RuntimeError("no peer-mapped transport is installed")
        if mode not in ("replicated", "sharded", "pipelined"):
            raise ValueError("mode must be 'replicated', 'sharded' or 'pipelined'")
        if mode != "sharded" and getattr(self, "_adam", None) is not None:
            raise ValueError("Adam runs on the sharded exchange only (the owner of a tile keeps its moments)")
        if mode == "pipelined" and not self._open_pipe(self.coll.group if self.coll is not None else None):
            raise ValueError("the N-worker pipelined step does not apply to this job (shape, optimizer, R != workers, "
                             "different data sets on the ranks, or regions that cannot be mapped)")
        self.p2p_mode = mode
        self._wire_p2p()

    def close_transports(self) -> None:
        """Unmap and release the peer-mapped regions (collective when there are peers: nobody may still be writing)."""
        if self.p2p is None and self.pipe_x is None:
            return
        torch.cuda.synchronize(self.state.device)
        if self.world > 1:
            if self.coll.backend == "nccl":
                dist.barrier(group=self.coll.group, device_ids=[self.state.device.index])
            else:
                dist.barrier(group=self.coll.group)
        if self.pipe_x is not None:
            self.pipe_x.close()
            self.pipe_x = None
        if self.p2p is not None:
            self.p2p.close()

    def p2p_candidates(self):
        """Peer-mapped transports this job could run (collective: may open the pipelined exchange)."""
        if self.p2p is None:
            return []
        if getattr(self, "_adam", None) is not None:
            return ["sharded"]
        out = ["replicated", "sharded"]
        if self._open_pipe(self.coll.group if self.coll is not None else None):
            out.append("pipelined")
        return out

    def _transport_late(self) -> bool:
        """Did a wait for a peer run into its limit (either family of regions)?  Synchronises with the device."""
        late = self.p2p.timed_out() if self.p2p is not None else False
        if self.pipe_x is not None and self.pipe_x.timed_out():
            late = True
        return late

    def select_transport(self, trial_steps: int = 192, graph_steps: int = 64, restore: bool = True):
        """Time every candidate transport on the machine the job runs on (collective) and keep the fastest: which of
        one hop with all the bytes, two hops with a fraction of them, or the one-launch step with its Gram work wins
        depends on link latency against link bandwidth against idle CUs, i.e. on the node.  Each candidate runs
        ``trial_steps`` steps to capture its graphs and ``trial_steps`` timed ones (maximum over the ranks).  With
        ``restore`` parameters and counters are put back afterwards, so a training run starts where it would have.
        A pipelined trial whose wait for a peer runs into its limit is recorded as failed and skipped; the tile
        transports keep their behaviour (the error is raised on every rank).  Returns the record for logs."""
        import time as _time
        st, prog = self.state, self.program
        cands = self.p2p_candidates()
        if len(cands) < 2:
            return None
        keep_p, keep_s = st.params.clone(), st.state.clone()
        keep_applied = self.updates_applied
        nccl = self.world > 1 and self.coll.backend == "nccl"

        def sync_all():
            torch.cuda.synchronize(st.device)
            if self.world > 1:
                dist.barrier(group=self.coll.group, device_ids=[st.device.index]) if nccl else dist.barrier(group=self.coll.group)
            torch.cuda.synchronize(st.device)

        times = {}
        for mode in cands:
            try:
                self.set_p2p_mode(mode)
                self.run(trial_steps, graph_steps)              # captures the graphs the timed trial replays
                best_dt = None
                for _ in range(2):                              # the better of two: a host hiccup does not decide
                    sync_all()
                    t0 = _time.perf_counter()
                    self.run(trial_steps, graph_steps, defer_checks=True)
                    torch.cuda.synchronize(st.device)
                    dt = _time.perf_counter() - t0
                    sync_all()
                    self.run_checks()
                    if self.world > 1:
                        t = torch.tensor([dt], dtype=torch.float64, device=st.device if nccl else "cpu")
                        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.coll.group)
                        dt = float(t.item())
                    best_dt = dt if best_dt is None else min(best_dt, dt)
                dt = best_dt
                times[mode] = dt
            except RuntimeError:
                if mode != "pipelined":
                    raise
                # raised on every rank together (the ranks agree on time-outs): forget the pipelined step for this job
                torch.cuda.synchronize(st.device)
                self.pipe_x.err.zero_()
                times[mode] = float("inf")
        best = min(times, key=times.get)                        # the same on every rank: maxima over the ranks
        self.set_p2p_mode(best)
        if restore:
            torch.cuda.synchronize(st.device)
            st.params.copy_(keep_p)
            st.state.copy_(keep_s)
            st.values_version += 1
            self.updates_applied = keep_applied
            prog._started = False
        rec = {"steps_each": trial_steps, "chosen": best}
        for mode, dt in times.items():
            rec[f"{mode}_us_per_step"] = (dt / trial_steps * 1e6) if dt != float("inf") else None
        return rec

    # ---- Adam on the sharded exchange: moments between the owners' buffers and the checkpointable slices
    def publish_state(self) -> None:
        """After the state was initialised or restored (collective): the owners' moment buffers take the moments the
        ranks' slices hold, and the update count the kernels derive from the step number is re-based."""
        if getattr(self, "_adam", None) is None:
            return
        st, a = self.state, self._adam
        full = st.gather_moments(self.coll)
        for key, src in (("m", full[0]), ("v", full[1])):
            a[key].copy_(src)
            a["loaded_" + key].copy_(src)
        a["t_offset"] = int(st.state[2].item()) - int(st.state[1].item())
        self._wire_p2p()

    def moments_to_state(self) -> None:
        """Before a checkpoint (collective): every element's moments come home from the rank that owns its tile (the
        others still hold what was published, so the sum of the changes is the owner's change) into the slices
        ``ShardedState`` saves, and the update counter follows the step counter."""
        if getattr(self, "_adam", None) is None:
            return
        st, a = self.state, self._adam
        torch.cuda.synchronize(st.device)
        lo, hi = st.shard
        for key, dst in (("m", st.m), ("v", st.v)):
            delta = a[key] - a["loaded_" + key]
            self.coll.all_reduce_sum(delta)
            dst.copy_((a["loaded_" + key] + delta)[lo:hi])
        st.state[2] = int(st.state[1].item()) + a["t_offset"]

    @property
    def transport(self) -> str:
        """What moves the gradients and parameters of a step between the ranks."""
        if self.general_on_device:
            return ("peer-mapped aggregation kernel per contribution (stamped gradients, first-R selection by the chief, "
                    "accepted-rank mask summed in rank order)")
        if not self.use_coll:
            return "none (one worker)"
        if self.p2p is not None and self.p2p_mode == "pipelined":
            return ("peer-mapped pipelined exchange over xGMI (one launch per step: dz1 and the output layer's shares "
                    "-> every rank's boxes, every rank applies the same sums to its replica)")
        if self.p2p is not None:
            where = "inside the backward launch" if self.p2p_fused else "separate update launch"
            if self.p2p_fused and self.p2p_mode == "sharded":
                return "peer-mapped sharded exchange over xGMI (tile -> owner, updated tile -> every replica), " + where
            return "peer-mapped replicated exchange over xGMI (every rank reads every peer's tiles), " + where
        if self.bucketed is not None:
            lib = "RCCL (native communicator)" if self.native_comm is not None else f"torch.distributed {self.coll.backend}"
            return f"{lib} reduce-scatter / all-gather per bucket behind the backward pass"
        lib = "RCCL (native communicator)" if self.native_comm is not None else f"torch.distributed {self.coll.backend}"
        return f"{lib} " + ("all-reduce" if self.collective == "all_reduce" else "reduce-scatter + all-gather")

    def _check_replicas(self, late: bool = False) -> None:
        """After the first runs over peer-mapped buffers: every rank must hold bit-identical parameters (the kernels
        add the ranks' gradients in one order everywhere).  A difference means the machine orders the peer traffic
        differently from what the in-launch exchange assumes: the ranks then take rank 0's parameters and continue
        with the separate update launch (acquire/release at system scope); a second difference is an error."""
        if self.p2p is None or self._replica_checks_left <= 0:
            return
        self._replica_checks_left -= 1
        p = self.state.params
        digest = (bool(late), float(p.double().sum().item()), float(p.double().square().sum().item()))
        seen = [None] * self.world
        dist.all_gather_object(seen, digest, group=self.coll.group)
        if any(d[0] for d in seen):                  # raised on every rank together
            raise RuntimeError("peer-mapped all-reduce: a peer's gradients did not arrive within the time limit "
                               f"(ranks {[r for r, d in enumerate(seen) if d[0]]})")
        if all(d == seen[0] for d in seen):
            return
        import warnings
        if self.p2p_mode == "pipelined":
            warnings.warn("the pipelined exchange left the replicas different on this machine; continuing from rank 0's "
                          "parameters with the separate update launch")
            self.p2p_mode, self.p2p_fused, self._pipe_refused = "replicated", False, True
            self.coll.broadcast(p, src=0)
            self.state.values_version += 1
            self._wire_p2p()
            self._replica_checks_left = 1
            return
        if not self.p2p_fused:
            raise RuntimeError(f"peer-mapped all-reduce: replicas differ after the update ({seen})")
        warnings.warn("peer exchange inside the backward launch left the replicas different on this machine; "
                      "continuing from rank 0's parameters with the separate update launch")
        self.p2p_fused = False
        self.coll.broadcast(p, src=0)
        self._wire_p2p()
        self._replica_checks_left = 1


    # ------------------------------------------------------------------ fast path (R == num_workers)
    def _aggregate_and_apply(self) -> None:
        st = self.state
        scale = 1.0 / self.R
        if self.collective == "all_reduce":
            self.coll.all_reduce_sum(st.grads)
            st.apply_full(scale)
        else:
            self.coll.reduce_scatter_sum(self._rs_out, st.grads)
            st.apply_shard(self._rs_out, scale)
            self.coll.all_gather(st.params, st.param_shard)

    def native_launch_ok(self) -> bool:
        """One worker, update inside the step's own launches: a run can be enqueued by one native call."""
        return bool(self.fast and self.world == 1 and not self.use_coll and self.p2p is None
                    and self.program.native_launch_ok())

    def run_native(self, n_steps: int) -> None:
        """``run`` for a short run of one worker without hipGraphs: ONE native call enqueues every kernel of the
        ``n_steps`` steps.  The first kernel starts ~5 us after the call where a graph's first launch needs ~16 us;
        beyond a hundred steps the two are equal (``profiles/r2_mlp_step/short_run_probe_*``)."""
        if not self.native_launch_ok():
            raise RuntimeError("native launch: one worker with the update inside the step's launches only")
        self.program.run_native(n_steps)
        self.updates_applied += n_steps

    def run(self, n_steps: int, graph_steps: int = 64, defer_checks: bool = False) -> None:
        """Run ``n_steps`` aggregated updates (each rank contributes as the R/num_workers ratio dictates).

        After a run over the peer-mapped transports the ranks agree on whether any wait ran into its limit (a flag
        read back from the device plus a 4-byte all-reduce, i.e. a device synchronisation and a collective per call).
        ``defer_checks=True`` enqueues the steps only and leaves that to :meth:`run_checks`, which the caller then
        owes before it uses the result (a benchmark calls it after its clock has stopped)."""
        if defer_checks and (self.fast or self.general_on_device) and self.p2p is not None:
            self.program.run(n_steps * (1 if self.fast else self.rounds), graph_steps)
            self.updates_applied += n_steps
            self._checks_owed = True
            return
        self._checks_owed = False
        if self.fast:
            # the peer-mapped update is an ordinary kernel: steps that use it replay from hipGraphs like the
            # one-rank step; library collectives stay out of captures unless asked for
            if self.use_coll and not self.capture_collectives and self.p2p is None:
                self.program.use_graphs = False
            if self.use_coll and self.program.use_graphs:
                try:
                    self.program.run(n_steps, graph_steps)
                except RuntimeError:
                    # collective capture not available in this build: same step, launched eagerly
                    self.program.use_graphs = False
                    self.program._graphs.clear()
                    torch.cuda.synchronize()
                    self.program.run(n_steps, graph_steps)
            else:
                self.program.run(n_steps, graph_steps)
            self.updates_applied += n_steps
            if self.p2p is not None:
                late = self._transport_late()
                self._runs_since_check += 1
                if self.world > 1 and self._runs_since_check >= self._replica_check_every:
                    self._replica_checks_left = max(self._replica_checks_left, 1)      # periodic: same call count everywhere
                if self._replica_checks_left > 0:
                    self._runs_since_check = 0
                    self._check_replicas(late)      # collective: every rank learns of any rank's time-out
                else:
                    if self.world > 1:              # one decision for all ranks (a 4-byte all-reduce per run() call)
                        t = torch.tensor([1 if late else 0], dtype=torch.int32,
                                         device=self.state.device if self.coll.backend == "nccl" else "cpu")
                        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.coll.group)
                        late = bool(t.item())
                    if late:
                        raise RuntimeError("peer-mapped all-reduce: a peer's gradients did not arrive within the time limit")
            return
        if self.general_on_device:
            # `rounds` program steps per update, every one of them compute launches + the aggregation kernel
            self.program.run(n_steps * self.rounds, graph_steps)
            self.updates_applied += n_steps
            late = self.p2p.timed_out()
            if self.world > 1:                      # one decision for all ranks
                t = torch.tensor([1 if late else 0], dtype=torch.int32,
                                 device=self.state.device if self.coll.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.coll.group)
                late = bool(t.item())
            if late:
                raise RuntimeError("sync replicas: a wait for the chief's decision or for a peer ran into its time limit")
            return
        for _ in range(n_steps):
            self._general_update()

    def run_checks(self) -> None:
        """The agreement of the ranks that ``run(..., defer_checks=True)`` left out (collective; no-op otherwise)."""
        if not getattr(self, "_checks_owed", False):
            return
        self._checks_owed = False
        late = self._transport_late()
        if self.fast:
            self._runs_since_check += 1
            if self.world > 1 and self._runs_since_check >= self._replica_check_every:
                self._replica_checks_left = max(self._replica_checks_left, 1)
            if self._replica_checks_left > 0:
                self._runs_since_check = 0
                self._check_replicas(late)
                return
        if self.world > 1:
            t = torch.tensor([1 if late else 0], dtype=torch.int32,
                             device=self.state.device if self.coll.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.coll.group)
            late = bool(t.item())
        if late:
            raise RuntimeError("peer-mapped transport: a wait for a peer ran into its time limit")

    @property
    def dropped(self) -> int:
        """Gradients of THIS rank that were computed but not aggregated (late or stale)."""
        if self.general_on_device:
            return int(self.state.state[3].item())
        return self._dropped

    @dropped.setter
    def dropped(self, v: int) -> None:
        self._dropped = v

    # ------------------------------------------------------------------ general path (R != num_workers)
    def _contribute(self):
        """One gradient computation of this rank; returns its completion time."""
        prog = self.program
        saved, prog.use_graphs = prog.use_graphs, False
        prog.run(1)
        prog.use_graphs = saved
        if self.state.device.type == "cuda":
            torch.cuda.synchronize()
        if self.arrival_delay is not None:
            time.sleep(self.arrival_delay(self.updates_applied))
        return time.time()

    def _select_earliest(self, t_done: float, k: int) -> bool:
        """True if this rank's gradient is among the k that arrived first (ties: lower rank)."""
        if k >= self.world:
            return True
        mine = torch.tensor([t_done], dtype=torch.float64)
        if self.coll.backend == "nccl":
            mine = mine.to(self.state.device)
        allt = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allt, mine, group=self.coll.group)
        times = [float(t.item()) for t in allt]
        order = sorted(range(self.world), key=lambda r: (times[r], r))
        return self.coll.rank in order[:k]

    def _general_update(self) -> None:
        st, W, R = self.state, self.world, self.R
        if self._acc is None:
            self._acc = torch.zeros_like(st.grads)
        rounds = (R + W - 1) // W
        self._acc.zero_()
        for rnd in range(rounds):
            need = min(W, R - rnd * W)              # contributions taken in this round
            t_done = self._contribute()
            take = self._select_earliest(t_done, need) if W > 1 else True
            if take:
                ops.accumulate(self._acc, st.grads, 1.0, 1.0)
            else:
                self.dropped += 1
        scale = 1.0 / R
        if W == 1:
            if st.optimizer == "sgd":
                ops.sgd_apply(st.params, self._acc, st.lr, scale)
            else:
                st.apply_shard(self._acc[st.shard[0]:st.shard[1]], scale)
        elif self.collective == "all_reduce":
            self.coll.all_reduce_sum(self._acc)
            ops.sgd_apply(st.params, self._acc, st.lr, scale)
        else:
            self.coll.reduce_scatter_sum(self._rs_out, self._acc)
            st.apply_shard(self._rs_out, scale)
            self.coll.all_gather(st.params, st.param_shard)
        st.state[0:1].add_(1)                       # one global_step increment per aggregated update
        self.updates_applied += 1

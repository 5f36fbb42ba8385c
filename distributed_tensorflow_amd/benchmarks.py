"""Benchmark driver behind ``bench.py`` (the JSON contract is documented there)."""
from __future__ import annotations

import os
import sys
import time

import torch
import torch.distributed as dist

from . import ops
from .data import mnist
from .data.device_dataset import DeviceDataset
from .models.mlp import MLPProgram, build_mlp_state
from .models.wide_mlp import WideMLPProgram
from .parallel import comm
from .parallel.sync_replicas import SyncReplicasEngine

METRIC = "examples/sec (whole node)"


def _setup_distributed(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.cpu or not torch.cuda.is_available():
        device = torch.device("cpu")
    else:
        if os.environ.get("DTF_SHARE_GPU") == "1":      # rehearsal of an N-rank launch on a box with fewer GPUs
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if world > 1:
        comm.init_from_env(device)
    return rank, world, device


def _sync_all(device, world):
    if device.type == "cuda":
        torch.cuda.synchronize()
    if world > 1:
        if device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize()


def _stop_clock(t0: float, device, world) -> float:
    """Closing bracket of a timed region: wait for this rank's device, take the time, THEN meet the other ranks (the
    result is the maximum over the ranks, so the rendezvous itself - an all-reduce plus its launch, tens of
    microseconds - does not belong into a run of a few hundred)."""
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _sync_all(device, world)
    return _max_over_ranks(dt, device, world)


def _max_over_ranks(value: float, device, world) -> float:
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64,
                     device=device if (device.type == "cuda" and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _rccl_probe(engine, rank, world, device):
    """How many ranks the native RCCL communicator of this job really has: its own count and an all-reduce of
    ones through it (None off RCCL).  The communicator is the engine's when it made one, else a fresh one."""
    if device.type != "cuda" or world < 1 or not dist.is_initialized() or dist.get_backend() != "nccl":
        return None
    from .parallel import native_comm as nc
    if not nc.enabled():
        return None
    comm, own = getattr(engine, "native_comm", None), False
    try:
        if comm is None:
            comm, own = nc.NativeComm(rank, world, device), True
        ones = torch.ones(4, dtype=torch.float32, device=device)
        comm.all_reduce_sum(ones)
        torch.cuda.synchronize()
        out = {"rccl_ranks": comm.count(), "rccl_allreduce_of_ones": float(ones[0].item())}
        if own:
            comm.destroy()
        return out
    except (RuntimeError, OSError) as e:
        sys.stderr.write(f"[bench] rank {rank}: RCCL probe failed ({e})\n")
        return None


def _assert_rccl_world(engine, rank, world, device):
    """N > 1 on RCCL: the communicator must really span N ranks - its own count and an all-reduce of ones both say N -
    or the run fails loudly (a degenerate communicator would otherwise time N independent one-rank jobs and report
    their sum as a scaling result).  Returns the probe record for the result line (None off RCCL)."""
    if world <= 1 or device.type != "cuda" or not dist.is_initialized() or dist.get_backend() != "nccl":
        return None
    probe = _rccl_probe(engine, rank, world, device)
    if probe is None:
        # no native communicator in this build / disabled: the same two facts through torch.distributed (also RCCL)
        ones = torch.ones(4, dtype=torch.float32, device=device)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        probe = {"rccl_ranks": dist.get_world_size(), "rccl_allreduce_of_ones": float(ones[0].item())}
    if probe["rccl_ranks"] != world or probe["rccl_allreduce_of_ones"] != float(world):
        raise SystemExit(f"[bench] rank {rank}: the RCCL communicator does not span the job: {probe} with --gpus {world}")
    return probe


def _steps_to_target(engine, prog, state, batch, per_update, args, rank, world, device, is_async=False):
    """Second half of the headline metric: global steps (aggregated updates in sync mode, all workers' updates in
    async mode, /root/reference/distributed.py:155) until the trailing-10 mean of the training loss that the loop
    prints every step (:152-153) is at or below ``--target_loss`` on EVERY worker.  ``engine`` / ``prog`` / ``state``
    are fresh (same seed as the timed job, nothing trained yet); runs after the timed segment, outside every clock."""
    target, limit, chunk = float(args.target_loss), int(args.target_max_steps), 16
    cpu_flags = world > 1 and dist.get_backend() != "nccl"
    losses, local_steps, hit = [], 0, None
    first = prog.steps_enqueued
    while local_steps < limit:
        engine.run(chunk, chunk)
        recs = prog.read_log(first + local_steps, chunk)
        local_steps += chunk
        losses += [r.loss for r in recs]
        # first step at which the trailing-10 mean was at or below the target, on this worker
        mine = None
        for i in range(max(10, local_steps - chunk), local_steps + 1):
            if i >= 10 and sum(losses[i - 10:i]) / 10.0 <= target:
                mine = i
                break
        above = 0 if mine is not None else 1
        if world > 1:
            t = torch.tensor([above], dtype=torch.int32, device="cpu" if cpu_flags else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)        # stop together, when every worker is below the target
            above = int(t.item())
        if not above:
            hit = mine
            if world > 1:                                    # the slowest worker's step speaks for the job
                t = torch.tensor([hit if hit is not None else local_steps], dtype=torch.int32,
                                 device="cpu" if cpu_flags else device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                hit = int(t.item())
            break
    if is_async and world > 1:
        engine.sync_params()
    out = {"target_loss": target, "lr": state.lr, "reached": hit is not None,
           "local_steps_per_worker": hit if hit is not None else local_steps,
           "loss_at_stop": sum(losses[-10:]) / max(1, min(10, len(losses)))}
    if hit is not None:
        # sync: one global step per aggregated update = per local step of every worker; async: every worker's
        # local step is an update
        out["steps_to_target"] = hit * (world if is_async else 1)
        out["examples_to_target"] = hit * batch * (world if is_async else per_update)
    else:
        out["steps_to_target"] = out["examples_to_target"] = None
    return out


class _FollowUpGuard:
    """Time limit over the untimed follow-up segments of a multi-worker run (steps to the target loss from a fresh
    state, the same job in the other update mode).  The headline number exists when they start; if they do not come
    back within the limit (a peer that stopped answering leaves the others in a collective), rank 0 writes the result
    line without them and every rank leaves with status 0, instead of the run ending with no line at all.  Every rank
    arms its timer right after the barrier that closes the timed segment, so they fire together.
    ``DTF_BENCH_FOLLOWUP_LIMIT_S`` (default 300; 0: no limit)."""

    def __init__(self, args, rank: int, world: int, core: dict):
        import threading
        self.limit = float(os.environ.get("DTF_BENCH_FOLLOWUP_LIMIT_S", "300"))
        self.emit = getattr(args, "_emit", None)
        self.rank, self.core = rank, core
        self.timer = None
        if world > 1 and self.limit > 0 and self.emit is not None:
            self.timer = threading.Timer(self.limit, self._fire)
            self.timer.daemon = True

    def _fire(self, why=None):
        why = why or f"not finished within {self.limit:.0f} s"
        sys.stderr.write(f"[bench] rank {self.rank}: follow-up segments {why}; reporting the timed segment alone\n")
        sys.stderr.flush()
        if self.rank == 0:
            self.core["follow_up_segments"] = f"{why} (steps_to_target / async_value missing)"
            self.emit(self.core)
        os._exit(0)

    def __enter__(self):
        if self.timer is not None:
            self.timer.start()
        return self

    def __exit__(self, etype, evalue, _tb):
        if self.timer is not None:
            self.timer.cancel()
            if etype is not None and issubclass(etype, Exception):
                # this rank's follow-up failed: the others are left in a collective and leave by their own timers
                self._fire(f"failed on rank {self.rank} ({etype.__name__}: {evalue})")
        return False


def _time_segment(engine, device, world, steps, graph_steps):
    _sync_all(device, world)
    t0 = time.perf_counter()
    engine.run(steps, graph_steps, defer_checks=True)
    dt = _stop_clock(t0, device, world)
    engine.run_checks()
    return dt


def _select_p2p_transport(engine, prog, args, rank, world, device, graph_steps):
    """The peer-mapped transports move the same update over mapped regions; which one is faster depends on the
    machine (link latency against link bandwidth against idle CUs), so it is measured here
    (``SyncReplicasEngine.select_transport``): a short timed run of each, maximum over the ranks, the fastest stays.
    Parameters and counters are put back, so the warm-up and the timed run start from the initial state.  Returns the
    record that goes into the result line."""
    return engine.select_transport(192, graph_steps, restore=True)


def bench_mlp_sync(args, rank, world, device, wide: bool = False):
    batch = args.batch_size or (100 if not wide else 4096)
    hidden_units = args.hidden_units or (100 if not wide else 4096)
    hidden = [hidden_units] * (7 if wide else 1)
    optimizer = args.optimizer or ("adam" if wide else "sgd")
    lr = 0.01 if optimizer == "sgd" else 1e-3
    train_rows = 55000 if not wide else max(8 * batch, 16384)
    # every worker holds its own full copy of the SAME data set and draws its own batches from it (independent
    # shuffles, no sharding by task: /root/reference/distributed.py:38, :137)
    ds = mnist.synthetic_data_sets(train_size=train_rows, validation_size=100, test_size=100, seed=0 if not wide else rank)
    force = bool(getattr(args, "force_collectives", False)) and world == 1 and device.type == "cuda"
    if force and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    R = getattr(args, "replicas_to_aggregate", None) or world
    fast = R == world
    fused = world == 1 and optimizer == "sgd" and not wide and not force and fast
    dtype_name = args.dtype or "fp32"
    graph_steps = args.graph_steps or 64

    def build():
        state = build_mlp_state(784, hidden, 10, device, world_size=world, rank=rank, optimizer=optimizer, lr=lr)
        state.initialize(seed=1234)                      # same seed everywhere == chief init + broadcast
        data = DeviceDataset(ds.train, device, shuffle_seed=1000 + rank)
        if wide:
            cdtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
            prog = WideMLPProgram(state, data, batch, hidden, loss_mode=ops.LOSS_REFERENCE, dtype=cdtype, gs_inc=1)
            args.no_post_update_accuracy = True
        else:
            prog = MLPProgram(state, data, batch, hidden, loss_mode=ops.LOSS_REFERENCE, fused_update=fused,
                              post_update_accuracy=not args.no_post_update_accuracy, gs_inc=1 if fast else 0)
        engine = SyncReplicasEngine(prog, state, collective=args.collective, force_collectives=force,
                                    capture_collectives=False if force else None,
                                    replicas_to_aggregate=getattr(args, "replicas_to_aggregate", None))
        return state, prog, engine

    state, prog, engine = build()
    prepared = False
    selection = None
    extra_steps = 0                                         # steps of the transport trials (before the warm-up)
    failed = 0
    if (world > 1 and engine.p2p is not None and engine.p2p_fused and fast and args.collective == "auto"
            and os.environ.get("DTF_P2P_MODE") is None):
        try:
            selection = _select_p2p_transport(engine, prog, args, rank, world, device, graph_steps)
        except RuntimeError as e:                           # a peer wait ran into its limit: raised on every rank
            sys.stderr.write(f"[bench] rank {rank}: transport trial failed ({e})\n")
            failed, selection = 1, None
    # --launch native: the run is enqueued by ONE native call instead of a graph replay.  In a loop the two are equal
    # (20 steps: 244 vs 242 us, tools/short_run_probe.py), but in a fresh process, which is what a benchmark run is,
    # the host's first pass through the launch loop is slow and noisy (13.4-18 us/step against 12.7 for the graph's
    # first launch): graphs stay the default.
    want = getattr(args, "launch", "auto")
    native = (world == 1 and not force and device.type == "cuda" and fast and want == "native" and engine.native_launch_ok())
    if want == "native" and not native:
        raise SystemExit("--launch native: needs one GPU worker whose update is inside the step's own launches")
    run = (lambda n: engine.run_native(n)) if native else (lambda n: engine.run(n, graph_steps))
    # timed segments: the ranks' agreement on the transport's health (a device read-back and a small all-reduce per
    # run() call) happens after the clock has stopped
    run_timed = run if native else (lambda n: engine.run(n, graph_steps, defer_checks=True))
    if world == 1 and not force and device.type == "cuda" and prog.use_graphs and args.warmup > 0 and fast and not native:
        # one worker: nothing is decided during the warm-up, so the graphs of the warm-up AND of the timed run are
        # captured before either (capture takes milliseconds of host time during which the GPU would fall idle and
        # drop its clocks right in front of the timed region)
        prog.prepare_graphs(args.warmup, graph_steps)
        prog.prepare_graphs(args.steps, graph_steps, start_offset=args.warmup)
        prepared = True
    _sync_all(device, world)
    # The warm-up doubles as the acceptance run of the peer-mapped transport on this machine: if any rank's wait
    # for a peer runs into its limit, EVERY rank rebuilds the job on RCCL and warms up again (one decision for all)
    try:
        if not failed and args.warmup > 0:
            run(args.warmup)
    except RuntimeError as e:
        if engine.p2p is None or world == 1:
            raise
        failed = 1
        sys.stderr.write(f"[bench] rank {rank}: {e}\n")
    if world > 1 and engine.p2p is not None:
        t = torch.tensor([failed], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t.item()):
            sys.stderr.write(f"[bench] rank {rank}: peer-mapped transport gave up during warm-up; continuing on RCCL\n")
            os.environ["DTF_P2P_ALLREDUCE"] = "0"
            if device.type == "cuda":
                torch.cuda.synchronize()
            state, prog, engine = build()
            extra_steps = 0
            _sync_all(device, world)
            engine.run(args.warmup, graph_steps) if args.warmup > 0 else None
    if not prepared and not native:
        _prepare_graphs(prog, args, graph_steps, device, 1 if engine.fast else engine.rounds)
    args._launch = "native schedule (one call enqueues every kernel of the run)" if native else (
        "hipGraph replay" if prog.use_graphs else "eager")
    _sync_all(device, world)
    t0 = time.perf_counter()
    run_timed(args.steps)
    dt = _stop_clock(t0, device, world)
    engine.run_checks()                                     # raises on every rank if any rank's wait for a peer timed out
    gs = state.global_step()
    assert gs == 1 + extra_steps + args.warmup + args.steps, (gs, extra_steps, args.warmup, args.steps)
    last = prog.read_log(prog.steps_enqueued - 1, 1)[0]
    # examples per aggregated update: R gradients of `batch` examples each (distributed.py:92-100)
    value = batch * args.steps * R / dt
    probe = _assert_rccl_world(engine, rank, world, device)
    transport, dropped = engine.transport, engine.dropped

    def result(second, target):
        return _sync_result(args, rank, world, device, wide, batch, hidden_units, optimizer, dtype_name, R, dt, state, prog,
                            engine, force, last, selection, probe, second, transport, dropped, target)

    target = second = None
    # What follows is outside every clock.  With several workers it runs under a time limit of its own: the timed
    # segment's result is complete here and is what gets reported if a follow-up segment does not come back.
    with _FollowUpGuard(args, rank, world, result(None, None)):
        if not wide and fast and getattr(args, "target_loss", 0):
            # steps to the target loss from a fresh state (same seed), same engine settings
            engine.close_transports()
            state_t, prog_t, engine_t = build()
            if selection is not None and engine_t.p2p is not None:
                engine_t.set_p2p_mode(selection["chosen"])
            target = _steps_to_target(engine_t, prog_t, state_t, batch, R, args, rank, world, device)
            engine_t.close_transports()
        if world > 1 and not wide and fast and not getattr(args, "no_second_mode", False):
            # the other update mode of the same job, timed the same way, so that one invocation yields both numbers
            if target is None:
                engine.close_transports()
            second = bench_mlp_async(_async_args(args), rank, world, device)
    return result(second, target)


def _async_args(args):
    import copy
    a = copy.copy(args)
    a.config, a.optimizer = "mlp_async", None
    return a


def _sync_result(args, rank, world, device, wide, batch, hidden_units, optimizer, dtype_name, R, dt, state, prog, engine, force,
                 last, selection, probe, second, transport, dropped, target=None):
    value = batch * args.steps * R / dt
    out = {
        "metric": METRIC, "value": value, "unit": "examples/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_name if wide else "fp32", "data": "synthetic MNIST-shape (resident in HBM), random-init weights",
        "config": {"model": (f"wide_mlp 784-7x{hidden_units}-10" if wide else f"reference MLP 784-{hidden_units}-10"),
                   "global_batch": batch * world, "per_worker_batch": batch, "seq_len": None,
                   "parallelism": (f"dp{world} sync_replicas R={R}, " +
                                   ("replicated SGD: every rank applies the same sums of the workers' exchanged dz1 / output-layer "
                                    "shares (80 KB per worker and step instead of 318 KB of gradients)"
                                    if engine.p2p_mode == "pipelined" else
                                    f"ps state sharded over {world} GPU(s) by weight-gradient tile (peer-mapped: tile -> owner, "
                                    "updated tile -> every replica)" if engine.p2p_mode == "sharded" and engine.p2p_fused
                                    else "replicated SGD after one all-reduce (stateless optimizer, 318 KB of parameters)"
                                    if (world > 1 or force) and engine.collective == "all_reduce"
                                    else f"ps state sharded over {world} GPU(s)")),
                   "optimizer": optimizer, "loss": "reference (softmax applied twice)", "mode": "sync",
                   "collective": engine.collective if (world > 1 or force) else "none",
                   "transport": transport,
                   "native_rccl": engine.native_comm is not None,
                   "peer_mapped_allreduce": getattr(engine, "p2p", None) is not None,
                   "peer_mapped_mode": engine.p2p_mode,
                   "peer_exchange_inside_backward_launch": bool(getattr(engine, "p2p_fused", False)) and getattr(engine, "p2p", None) is not None,
                   "step_program": ("one launch per step, pipelined over the workers (mlp_pipe_multi.hip)"
                                    if getattr(prog, "pipe_multi", None) is not None
                                    else "one launch per step, pipelined (mlp_pipe.hip)" if getattr(prog, "pipe", False)
                                    else "two launches per step (mlp2_fused.hip)" if getattr(prog, "fused2", False)
                                    else "general launch sequence"),
                   "post_update_accuracy_pass": not args.no_post_update_accuracy,
                   "hip_graph": bool(prog.use_graphs) and not str(getattr(args, "_launch", "")).startswith("native"),
                   "launch": getattr(args, "_launch", None), "params": state.layout.used},
        "final_loss": last.loss,
    }
    if R != world:
        out["config"]["dropped_gradients_this_rank"] = dropped
    if selection is not None:
        out["config"]["transport_selection"] = selection
    if probe is not None:
        out.update(probe)
    if target is not None:
        out["target_loss"] = target["target_loss"]
        out["steps_to_target"] = target["steps_to_target"]
        out["examples_to_target"] = target["examples_to_target"]
        out["steps_to_target_detail"] = target
    if second is not None:
        out["async_value"] = second["value"]
        out["async_ms_per_step"] = second["ms_per_step"]
        out["async_config"] = second["config"]
        for k in ("staleness_max", "staleness_mean", "global_step", "steps_to_target", "examples_to_target",
                  "staleness_bound", "forced_pulls"):
            if k in second:
                out["async_" + k] = second[k]
    return out


def _prepare_graphs(prog, args, graph_steps, device, steps_per_update: int = 1) -> None:
    """Build the hipGraphs of the timed run before the clock starts (after a warm-up run only: by then the engine
    has decided whether this program's steps are replayed at all).  Every timed step still runs inside the timed
    region; what moves out is graph capture and instantiation."""
    if args.warmup <= 0 or device.type != "cuda" or not getattr(prog, "use_graphs", False):
        return
    try:
        prog.prepare_graphs(args.steps * steps_per_update, graph_steps if graph_steps else 64)
    except RuntimeError as e:
        sys.stderr.write(f"[bench] graphs not prepared ahead of the timed run ({e})\n")
        prog._graphs.clear()
        torch.cuda.synchronize()


def _own_wgrad_calls() -> int:
    from .models.resnet import conv1x1_calls
    return conv1x1_calls["own_wgrad"]


def bench_resnet50_sync(args, rank, world, device):
    from .models.resnet import TorchModuleProgram, resnet50
    from .ops.batchnorm import calls as bn_calls
    from .ops.classifier import calls as head_calls
    from .ops.maxpool import calls as pool_calls
    batch = args.batch_size or 256
    optimizer = args.optimizer or "adam"
    dtype_name = args.dtype or "bf16"
    if device.type == "cuda" and os.environ.get("DTF_MIOPEN_FIND", "1") == "1":
        # let MIOpen time its convolution algorithms once per shape (costs ~2 minutes of warm-up on a fresh box,
        # measured 25.4 -> 22.0 ms/step at batch 128)
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    model = resnet50()
    g = torch.Generator().manual_seed(100 + rank)
    n = 2 * batch
    images = torch.rand(n, 3, 224, 224, generator=g)
    labels = torch.randint(0, 1000, (n,), generator=g)
    autocast = torch.bfloat16 if dtype_name == "bf16" and device.type == "cuda" else None
    prog = TorchModuleProgram(model, device, images, labels, batch, world_size=world, rank=rank, optimizer=optimizer,
                              lr=1e-3 if optimizer == "adam" else 0.1, autocast=autocast)
    # --force_collectives on one GPU: the step the workers of a multi-GPU job run - gradients in buckets, per bucket
    # reduce-scatter -> optimizer on the owned slice -> all-gather on a one-rank RCCL communicator
    force = bool(getattr(args, "force_collectives", False)) and world == 1 and device.type == "cuda"
    if force and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29535")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    engine = SyncReplicasEngine(prog, prog.state, collective="bucketed" if force else args.collective, force_collectives=force,
                                capture_collectives=False if force else None)
    _sync_all(device, world)
    engine.run(args.warmup) if args.warmup > 0 else None
    _sync_all(device, world)
    t0 = time.perf_counter()
    engine.run(args.steps)
    dt = _stop_clock(t0, device, world)
    last = prog.read_log(prog.steps_enqueued - 1, 1)[0]
    return {
        "metric": METRIC, "value": batch * args.steps * world / dt, "unit": "examples/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_name if autocast is not None else "fp32",
        "data": "synthetic ImageNet-shape (resident in HBM), random-init weights",
        "config": {"model": "ResNet-50", "global_batch": batch * world, "per_worker_batch": batch, "seq_len": None,
                   "parallelism": f"dp{world} sync_replicas, optimizer state sharded over {world} GPU(s)",
                   "optimizer": optimizer, "mode": "sync", "collective": engine.collective if (world > 1 or force) else "none",
                   "transport": engine.transport, "gradient_buckets": len(prog._groups) if prog._groups else None,
                   "params": prog.state.layout.used, "convolutions": ("MIOpen (library); weight gradients of the stride-1 1x1 layers: own kernel (conv1x1.hip)"
                                   if _own_wgrad_calls() else "MIOpen (library)"),
                   "classifier": ("own kernels (MFMA GEMM + fused softmax cross-entropy)"
                                  if head_calls["kernel"] and not head_calls["operators"]
                                  else "PyTorch operators" if not head_calls["kernel"] else "mixed"),
                   "batch_norm": ("fused kernels (batchnorm.hip)" if bn_calls["kernel"] and not bn_calls["operators"]
                                  else "PyTorch operators" if not bn_calls["kernel"] else "mixed"),
                   "max_pool": "own kernels (maxpool.hip)" if pool_calls["kernel"] else "PyTorch operator",
                   "gradients": (("handed over by autograd, one multi-tensor copy per bucket into the flat buffer on the "
                                  "weight-gradient stream, bf16 weight copies" if prog._groups else
                                  "handed over by autograd, multi-tensor copies into the flat buffer, bf16 weight copies")
                                 if prog._direct_now else "accumulated into views of the flat buffer")},
        "final_loss": last.loss,
    }


def bench_mlp_async(args, rank, world, device, wide: bool = False):
    from .parallel.async_peer import make_async_engine
    batch = args.batch_size or (100 if not wide else 4096)
    hidden_units = args.hidden_units or (100 if not wide else 4096)
    hidden = [hidden_units] * (7 if wide else 1)
    optimizer = args.optimizer or ("adam" if wide else "sgd")
    lr = 0.01 if optimizer == "sgd" else 1e-3
    train_rows = 55000 if not wide else max(8 * batch, 16384)
    dtype_name = (args.dtype or "fp32") if wide else "fp32"
    # the wide model spreads its 416 MB of state over every rank (each owns a slice), the reference model keeps
    # the one logical ps shard of the named configuration
    shards = world if wide else 1
    # reference behaviour: every step reads the ps-resident variables (staleness 0); --staleness k pulls every (k+1)-th step
    staleness = args.staleness if getattr(args, "staleness", None) is not None else 0
    bound = getattr(args, "staleness_bound", None)
    ds = mnist.synthetic_data_sets(train_size=train_rows, validation_size=100, test_size=100, seed=0 if not wide else rank)
    force = bool(getattr(args, "force_collectives", False)) and world == 1 and device.type == "cuda" and not wide

    def make():
        state = build_mlp_state(784, hidden, 10, device, world_size=world, rank=rank, optimizer=optimizer, lr=lr)
        state.initialize(seed=1234)
        data = DeviceDataset(ds.train, device, shuffle_seed=1000 + rank)

        def program(fused, gs_inc):
            if wide:
                cdtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
                args.no_post_update_accuracy = True
                return WideMLPProgram(state, data, batch, hidden, loss_mode=ops.LOSS_REFERENCE, dtype=cdtype, gs_inc=gs_inc)
            return MLPProgram(state, data, batch, hidden, fused_update=fused, gs_inc=gs_inc,
                              post_update_accuracy=not args.no_post_update_accuracy)

        if force:
            # N=1 only: the multi-rank code path (gradients stored, push/pull kernel through the exported region)
            from .parallel.async_peer import PeerMappedAsyncEngine
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29534")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            prog = program(fused=False, gs_inc=0)
            engine = PeerMappedAsyncEngine(prog, state, 1, staleness=staleness, staleness_bound=bound)
            finish = engine.finish
        elif world == 1:
            # one worker and its own ps: every update is applied before the next gradient is computed
            prog = program(fused=optimizer == "sgd", gs_inc=1)
            engine = SyncReplicasEngine(prog, state)
            finish = lambda: None
        else:
            prog = program(fused=False, gs_inc=0)
            engine = make_async_engine(prog, state, shards, staleness, staleness_bound=bound)
            finish = engine.finish
        return state, prog, engine, finish

    state, prog, engine, finish = make()
    _sync_all(device, world)
    engine.run(args.warmup, 64)
    if type(engine).__name__ in ("PeerMappedAsyncEngine", "SyncReplicasEngine"):    # engines whose steps replay from graphs
        _prepare_graphs(prog, args, 64, device)
    _sync_all(device, world)
    t0 = time.perf_counter()
    engine.run(args.steps, 64)
    if world > 1 or force:
        engine.sync_params()
    dt = _stop_clock(t0, device, world)
    stale = None
    if hasattr(engine, "staleness_stats"):
        smax, smean = engine.staleness_stats(args.warmup, args.steps)
        # every rank's view, worst case over the job
        smax = int(_max_over_ranks(float(smax), device, world))
        smean = _max_over_ranks(smean, device, world)
        stale = (smax, smean)
    forced = waits = None
    if hasattr(engine, "forced_pulls"):
        forced = int(_max_over_ranks(float(engine.forced_pulls()), device, world))
        waits = int(_max_over_ranks(float(engine.bound_waits()), device, world))
    finish()
    target = None
    if not wide and getattr(args, "target_loss", 0):
        state_t, prog_t, engine_t, finish_t = make()
        target = _steps_to_target(engine_t, prog_t, state_t, batch, 1, args, rank, world, device, is_async=True)
        finish_t()
    out = {
        "metric": METRIC, "value": batch * args.steps * world / dt, "unit": "examples/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_name,
        "data": "synthetic MNIST-shape (resident in HBM), random-init weights",
        "config": {"model": (f"wide_mlp 784-7x{hidden_units}-10" if wide else f"reference MLP 784-{hidden_units}-10"),
                   "global_batch": batch * world, "per_worker_batch": batch, "seq_len": None,
                   "parallelism": (f"dp{world} async push/pull, {shards} ps shard(s), pull every "
                                   f"{getattr(engine, 'staleness', 0) + 1} step(s)"),
                   "optimizer": optimizer, "mode": "async", "engine": type(engine).__name__,
                   "staleness_configured": getattr(engine, "staleness", 0),
                   "push_pull_inside_backward_launch": bool(getattr(engine, "fused", False)), "params": state.layout.used},
        "global_step": state.global_step() if rank == 0 else None,
    }
    if stale is not None:
        # measured on the device: updates of OTHER workers that the replica a gradient was computed on had not seen
        out["staleness_max"], out["staleness_mean"] = stale
    if getattr(engine, "staleness_bound", None) is not None:
        out["staleness_bound"] = engine.staleness_bound
    if forced is not None:
        out["forced_pulls"] = forced
    if waits is not None:
        out["pushes_that_waited_for_a_slower_worker"] = waits
    if target is not None:
        out["target_loss"] = target["target_loss"]
        out["steps_to_target"] = target["steps_to_target"]
        out["examples_to_target"] = target["examples_to_target"]
        out["steps_to_target_detail"] = target
    return out


def run_benchmark(args):
    rank, world, device = _setup_distributed(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's process count and --gpus must agree")
    if args.config == "mlp_sync":
        res = bench_mlp_sync(args, rank, world, device, wide=False)
    elif args.config == "wide_mlp_sync":
        res = bench_mlp_sync(args, rank, world, device, wide=True)
    elif args.config == "resnet50_sync":
        res = bench_resnet50_sync(args, rank, world, device)
    elif args.config == "mlp_async":
        res = bench_mlp_async(args, rank, world, device)
    elif args.config == "wide_mlp_async":
        res = bench_mlp_async(args, rank, world, device, wide=True)
    else:
        raise SystemExit(f"config {args.config} is not wired into bench.py yet")
    comm.shutdown()                 # also the one-rank group some configurations create
    return res if rank == 0 else None

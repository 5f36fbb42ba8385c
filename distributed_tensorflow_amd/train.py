"""Training driver with the behaviour of /root/reference/distributed.py:37-165.

Same flags, same role handling, same log lines (typos included), same loop
rules: validation accuracy every 10000 local steps including step 0 (:140-143),
per-step line with loss and training accuracy (:152-153), stop once the
``global_step`` fetched with the train op reaches ``--train_steps`` (:155-156),
then elapsed time and test accuracy (:158-165).

What differs is where the work happens: steps are replayed in hipGraph
launches of many steps each, logs come back in batches from a device-side
ring, and the per-step host round trips of the reference (two session runs, two
feeds, one global_step fetch) do not exist.
"""
from __future__ import annotations

import sys
import time
from typing import List, Optional

import torch

from . import flags as flags_lib
from . import ops
from .cluster.server import Server
from .cluster.spec import ClusterSpec, Role, split_hosts, validate_role
from .data import mnist
from .data.device_dataset import DeviceDataset
from .models.mlp import MLPProgram, StepRecord, build_mlp_state
from .parallel.async_peer import make_async_engine
from .utils import trace
from .parallel.async_ps import AsyncParameterServerEngine
from .parallel.sync_replicas import SyncReplicasEngine
from .supervisor.supervisor import Supervisor

FLAGS = flags_lib.FLAGS
IMAGE_PIXELS = 28   # distributed.py:35

_defined = False


def define_flags() -> None:
    global _defined
    if not _defined:
        flags_lib.define_reference_flags()
        flags_lib.define_framework_flags()
        _defined = True


def resolve_device(role: Optional[Role], requested: str) -> torch.device:
    """``allow_soft_placement=True`` (distributed.py:114): use the GPU when there is one, else the CPU."""
    if requested == "cpu":
        return torch.device("cpu")
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        idx = role.local_rank() % n if role is not None and role.is_worker else 0
        return torch.device("cuda", idx)
    if requested == "gpu":
        raise RuntimeError("--device=gpu but no GPU is visible")
    return torch.device("cpu")


def shuffle_seed(seed_flag: int, task_index: int) -> int:
    """Seed of a worker's data shuffle: derived from ``--seed``; with ``--seed=0`` from the clock and the process, like
    the parameter initialisation (the reference sets no seed anywhere: every run and every worker shuffles differently)."""
    import os
    base = seed_flag if seed_flag else ((int(time.time() * 1e6) ^ (os.getpid() << 16)) & 0x3FFFFFFF)
    return (base * 7919 + task_index) & 0x7FFFFFFF


def local_hosts(hosts) -> bool:
    """True when every ``host:port`` entry names this machine (loopback or one of its own addresses)."""
    import socket
    mine = {"localhost", "127.0.0.1", "0.0.0.0", ""}
    try:
        name = socket.gethostname()
        mine.add(name)
        mine.update(socket.gethostbyname_ex(name)[2])
    except OSError:
        pass
    return all(h.rsplit(":", 1)[0] in mine or h.rsplit(":", 1)[0].startswith("127.") for h in hosts)


def _emit(line: str) -> None:
    sys.stdout.write(line + "\n")


class TrainingLoop:
    """The ``while True`` loop of distributed.py:135-156 in launches of many steps."""

    def __init__(self, task_index: int, program: MLPProgram, engine, train_steps: int, validation_every: int,
                 log_every: int, steps_per_launch: int, validation, test, supervisor: Optional[Supervisor],
                 is_async: bool, world: int, emit=_emit):
        self.task, self.prog, self.engine = task_index, program, engine
        self.train_steps, self.validation_every = train_steps, max(1, validation_every)
        self.log_every = max(1, log_every)
        self.steps_per_launch = steps_per_launch
        self.validation, self.test = validation, test
        self.sv = supervisor
        self.is_async, self.world = is_async, world
        self.emit = emit
        self.local_step = 0
        self.records: List[StepRecord] = []      # everything printed, for callers that want the trajectory
        self.keep_records = False
        self.validation_accuracies: List[float] = []
        self.stop_check = None                   # optional callable: True once the engine was told to stop
        # Gradient contributions (= program steps = the reference's local steps) per aggregated update: more than one
        # when replicas_to_aggregate exceeds the number of workers.  engine.run(n) runs n updates.
        self.rounds = 1 if getattr(engine, "fast", True) else max(1, int(getattr(engine, "rounds", 1)))

    def _print_record(self, r: StepRecord) -> None:
        if self.keep_records:
            self.records.append(r)
        if r.local_step % self.log_every == 0:
            self.emit('Worker %d: traing step %d (global step:%d) loss %f training accuracy %g'
                      % (self.task, r.local_step, r.global_step, r.loss, r.train_accuracy))

    def _validate(self) -> None:
        if self.is_async:
            self.engine.sync_params()
        acc = self.prog.accuracy(*self.validation)
        self.validation_accuracies.append(acc)
        self.emit('Worker %d: validation accuracy %g' % (self.task, acc))

    def _plan(self, last_seen_gs: int, local_step: Optional[int] = None) -> int:
        local_step = self.local_step if local_step is None else local_step
        to_validation = self.validation_every - (local_step % self.validation_every)
        if self.is_async:
            # other workers advance global_step too: approach the limit in shrinking launches
            remaining = max(1, (self.train_steps - last_seen_gs + 1) // max(1, self.world))
        else:
            # the fetched value is global_step before the step's own increment
            remaining = max(1, self.train_steps - last_seen_gs + 1)
        cap = self.steps_per_launch if self.steps_per_launch > 0 else 256
        cap = min(cap, self.prog.log_cap - 1, self.prog.data.max_steps_per_launch(self.prog.B))
        # cap and to_validation count local steps, `remaining` counts updates
        return max(1, min(cap // self.rounds, remaining, to_validation // self.rounds))

    def _can_overlap(self) -> bool:
        """Launch ahead of the log processing?  Only where the launches do not depend on what the log says: sync mode
        with every update made of one step per worker, or one worker alone in async mode (global_step then advances
        by exactly one per step), on a GPU, with nothing that may end the training from outside.  Measured: one worker,
        200 000 steps with every step printed, 2.04 -> 1.39 s; two sync workers sharing one GPU, 3000 steps,
        0.39 -> 0.19 s (tools/cli_two_workers.sh), output identical in both cases.  DTF_LOOP_OVERLAP=0: plain loop."""
        import os
        mode = os.environ.get("DTF_LOOP_OVERLAP", "1")
        # ("cpu": the same control flow without a device to overlap with, for tests that have no GPU)
        return (mode != "0" and (self.world == 1 or not self.is_async) and self.stop_check is None
                and self.rounds == 1 and isinstance(self.prog, MLPProgram) and (self.prog.is_cuda or mode == "cpu"))

    def _run_overlapped(self) -> float:
        """``run`` with up to two launches in flight: while the host decodes and prints the records of one launch, the
        device runs the next.  Records are read on a stream of their own behind an event recorded after their launch.
        Order and content of everything printed are those of the plain loop; launches stop at validation boundaries
        and at the last update, where the loop drains."""
        import collections
        prog = self.prog
        time_begin = time.time()
        held: Optional[StepRecord] = None
        gs_seen = prog.state.global_step()
        planned_gs, planned_local = gs_seen, self.local_step
        validated_at = -1
        inflight = collections.deque()           # (first local step, steps, event behind the launch, global_step before it)
        done = False
        launched_any = False                     # the reference checks the stop rule after a step: at least one runs
        while not done:
            while len(inflight) < 2 and (planned_gs <= self.train_steps or not launched_any):
                if planned_local % self.validation_every == 0 and validated_at != planned_local:
                    break                        # validation first: it reads the parameters of exactly this step
                # two launches share the log ring: the one being read and the one running must both fit
                n = min(self._plan(planned_gs, planned_local), max(1, (prog.log_cap - 1) // 2))
                with trace.annotate(f"launch {n} updates"):
                    self.engine.run(n, min(n, 64) if n >= 2 else 1)
                ev = None
                if prog.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                inflight.append((planned_local, n, ev, planned_gs))
                launched_any = True
                planned_local += n
                planned_gs += n
            if not inflight:
                if planned_gs > self.train_steps and launched_any:
                    break
                if held is not None:             # keep the reference's line order: step line, then validation
                    held.post_correct = prog.post_correct_of_last_step()
                    self._print_record(held)
                    held = None
                with trace.annotate("validation"):
                    self._validate()
                validated_at = planned_local
                continue
            first, n, ev, gs_launch = inflight.popleft()
            self.local_step = first + n
            with trace.annotate("read step log"):
                recs = prog.read_log(first, n, after=ev)
            if held is not None:
                held.post_correct = prog.first_post_of_last_read
                self._print_record(held)
                held = None
            for r in recs:
                stop_here = r.global_step >= self.train_steps
                if prog.post and r.post_correct is None:          # only the last record of a launch
                    if not stop_here:
                        held = r                 # resolved by the next launch's first step, or in front of a validation
                        break
                    r.post_correct = prog.post_correct_of_last_step()
                self._print_record(r)
                if stop_here:
                    done = True
                    break
            gs_seen = recs[-1].global_step + 1
            delta = gs_seen - (gs_launch + n)                     # (zero unless something else moved global_step)
            if delta:
                planned_gs += delta
                # the launch still in flight was planned with the uncorrected value: correct it once, here
                inflight = collections.deque((f, k, e, g + delta) for f, k, e, g in inflight)
            if self.sv is not None:
                with trace.annotate("checkpoint check"):
                    self.sv.maybe_checkpoint()
                self.sv.report_step_rate(gs_seen)
        time_end = time.time()
        return time_end - time_begin

    def run(self) -> float:
        if self._can_overlap():
            return self._run_overlapped()
        prog = self.prog
        time_begin = time.time()
        held: Optional[StepRecord] = None        # last record of the previous launch, waiting for its post-update accuracy
        done = False
        gs_seen = prog.state.global_step()
        while not done:
            if self.local_step % self.validation_every == 0:
                if held is not None:             # keep the reference's line order: step line, then validation
                    held.post_correct = prog.post_correct_of_last_step()
                    self._print_record(held)
                    held = None
                with trace.annotate("validation"):
                    self._validate()
            n_updates = self._plan(gs_seen)
            first = self.local_step
            with trace.annotate(f"launch {n_updates} updates"):
                self.engine.run(n_updates, min(n_updates * self.rounds, 64) if n_updates * self.rounds >= 2 else 1)
            n = n_updates * self.rounds
            if self.stop_check is not None and self.stop_check():
                n = prog.steps_enqueued - first  # the service ended training: keep what actually ran
                done = True
                if n <= 0:
                    break
            self.local_step += n
            with trace.annotate("read step log"):
                recs = prog.read_log(first, n)       # synchronises with the device
            if held is not None:
                # the accuracy-only rows of this launch's first step re-scored the held step's batch
                held.post_correct = self._post_of_previous(first)
                self._print_record(held)
                held = None
            for i, r in enumerate(recs):
                # the stop rule is evaluated where an update completes (every step, unless an update takes several)
                stop_here = r.global_step >= self.train_steps and (i + 1) % self.rounds == 0
                if prog.post and r.post_correct is None:          # only the last record of a launch
                    if not stop_here:
                        held = r
                        break
                    r.post_correct = prog.post_correct_of_last_step()
                self._print_record(r)
                if stop_here:
                    done = True
                    break
            gs_seen = recs[-1].global_step + 1
            if self.sv is not None:
                with trace.annotate("checkpoint check"):
                    self.sv.maybe_checkpoint()
                self.sv.report_step_rate(gs_seen)
        time_end = time.time()
        return time_end - time_begin

    def _post_of_previous(self, first_local_step: int) -> int:
        return self.prog.post_count_logged_at(first_local_step)


def build_worker(role: Optional[Role], device: torch.device, datasets, hidden_units: int, batch_size: int,
                 learning_rate: float, sync_replicas: bool, replicas_to_aggregate: Optional[int], world: int, rank: int,
                 optimizer: str = "sgd", loss: str = "reference", post_update_accuracy: bool = True, hidden_layers: int = 1,
                 staleness: int = 0, num_ps: int = 1, seed: int = 0, collective: str = "auto", model: str = "mlp",
                 dtype: str = "fp32", staleness_bound: Optional[int] = None):
    """Graph construction of distributed.py:60-102, as device-side state + step program + engine."""
    if model not in ("mlp", "wide_mlp"):
        raise ValueError(f"--model={model}: this script trains 'mlp' and 'wide_mlp' on MNIST-shape data "
                         "(ResNet-50 runs through bench.py --config resnet50_sync)")
    hidden = [hidden_units] * max(1, hidden_layers)

    def make_program(fused: bool, gs_inc: int):
        if model == "wide_mlp":
            from .models.wide_mlp import WideMLPProgram
            cdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
            return WideMLPProgram(state, data, batch_size, hidden, loss_mode=loss_mode, dtype=cdtype, gs_inc=gs_inc)
        return MLPProgram(state, data, batch_size, hidden, loss_mode=loss_mode, fused_update=fused,
                          post_update_accuracy=post_update_accuracy, gs_inc=gs_inc)

    state = build_mlp_state(IMAGE_PIXELS * IMAGE_PIXELS, hidden, 10, device, world_size=world, rank=rank,
                            optimizer=optimizer, lr=learning_rate)
    data = DeviceDataset(datasets.train, device, shuffle_seed=shuffle_seed(seed, rank))
    loss_mode = ops.LOSS_REFERENCE if loss == "reference" else ops.LOSS_STANDARD
    if sync_replicas or world == 1:
        R = replicas_to_aggregate if (sync_replicas and replicas_to_aggregate is not None) else world
        if not sync_replicas:
            R = 1
        fast = (R == world)
        fused = fast and world == 1 and optimizer == "sgd" and model == "mlp"
        prog = make_program(fused, 1 if fast else 0)
        engine = SyncReplicasEngine(prog, state, replicas_to_aggregate=R, collective=collective)
        is_async = False
    else:
        prog = make_program(False, 0)
        engine = make_async_engine(prog, state, num_ps, staleness, staleness_bound=staleness_bound)   # device-side push/pull where it applies
        is_async = True
    return state, prog, engine, is_async


def main(unused_argv=None) -> int:
    define_flags()
    datasets = mnist.load(FLAGS.data_dir, FLAGS.synthetic_data, seed=FLAGS.seed,
                          synthetic_train_size=FLAGS.synthetic_train_size)                 # distributed.py:38

    if FLAGS.single_process:
        if FLAGS.job_name is None:
            FLAGS.job_name = "worker"
        if FLAGS.task_index is None:
            FLAGS.task_index = 0
    validate_role(FLAGS.job_name, FLAGS.task_index, echo=print)                            # :40-47

    if FLAGS.single_process:
        ps_spec, worker_spec = [], ["127.0.0.1:0"]
    else:
        ps_spec = split_hosts(FLAGS.ps_hosts)                                              # :49-50
        worker_spec = split_hosts(FLAGS.worker_hosts)
    num_workers = len(worker_spec)                                                         # :52
    cluster = ClusterSpec({'ps': ps_spec, 'worker': worker_spec})                          # :53
    role = Role(FLAGS.job_name, FLAGS.task_index, cluster)
    device = resolve_device(role, FLAGS.device)
    # auto: the parameter-server state lives in the workers' HBM whenever every worker is on this machine (the ps
    # entries then only name logical shards); workers on other machines need the ps processes of the host backend
    if FLAGS.ps_backend != "auto":
        ps_backend = FLAGS.ps_backend
    else:
        ps_backend = "hbm" if (FLAGS.single_process or not ps_spec or local_hosts(worker_spec)) else "host"

    if ps_backend == "host":
        from .parallel import host_ps
        return host_ps.run_role(role, device, datasets, FLAGS)

    server = Server(cluster, FLAGS.job_name, FLAGS.task_index, ps_backend=ps_backend, device=device,
                    dist_backend=FLAGS.dist_backend, start=not FLAGS.single_process)       # :54
    if FLAGS.job_name == 'ps':
        server.join()                                                                      # :55-56
        return 0

    is_chief = (FLAGS.task_index == 0)                                                     # :58
    state, prog, engine, is_async = build_worker(
        role, device, datasets, FLAGS.hidden_units, FLAGS.batch_size, FLAGS.learning_rate, FLAGS.sync_replicas,
        FLAGS.replicas_to_aggregate, num_workers, FLAGS.task_index, optimizer=FLAGS.optimizer, loss=FLAGS.loss,
        post_update_accuracy=FLAGS.post_update_accuracy, hidden_layers=FLAGS.hidden_layers, staleness=FLAGS.staleness,
        num_ps=max(1, len(ps_spec)), seed=FLAGS.seed, model=FLAGS.model, dtype=FLAGS.dtype,
        staleness_bound=FLAGS.staleness_bound)

    sys.stderr.write("Worker %d: update engine %s\n" % (FLAGS.task_index, type(engine).__name__))   # not a reference line
    seed = FLAGS.seed if FLAGS.seed else (int(time.time() * 1000) & 0x7FFFFFFF)
    sv = Supervisor(is_chief=is_chief, logdir=FLAGS.train_dir, init_fn=lambda: state.initialize(seed), state=state,
                    recovery_wait_secs=1, save_model_secs=FLAGS.save_model_secs, store=server.store,
                    group_size=num_workers,                                                # :108-111
                    examples_per_global_step=FLAGS.batch_size * ((FLAGS.replicas_to_aggregate or num_workers)
                                                                 if FLAGS.sync_replicas else 1))
    sv.engine = engine
    if is_chief:
        print('Worker %d: Initailizing session...' % FLAGS.task_index)                     # :122
    else:
        print('Worker %d: Waiting for session to be initaialized...' % FLAGS.task_index)   # :124
    sv.prepare_or_wait_for_session()                                                       # :125
    if hasattr(engine, "publish_state"):
        engine.publish_state()        # ps-resident copies of the variables are taken from the initialised state
    if (FLAGS.sync_replicas and num_workers > 1 and getattr(engine, "p2p", None) is not None and getattr(engine, "fast", False)
            and hasattr(engine, "select_transport")):
        # Which peer-mapped transport is fastest depends on the node (link latency against link bandwidth against idle
        # CUs): a short timed trial of each (collective; parameters and counters are put back), as bench.py does.
        # DTF_P2P_MODE pins one and skips the trial.
        import os
        if os.environ.get("DTF_P2P_MODE") is None:
            rec = engine.select_transport(192, 64, restore=True)
            if rec is not None:
                sys.stderr.write("Worker %d: sync transport '%s' after a trial of %d steps each (%s)\n" % (
                    FLAGS.task_index, rec["chosen"], rec["steps_each"],
                    ", ".join("%s %.1f us/step" % (k[:-len("_us_per_step")], v) for k, v in sorted(rec.items())
                              if k.endswith("_us_per_step") and v is not None)))
        else:
            sys.stderr.write("Worker %d: sync transport '%s' (DTF_P2P_MODE)\n" % (FLAGS.task_index, engine.p2p_mode))
    print('Worker %d: Session initialization  complete.' % FLAGS.task_index)               # :126

    if FLAGS.sync_replicas and is_chief:
        print('Starting chief queue runner and running init_tokens_op')                    # :129

    def up(a):
        t = torch.from_numpy(a)
        return ops.h2d_verified(torch.empty(t.shape, dtype=t.dtype, device=device), t, "evaluation split")

    val = (up(datasets.validation.images), up(datasets.validation.labels))
    test = (up(datasets.test.images), up(datasets.test.labels))
    loop = TrainingLoop(FLAGS.task_index, prog, engine, FLAGS.train_steps, FLAGS.validation_every, FLAGS.log_every,
                        FLAGS.steps_per_launch, val, test, sv, is_async, num_workers)
    train_time = loop.run()                                                                # :133-158
    print('Training elapsed time:%f s' % train_time)                                       # :161
    if is_async:
        engine.sync_params()
    test_accuracy = prog.accuracy(*test)                                                   # :163-164
    print('Worker %d: test accuracy %g' % (FLAGS.task_index, test_accuracy))               # :165
    examples = loop.local_step * FLAGS.batch_size
    print('Worker %d: %d local steps, %.1f examples/sec on this worker' % (FLAGS.task_index, loop.local_step,
                                                                           examples / max(train_time, 1e-9)))
    # The final checkpoint comes BEFORE the engine is torn down: with Adam the chief reads the slot variables from the
    # owners (mapped regions / their service threads), which finish() releases / lets return.
    sv.stop()
    if is_async:
        engine.finish()
    server.shutdown()
    sys.stdout.flush()
    return 0


def run() -> None:
    """Entry point: ``tf.app.run()`` (distributed.py:167-168)."""
    define_flags()
    flags_lib.run(main)

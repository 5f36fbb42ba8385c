"""Host-memory parameter-server backend (``--ps_backend=host``).

This is the reference's own topology: ``ps`` processes own the variables in
host memory and every worker talks to every ps task over TCP each step
(/root/reference/distributed.py:54-64, :102, :116-117 — a star, workers never
talk to each other).  The service itself is native code
(``csrc/runtime/ps_service.cpp``); this module is its client side and the two
update protocols:

* async (default): pull all variables, compute, push each gradient — the ps
  applies it on arrival without locking — then bump ``global_step``.
* sync (``--sync_replicas``): gradients go into per-variable accumulators
  stamped with the worker's local step (stale ones are dropped); the chief's
  queue-runner thread takes ``replicas_to_aggregate`` gradients per variable,
  applies their mean, bumps ``global_step`` once, stamps the accumulators and
  releases ``total_num_replicas`` tokens; every worker blocks on a token before
  its next step (:97-106, :128-131).

Variables are placed on ps tasks round-robin in creation order
(``global_step, hid_w, hid_b, sm_w, sm_b``), whole variables only — what
``tf.train.replica_device_setter`` does (:60-64).

The HBM-sharded backend is the fast path on MI355X; this one exists for
fidelity (CPU parameter servers, multi-node over plain TCP) and as the
executable specification of the SyncReplicas semantics.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from ..cluster.spec import Role, parse_host_port, round_robin_placement
from .. import ops
from ..ops import _lib

(OP_REGISTER, OP_INIT, OP_PULL, OP_PUSH_ASYNC, OP_INC_STEP, OP_GET_STEP, OP_SET_STEP, OP_ACC_APPLY, OP_TAKE_APPLY,
 OP_ACC_SET_STEP, OP_TOKEN_ENQUEUE, OP_TOKEN_DEQUEUE, OP_SET_READY, OP_IS_READY, OP_SHUTDOWN, OP_STATS,
 OP_WORKER_DONE) = range(1, 18)
STATUS_STOPPING = 4

_declared = False


def _rt():
    global _declared
    lib = _lib.runtime()
    if not _declared:
        lib.dtf_ps_server_start.restype = C.c_void_p
        lib.dtf_ps_server_start.argtypes = [C.c_int, C.POINTER(C.c_int)]
        lib.dtf_ps_server_start_on.restype = C.c_void_p
        lib.dtf_ps_server_start_on.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        lib.dtf_ps_server_wait.restype = C.c_int
        lib.dtf_ps_server_wait.argtypes = [C.c_void_p, C.c_double]
        lib.dtf_ps_server_stop.restype = None
        lib.dtf_ps_server_stop.argtypes = [C.c_void_p]
        lib.dtf_ps_connect.restype = C.c_void_p
        lib.dtf_ps_connect.argtypes = [C.c_char_p, C.c_int, C.c_double]
        lib.dtf_ps_call.restype = C.c_int
        lib.dtf_ps_call.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_double, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_uint32)]
        lib.dtf_ps_close.restype = None
        lib.dtf_ps_close.argtypes = [C.c_void_p]
        _declared = True
    return lib


class ParameterService:
    """One ps task: owns its variables, accumulators, the token queue and global_step."""

    def __init__(self, port: int, num_workers: int = 0, host: str = ""):
        """``host``: the address of this task in ``--ps_hosts``; the service listens there only.  A host name is
        resolved to its IPv4 address; an empty string means the loopback interface; every interface has to be asked
        for as ``0.0.0.0``."""
        self.port, self.num_workers = port, num_workers
        import re
        import socket
        host = host or ""
        if host and not re.fullmatch(r"\d+\.\d+\.\d+\.\d+", host):
            try:
                host = socket.gethostbyname(host)
            except OSError as e:
                raise OSError(f"parameter service: cannot resolve the ps host {host!r} of the cluster description") from e
        self.host = host
        self._h = None

    def start(self) -> int:
        actual = C.c_int(0)
        self._h = _rt().dtf_ps_server_start_on(self.host.encode() or None, self.port, C.byref(actual))
        if not self._h:
            raise OSError(f"parameter service could not listen on {self.host or '*'}:{self.port}")
        self.port = actual.value
        return self.port

    def wait(self, timeout: float = -1.0) -> bool:
        """``server.join()``: returns once a client requested shutdown."""
        if timeout >= 0:
            return bool(_rt().dtf_ps_server_wait(self._h, timeout))
        while not _rt().dtf_ps_server_wait(self._h, 0.5):   # short waits keep Ctrl-C responsive
            pass
        return True

    def stop(self) -> None:
        if self._h:
            _rt().dtf_ps_server_stop(self._h)
            self._h = None


class PSStopped(RuntimeError):
    """The service is shutting down (a blocked dequeue / take was released)."""


class PSClient:
    def __init__(self, host: str, port: int, timeout: float = 600.0):
        self._h = _rt().dtf_ps_connect(host.encode(), port, timeout)
        if not self._h:
            raise ConnectionError(f"cannot reach the parameter service at {host}:{port}")
        self._val, self._aux, self._nb = C.c_int64(), C.c_uint32(), C.c_uint64()

    def call(self, op, var=0, aux=0, value=0, scalar=0.0, payload: Optional[np.ndarray] = None,
             out: Optional[np.ndarray] = None):
        pp, pn = (payload.ctypes.data_as(C.c_void_p), payload.nbytes) if payload is not None else (None, 0)
        op_, on = (out.ctypes.data_as(C.c_void_p), out.nbytes) if out is not None else (None, 0)
        rc = _rt().dtf_ps_call(self._h, op, var, aux, value, scalar, pp, pn, op_, on, C.byref(self._nb),
                               C.byref(self._val), C.byref(self._aux))
        if rc == STATUS_STOPPING or rc < 0:
            # status 4: a blocked call was released by shutdown; negative: the connection is gone,
            # which only happens once the chief has shut the service down
            raise PSStopped()
        if rc != 0:
            raise RuntimeError(f"parameter service call {op} failed with code {rc}")
        return self._val.value, self._aux.value

    def close(self):
        if self._h:
            _rt().dtf_ps_close(self._h)
            self._h = None


class HostPSEngine:
    """Worker-side engine for the host backend (async and sync protocols)."""

    def __init__(self, program, state, role: Role, sync: bool, replicas_to_aggregate: Optional[int] = None,
                 init_tokens: Optional[int] = None, connect_timeout: float = 600.0):
        self.program, self.state, self.role = program, state, role
        if program.fused or program.gs_inc != 0:
            raise ValueError("host backend needs a program that stores gradients and leaves global_step to the ps")
        self.sync = sync
        self.N = role.num_workers
        self.R = replicas_to_aggregate if replicas_to_aggregate is not None else self.N
        self.init_tokens = self.R if init_tokens is None else init_tokens
        if self.init_tokens < self.R - self.N:
            raise ValueError("init_tokens must be at least replicas_to_aggregate - num_workers")
        self.names = ["global_step"] + state.names()
        self.placement = round_robin_placement(self.names, role.num_ps)
        ps_addrs = [parse_host_port(a) for a in role.cluster.job_tasks("ps")]
        self.clients = [PSClient(h, p, connect_timeout) for h, p in ps_addrs]
        self._ps_addrs = ps_addrs
        self.var_ids: Dict[str, int] = {}
        self._host_params = np.zeros(state.layout.total, dtype=np.float32)
        self._host_grads = np.zeros(state.layout.total, dtype=np.float32)
        self._slices = {}
        for v in state.layout.variables:
            off = state.layout.offsets[v.name]
            self._slices[v.name] = slice(off, off + v.numel)
        self.local_stamp = 0               # the worker's SyncReplicas local step (starts at 0)
        self.dropped = 0
        self.stopped = False
        self._runner: Optional[threading.Thread] = None
        self._runner_stop = threading.Event()
        self.aggregates = 0

    # ------------------------------------------------------------------ registration / init / readiness
    def _client_of(self, name: str) -> PSClient:
        return self.clients[self.placement[name]]

    def register(self) -> None:
        for v in self.state.layout.variables:
            vid, _ = self._client_of(v.name).call(OP_REGISTER, value=v.numel, payload=np.frombuffer(v.name.encode(), np.uint8))
            self.var_ids[v.name] = vid

    def chief_initialize(self) -> None:
        """Chief: push the initial values, set global_step, publish readiness (init_op, distributed.py:108)."""
        host = self.state.params.detach().cpu().numpy()
        for name, sl in self._slices.items():
            self._client_of(name).call(OP_INIT, var=self.var_ids[name], payload=np.ascontiguousarray(host[sl]))
        self._client_of("global_step").call(OP_SET_STEP, value=self.state.global_step())
        for c in self.clients:
            c.call(OP_SET_READY)

    def wait_ready(self, recovery_wait_secs: float = 1.0) -> int:
        """Non-chief: poll every ps until the chief has initialised everything (:110-111, :125)."""
        polls = 0
        while True:
            if all(c.call(OP_IS_READY)[1] == 1 for c in self.clients):
                return polls
            polls += 1
            time.sleep(recovery_wait_secs)

    # ------------------------------------------------------------------ data movement
    def pull(self) -> int:
        for name, sl in self._slices.items():
            self._client_of(name).call(OP_PULL, var=self.var_ids[name], out=self._host_params[sl])
        gs, _ = self._client_of("global_step").call(OP_GET_STEP)
        self.state.params.copy_(torch.from_numpy(self._host_params))
        self.state.state[0] = int(gs)
        return int(gs)

    def _grads_to_host(self) -> None:
        self._host_grads[:] = self.state.grads.detach().cpu().numpy()

    # ------------------------------------------------------------------ steps
    def step(self) -> None:
        if self.stopped:
            return
        prog = self.program
        try:
            self.pull()
            saved, prog.use_graphs = prog.use_graphs, False
            prog.run(1)
            prog.use_graphs = saved
            self._grads_to_host()
            if not self.sync:
                for name, sl in self._slices.items():
                    self._client_of(name).call(OP_PUSH_ASYNC, var=self.var_ids[name], scalar=self.state.lr,
                                               payload=self._host_grads[sl])
                self._client_of("global_step").call(OP_INC_STEP)
            else:
                for name, sl in self._slices.items():
                    _, accepted = self._client_of(name).call(OP_ACC_APPLY, var=self.var_ids[name], value=self.local_stamp,
                                                             payload=self._host_grads[sl])
                    if not accepted and name == self.state.names()[0]:
                        self.dropped += 1
                token, _ = self._client_of("global_step").call(OP_TOKEN_DEQUEUE)
                self.local_stamp = int(token)
        except PSStopped:
            self.stopped = True

    def run(self, n_steps: int, graph_steps: int = 0) -> None:
        for _ in range(n_steps):
            self.step()

    def sync_params(self) -> int:
        try:
            return self.pull()
        except PSStopped:
            self.stopped = True
            return self.state.global_step()

    # ------------------------------------------------------------------ chief queue runner (sync mode)
    def start_queue_runner(self) -> None:
        """``sv.start_queue_runners(sess, [chief_queue_runner])`` + ``init_tokens_op`` (distributed.py:128-131)."""
        gs_client = PSClient(*self._ps_addrs[self.placement["global_step"]])
        gs, _ = gs_client.call(OP_GET_STEP)
        if self.init_tokens > 0:
            gs_client.call(OP_TOKEN_ENQUEUE, aux=self.init_tokens, value=gs)
        clients = [PSClient(h, p) for h, p in self._ps_addrs]   # the runner blocks in TAKE: own connections

        def loop():
            try:
                while not self._runner_stop.is_set():
                    for name in self.state.names():
                        clients[self.placement[name]].call(OP_TAKE_APPLY, var=self.var_ids[name], value=self.R,
                                                           scalar=self.state.lr)
                    new_gs, _ = gs_client.call(OP_INC_STEP)
                    for name in self.state.names():
                        clients[self.placement[name]].call(OP_ACC_SET_STEP, var=self.var_ids[name], value=new_gs)
                    gs_client.call(OP_TOKEN_ENQUEUE, aux=self.N, value=new_gs)
                    self.aggregates += 1
            except (PSStopped, RuntimeError):
                pass
            finally:
                for c in clients + [gs_client]:
                    c.close()

        self._runner = threading.Thread(target=loop, name="dtf-chief-queue-runner", daemon=True)
        self._runner.start()

    # ------------------------------------------------------------------ end of training
    def finish(self, grace_secs: float = 5.0) -> None:
        """Clean termination (the reference's ps never exits and its sync workers can hang when the
        chief leaves): every worker reports done; the chief waits for the others, then shuts the
        service down, which also releases anyone still blocked on a token."""
        gs_client = self._client_of("global_step")
        try:
            gs_client.call(OP_WORKER_DONE)
        except (PSStopped, RuntimeError):
            pass
        if self.role.is_chief:
            deadline = time.time() + grace_secs
            while time.time() < deadline:
                try:
                    stats = np.zeros(5, dtype=np.int64)
                    gs_client.call(OP_STATS, out=stats)
                    if stats[4] >= self.N:
                        break
                except (PSStopped, RuntimeError):
                    break
                time.sleep(0.05)
            self._runner_stop.set()
            for c in self.clients:
                try:
                    c.call(OP_SHUTDOWN)
                except (PSStopped, RuntimeError):
                    pass
            if self._runner is not None:
                self._runner.join(timeout=10)
        for c in self.clients:
            c.close()

    def stats(self) -> Dict[str, int]:
        st = np.zeros(5, dtype=np.int64)
        gs, _ = self._client_of("global_step").call(OP_STATS, out=st)
        return {"pushes": int(st[0]), "pulls": int(st[1]), "dropped": int(st[2]), "updates": int(st[3]),
                "workers_done": int(st[4]), "global_step": int(gs)}


def run_role(role: Role, device, datasets, FLAGS) -> int:
    """``main`` for the host backend: ps tasks serve, workers train (distributed.py:54-165)."""
    from .. import ops, train
    from ..data.device_dataset import DeviceDataset
    from ..models.mlp import MLPProgram, build_mlp_state
    from ..supervisor.supervisor import Supervisor

    # what this backend trains is the reference configuration: say so instead of silently ignoring a flag
    unsupported = [f"--{n}={getattr(FLAGS, n)}" for n, dflt in (("optimizer", "sgd"), ("model", "mlp"), ("dtype", "fp32"))
                   if getattr(FLAGS, n) != dflt]
    if unsupported:
        raise ValueError("--ps_backend=host trains the reference MLP with SGD in fp32; not supported there: "
                         + " ".join(unsupported))
    if role.is_ps:
        host, port = parse_host_port(role.address)
        svc = ParameterService(port, role.num_workers, host)
        svc.start()
        svc.wait()                                             # server.join()
        time.sleep(0.2)                                        # let the last responses drain
        svc.stop()
        return 0

    if role.num_ps < 1:
        raise ValueError("--ps_backend=host needs at least one entry in --ps_hosts")
    is_chief = role.is_chief
    hidden = [FLAGS.hidden_units] * max(1, FLAGS.hidden_layers)
    state = build_mlp_state(train.IMAGE_PIXELS ** 2, hidden, 10, device, optimizer="sgd", lr=FLAGS.learning_rate)
    data = DeviceDataset(datasets.train, device, shuffle_seed=train.shuffle_seed(FLAGS.seed, role.task_index))
    loss_mode = ops.LOSS_REFERENCE if FLAGS.loss == "reference" else ops.LOSS_STANDARD
    # (the training accuracy of a step is the one of its training pass: the parameters live on the ps, a second
    # pass would need a second pull; --post_update_accuracy has no effect on this backend)
    prog = MLPProgram(state, data, FLAGS.batch_size, hidden, loss_mode=loss_mode, fused_update=False,
                      post_update_accuracy=False, gs_inc=0)
    engine = HostPSEngine(prog, state, role, sync=FLAGS.sync_replicas, replicas_to_aggregate=FLAGS.replicas_to_aggregate)
    engine.register()
    seed = FLAGS.seed if FLAGS.seed else (int(time.time() * 1000) & 0x7FFFFFFF)
    sv = Supervisor(is_chief=is_chief, logdir=FLAGS.train_dir, init_fn=lambda: state.initialize(seed), state=state,
                    save_model_secs=FLAGS.save_model_secs)
    if is_chief:
        print('Worker %d: Initailizing session...' % role.task_index)
        sv.prepare_or_wait_for_session()                       # restore-or-init locally, then publish to the ps tasks
        engine.chief_initialize()
    else:
        print('Worker %d: Waiting for session to be initaialized...' % role.task_index)
        engine.wait_ready(1.0)
    print('Worker %d: Session initialization  complete.' % role.task_index)
    if FLAGS.sync_replicas and is_chief:
        print('Starting chief queue runner and running init_tokens_op')
        engine.start_queue_runner()

    def up(a):
        t = torch.from_numpy(a)
        return ops.h2d_verified(torch.empty(t.shape, dtype=t.dtype, device=device), t, "evaluation split")

    val = (up(datasets.validation.images), up(datasets.validation.labels))
    test = (up(datasets.test.images), up(datasets.test.labels))
    loop = train.TrainingLoop(role.task_index, prog, engine, FLAGS.train_steps, FLAGS.validation_every, FLAGS.log_every,
                              1, val, test, sv if is_chief else None, True, role.num_workers)
    loop.stop_check = lambda: engine.stopped
    train_time = loop.run()
    print('Training elapsed time:%f s' % train_time)
    engine.sync_params()
    print('Worker %d: test accuracy %g' % (role.task_index, prog.accuracy(*test)))
    if is_chief:
        # snapshot for the final checkpoint comes from the ps, which holds the authoritative values
        sv.stop()
    engine.finish()
    return 0

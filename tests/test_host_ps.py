"""Native parameter service: protocol unit tests + end-to-end CLI runs with real ps processes."""
import threading
import time

import numpy as np
import pytest

from distributed_tensorflow_amd.parallel import host_ps as H
from test_cli import COMMON, _finish, _launch, _steps
from dist_utils import free_port


@pytest.fixture
def service():
    svc = H.ParameterService(0)
    port = svc.start()
    yield svc, port
    svc.stop()


def _reg(c, name, n):
    return c.call(H.OP_REGISTER, value=n, payload=np.frombuffer(name.encode(), np.uint8))[0]


def test_register_init_pull_push_async(service):
    _, port = service
    c = H.PSClient("127.0.0.1", port)
    vid = _reg(c, "hid_w", 1000)
    assert _reg(c, "hid_w", 1000) == vid and _reg(c, "sm_w", 10) == vid + 1
    assert c.call(H.OP_IS_READY)[1] == 0
    w = np.arange(1000, dtype=np.float32)
    c.call(H.OP_INIT, var=vid, payload=w)
    c.call(H.OP_INIT, var=vid + 1, payload=np.zeros(10, np.float32))
    c.call(H.OP_SET_READY)
    assert c.call(H.OP_IS_READY)[1] == 1
    out = np.zeros(1000, np.float32)
    gs, _ = c.call(H.OP_PULL, var=vid, out=out)
    assert gs == 1 and np.array_equal(out, w)                 # global_step starts at 1
    g = np.ones(1000, np.float32)
    c.call(H.OP_PUSH_ASYNC, var=vid, scalar=0.5, payload=g)
    assert c.call(H.OP_INC_STEP)[0] == 2
    c.call(H.OP_PULL, var=vid, out=out)
    np.testing.assert_allclose(out, w - 0.5)
    with pytest.raises(RuntimeError):
        c.call(H.OP_INIT, var=vid, payload=np.zeros(3, np.float32))    # wrong size
    c.close()


def test_accumulator_mean_stale_drop_and_tokens(service):
    _, port = service
    a, b, chief = (H.PSClient("127.0.0.1", port) for _ in range(3))
    vid = _reg(a, "v", 4)
    a.call(H.OP_INIT, var=vid, payload=np.zeros(4, np.float32))
    took = {}

    def take():
        took["n"] = chief.call(H.OP_TAKE_APPLY, var=vid, value=2, scalar=1.0)[1]

    t = threading.Thread(target=take)
    t.start()
    time.sleep(0.1)
    assert t.is_alive()                                              # blocks until 2 gradients arrived
    assert a.call(H.OP_ACC_APPLY, var=vid, value=0, payload=np.full(4, 2.0, np.float32))[1] == 1
    assert t.is_alive()
    assert b.call(H.OP_ACC_APPLY, var=vid, value=0, payload=np.full(4, 4.0, np.float32))[1] == 1
    t.join(timeout=5)
    assert took["n"] == 2
    out = np.zeros(4, np.float32)
    a.call(H.OP_PULL, var=vid, out=out)
    np.testing.assert_allclose(out, -3.0)                            # mean of (2, 4), lr 1: averaged, not summed
    chief.call(H.OP_ACC_SET_STEP, var=vid, value=5)
    assert a.call(H.OP_ACC_APPLY, var=vid, value=4, payload=np.ones(4, np.float32))[1] == 0   # stale: dropped
    assert a.call(H.OP_ACC_APPLY, var=vid, value=5, payload=np.ones(4, np.float32))[1] == 1
    st = np.zeros(5, np.int64)
    a.call(H.OP_STATS, out=st)
    assert st[2] == 1
    # token queue: FIFO, blocking dequeue
    chief.call(H.OP_TOKEN_ENQUEUE, aux=2, value=7)
    assert a.call(H.OP_TOKEN_DEQUEUE)[0] == 7 and b.call(H.OP_TOKEN_DEQUEUE)[0] == 7
    got = {}
    t2 = threading.Thread(target=lambda: got.setdefault("t", a.call(H.OP_TOKEN_DEQUEUE)[0]))
    t2.start()
    time.sleep(0.1)
    assert t2.is_alive()
    chief.call(H.OP_TOKEN_ENQUEUE, aux=1, value=8)
    t2.join(timeout=5)
    assert got["t"] == 8
    for c in (a, b, chief):
        c.close()


def test_blocked_dequeue_is_released_on_stop():
    svc = H.ParameterService(0)
    port = svc.start()
    c = H.PSClient("127.0.0.1", port)
    res = {}

    def deq():
        try:
            c.call(H.OP_TOKEN_DEQUEUE)
        except (H.PSStopped, RuntimeError) as e:
            res["e"] = e

    t = threading.Thread(target=deq)
    t.start()
    time.sleep(0.1)
    svc.stop()
    t.join(timeout=5)
    assert not t.is_alive() and "e" in res


def _cluster_args(n_ps, n_workers, extra):
    ps = [free_port() for _ in range(n_ps)]
    wk = [free_port() for _ in range(n_workers)]
    return ["--ps_backend=host", "--ps_hosts=" + ",".join("127.0.0.1:%d" % p for p in ps),
            "--worker_hosts=" + ",".join("127.0.0.1:%d" % p for p in wk)] + extra + COMMON


@pytest.mark.slow
def test_host_backend_async_two_ps_two_workers():
    args = _cluster_args(2, 2, ["--train_steps=30"])
    ps = [_launch(["--job_name=ps", "--task_index=%d" % i] + args) for i in range(2)]
    w1 = _launch(["--job_name=worker", "--task_index=1"] + args)
    w0 = _launch(["--job_name=worker", "--task_index=0"] + args)
    l0, l1 = _finish(w0), _finish(w1)
    for p in ps:
        _finish(p)                                   # ps tasks exit cleanly once the chief shuts the service down
    s0, s1 = _steps(l0), _steps(l1)
    assert s0 and s1 and max(s[1] for s in s0 + s1) >= 30
    assert len(s0) + len(s1) <= 30 + 4               # global_step counts both workers' updates
    assert s0[-1][2] < s0[0][2] - 0.2
    assert "Worker 1: Waiting for session to be initaialized..." in l1


@pytest.mark.slow
def test_host_backend_sync_one_ps_two_workers():
    args = _cluster_args(1, 2, ["--train_steps=12", "--sync_replicas"])
    ps = _launch(["--job_name=ps", "--task_index=0"] + args)
    w0 = _launch(["--job_name=worker", "--task_index=0"] + args)
    w1 = _launch(["--job_name=worker", "--task_index=1"] + args)
    l0, l1 = _finish(w0), _finish(w1)
    _finish(ps)
    assert "Starting chief queue runner and running init_tokens_op" in l0
    s0, s1 = _steps(l0), _steps(l1)
    # tokens pace the workers: both see global_step reach train_steps, one increment per aggregate
    assert s0[-1][1] >= 12 and s1[-1][1] >= 12
    assert s0[-1][2] < s0[0][2] - 0.2
    gs0 = [s[1] for s in s0]
    assert gs0 == sorted(gs0)


def test_service_refuses_sizes_it_cannot_trust(service):
    """Sizes arrive over the wire: a negative or absurd element count, a payload larger than the variable it is
    for, or a flood of tokens gets an error status (or a closed connection), and the service keeps serving."""
    _, port = service
    c = H.PSClient("127.0.0.1", port)
    for bad in (-1, 0, 1 << 40):
        with pytest.raises(RuntimeError):
            c.call(H.OP_REGISTER, value=bad, payload=np.frombuffer(b"w", np.uint8))
    vid = _reg(c, "v", 8)
    with pytest.raises(RuntimeError):
        c.call(H.OP_TOKEN_ENQUEUE, aux=1 << 20, value=1)
    # a payload that does not fit the variable: refused before it is read, the connection ends
    with pytest.raises(RuntimeError):
        c.call(H.OP_INIT, var=vid, payload=np.zeros(1 << 16, np.float32))
    c.close()
    d = H.PSClient("127.0.0.1", port)                      # the service is still there
    assert _reg(d, "v", 8) == vid
    d.call(H.OP_INIT, var=vid, payload=np.arange(8, dtype=np.float32))
    out = np.zeros(8, np.float32)
    d.call(H.OP_PULL, var=vid, out=out)
    assert np.array_equal(out, np.arange(8, dtype=np.float32))
    d.close()


def test_service_listens_on_its_configured_address_only():
    svc = H.ParameterService(0, host="127.0.0.1")
    port = svc.start()
    c = H.PSClient("127.0.0.1", port)
    assert c.call(H.OP_GET_STEP)[0] == 1
    c.close()
    svc.stop()

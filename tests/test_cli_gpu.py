"""The training script on a GPU: the loop that launches ahead of its log processing prints, line for line, what the
plain loop prints (same trajectory: consecutive launches continue the pipelined step bit for bit)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_RE = re.compile(r'^Worker 0: traing step (\d+) \(global step:(\d+)\) loss (\d+\.\d{6}) training accuracy (\S+)$')


def _run(args, overlap):
    env = dict(os.environ, PYTHONUNBUFFERED="1", DTF_LOOP_OVERLAP="1" if overlap else "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "distributed.py")] + args, capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return [l for l in p.stdout.splitlines() if not l.startswith("Training elapsed time") and "examples/sec" not in l]


@pytest.mark.parametrize("extra,steps", [(["--sync_replicas", "--validation_every=250", "--steps_per_launch=64"], 700),
                                         (["--sync_replicas", "--validation_every=97"], 1000),
                                         (["--sync_replicas", "--validation_every=100000", "--nopost_update_accuracy"], 600),
                                         (["--validation_every=300"], 900)])       # the script's default: async mode
def test_overlapped_loop_prints_what_the_plain_loop_prints(extra, steps):
    args = ["--single_process", "--train_steps=%d" % steps, "--synthetic_data", "--synthetic_train_size=4000",
            "--learning_rate=0.5", "--seed=7"] + extra
    plain, ahead = _run(args, False), _run(args, True)
    assert plain == ahead
    recs = [STEP_RE.match(l) for l in ahead]
    recs = [(int(m.group(1)), int(m.group(2)), float(m.group(3))) for m in recs if m]
    assert [r[0] for r in recs] == list(range(1, steps + 1))          # every local step, in order, none twice
    assert [r[1] for r in recs] == list(range(1, steps + 1))          # global_step starts at 1; the last fetched value is train_steps
    assert recs[-1][2] < recs[0][2]
    assert any(l.startswith("Worker 0: test accuracy") for l in ahead)


@pytest.mark.parametrize("transport", ["pipelined", "sharded"])
def test_two_sync_workers_sharing_the_gpu_print_the_same_with_both_loops(transport):
    """1 ps + 2 workers as processes of the script, the workers on the one GPU of the box (gloo process group, peer-mapped
    sync step, the transport pinned so that both runs take the same one): the loop that launches ahead prints what the
    plain loop prints, on both workers."""
    from dist_utils import free_port

    def run(mode):
        ports = [free_port() for _ in range(3)]
        common = ["--ps_hosts=127.0.0.1:%d" % ports[2], "--worker_hosts=127.0.0.1:%d,127.0.0.1:%d" % (ports[0], ports[1]),
                  "--sync_replicas", "--train_steps=1200", "--synthetic_data", "--synthetic_train_size=4000", "--learning_rate=0.5",
                  "--seed=7", "--validation_every=500", "--log_every=25", "--dist_backend=gloo"]
        env = dict(os.environ, PYTHONUNBUFFERED="1", DTF_LOOP_OVERLAP=mode, DTF_SHARE_GPU="1", DTF_P2P_ALLREDUCE="1",
                   DTF_P2P_MODE=transport)
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "distributed.py"), "--job_name=%s" % job, "--task_index=%d" % idx]
                                  + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
                 for job, idx in (("ps", 0), ("worker", 1), ("worker", 0))]
        outs = []
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-3000:]
            if p is not procs[0]:
                assert "sync transport '%s' (DTF_P2P_MODE)" % transport in err
            outs.append([l for l in out.splitlines() if not l.startswith("Training elapsed time") and "examples/sec" not in l])
        return outs[1], outs[2]

    (w1_plain, w0_plain), (w1_ahead, w0_ahead) = run("0"), run("1")
    assert w0_plain == w0_ahead and w1_plain == w1_ahead
    steps = [STEP_RE.match(l) for l in w0_ahead]
    steps = [(int(m.group(1)), int(m.group(2))) for m in steps if m]
    assert steps and steps[-1][1] <= 1200 and all(a[0] < b[0] for a, b in zip(steps, steps[1:]))
    assert any(l.startswith("Worker 0: test accuracy") for l in w0_ahead)


def test_sync_workers_time_the_transports_at_start_up_and_say_which_one_they_keep():
    """Without a pinned transport the workers run a short trial of every peer-mapped transport before the first
    training step (parameters and counters are put back: the step lines still start at 1) and name the winner."""
    from dist_utils import free_port
    ports = [free_port() for _ in range(3)]
    common = ["--ps_hosts=127.0.0.1:%d" % ports[2], "--worker_hosts=127.0.0.1:%d,127.0.0.1:%d" % (ports[0], ports[1]),
              "--sync_replicas", "--train_steps=300", "--synthetic_data", "--synthetic_train_size=4000", "--learning_rate=0.5",
              "--seed=7", "--validation_every=500", "--dist_backend=gloo"]
    env = {k: v for k, v in os.environ.items() if k != "DTF_P2P_MODE"}
    env.update(PYTHONUNBUFFERED="1", DTF_SHARE_GPU="1", DTF_P2P_ALLREDUCE="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "distributed.py"), "--job_name=%s" % job, "--task_index=%d" % idx]
                              + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
             for job, idx in (("ps", 0), ("worker", 1), ("worker", 0))]
    res = [p.communicate(timeout=300) for p in procs]
    for p, (_, err) in zip(procs, res):
        assert p.returncode == 0, err[-3000:]
    chosen = []
    for out, err in res[1:]:
        m = re.search(r"sync transport '(replicated|sharded|pipelined)' after a trial of 192 steps each \((.*)\)", err)
        assert m, err[-2000:]
        assert "pipelined" in m.group(2) and "sharded" in m.group(2) and "replicated" in m.group(2)
        chosen.append(m.group(1))
        steps = [STEP_RE.match(l.replace("Worker 1", "Worker 0")) for l in out.splitlines()]
        steps = [(int(x.group(1)), int(x.group(2))) for x in steps if x]
        assert steps[0] == (1, 1) and steps[-1][1] == 300
    assert chosen[0] == chosen[1]

"""End-to-end CLI runs on CPU (gloo): log lines, stop rule, sync trajectory equivalence, ps role."""
import os
import re
import subprocess
import sys

import pytest

from dist_utils import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_RE = re.compile(r'^Worker (\d+): traing step (\d+) \(global step:(\d+)\) loss (\d+\.\d{6}) training accuracy (\S+)$')
COMMON = ["--synthetic_data", "--synthetic_train_size=2000", "--device=cpu", "--learning_rate=0.5", "--seed=7",
          "--validation_every=10"]


def _launch(args, timeout=300):
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONUNBUFFERED="1")
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "distributed.py")] + args, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)


def _finish(p, timeout=300):
    out, err = p.communicate(timeout=timeout)
    assert p.returncode == 0, f"exit {p.returncode}\nSTDOUT:\n{out}\nSTDERR:\n{err[-3000:]}"
    return out.splitlines()


def _steps(lines):
    return [(int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5)))
            for m in (STEP_RE.match(l) for l in lines) if m]


def test_single_worker_log_format_and_stop_rule():
    lines = _finish(_launch(["--job_name=worker", "--task_index=0", "--worker_hosts=127.0.0.1:%d" % free_port(),
                             "--ps_hosts=", "--train_steps=23"] + COMMON))
    assert lines[0] == "job_name : worker" and lines[1] == "task_index : 0"
    assert lines[2] == "Worker 0: Initailizing session..."
    assert lines[3] == "Worker 0: Session initialization  complete."
    assert re.match(r"^Worker 0: validation accuracy \S+$", lines[4])
    steps = _steps(lines)
    # global_step starts at 1 and the loop stops at the first fetched value >= train_steps
    assert [s[0] for s in steps] == list(range(1, 24))
    assert [s[1] for s in steps] == list(range(1, 24))
    assert steps[-1][2] < steps[0][2]
    # validation at local steps 0, 10, 20 — each before the step line that follows it
    vidx = [i for i, l in enumerate(lines) if "validation accuracy" in l]
    assert len(vidx) == 3
    assert lines[vidx[1] - 1].startswith("Worker 0: traing step 10 ") and lines[vidx[1] + 1].startswith("Worker 0: traing step 11 ")
    assert any(re.match(r"^Training elapsed time:\d+\.\d{6} s$", l) for l in lines)
    assert any(re.match(r"^Worker 0: test accuracy \S+$", l) for l in lines)


def test_missing_job_name_is_a_value_error():
    p = _launch(["--task_index=0"] + COMMON)
    out, err = p.communicate(timeout=120)
    assert p.returncode != 0 and "ValueError: Must specify an explicit job_name !" in err


@pytest.mark.slow
def test_one_ps_two_workers_sync_matches_single_process_big_batch():
    """The CPU correctness configuration: 1 ps + 2 workers on CPU, sync mode.  The mean of the two workers'
    gradients applied once equals one step on the union batch, so every worker sees the same
    global steps and the run converges like a batch-200 run."""
    port_w0, port_w1, port_ps = free_port(), free_port(), free_port()
    hosts = ["--ps_hosts=127.0.0.1:%d" % port_ps, "--worker_hosts=127.0.0.1:%d,127.0.0.1:%d" % (port_w0, port_w1),
             "--sync_replicas", "--train_steps=15"] + COMMON
    ps = _launch(["--job_name=ps", "--task_index=0"] + hosts)
    w1 = _launch(["--job_name=worker", "--task_index=1"] + hosts)
    w0 = _launch(["--job_name=worker", "--task_index=0"] + hosts)
    l0, l1, lps = _finish(w0), _finish(w1), _finish(ps)
    assert "Worker 0: Initailizing session..." in l0
    assert "Starting chief queue runner and running init_tokens_op" in l0
    assert "Worker 1: Waiting for session to be initaialized..." in l1
    assert "Starting chief queue runner and running init_tokens_op" not in l1
    assert lps[:2] == ["job_name : ps", "task_index : 0"]
    s0, s1 = _steps(l0), _steps(l1)
    # one global_step increment per aggregated update: both workers log 1..15
    assert [s[1] for s in s0] == list(range(1, 16)) == [s[1] for s in s1]
    assert s0[-1][2] < s0[0][2] - 0.3 and s1[-1][2] < s1[0][2] - 0.3
    t0 = [l for l in l0 if "test accuracy" in l][0].split()[-1]
    t1 = [l for l in l1 if "test accuracy" in l][0].split()[-1]
    assert t0 == t1                        # both replicas hold identical parameters


@pytest.mark.slow
def test_three_workers_async_counts_all_updates():
    ports = [free_port() for _ in range(4)]
    hosts = ["--ps_hosts=127.0.0.1:%d" % ports[3], "--worker_hosts=" + ",".join("127.0.0.1:%d" % p for p in ports[:3]),
             "--train_steps=40", "--staleness=2"] + COMMON
    procs = [_launch(["--job_name=worker", "--task_index=%d" % i] + hosts) for i in (2, 1, 0)]
    ps = _launch(["--job_name=ps", "--task_index=0"] + hosts)
    outs = [_finish(p) for p in procs] + [_finish(ps)]
    total_steps = 0
    for lines in outs[:3]:
        st = _steps(lines)
        assert st, lines
        total_steps += len(st)
        gs = [s[1] for s in st]
        assert gs == sorted(gs) and gs[-1] >= 40          # stop rule on the shared counter
        assert all(g <= 40 + 3 * 3 for g in gs)
    # global_step counts the updates of all workers: nobody needs 40 local steps
    assert 40 <= total_steps <= 40 + 3 * 4


def test_one_worker_sync_r3_logs_every_contribution_and_counts_updates():
    """replicas_to_aggregate=3 on one worker: three gradient contributions (local steps) per aggregated update;
    every contribution prints its line with the global_step fetched with it, and exactly train_steps updates run."""
    lines = _finish(_launch(["--job_name=worker", "--task_index=0", "--worker_hosts=127.0.0.1:%d" % free_port(),
                             "--ps_hosts=", "--train_steps=7", "--sync_replicas", "--replicas_to_aggregate=3"] + COMMON))
    steps = _steps(lines)
    assert [s[0] for s in steps] == list(range(1, 22))                       # 7 updates x 3 contributions
    assert [s[1] for s in steps] == [1 + i // 3 for i in range(21)]          # one increment per aggregate
    assert steps[-1][2] < steps[0][2]


@pytest.mark.slow
def test_two_workers_sync_r3_runs_exactly_train_steps_updates():
    port_w0, port_w1, port_ps = free_port(), free_port(), free_port()
    hosts = ["--ps_hosts=127.0.0.1:%d" % port_ps, "--worker_hosts=127.0.0.1:%d,127.0.0.1:%d" % (port_w0, port_w1),
             "--sync_replicas", "--replicas_to_aggregate=3", "--train_steps=6"] + COMMON
    ps = _launch(["--job_name=ps", "--task_index=0"] + hosts)
    w1 = _launch(["--job_name=worker", "--task_index=1"] + hosts)
    w0 = _launch(["--job_name=worker", "--task_index=0"] + hosts)
    l0, l1, _ = _finish(w0), _finish(w1), _finish(ps)
    for st in (_steps(l0), _steps(l1)):
        assert [s[0] for s in st] == list(range(1, 13))                      # 6 updates x 2 rounds per worker
        assert [s[1] for s in st] == [1 + i // 2 for i in range(12)]
    t0 = [l for l in l0 if "test accuracy" in l][0].split()[-1]
    t1 = [l for l in l1 if "test accuracy" in l][0].split()[-1]
    assert t0 == t1


@pytest.mark.parametrize("extra,steps", [(["--sync_replicas", "--validation_every=10"], 47), (["--validation_every=7", "--steps_per_launch=5"], 30),
                                         (["--sync_replicas", "--validation_every=1000", "--nopost_update_accuracy"], 33)])
def test_loop_that_launches_ahead_prints_what_the_plain_loop_prints(extra, steps):
    """The control flow of the GPU loop that keeps two launches in flight (train.TrainingLoop._run_overlapped), run on
    the CPU: every line it prints is the plain loop's."""
    args = ["--single_process", "--train_steps=%d" % steps, "--synthetic_data", "--synthetic_train_size=2000", "--device=cpu",
            "--learning_rate=0.5", "--seed=7"] + extra

    def run(mode):
        env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONUNBUFFERED="1", DTF_LOOP_OVERLAP=mode)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "distributed.py")] + args, capture_output=True, text=True,
                           env=env, cwd=ROOT, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        return [l for l in p.stdout.splitlines() if not l.startswith("Training elapsed time") and "examples/sec" not in l]

    plain, ahead = run("0"), run("cpu")
    assert plain == ahead
    assert [s[0] for s in _steps(ahead)] == list(range(1, steps + 1))

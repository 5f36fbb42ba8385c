"""``python bench.py --gpus N`` with no launcher around it: the script starts its own N worker processes
(one per GPU, like the reference's launch recipe /root/reference/README.md:7-15), prints rank 0's ONE JSON
line and fails when a worker fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    return r


def _one_json_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                          # nothing but the result on stdout
    return json.loads(lines[0])


@pytest.mark.slow
def test_bench_starts_its_own_two_cpu_workers():
    d = _one_json_line(_bench(["--gpus", "2", "--cpu", "--steps", "6", "--warmup", "2"]))
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0
    assert d["config"]["global_batch"] == 200 and d["config"]["mode"] == "sync"
    assert abs(d["value"] - 200 * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-6 * d["value"]


@pytest.mark.slow
def test_bench_exits_nonzero_when_a_worker_fails():
    # an impossible configuration makes every worker raise; the parent must not print a result
    r = _bench(["--gpus", "2", "--cpu", "--steps", "2", "--warmup", "0", "--config", "mlp_sync", "--optimizer", "nonsense"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_more_ranks_than_gpus_without_the_rehearsal_switch():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"DTF_SHARE_GPU": "0"})
    assert r.returncode != 0 and "GPU" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["mlp_sync", "mlp_async"])
def test_bench_starts_its_own_two_gpu_workers_sharing_the_gpu(config):
    """Two ranks on the one GPU of the test box (gloo carries the process group, the peer-mapped kernels carry
    the updates): the path the 8-GPU scaling run takes, launcher included."""
    d = _one_json_line(_bench(["--gpus", "2", "--steps", "60", "--warmup", "20", "--config", config],
                              {"DTF_SHARE_GPU": "1", "DTF_DIST_BACKEND": "gloo", "DTF_P2P_ALLREDUCE": "1", "DTF_PEER_ASYNC": "1"}))
    assert d["n_gpus"] == 2 and d["steps"] == 60 and d["value"] > 0
    assert d["config"]["global_batch"] == 200


@pytest.mark.gpu
def test_bench_runs_the_one_launch_step_at_hidden_256_batch_256():
    """--hidden_units / --batch_size are free integers in the reference (/root/reference/distributed.py:11-13): up to
    256 / 256 one worker keeps the one-launch pipelined step."""
    d = _one_json_line(_bench(["--steps", "40", "--warmup", "10", "--hidden_units", "256", "--batch_size", "256",
                               "--target_loss", "0"]))
    assert d["config"]["step_program"] == "one launch per step, pipelined (mlp_pipe.hip)"
    assert d["config"]["global_batch"] == 256 and d["value"] > 0


def test_bench_launcher_stops_its_workers_when_it_is_told_to_stop():
    """SIGTERM to the self-launching bench.py ends the worker processes it started (nothing is left behind for the
    next job on the machine)."""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu", "--steps", "100000",
                          "--warmup", "1", "--no_second_mode"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 120
    kids = []
    while time.time() < deadline and len(kids) < 2:             # wait until both workers exist
        out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        kids = [int(k) for k in out]
        time.sleep(0.2)
    assert len(kids) == 2, kids
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    deadline = time.time() + 20
    while time.time() < deadline and any(os.path.exists(f"/proc/{k}") for k in kids):
        time.sleep(0.2)
    assert not any(os.path.exists(f"/proc/{k}") for k in kids)


def test_launcher_counts_gpus_in_sysfs_without_the_hip_runtime(tmp_path, monkeypatch):
    """The parent's GPU count comes from the driver's topology files: nodes with SIMDs, narrowed by a visibility list."""
    sys.path.insert(0, ROOT)
    import bench
    import glob as globmod
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 256, 256, 0, 256]):                 # two CPU agents, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    real_glob, real_isdir = globmod.glob, os.path.isdir
    monkeypatch.setattr(globmod, "glob", lambda pat: real_glob(str(nodes / "*" / "properties")) if "kfd" in pat else real_glob(pat))
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == "/sys/class/kfd" else real_isdir(p))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.count_gpus_without_hip() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_hip() == 2
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    assert bench.count_gpus_without_hip() == 0


@pytest.mark.slow
@pytest.mark.parametrize("extra", [[], ["--config", "mlp_async"], ["--replicas_to_aggregate", "6"],
                                   ["--replicas_to_aggregate", "12"]],
                         ids=["sync", "async", "R6_backup_workers", "R12_two_rounds"])
def test_bench_at_world_eight_on_cpu(extra):
    """The world size of the target node (8 workers, /root/reference/README.md:7-15 has 4), as gloo processes: sync,
    async, fewer replicas to aggregate than workers (two gradients dropped per update) and more (two rounds)."""
    d = _one_json_line(_bench(["--gpus", "8", "--cpu", "--steps", "6", "--warmup", "2", "--target_loss", "0"] + extra,
                              timeout=600))
    assert d["n_gpus"] == 8 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0
    assert d["config"]["global_batch"] == 800
    if "--replicas_to_aggregate" in extra:
        r = int(extra[1])
        assert f"R={r}" in d["config"]["parallelism"]
        # examples per update = R x batch (distributed.py:92-100)
        assert abs(d["value"] - 100 * r * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-6 * d["value"]
    elif extra:
        assert d["config"]["mode"] == "async" and d["global_step"] == 1 + 8 * 8      # every worker's update counts
    else:
        assert d["async_global_step"] == 1 + 8 * 8


@pytest.mark.slow
def test_bench_reports_steps_to_target_loss():
    """Second half of the headline metric, in the result line: steps until the printed training loss is below the
    target.  Two sync workers average two gradients per update: never more updates than one worker needs (+ noise)."""
    one = _one_json_line(_bench(["--gpus", "1", "--cpu", "--steps", "4", "--warmup", "1", "--target_loss", "2.2"]))
    two = _one_json_line(_bench(["--gpus", "2", "--cpu", "--steps", "4", "--warmup", "1", "--target_loss", "2.2"]))
    for d in (one, two):
        assert d["target_loss"] == 2.2 and d["steps_to_target"] > 10
        assert d["examples_to_target"] == d["steps_to_target"] * 100 * d["n_gpus"]
    assert two["steps_to_target"] <= one["steps_to_target"] + 8
    assert two["async_steps_to_target"] > 10 and two["async_examples_to_target"] == two["async_steps_to_target"] * 100


def test_a_follow_up_segment_that_does_not_come_back_costs_no_result():
    """The timed segment's line is written (without steps_to_target / async_value) and the run ends with status 0 when
    the untimed follow-up segments exceed their limit - here a limit no run can meet."""
    r = _bench(["--gpus", "2", "--cpu", "--steps", "6", "--warmup", "2"], {"DTF_BENCH_FOLLOWUP_LIMIT_S": "0.0001"})
    d = _one_json_line(r)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 6
    assert "not finished" in d["follow_up_segments"]
    assert "steps_to_target" not in d and "async_value" not in d
    assert "follow-up segments not finished" in r.stderr

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

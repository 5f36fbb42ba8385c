"""Supervisor: chief init-or-restore, non-chief readiness polling, periodic checkpoints, step-rate summary."""
import json
import os
import threading
import time

import torch
import torch.distributed as dist

from dist_utils import free_port
from distributed_tensorflow_amd.checkpoint import pyformat
from distributed_tensorflow_amd.checkpoint.saver import latest_checkpoint
from distributed_tensorflow_amd.models.mlp import build_mlp_state
from distributed_tensorflow_amd.supervisor.supervisor import Supervisor


def _state():
    return build_mlp_state(784, [100], 10, "cpu", lr=0.01)


def test_chief_initializes_then_restores_from_checkpoint(tmp_path):
    st = _state()
    sv = Supervisor(True, str(tmp_path), lambda: st.initialize(5), st, save_model_secs=600)
    sv.prepare_or_wait_for_session()
    assert st.initialized and st.global_step() == 1 and sv.restored_from is None
    st.state[0] = 4321
    st.var("hid_b").fill_(0.25)
    path = sv.save()
    assert path.endswith("model.ckpt-4321")
    # the file is a TF-Saver-layout bundle readable by the independent Python reader
    got = pyformat.read_bundle(path)
    assert sorted(got) == ["global_step", "hid_b", "hid_w", "sm_b", "sm_w"]
    assert int(got["global_step"]) == 4321 and got["global_step"].dtype.name == "int32" and got["hid_w"].shape == (784, 100)
    # a new chief pointed at the same directory resumes instead of re-initialising
    st2 = _state()
    called = []
    sv2 = Supervisor(True, str(tmp_path), lambda: called.append(1), st2, save_model_secs=600)
    sv2.prepare_or_wait_for_session()
    assert not called and sv2.restored_from == latest_checkpoint(str(tmp_path))
    assert st2.global_step() == 4321
    torch.testing.assert_close(st2.params, st.params)


def test_default_logdir_is_a_fresh_temp_dir():
    st = _state()
    a = Supervisor(True, None, lambda: st.initialize(1), st)
    b = Supervisor(True, None, lambda: st.initialize(1), st)
    assert a.logdir != b.logdir and os.path.isdir(a.logdir)      # tempfile.mkdtemp per process per run


def test_non_chief_polls_until_chief_is_ready():
    port = free_port()
    master = dist.TCPStore("127.0.0.1", port, None, True, wait_for_workers=False)
    client = dist.TCPStore("127.0.0.1", port, None, False)
    st_w = _state()
    worker = Supervisor(False, None, lambda: (_ for _ in ()).throw(AssertionError("non-chief must not init")), st_w,
                        recovery_wait_secs=0.05, store=client)
    t = threading.Thread(target=worker.prepare_or_wait_for_session)
    t.start()
    time.sleep(0.3)
    assert t.is_alive()                                            # still waiting: the chief has not initialised
    st_c = _state()
    chief = Supervisor(True, None, lambda: st_c.initialize(3), st_c, store=master)
    chief.prepare_or_wait_for_session()
    t.join(timeout=5)
    assert not t.is_alive() and worker.waited_polls >= 3


def test_periodic_checkpoint_and_step_rate(tmp_path):
    st = _state()
    sv = Supervisor(True, str(tmp_path), lambda: st.initialize(5), st, save_model_secs=0.2, save_summaries_secs=0.1)
    sv.prepare_or_wait_for_session()
    assert sv.maybe_checkpoint() is None                           # too early
    sv.report_step_rate(1)
    time.sleep(0.25)
    st.state[0] = 50
    assert sv.maybe_checkpoint().endswith("model.ckpt-50")
    rate = sv.report_step_rate(50)
    assert rate is not None and rate > 0
    sv.stop()
    ev = [json.loads(l) for l in open(tmp_path / "events.jsonl")]
    assert ev[0]["step"] == 50 and "global_step/sec" in ev[0]
    non_chief = Supervisor(False, str(tmp_path), lambda: None, st)
    assert non_chief.save() is None                                # only the chief writes checkpoints

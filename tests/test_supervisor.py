"""Supervisor: chief init-or-restore, non-chief readiness polling, periodic checkpoints, step-rate summary."""
import json
import os
import threading
import time

import pytest
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
    # the reference's five variables, plus the saving worker's local step (where its data pipeline stands)
    assert sorted(got) == ["dtf/local_step", "global_step", "hid_b", "hid_w", "sm_b", "sm_w"]
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


def test_event_file_is_tensorboard_framed(tmp_path):
    """The chief's global_step/sec summaries go to an events.out.tfevents.* file (TFRecord framing with masked
    crc32c, Event/Summary protobufs) written by the native runtime and parsed back by an independent reader."""
    import glob
    from distributed_tensorflow_amd.utils import native
    from distributed_tensorflow_amd.utils.tfevents import EventFileWriter, read_events
    if not native.available():
        pytest.skip("native runtime not built")
    w = EventFileWriter(str(tmp_path), now=1700000000.25)
    w.add_scalar("global_step/sec", 1234.5, 100, 1700000001.0)
    w.add_scalar("examples/sec", 8.5e6, 300, 1700000002.0)
    w.close()
    (path,) = glob.glob(str(tmp_path / "events.out.tfevents.1700000000.*"))
    ev = read_events(path)
    assert ev[0]["file_version"] == "brain.Event:2" and ev[0]["wall_time"] == 1700000000.25
    assert ev[1]["step"] == 100 and abs(ev[1]["scalars"]["global_step/sec"] - 1234.5) < 1e-3
    assert ev[2]["step"] == 300 and ev[2]["scalars"]["examples/sec"] == pytest.approx(8.5e6)
    # first record byte for byte: length 24, Event{wall_time, file_version}
    raw = open(path, "rb").read()
    assert raw[:8] == (24).to_bytes(8, "little") and raw[12:13] == b"\x09" and raw[21:23] == b"\x1a\x0d"
    # a flipped byte is detected
    bad = bytearray(raw)
    bad[40] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        read_events(path)

"""Checkpoint -> restore continues the SAME trajectory: 20 steps, save, restore into a fresh state, 20 more steps
must equal 40 uninterrupted steps bit for bit - SGD and Adam (slot variables under TF's names), one worker and two
gloo workers with the state sharded between them; the async send/recv engine resumes its counters and moments."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from dist_utils import free_port, run_ranks
from distributed_tensorflow_amd.checkpoint import pyformat
from distributed_tensorflow_amd.checkpoint.saver import latest_checkpoint, read_checkpoint_state


def _build(rank, world, optimizer, store=None, logdir=None, hidden=(32,)):
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    from distributed_tensorflow_amd.supervisor.supervisor import Supervisor
    ds = mnist.synthetic_data_sets(train_size=700, validation_size=50, test_size=50, seed=40 + rank)
    st = build_mlp_state(784, list(hidden), 10, "cpu", world_size=world, rank=rank, optimizer=optimizer,
                         lr=0.3 if optimizer == "sgd" else 0.01)
    sv = Supervisor(rank == 0, logdir, lambda: st.initialize(9), st, recovery_wait_secs=0.05, save_model_secs=600,
                    store=store, group_size=world)
    sv.prepare_or_wait_for_session()
    prog = MLPProgram(st, DeviceDataset(ds.train, "cpu", shuffle_seed=40 + rank), 50, list(hidden), fused_update=False,
                      post_update_accuracy=False)
    eng = SyncReplicasEngine(prog, st, collective="reduce_scatter" if world > 1 else "auto")
    sv.engine = eng
    return st, prog, eng, sv


def _resume_rank(rank, world, port, optimizer, logdir):
    store = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        store = dist.distributed_c10d._get_default_store()
    # uninterrupted run
    st_a, prog_a, eng_a, _ = _build(rank, world, optimizer, store, os.path.join(logdir, "a"))
    eng_a.run(40)
    # 20 steps, checkpoint (every rank calls save: with sharded Adam moments the gather is a collective)
    st_b, prog_b, eng_b, sv_b = _build(rank, world, optimizer, store, os.path.join(logdir, "b"))
    eng_b.run(20)
    path = sv_b.save()
    if world > 1:
        dist.barrier()
    # a new job on the same directory: restored, not initialised
    st_c, prog_c, eng_c, sv_c = _build(rank, world, optimizer, store, os.path.join(logdir, "b"))
    restored = sv_c.restored_from is not None
    gs_restored, ls_restored = st_c.global_step(), st_c.local_step()
    eng_c.run(20)
    out = {"a": st_a.params.numpy().copy(), "c": st_c.params.numpy().copy(), "restored": restored, "gs": st_c.global_step(),
           "gs_restored": gs_restored, "ls_restored": ls_restored, "path": path,
           "loss_a": [r.loss for r in prog_a.read_log(20, 20)], "loss_c": [r.loss for r in prog_c.read_log(20, 20)],
           "m_equal": (st_a.m is None) or bool((st_a.m == st_c.m).all() and (st_a.v == st_c.v).all())}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_one_worker_resume_is_bit_identical(optimizer, tmp_path):
    (r,) = run_ranks(_resume_rank, 1, optimizer, str(tmp_path))
    assert r["restored"] and r["gs_restored"] == 21 and r["ls_restored"] == 20 and r["gs"] == 41
    assert (r["a"] == r["c"]).all() and r["m_equal"]
    assert r["loss_a"] == r["loss_c"]
    got = pyformat.read_bundle(r["path"])
    if optimizer == "adam":
        # a TF AdamOptimizer's slot variables, by name
        for name in ("hid_w/Adam", "hid_w/Adam_1", "sm_b/Adam", "sm_b/Adam_1", "beta1_power", "beta2_power"):
            assert name in got
        assert got["hid_w/Adam"].shape == (784, 32) and abs(float(got["beta1_power"]) - 0.9 ** 21) < 1e-6
    else:
        assert "beta1_power" not in got


@pytest.mark.slow
@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_two_workers_resume_is_bit_identical(optimizer, tmp_path):
    res = run_ranks(_resume_rank, 2, optimizer, str(tmp_path), timeout=200)
    for r in res:
        assert r["restored"] == (r is res[0])                     # only the chief reads the checkpoint ...
        assert r["gs_restored"] == 21 and r["ls_restored"] == 20  # ... and everybody receives what it held
        assert (r["a"] == r["c"]).all() and r["m_equal"] and r["gs"] == 41
    assert (res[0]["c"] == res[1]["c"]).all()


def test_saver_continues_the_kept_list_of_a_previous_job(tmp_path):
    from distributed_tensorflow_amd.checkpoint.saver import Saver
    a = Saver(max_to_keep=3)
    t = {"global_step": np.int32(1), "w": np.ones(4, np.float32)}
    for gs in (1, 2):
        a.save(t, str(tmp_path), gs)
    b = Saver(max_to_keep=3)                                      # the restarted job
    for gs in (3, 4):
        b.save(t, str(tmp_path), gs)
    st = read_checkpoint_state(str(tmp_path))
    assert st["all_model_checkpoint_paths"] == ["model.ckpt-2", "model.ckpt-3", "model.ckpt-4"]
    assert not os.path.exists(tmp_path / "model.ckpt-1.index") and os.path.exists(tmp_path / "model.ckpt-2.index")
    assert latest_checkpoint(str(tmp_path)).endswith("model.ckpt-4")


def _async_resume_rank(rank, world, port, logdir):
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterServerEngine
    from distributed_tensorflow_amd.supervisor.supervisor import Supervisor
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    out = {}
    for phase in (0, 1):
        ds = mnist.synthetic_data_sets(train_size=700, validation_size=50, test_size=50, seed=40 + rank)
        st = build_mlp_state(784, [32], 10, "cpu", world_size=world, rank=rank, optimizer="adam", lr=0.01)
        sv = Supervisor(rank == 0, logdir, lambda: st.initialize(9), st, recovery_wait_secs=0.05, save_model_secs=600,
                        store=store, group_size=world)
        sv.prepare_or_wait_for_session()
        prog = MLPProgram(st, DeviceDataset(ds.train, "cpu", shuffle_seed=40 + rank), 50, [32], fused_update=False, gs_inc=0,
                          post_update_accuracy=False)
        eng = AsyncParameterServerEngine(prog, st, num_shards=2, staleness=1)
        sv.engine = eng
        eng.publish_state()
        out[f"gs_start{phase}"] = st.global_step()            # (the engine's own counters move with the first push of a peer)
        out[f"t_start{phase}"] = int(st.state[2].item())
        eng.run(10)
        eng.sync_params()                             # drains this worker's own pushes
        dist.barrier()                                # ... and now everybody's have been applied
        gs = eng.sync_params()
        if rank == 0:
            sv.save()                                 # the chief asks the other owner for its moments: not a collective
        dist.barrier()
        eng.finish()
        out[f"gs_end{phase}"] = gs
        dist.barrier()
    dist.destroy_process_group()
    return out


@pytest.mark.slow
def test_async_send_recv_engine_resumes_counters_and_moments(tmp_path):
    res = run_ranks(_async_resume_rank, 2, str(tmp_path), timeout=200)
    assert res[0]["gs_start0"] == 1 and res[0]["t_start0"] == 0
    # second job: global_step and Adam's update count continue where the checkpoint left them
    assert res[0]["gs_start1"] == 21 and res[0]["t_start1"] == 20
    got = pyformat.read_bundle(latest_checkpoint(str(tmp_path)))
    assert int(got["global_step"]) == 41
    assert np.abs(got["hid_w/Adam"]).sum() > 0 and np.abs(got["sm_w/Adam_1"]).sum() > 0     # both owners' slices arrived


def _async_cli_order_rank(rank, world, port, logdir):
    """The order of the training script's last lines: every worker on its own does run -> sync_params -> final
    checkpoint (chief) -> finish; no barrier in between, so the chief asks for the other owner's moments while that
    worker may already sit in finish()."""
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterServerEngine
    from distributed_tensorflow_amd.supervisor.supervisor import Supervisor
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    ds = mnist.synthetic_data_sets(train_size=700, validation_size=50, test_size=50, seed=40 + rank)
    st = build_mlp_state(784, [32], 10, "cpu", world_size=world, rank=rank, optimizer="adam", lr=0.01)
    sv = Supervisor(rank == 0, logdir, lambda: st.initialize(9), st, recovery_wait_secs=0.05, save_model_secs=600,
                    store=store, group_size=world)
    sv.prepare_or_wait_for_session()
    prog = MLPProgram(st, DeviceDataset(ds.train, "cpu", shuffle_seed=40 + rank), 50, [32], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = AsyncParameterServerEngine(prog, st, num_shards=2, staleness=0)
    sv.engine = eng
    eng.publish_state()
    eng.run(6 if rank == 0 else 3)                    # the chief is the last to finish training
    eng.sync_params()
    sv.stop()
    everyone = eng.finish(timeout_s=60)
    refused = False
    try:
        eng.gather_moments()
    except RuntimeError:
        refused = True
    dist.barrier()
    dist.destroy_process_group()
    return {"everyone": everyone, "refused": refused}


@pytest.mark.slow
def test_async_final_checkpoint_in_the_training_scripts_order(tmp_path):
    res = run_ranks(_async_cli_order_rank, 2, str(tmp_path), timeout=200)
    assert all(r["everyone"] and r["refused"] for r in res)
    got = pyformat.read_bundle(latest_checkpoint(str(tmp_path)))
    # both owners' slot variables are in the final checkpoint (sm_w lies in the second shard)
    assert np.abs(got["hid_w/Adam"]).sum() > 0 and np.abs(got["sm_w/Adam_1"]).sum() > 0

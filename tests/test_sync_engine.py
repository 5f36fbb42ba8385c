"""SyncReplicas semantics on gloo: R == N, R < N (backup workers), R > N; averaged, not summed;
one global_step increment per aggregated update."""
import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from dist_utils import run_ranks

LR = 0.5
B = 50


def _sync_worker(rank, world, port, R, collective, steps, slow_rank):
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ds = mnist.synthetic_data_sets(train_size=600, validation_size=50, test_size=50, seed=100 + rank)
    st = build_mlp_state(784, [16], 10, "cpu", world_size=world, rank=rank, lr=LR)
    init = oracle.init_params(16, seed=3)
    st.load_values({k: torch.from_numpy(v).float() for k, v in init.items()})
    fast = R == world
    prog = MLPProgram(st, DeviceDataset(ds.train, "cpu", shuffle_seed=100 + rank), B, [16], fused_update=False,
                      gs_inc=1 if fast else 0, post_update_accuracy=False)
    delay = (lambda step: 0.4) if rank == slow_rank else (lambda step: 0.0)
    eng = SyncReplicasEngine(prog, st, replicas_to_aggregate=R, collective=collective, arrival_delay=None if fast else delay)
    eng.run(steps)
    out = {"params": {n: st.var(n).clone().numpy() for n in st.names()}, "gs": st.global_step(), "ls": st.local_step(),
           "dropped": eng.dropped}
    dist.barrier()
    dist.destroy_process_group()
    return out


def _batches(rank, n_contrib):
    from distributed_tensorflow_amd.data import mnist
    ds = mnist.synthetic_data_sets(train_size=600, validation_size=50, test_size=50, seed=100 + rank)
    out = []
    for s in range(n_contrib):
        rows = ds.train.batch_rows(s * B, B)
        out.append((ds.train.images[rows].astype(np.float64), ds.train.labels[rows].astype(np.float64)))
    return out


def _check_equal_replicas(res):
    for r in res[1:]:
        for k in res[0]["params"]:
            np.testing.assert_array_equal(r["params"][k], res[0]["params"][k])


@pytest.mark.slow
@pytest.mark.parametrize("collective,world,steps", [("all_reduce", 2, 6), ("reduce_scatter", 2, 6), ("reduce_scatter", 4, 3),
                                                    ("all_reduce", 8, 3)])
def test_r_equals_n_is_the_mean_of_worker_gradients(collective, world, steps):
    res = run_ranks(_sync_worker, world, world, collective, steps, -1)
    _check_equal_replicas(res)
    params = oracle.init_params(16, seed=3)
    data = [_batches(r, steps) for r in range(world)]
    for s in range(steps):
        grads = [oracle.loss_and_grads(params, *data[r][s])[2] for r in range(world)]
        mean = {k: sum(g[k] for g in grads) / world for k in params}
        params = oracle.sgd(params, mean, LR)
    for k, v in params.items():
        np.testing.assert_allclose(res[0]["params"][k], v, rtol=5e-4, atol=5e-6)
    assert all(r["gs"] == 1 + steps and r["ls"] == steps for r in res)     # one increment per aggregate


@pytest.mark.slow
def test_backup_worker_gradient_is_dropped_when_r_less_than_n():
    world, R, steps, slow = 3, 2, 4, 2
    res = run_ranks(_sync_worker, world, R, "all_reduce", steps, slow)
    _check_equal_replicas(res)
    assert res[slow]["dropped"] == steps and res[0]["dropped"] == 0 and res[1]["dropped"] == 0
    params = oracle.init_params(16, seed=3)
    data = [_batches(r, steps) for r in range(world)]
    for s in range(steps):
        grads = [oracle.loss_and_grads(params, *data[r][s])[2] for r in (0, 1)]      # the slow rank never makes it
        mean = {k: sum(g[k] for g in grads) / R for k in params}
        params = oracle.sgd(params, mean, LR)
    for k, v in params.items():
        np.testing.assert_allclose(res[0]["params"][k], v, rtol=5e-4, atol=5e-6)
    assert all(r["gs"] == 1 + steps for r in res)


@pytest.mark.slow
def test_r_greater_than_n_accumulates_several_rounds():
    world, R, steps = 2, 3, 3          # every update: both workers once, then the faster one again
    res = run_ranks(_sync_worker, world, R, "reduce_scatter", steps, 1)
    _check_equal_replicas(res)
    params = oracle.init_params(16, seed=3)
    data = [_batches(r, 2 * steps) for r in range(world)]
    for s in range(steps):
        contrib = [data[0][2 * s], data[1][2 * s], data[0][2 * s + 1]]             # rank 1 is slow in the second pass
        grads = [oracle.loss_and_grads(params, *c)[2] for c in contrib]
        mean = {k: sum(g[k] for g in grads) / R for k in params}
        params = oracle.sgd(params, mean, LR)
    for k, v in params.items():
        np.testing.assert_allclose(res[0]["params"][k], v, rtol=5e-4, atol=5e-6)
    assert all(r["gs"] == 1 + steps and r["ls"] == 2 * steps for r in res)
    assert res[1]["dropped"] == steps


def test_single_rank_sync_r3_accumulates_three_batches():
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    ds = mnist.synthetic_data_sets(train_size=600, validation_size=50, test_size=50, seed=100)
    st = build_mlp_state(784, [16], 10, "cpu", lr=LR)
    init = oracle.init_params(16, seed=3)
    st.load_values({k: torch.from_numpy(v).float() for k, v in init.items()})
    prog = MLPProgram(st, DeviceDataset(ds.train, "cpu", shuffle_seed=100), B, [16], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = SyncReplicasEngine(prog, st, replicas_to_aggregate=3)
    eng.run(2)
    params, data = init, _batches(0, 6)
    for s in range(2):
        grads = [oracle.loss_and_grads(params, *data[3 * s + j])[2] for j in range(3)]
        params = oracle.sgd(params, {k: sum(g[k] for g in grads) / 3 for k in params}, LR)
    for k, v in params.items():
        np.testing.assert_allclose(st.var(k).numpy(), v, rtol=5e-4, atol=5e-6)
    assert st.global_step() == 3 and st.local_step() == 6

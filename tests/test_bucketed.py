"""Bucketed aggregation behind the backward pass: bucket geometry, and two-rank runs whose replicas must be
bit-identical to the single reduce-scatter after the backward pass (same sums, different partition)."""
import numpy as np
import pytest
import torch
import torch.distributed as dist

from dist_utils import run_ranks
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import build_mlp_state
from distributed_tensorflow_amd.models.resnet import ResNet, TorchModuleProgram
from distributed_tensorflow_amd.models.wide_mlp import WideMLPProgram
from distributed_tensorflow_amd.parallel import bucketed
from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine

HIDDEN = [64, 48, 32]


def test_bucket_ranges_tile_the_flat_buffer():
    st = build_mlp_state(784, HIDDEN, 10, "cpu", world_size=4, rank=1)
    groups = [["hid_w", "hid_b"], ["hid2_w", "hid2_b"], ["hid3_w", "hid3_b"], ["sm_w", "sm_b"]]
    r = bucketed.bucket_ranges(st, groups)
    assert r[0][0] == 0 and r[-1][1] == st.layout.total
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert all((hi - lo) % 4 == 0 for lo, hi in r)
    with pytest.raises(ValueError):
        bucketed.bucket_ranges(st, groups[:-1])                              # a variable without a bucket
    with pytest.raises(ValueError):
        bucketed.bucket_ranges(st, [["hid_w", "hid2_w"]] + groups[2:])        # not consecutive
    bu = bucketed.BucketedUpdate(st, groups, None, 1.0)
    assert sum(hi - lo for lo, hi in bu.own) == st.layout.shard_size
    assert bu.local[-1][1] == st.layout.shard_size


def test_group_by_bytes_covers_every_variable_once():
    st = build_mlp_state(784, HIDDEN, 10, "cpu")
    g = bucketed.group_by_bytes(st, 64 << 10)
    assert [n for grp in g for n in grp] == st.names()
    assert len(g) > 1
    bucketed.bucket_ranges(st, g)


def test_missing_bucket_signal_is_an_error():
    st = build_mlp_state(784, [64], 10, "cpu")
    bu = bucketed.BucketedUpdate(st, [["hid_w", "hid_b"], ["sm_w", "sm_b"]], None, 1.0)
    bu.ready(1)
    with pytest.raises(RuntimeError):
        bu.finish()
    with pytest.raises(RuntimeError):
        bu.ready(1)


def _wide_rank(rank, world, port, collective, optimizer, device="cpu", dtype=torch.float32):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ds = mnist.synthetic_data_sets(train_size=512, validation_size=10, test_size=10, seed=rank)
    ref = build_mlp_state(784, HIDDEN, 10, "cpu", world_size=world, rank=rank, optimizer=optimizer)
    ref.initialize(seed=5)
    st = build_mlp_state(784, HIDDEN, 10, device, world_size=world, rank=rank, optimizer=optimizer,
                         lr=0.05 if optimizer == "sgd" else 1e-3)
    st.load_values({n: ref.var(n) for n in ref.names()})
    prog = WideMLPProgram(st, DeviceDataset(ds.train, device, shuffle_seed=rank), 32, HIDDEN, loss_mode=ops.LOSS_STANDARD,
                          dtype=dtype)
    eng = SyncReplicasEngine(prog, st, collective=collective)
    eng.run(5, 4)
    assert (eng.bucketed is not None) == (collective == "bucketed")
    if device != "cpu":
        torch.cuda.synchronize()
    out = (st.params.cpu().numpy().copy(), st.global_step(), [r.loss for r in prog.read_log(0, 5)])
    dist.barrier()
    dist.destroy_process_group()
    return out


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_two_ranks_bucketed_equals_single_reduce_scatter(optimizer):
    a = run_ranks(_wide_rank, 2, "bucketed", optimizer)
    b = run_ranks(_wide_rank, 2, "reduce_scatter", optimizer)
    assert np.array_equal(a[0][0], a[1][0])                 # replicas identical
    assert np.array_equal(a[0][0], b[0][0])                 # and equal to the unbucketed engine
    assert a[0][1] == b[0][1] == 6
    assert a[0][2][-1] < a[0][2][0]


def _resnet_rank(rank, world, port, collective):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = ResNet(layers=(1, 1, 1, 1), num_classes=10, width=8)
    g = torch.Generator().manual_seed(10 + rank)
    x, y = torch.rand(16, 3, 32, 32, generator=g), torch.randint(0, 10, (16,), generator=g)
    prog = TorchModuleProgram(model, "cpu", x, y, batch=8, world_size=world, rank=rank, optimizer="adam", lr=1e-3,
                              channels_last=False, bucket_bytes=64 << 10)
    eng = SyncReplicasEngine(prog, prog.state, collective=collective)
    eng.run(3)
    nb = len(eng.bucketed.ranges) if eng.bucketed is not None else 0
    out = (prog.state.params.numpy().copy(), prog.state.global_step(), nb)
    dist.barrier()
    dist.destroy_process_group()
    return out


def test_two_ranks_module_hooks_release_buckets_in_one_order():
    a = run_ranks(_resnet_rank, 2, "bucketed")
    b = run_ranks(_resnet_rank, 2, "reduce_scatter")
    assert a[0][2] > 2                                      # several buckets, released from autograd hooks
    assert np.array_equal(a[0][0], a[1][0])
    assert np.array_equal(a[0][0], b[0][0])
    assert a[0][1] == 4

"""The one-launch pipelined step of N synchronous workers (csrc/kernels/mlp_pipe_multi.hip): one rank exchanging with
itself through its own region against the two-launch step and the NumPy oracle; 2, 3 and 8 ranks sharing the test
box's GPU (IPC-mapped regions of the same device) against the NumPy oracle of mean-gradient SGD
(/root/reference/distributed.py:91-102) and against the sharded exchange of the two-launch step."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from dist_utils import run_ranks
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state

pytestmark = pytest.mark.gpu


def _one_rank(pipelined, hidden=100, batch=100, lr=0.5, post=True, loss_mode=ops.LOSS_REFERENCE, n_train=2000, in_dim=784,
              classes=10, use_graphs=None, seed=5):
    """One worker: through its own exchange region (`pipelined`), or the two-launch step with SGD in its epilogue."""
    from distributed_tensorflow_amd.parallel.pipe_exchange import PipeExchange
    dev = torch.device("cuda", 0)
    if in_dim == 784 and classes == 10:
        train = mnist.synthetic_data_sets(train_size=n_train, validation_size=100, test_size=100, seed=seed).train
    else:
        rng = np.random.default_rng(seed)
        images = rng.random((n_train, in_dim), dtype=np.float32)
        labels = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, n_train)]
        train = mnist.DataSet(images, labels)
    st = build_mlp_state(in_dim, [hidden], classes, dev, lr=lr)
    st.initialize(seed=11)
    old = os.environ.get("DTF_PIPELINED_STEP")
    os.environ["DTF_PIPELINED_STEP"] = "0"            # the constructor builds the two-launch program either way
    try:
        data = DeviceDataset(train, dev, shuffle_seed=seed)
        prog = MLPProgram(st, data, batch, [hidden], classes=classes, loss_mode=loss_mode, post_update_accuracy=post,
                          use_graphs=use_graphs, fused_update=not pipelined)
    finally:
        if old is None:
            os.environ.pop("DTF_PIPELINED_STEP", None)
        else:
            os.environ["DTF_PIPELINED_STEP"] = old
    if pipelined:
        data.enable_peer_shuffles([seed], 0)
        prog.use_pipe_exchange(PipeExchange(dev, 0, 1, timeout_s=5.0), lr)
        assert prog.pipe and prog.pipe_multi is not None
    return st, prog


def _compare(a, b, steps, runs, graph_steps=8, post=True, loss_tol=2e-5, param_tol=2e-5):
    (sta, pa), (stb, pb) = a, b
    for n in runs:
        pa.run(n, graph_steps)
        pb.run(n, graph_steps)
    torch.cuda.synchronize()
    assert not pa.pipe_multi.timed_out()
    ra, rb = pa.read_log(0, steps), pb.read_log(0, steps)
    assert sta.global_step() == stb.global_step() == steps + 1
    assert sta.local_step() == stb.local_step() == steps
    for s in range(steps):
        assert ra[s].global_step == rb[s].global_step == s + 1
        assert abs(ra[s].loss - rb[s].loss) < loss_tol, (s, ra[s].loss, rb[s].loss)
        assert abs(ra[s].train_correct - rb[s].train_correct) <= 1, (s, ra[s].train_correct, rb[s].train_correct)
        if post and s < steps - 1:
            assert abs(ra[s].post_correct - rb[s].post_correct) <= 1, (s, ra[s].post_correct, rb[s].post_correct)
    for name in sta.names():
        torch.testing.assert_close(sta.var(name), stb.var(name), rtol=1e-4, atol=param_tol, msg=name)
    assert abs(pa.post_correct_of_last_step() - pb.post_correct_of_last_step()) <= 1
    assert ra[-1].loss < ra[0].loss


@pytest.mark.parametrize("loss", ["reference", "standard"])
def test_one_rank_through_its_own_region_follows_the_two_launch_trajectory(loss):
    mode = ops.LOSS_REFERENCE if loss == "reference" else ops.LOSS_STANDARD
    _compare(_one_rank(True, loss_mode=mode), _one_rank(False, loss_mode=mode), steps=45, runs=(19, 1, 25))


def test_one_rank_across_epoch_boundaries_and_without_accuracy_rows():
    _compare(_one_rank(True, n_train=1000), _one_rank(False, n_train=1000), steps=47, runs=(23, 24))
    _compare(_one_rank(True, post=False), _one_rank(False, post=False), steps=20, runs=(20,), post=False)


def test_one_rank_eager_single_steps():
    _compare(_one_rank(True, use_graphs=False), _one_rank(False, use_graphs=False), steps=5, runs=(3, 2))


@pytest.mark.parametrize("hidden,batch,classes,in_dim", [(31, 32, 5, 20), (127, 128, 16, 64), (64, 50, 10, 784)])
def test_one_rank_other_shapes(hidden, batch, classes, in_dim):
    _compare(_one_rank(True, hidden=hidden, batch=batch, classes=classes, in_dim=in_dim, n_train=16 * batch),
             _one_rank(False, hidden=hidden, batch=batch, classes=classes, in_dim=in_dim, n_train=16 * batch),
             steps=21, runs=(21,), graph_steps=6)


def _sync_ranks(rank, world, port, mode, steps, runs, want_params=True):
    """`world` sync workers sharing the GPU, the SAME data set on every rank, independent shuffles."""
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    os.environ["DTF_SHARE_GPU"] = "1"
    os.environ["DTF_P2P_TIMEOUT_S"] = "20"
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=2000, validation_size=100, test_size=100, seed=5)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    params = oracle.init_params(100, seed=11)
    st.load_values({k: torch.from_numpy(v).float() for k, v in params.items()})
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=40 + rank), 100, [100], fused_update=False)
    eng = SyncReplicasEngine(prog, st, collective=mode)
    for n in runs:
        eng.run(n, 8)
    torch.cuda.synchronize()
    first = max(0, steps - 64)                       # (the log ring holds 2048 records)
    recs = prog.read_log(first, steps - first)
    out = {"mode": eng.p2p_mode, "program": "pipe" if prog.pipe_multi is not None else "mlp2", "gs": st.global_step(),
           "loss": [r.loss for r in recs], "correct": [r.train_correct for r in recs],
           "post": [r.post_correct for r in recs[:-1]], "transport": eng.transport, "primes": prog.pipe_primes}
    if want_params:
        out["params"] = st.params.cpu().numpy().copy()
        out["named"] = {k: st.var(k).cpu().numpy().copy() for k in st.names()}
    else:
        out["digest"] = (float(st.params.double().sum().item()), float(st.params.double().square().sum().item()))
    dist.barrier()
    eng.close_transports()
    dist.destroy_process_group()
    return out


def _oracle_sync(world, steps):
    """Mean-gradient SGD of `world` workers (shuffle seeds 40 + rank) in NumPy float64."""
    ds = mnist.synthetic_data_sets(train_size=2000, validation_size=100, test_size=100, seed=5)
    params = oracle.init_params(100, seed=11)
    from distributed_tensorflow_amd.data.mnist import epoch_permutation
    losses = [[] for _ in range(world)]
    posts = [[] for _ in range(world)]
    n = ds.train.num_examples

    def rows(rank, s):
        pos = np.arange(s * 100, (s + 1) * 100)
        ep, off = pos // n, pos % n
        out = np.empty(100, dtype=np.int64)
        for e in np.unique(ep):
            msk = ep == e
            out[msk] = epoch_permutation(40 + rank, int(e), n)[off[msk]]
        return out

    for s in range(steps):
        total, batches = None, []
        for r in range(world):
            rr = rows(r, s)
            x, y = ds.train.images[rr].astype(np.float64), ds.train.labels[rr].astype(np.float64)
            l, _, grads = oracle.loss_and_grads(params, x, y, "reference")
            losses[r].append(l)
            batches.append((x, y))
            total = grads if total is None else {k: total[k] + grads[k] for k in grads}
        params = oracle.sgd(params, {k: v / world for k, v in total.items()}, 0.5)
        for r, (x, y) in enumerate(batches):
            posts[r].append(round(oracle.accuracy(params, x, y) * 100))
    return params, losses, posts


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_the_gpu_match_the_oracle_of_mean_gradient_sgd(world):
    steps = 30
    res = run_ranks(_sync_ranks, world, "p2p_pipelined", steps, (11, 1, 18), timeout=240)
    want, losses, posts = _oracle_sync(world, steps)
    for r, out in enumerate(res):
        assert out["mode"] == "pipelined" and out["program"] == "pipe" and out["gs"] == steps + 1
        assert (out["params"] == res[0]["params"]).all()            # replicas: the same sums in the same order
        for s in range(steps):
            assert abs(out["loss"][s] - losses[r][s]) < 3e-5, (r, s, out["loss"][s], losses[r][s])
        for s in range(steps - 1):
            assert abs(out["post"][s] - posts[r][s]) <= 1
        for k, v in want.items():
            np.testing.assert_allclose(out["named"][k], v, rtol=3e-4, atol=3e-6)
        assert out["primes"] == 1                                    # later runs continue the pipeline


@pytest.mark.parametrize("world", [2, 3])
def test_long_graph_replayed_run_keeps_replicas_identical_and_equals_the_sharded_exchange(world):
    steps = 4000
    a = run_ranks(_sync_ranks, world, "p2p_pipelined", steps, (steps,), timeout=300)
    b = run_ranks(_sync_ranks, world, "p2p_sharded", steps, (steps,), timeout=300)
    assert all(x["program"] == "pipe" for x in a) and all(x["program"] == "mlp2" and x["mode"] == "sharded" for x in b)
    for x in a:
        assert (x["params"] == a[0]["params"]).all() and x["gs"] == steps + 1
    # another association of the same sums: fp32 rounding apart after 4000 updates at lr 0.5
    np.testing.assert_allclose(a[0]["params"], b[0]["params"], rtol=2e-3, atol=2e-4)
    for r in range(world):
        assert max(abs(p - q) for p, q in zip(a[r]["loss"][-64:], b[r]["loss"][-64:])) < 2e-4


def test_eight_ranks_sharing_the_gpu():
    """The world size of the target node: flag / box indexing for eight sources, K split of the weight update over
    eight waves, 112 KB of Gram strips per row workgroup; only the row workgroups of a launch ever wait, so eight
    ranks' launches fit on one GPU together."""
    steps = 12
    res = run_ranks(_sync_ranks, 8, "p2p_pipelined", steps, (5, 7), False, timeout=400)
    want, losses, _ = _oracle_sync(8, steps)
    for r, out in enumerate(res):
        assert out["mode"] == "pipelined" and out["gs"] == steps + 1
        assert out["digest"] == res[0]["digest"]
        for s in range(steps):
            assert abs(out["loss"][s] - losses[r][s]) < 3e-5, (r, s)


def _diverging_replica(rank, world, port):
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    os.environ["DTF_SHARE_GPU"] = "1"
    os.environ["DTF_P2P_TIMEOUT_S"] = "20"
    import warnings
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=2000, validation_size=100, test_size=100, seed=5)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=40 + rank), 100, [100], fused_update=False)
    eng = SyncReplicasEngine(prog, st, collective="all_reduce")     # two workers: the pipelined exchange by default
    first = eng.p2p_mode
    eng.run(6, 4)
    torch.cuda.synchronize()
    if rank == 1:
        st.params[5] += 1.0              # an edit of the flat tensor on one rank: not a reason to prime (see _pipe_token)
    eng._replica_checks_left = 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.run(6, 4)
    torch.cuda.synchronize()
    quiet = not any("replicas different" in str(x.message) for x in w)
    same = float(st.params[5].item())
    # what a machine that orders the peer traffic differently would leave behind: one rank's working copy off
    if rank == 1:
        prog.pbuf.wm[:, 5, 3] += 1.0
    eng._replica_checks_left = 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.run(6, 4)
    torch.cuda.synchronize()
    warned = any("pipelined exchange left the replicas different" in str(x.message) for x in w)
    eng.run(8, 4)
    torch.cuda.synchronize()
    out = (st.params.cpu().numpy().copy(), first, quiet, same, warned, eng.p2p_mode, eng.p2p_fused, st.global_step(),
           prog.pipe_multi is None)
    dist.barrier()
    eng.close_transports()
    dist.destroy_process_group()
    return out


def test_one_rank_off_is_noticed_and_the_ranks_continue_on_the_update_launch():
    a, b = run_ranks(_diverging_replica, 2, timeout=200)
    for x in (a, b):
        assert x[1] == "pipelined" and x[2]            # the flat-tensor edit neither primed one rank alone nor survived
        assert x[4] and x[5] == "replicated" and not x[6] and x[8] and x[7] == 27
    assert a[3] == b[3]
    assert (a[0] == b[0]).all()

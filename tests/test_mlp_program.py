"""Step program vs the NumPy oracle (CPU) and GPU vs CPU trajectories."""
import numpy as np
import pytest
import torch

import oracle
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state


@pytest.fixture
def one_rank_rccl_group():
    """A one-rank RCCL process group for the tests that exercise the collective code paths on a single GPU; torn
    down afterwards so that it neither leaks into later tests nor outlives the interpreter."""
    import torch.distributed as dist
    from dist_utils import init_one_rank_group
    created = not dist.is_initialized()
    if created:
        init_one_rank_group("nccl", device_id=torch.device("cuda", 0))
    yield
    if created and dist.is_initialized():
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _make(device, hidden=(100,), batch=100, lr=0.01, post=True, loss_mode=ops.LOSS_REFERENCE, n_train=1000, seed=5,
          fused=True, use_graphs=None):
    ds = mnist.synthetic_data_sets(train_size=n_train, validation_size=200, test_size=300, seed=seed)
    st = build_mlp_state(784, list(hidden), 10, device, lr=lr)
    init = oracle.init_params(hidden[0], seed=11) if len(hidden) == 1 else None
    if init is not None:
        st.load_values({k: torch.from_numpy(v).float() for k, v in init.items()})
    else:
        st.initialize(seed=11)
    dd = DeviceDataset(ds.train, device, shuffle_seed=seed)
    prog = MLPProgram(st, dd, batch, list(hidden), loss_mode=loss_mode, post_update_accuracy=post, fused_update=fused,
                      use_graphs=use_graphs)
    return ds, st, prog, init


@pytest.mark.parametrize("loss", ["reference", "standard"])
def test_cpu_program_matches_numpy_oracle(loss):
    mode = ops.LOSS_REFERENCE if loss == "reference" else ops.LOSS_STANDARD
    ds, st, prog, params = _make("cpu", lr=0.5, loss_mode=mode)
    steps = 25
    prog.run(steps)
    recs = prog.read_log(0, steps)
    for s in range(steps):
        rows = ds.train.batch_rows(s * 100, 100)
        x, y = ds.train.images[rows].astype(np.float64), ds.train.labels[rows].astype(np.float64)
        l, acc, grads = oracle.loss_and_grads(params, x, y, loss)
        params = oracle.sgd(params, grads, 0.5)
        assert recs[s].global_step == s + 1          # global_step starts at 1 (distributed.py:65)
        assert recs[s].local_step == s + 1
        assert abs(recs[s].loss - l) < 2e-5, (s, recs[s].loss, l)
        assert recs[s].train_correct == round(acc * 100)
        if s < steps - 1:
            assert recs[s].post_correct == round(oracle.accuracy(params, x, y) * 100)
    for k, v in params.items():
        np.testing.assert_allclose(st.var(k).numpy(), v, rtol=2e-4, atol=2e-6)
    assert st.global_step() == steps + 1
    assert prog.post_correct_of_last_step() == round(oracle.accuracy(params, x, y) * 100)


def test_double_softmax_loss_is_bounded():
    ds, st, prog, _ = _make("cpu", lr=2.0, n_train=500)
    prog.run(60)
    recs = prog.read_log(0, 60)
    assert abs(recs[0].loss - np.log(10)) < 0.02
    assert all(1.4611 <= r.loss <= 2.4612 for r in recs)    # -log(e/(e+9)) .. -log(1/(e+9))
    assert recs[-1].loss < 1.6


def test_epoch_rollover_batches_match_host_next_batch():
    ds, st, prog, _ = _make("cpu", batch=64, n_train=200)    # 200 rows, batch 64: epochs end inside batches
    prog.start()
    for s in range(10):
        want = ds.train.batch_rows(s * 64, 64)
        got = prog.ridx[s & 1][:64].numpy()
        np.testing.assert_array_equal(got, want)
        x, _ = ds.train.next_batch(64)
        np.testing.assert_array_equal(x, ds.train.images[want])
        prog.run(1)


def test_deep_mlp_cpu_matches_autograd():
    ds, st, prog, _ = _make("cpu", hidden=(64, 48, 32), lr=0.1, loss_mode=ops.LOSS_STANDARD, fused=True)
    w = {n: st.var(n).clone().requires_grad_(True) for n in st.names()}
    rows = torch.from_numpy(ds.train.batch_rows(0, 100)).long()
    x, y = torch.from_numpy(ds.train.images)[rows], torch.from_numpy(ds.train.labels)[rows]
    h = x
    for n in ("hid", "hid2", "hid3"):
        h = torch.relu(h @ w[f"{n}_w"] + w[f"{n}_b"])
    z = h @ w["sm_w"] + w["sm_b"]
    loss = -(y * torch.log_softmax(z, 1)).sum(1).mean()
    loss.backward()
    prog.run(1)
    rec = prog.read_log(0, 1)[0]
    assert abs(rec.loss - loss.item()) < 1e-5
    for n in st.names():
        torch.testing.assert_close(st.var(n), (w[n] - 0.1 * w[n].grad).detach(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_gpu_prepared_graphs_are_the_ones_the_run_replays():
    """prepare_graphs captures, without launching, exactly the graphs the next run needs: the run creates none and
    the trajectory is the one of a program that captured on the fly."""
    _, st_a, prog_a, _ = _make("cuda", lr=0.3, use_graphs=True)
    _, st_b, prog_b, _ = _make("cuda", lr=0.3, use_graphs=True)
    for prog in (prog_a, prog_b):
        prog.run(7, graph_steps=4)
    before = prog_a.state.global_step()
    sizes = list(prog_a._launch_sizes(23, 8, exact_tail=True))    # 8 + 8 + 7 (6 + 6 + 6 + 5 when the step looks further ahead)
    prog_a.prepare_graphs(23, graph_steps=8)
    torch.cuda.synchronize()
    assert prog_a.state.global_step() == before and prog_a.steps_enqueued == 7       # nothing ran
    n_graphs = len(prog_a._graphs)
    prog_a.run(23, graph_steps=8)
    prog_b.run(23, graph_steps=8)
    assert len(prog_a._graphs) == n_graphs           # the run replays what was prepared and captures nothing
    assert sum(sizes) == 23 and sizes[-1] in (7, 5)
    # the rest of the run is ONE graph of exactly that length (ending with the flush launch of a pipelined run)
    assert prog_a._graph_key(sizes[-1], 7 + 23 - sizes[-1], last=True) in prog_a._graphs
    ra, rb = prog_a.read_log(0, 30), prog_b.read_log(0, 30)
    assert st_a.global_step() == 31
    # (runs of the pipelined step continue the pipeline: however a trajectory is cut into graphs, it is the same bit for bit)
    assert [r.loss for r in ra] == [r.loss for r in rb]
    assert torch.equal(st_a.params, st_b.params)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,post,graphs", [((100,), True, True), ((100,), False, False), ((64, 48), True, True)])
def test_gpu_trajectory_matches_cpu(hidden, post, graphs):
    steps = 37
    _, st_c, prog_c, _ = _make("cpu", hidden=hidden, lr=0.3, post=post)
    _, st_g, prog_g, _ = _make("cuda", hidden=hidden, lr=0.3, post=post, use_graphs=graphs)
    if len(hidden) > 1:
        st_g.load_values({n: st_c.var(n) for n in st_c.names()})
    prog_c.run(steps)
    prog_g.run(steps, graph_steps=8)
    rc, rg = prog_c.read_log(0, steps), prog_g.read_log(0, steps)
    for a, b in zip(rc, rg):
        assert a.global_step == b.global_step and a.local_step == b.local_step
        assert abs(a.loss - b.loss) < 1e-4
        assert abs(a.train_correct - b.train_correct) <= 1
        if post and a.post_correct is not None:
            assert abs(a.post_correct - b.post_correct) <= 1
    for n in st_c.names():
        torch.testing.assert_close(st_g.var(n).cpu(), st_c.var(n), rtol=1e-3, atol=1e-5)
    assert st_g.global_step() == steps + 1 and st_g.local_step() == steps
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=200, test_size=300, seed=5)
    xt, yt = torch.from_numpy(ds.test.images), torch.from_numpy(ds.test.labels)
    assert abs(prog_g.accuracy(xt.cuda(), yt.cuda()) - prog_c.accuracy(xt, yt)) < 0.01
    assert prog_g.post_correct_of_last_step() == prog_c.post_correct_of_last_step()


@pytest.mark.gpu
@pytest.mark.parametrize("collective,optimizer", [("all_reduce", "sgd"), ("reduce_scatter", "sgd"), ("reduce_scatter", "adam")])
def test_gpu_rccl_collectives_in_captured_step(collective, optimizer, one_rank_rccl_group):
    """One-rank RCCL group: the reduce-scatter / all-gather / all-reduce calls of the sync engine,
    captured in the step hipGraph, must leave the trajectory identical to the fused local update."""
    import torch.distributed as dist
    from dist_utils import free_port
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    steps = 24
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=200, test_size=300, seed=5)

    def make(fused, coll):
        st = build_mlp_state(784, [100], 10, "cuda", lr=0.3 if optimizer == "sgd" else 1e-3, optimizer=optimizer)
        st.initialize(seed=11)
        prog = MLPProgram(st, DeviceDataset(ds.train, "cuda", shuffle_seed=5), 100, [100], fused_update=fused)
        eng = SyncReplicasEngine(prog, st, collective=coll or "auto", force_collectives=coll is not None)
        return st, prog, eng

    st_a, prog_a, eng_a = make(optimizer == "sgd", None)
    st_b, prog_b, eng_b = make(False, collective)
    eng_a.run(steps, 8)
    eng_b.run(steps, 8)
    torch.cuda.synchronize()
    ra, rb = prog_a.read_log(0, steps), prog_b.read_log(0, steps)
    for a, b in zip(ra, rb):
        assert abs(a.loss - b.loss) < 1e-5 and a.global_step == b.global_step
    torch.testing.assert_close(st_a.params, st_b.params, rtol=1e-5, atol=1e-6)
    assert st_b.global_step() == steps + 1


@pytest.mark.gpu
def test_gpu_cli_single_worker_runs_hip_path(capsys):
    from distributed_tensorflow_amd import flags as F, train
    from distributed_tensorflow_amd.ops import _lib
    train.define_flags()
    F.FLAGS.reset()
    F.FLAGS.parse(["--single_process", "--synthetic_data", "--synthetic_train_size=3000", "--train_steps=300",
                   "--learning_rate=0.5", "--seed=3", "--validation_every=100", "--device=gpu", "--save_model_secs=0"])
    assert train.main([]) == 0
    out = capsys.readouterr().out.splitlines()
    steps = [l for l in out if "traing step" in l]
    assert len(steps) == 300 and steps[-1].startswith("Worker 0: traing step 300 (global step:300)")
    assert float(steps[-1].split("loss ")[1].split()[0]) < 1.6
    assert sum("validation accuracy" in l for l in out) == 3
    assert _lib._kernels is not None            # the HIP library did the work
    F.FLAGS.reset()


@pytest.mark.gpu
def test_gpu_native_rccl_update_entry_is_used_and_matches_torch_collectives(monkeypatch, one_rank_rccl_group):
    """The engine's GPU fast path must run through the native RCCL communicator (one-rank group here);
    with DTF_NATIVE_COMM=0 the torch.distributed path must give the same trajectory."""
    import torch.distributed as dist
    from dist_utils import free_port
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=200, test_size=300, seed=5)

    def make(native, coll, optimizer):
        monkeypatch.setenv("DTF_NATIVE_COMM", "1" if native else "0")
        st = build_mlp_state(784, [100], 10, "cuda", lr=0.3 if optimizer == "sgd" else 1e-3, optimizer=optimizer)
        st.initialize(seed=11)
        prog = MLPProgram(st, DeviceDataset(ds.train, "cuda", shuffle_seed=5), 100, [100], fused_update=False)
        eng = SyncReplicasEngine(prog, st, collective=coll, force_collectives=True, capture_collectives=False)
        eng.run(16)
        torch.cuda.synchronize()
        return st, prog, eng

    for coll, optimizer in (("all_reduce", "sgd"), ("reduce_scatter", "sgd"), ("reduce_scatter", "adam")):
        st_n, prog_n, eng_n = make(True, coll, optimizer)
        st_t, prog_t, eng_t = make(False, coll, optimizer)
        assert eng_n.native_comm is not None and prog_n.after_compute is None       # update rides in the native schedule
        assert eng_t.native_comm is None and prog_t.after_compute is not None
        torch.testing.assert_close(st_n.params, st_t.params, rtol=1e-6, atol=1e-7)
        assert st_n.global_step() == st_t.global_step() == 17
        if optimizer == "adam":
            assert int(st_n.state[2]) == 16


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,batch,post,fused,loss", [
    (100, 100, True, True, "reference"), (100, 100, True, False, "reference"), (50, 37, False, True, "standard"),
    (128, 128, True, True, "reference"), (20, 16, True, False, "standard"), (72, 60, False, False, "reference")])
def test_gpu_two_launch_step_matches_three_launch_step(monkeypatch, hidden, batch, post, fused, loss):
    """mlp2_fused.hip (2 launches/step) against the general kernels (3 launches/step) and the CPU program:
    same losses, counts, parameters (update fused) or gradients (gradients stored)."""
    mode = ops.LOSS_REFERENCE if loss == "reference" else ops.LOSS_STANDARD
    steps = 21

    monkeypatch.setenv("DTF_PIPELINED_STEP", "0")       # this test is about the two-launch step

    def make(device, two_launch):
        monkeypatch.setenv("DTF_FUSED_STEP", "1" if two_launch else "0")
        ds, st, prog, _ = _make(device, hidden=(hidden,), batch=batch, lr=0.2, post=post, loss_mode=mode, fused=fused,
                                n_train=1000)
        if not fused:
            prog.after_compute = lambda: st.apply_shard(None, 1.0)      # plain SGD from the stored gradients
        return st, prog

    st_2, prog_2 = make("cuda", True)
    st_3, prog_3 = make("cuda", False)
    st_c, prog_c = make("cpu", False)
    assert prog_2.fused2 and not prog_3.fused2
    for p in (prog_2, prog_3, prog_c):
        p.run(steps, graph_steps=4)
    r2, r3, rc = prog_2.read_log(0, steps), prog_3.read_log(0, steps), prog_c.read_log(0, steps)
    for a, b, c in zip(r2, r3, rc):
        assert a.global_step == b.global_step == c.global_step and a.local_step == c.local_step
        assert abs(a.loss - b.loss) < 2e-5 and abs(a.loss - c.loss) < 1e-4
        assert abs(a.train_correct - c.train_correct) <= 1
        if post and c.post_correct is not None:
            assert abs(a.post_correct - c.post_correct) <= 1
    torch.testing.assert_close(st_2.params, st_3.params, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(st_2.params.cpu(), st_c.params, rtol=1e-3, atol=1e-5)
    if not fused:
        torch.testing.assert_close(st_2.grads, st_3.grads, rtol=1e-4, atol=1e-7)
    assert st_2.global_step() == steps + 1 and st_2.local_step() == steps
    assert prog_2.post_correct_of_last_step() == prog_3.post_correct_of_last_step()


@pytest.mark.gpu
def test_gpu_native_multi_step_launch_equals_graph_replay(monkeypatch):
    """One native call that enqueues every kernel of a run (bench.py's path for short one-worker runs) against the
    hipGraph replay of the same steps: same log, same parameters, bit for bit."""
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    monkeypatch.setenv("DTF_PIPELINED_STEP", "0")       # the native launcher drives the two-launch step
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=200, test_size=300, seed=5)
    out = []
    for native in (True, False):
        st = build_mlp_state(784, [100], 10, "cuda", lr=0.3)
        st.initialize(seed=11)
        prog = MLPProgram(st, DeviceDataset(ds.train, "cuda", shuffle_seed=5), 100, [100], fused_update=True)
        eng = SyncReplicasEngine(prog, st)
        assert eng.native_launch_ok()
        if native:
            eng.run_native(5)
            eng.run_native(20)
        else:
            eng.run(5, 64)
            eng.run(20, 64)
        torch.cuda.synchronize()
        assert eng.updates_applied == 25 and st.global_step() == 26
        out.append((st.params.clone(), prog.read_log(0, 25)))
    assert torch.equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert a == b


def _first_batch_problems(prog, split, seed):
    """What is wrong with a started program's view of its first batch (empty list = nothing): the uploaded split, the row
    indices the step tail wrote, the rows and labels gathered for the two-launch step."""
    from distributed_tensorflow_amd.data.mnist import epoch_permutation
    torch.cuda.synchronize()
    B, q = prog.B, prog.steps_enqueued % prog.P
    img, lab = torch.from_numpy(split.images), torch.from_numpy(split.labels)
    want_rows = torch.from_numpy(epoch_permutation(seed, 0, prog.data.n)[:B].astype("int64"))
    out = []
    d_img, d_lab = prog.data.images.cpu(), prog.data.labels.cpu()
    if not torch.equal(d_img, img):
        out.append(f"uploaded images differ from the host split in rows {(d_img != img).any(1).nonzero().flatten().tolist()[:8]}")
    if not torch.equal(d_lab, lab):
        out.append(f"uploaded labels differ from the host split in rows {(d_lab != lab).any(1).nonzero().flatten().tolist()[:8]}")
    if not torch.equal(prog.data.perm[0].cpu().long(), torch.from_numpy(epoch_permutation(seed, 0, prog.data.n).astype("int64"))):
        out.append("the resident permutation of epoch 0 differs from the host's")
    got_rows = prog.ridx[q][:B].cpu().long()
    if not torch.equal(got_rows, want_rows):
        out.append(f"row indices of the first batch differ at positions {(got_rows != want_rows).nonzero().flatten().tolist()[:8]}")
    if prog.fused2:
        xg = prog.xg[q][:B].cpu()
        if not torch.equal(xg, img[want_rows]):
            out.append(f"gathered rows differ at batch positions {(xg != img[want_rows]).any(1).nonzero().flatten().tolist()[:8]}")
        lg = prog.labg[q][:prog.C, :B].cpu()
        if not torch.equal(lg, lab[want_rows].t()):
            out.append(f"gathered labels differ at batch positions {(lg != lab[want_rows].t()).any(0).nonzero().flatten().tolist()[:8]}")
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["1", "0", "sharded"])
def test_gpu_peer_mapped_update_entry_on_one_rank(monkeypatch, fused, one_rank_rccl_group):
    """DTF_P2P_ALLREDUCE=1 on a one-rank group: gradients go through the exported slot buffers and the one-shot
    update kernel (captured in the step graph); the trajectory must equal the fused local update."""
    import torch.distributed as dist
    from dist_utils import free_port
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=200, test_size=300, seed=5)

    def make(p2p):
        monkeypatch.setenv("DTF_P2P_ALLREDUCE", "1" if p2p else "0")
        monkeypatch.setenv("DTF_P2P_FUSED", "1" if fused == "sharded" else fused)
        monkeypatch.setenv("DTF_P2P_MODE", "sharded" if fused == "sharded" else "replicated")
        st = build_mlp_state(784, [100], 10, "cuda", lr=0.3)
        st.initialize(seed=11)
        prog = MLPProgram(st, DeviceDataset(ds.train, "cuda", shuffle_seed=5), 100, [100], fused_update=not p2p)
        eng = SyncReplicasEngine(prog, st, collective="all_reduce", force_collectives=p2p)
        prog.start()
        problems = _first_batch_problems(prog, ds.train, 5)
        assert not problems, (p2p, problems)
        initial.append(st.params.clone())
        eng.run(30, 8)
        torch.cuda.synchronize()
        return st, prog, eng

    initial = []
    st_p, prog_p, eng_p = make(True)
    st_f, prog_f, eng_f = make(False)
    assert torch.equal(initial[0], initial[1]), "the two programs started from different parameters"
    assert eng_p.p2p is not None and prog_p.after_compute is None and eng_f.p2p is None
    assert eng_p.p2p_fused == (fused != "0") and eng_p.p2p_mode == ("sharded" if fused == "sharded" else "replicated")
    for a, b in zip(prog_p.read_log(0, 30), prog_f.read_log(0, 30)):
        assert abs(a.loss - b.loss) < 1e-5 and a.global_step == b.global_step
    torch.testing.assert_close(st_p.params, st_f.params, rtol=1e-5, atol=1e-6)


def _make_classes(device, classes, hidden=(48,), batch=40, post=True, use_graphs=None):
    """A program over synthetic data with `classes` label columns (beyond the 16 of the one-workgroup head kernel)."""
    import numpy as np
    from distributed_tensorflow_amd.data.mnist import DataSet
    from distributed_tensorflow_amd.parallel.sharded_state import ShardedState, mlp_variables
    rng = np.random.RandomState(3)
    n = 400
    images = rng.rand(n, 784).astype(np.float32)
    lab = rng.randint(0, classes, n)
    labels = np.zeros((n, classes), np.float32)
    labels[np.arange(n), lab] = 1.0
    split = DataSet(images, labels, seed=1)
    st = ShardedState(mlp_variables(784, list(hidden), classes), device, lr=0.2)
    st.initialize(seed=4)
    prog = MLPProgram(st, DeviceDataset(split, device, shuffle_seed=2), batch, list(hidden), classes=classes,
                      loss_mode=ops.LOSS_STANDARD, post_update_accuracy=post, fused_update=True, use_graphs=use_graphs)
    return st, prog


def test_cpu_program_with_40_classes_trains():
    st, prog = _make_classes("cpu", 40)
    prog.run(12)
    recs = prog.read_log(0, 12)
    assert abs(recs[0].loss - np.log(40)) < 0.5 and recs[-1].loss < recs[0].loss


@pytest.mark.gpu
@pytest.mark.parametrize("classes,graphs", [(17, False), (40, True), (1000, True)])
def test_gpu_program_with_more_than_16_classes_matches_cpu(classes, graphs):
    """The last layer beyond 16 classes: small GEMM + general softmax cross-entropy kernel + masked GEMM, also
    captured in step graphs; same trajectory as the CPU statements."""
    steps = 9
    st_c, prog_c = _make_classes("cpu", classes)
    st_g, prog_g = _make_classes("cuda", classes, use_graphs=graphs)
    st_g.load_values({n: st_c.var(n) for n in st_c.names()})
    prog_c.run(steps)
    prog_g.run(steps, graph_steps=4)
    for a, b in zip(prog_c.read_log(0, steps), prog_g.read_log(0, steps)):
        assert abs(a.loss - b.loss) < 2e-4 and a.global_step == b.global_step
        assert abs(a.train_correct - b.train_correct) <= 1
        if a.post_correct is not None:
            assert abs(a.post_correct - b.post_correct) <= 1
    for n in st_c.names():
        torch.testing.assert_close(st_g.var(n).cpu(), st_c.var(n), rtol=2e-3, atol=2e-5)

"""One-launch pipelined step (csrc/kernels/mlp_pipe.hip) against the two-launch step and the NumPy oracle.

The pipelined step computes the same SGD trajectory with another association of the products (Q + Gram-matrix
corrections instead of x @ W1 with materialised weights), so losses and parameters agree to fp32 rounding, not bit
for bit; prediction counts may differ by a near-tie."""
import os

import numpy as np
import pytest
import torch

import oracle
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state

pytestmark = pytest.mark.gpu


def _program(pipelined, hidden=100, batch=100, lr=0.5, post=True, loss_mode=ops.LOSS_REFERENCE, n_train=2000, in_dim=784,
             classes=10, use_graphs=None, seed=5):
    dev = torch.device("cuda", 0)
    if in_dim == 784 and classes == 10:
        ds = mnist.synthetic_data_sets(train_size=n_train, validation_size=100, test_size=100, seed=seed)
        train = ds.train
    else:
        rng = np.random.default_rng(seed)
        images = rng.random((n_train, in_dim), dtype=np.float32)
        labels = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, n_train)]
        train = mnist.DataSet(images, labels)
    st = build_mlp_state(in_dim, [hidden], classes, dev, lr=lr)
    st.initialize(seed=11)
    old = os.environ.get("DTF_PIPELINED_STEP")
    os.environ["DTF_PIPELINED_STEP"] = "1" if pipelined else "0"
    try:
        prog = MLPProgram(st, DeviceDataset(train, dev, shuffle_seed=seed), batch, [hidden], classes=classes,
                          loss_mode=loss_mode, post_update_accuracy=post, use_graphs=use_graphs)
    finally:
        if old is None:
            os.environ.pop("DTF_PIPELINED_STEP", None)
        else:
            os.environ["DTF_PIPELINED_STEP"] = old
    assert prog.pipe == pipelined
    return st, prog


def _compare(a, b, steps, runs, graph_steps=8, post=True, loss_tol=2e-5, param_tol=2e-5):
    (sta, pa), (stb, pb) = a, b
    for n in runs:
        pa.run(n, graph_steps)
        pb.run(n, graph_steps)
    torch.cuda.synchronize()
    ra, rb = pa.read_log(0, steps), pb.read_log(0, steps)
    assert sta.global_step() == stb.global_step() == steps + 1
    assert sta.local_step() == stb.local_step() == steps
    for s in range(steps):
        assert ra[s].global_step == rb[s].global_step == s + 1
        assert ra[s].local_step == rb[s].local_step == s + 1
        assert abs(ra[s].loss - rb[s].loss) < loss_tol, (s, ra[s].loss, rb[s].loss)
        assert abs(ra[s].train_correct - rb[s].train_correct) <= 1, (s, ra[s].train_correct, rb[s].train_correct)
        if post and s < steps - 1:
            assert abs(ra[s].post_correct - rb[s].post_correct) <= 1, (s, ra[s].post_correct, rb[s].post_correct)
    for name in sta.names():
        torch.testing.assert_close(sta.var(name), stb.var(name), rtol=1e-4, atol=param_tol, msg=name)
    assert abs(pa.post_correct_of_last_step() - pb.post_correct_of_last_step()) <= 1
    assert ra[-1].loss < ra[0].loss


@pytest.mark.parametrize("loss", ["reference", "standard"])
def test_pipelined_step_follows_the_two_launch_trajectory(loss):
    mode = ops.LOSS_REFERENCE if loss == "reference" else ops.LOSS_STANDARD
    # runs of different lengths: graphs of 8, an odd tail, a single eager step, a second run starting on another slot
    _compare(_program(True, loss_mode=mode), _program(False, loss_mode=mode), steps=45, runs=(19, 1, 25))


def test_pipelined_step_across_epoch_boundaries():
    # 1000 rows, batch 100: an epoch ends every 10 steps; the gather runs three batches ahead of the step
    _compare(_program(True, n_train=1000), _program(False, n_train=1000), steps=47, runs=(23, 24))


def test_pipelined_step_without_accuracy_only_rows():
    _compare(_program(True, post=False), _program(False, post=False), steps=20, runs=(20,), post=False)


def test_pipelined_step_eager_single_steps():
    _compare(_program(True, use_graphs=False), _program(False, use_graphs=False), steps=5, runs=(3, 2))


@pytest.mark.parametrize("hidden,batch,classes,in_dim", [(31, 32, 5, 20), (127, 128, 16, 64), (64, 50, 10, 784), (100, 100, 10, 788),
                                                         # the 16 x 16-tile geometry: hidden up to 256, batch up to 256
                                                         (256, 256, 10, 784), (200, 128, 10, 784), (128, 100, 10, 784),
                                                         (100, 200, 10, 784), (255, 250, 16, 100), (144, 60, 3, 36)])
def test_pipelined_step_other_shapes(hidden, batch, classes, in_dim):
    _compare(_program(True, hidden=hidden, batch=batch, classes=classes, in_dim=in_dim, n_train=16 * batch),
             _program(False, hidden=hidden, batch=batch, classes=classes, in_dim=in_dim, n_train=16 * batch),
             steps=21, runs=(21,), graph_steps=6)


def test_pipelined_step_matches_numpy_oracle():
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=2000, validation_size=100, test_size=100, seed=5)
    st = build_mlp_state(784, [100], 10, dev, lr=0.5)
    params = oracle.init_params(100, seed=11)
    st.load_values({k: torch.from_numpy(v).float() for k, v in params.items()})
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=5), 100, [100])
    assert prog.pipe
    steps = 30
    prog.run(steps, 8)
    recs = prog.read_log(0, steps)
    for s in range(steps):
        rows = ds.train.batch_rows(s * 100, 100)
        x, y = ds.train.images[rows].astype(np.float64), ds.train.labels[rows].astype(np.float64)
        l, acc, grads = oracle.loss_and_grads(params, x, y, "reference")
        params = oracle.sgd(params, grads, 0.5)
        assert abs(recs[s].loss - l) < 2e-5, (s, recs[s].loss, l)
        assert abs(recs[s].train_correct - round(acc * 100)) <= 1
        if s < steps - 1:
            assert abs(recs[s].post_correct - round(oracle.accuracy(params, x, y) * 100)) <= 1
    for k, v in params.items():
        np.testing.assert_allclose(st.var(k).cpu().numpy(), v, rtol=2e-4, atol=2e-6)


def test_parameters_are_materialised_between_runs():
    """Evaluation and checkpoints read the flat parameter buffer: after every run() it holds the parameters of the
    last step (the flush launch), and a run continues from values written into it by the host."""
    st, prog = _program(True)
    prog.run(10, 4)
    w = {n: st.var(n).clone() for n in st.names()}
    st2, prog2 = _program(False)
    prog2.run(10, 4)
    for n in st.names():
        torch.testing.assert_close(w[n], st2.var(n), rtol=1e-4, atol=2e-5)
    # the host rewrites the parameters (a restore): the next run must start from them
    for stx in (st, st2):
        stx.var("hid_w").mul_(0.5)
        stx.var("sm_b").add_(0.25)
    prog.run(6, 4)
    prog2.run(6, 4)
    torch.cuda.synchronize()
    for n in st.names():
        torch.testing.assert_close(st.var(n), st2.var(n), rtol=1e-4, atol=2e-5, msg=n)
    la, lb = prog.read_log(10, 6), prog2.read_log(10, 6)
    for x, y in zip(la, lb):
        assert abs(x.loss - y.loss) < 2e-5


def test_runs_continue_the_pipeline():
    """A run that follows another with nothing rewritten in between starts without a prime launch and computes bit
    for bit what one longer run computes; the last graph of a run holds the flush launch."""
    st, prog = _program(True)
    for n in (7, 1, 12, 20):
        prog.run(n, 8)
    st2, prog2 = _program(True)
    prog2.run(40, 8)
    torch.cuda.synchronize()
    assert prog.pipe_primes == 1 and prog2.pipe_primes == 1
    for n in st.names():
        assert torch.equal(st.var(n), st2.var(n)), n
    ra, rb = prog.read_log(0, 40), prog2.read_log(0, 40)
    assert [r.loss for r in ra] == [r.loss for r in rb]
    assert any(k[-1] == "flush" for k in prog._graphs)
    # counters rewritten by the host (a restore sets global_step): primed again, logged from the new value on
    st.load_values({n: st.var(n).clone() for n in st.names()}, global_step=1000)
    prog.run(3, 8)
    torch.cuda.synchronize()
    assert prog.pipe_primes == 2
    assert [r.global_step for r in prog.read_log(40, 3)] == [1000, 1001, 1002]

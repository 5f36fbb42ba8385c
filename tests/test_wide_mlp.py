"""Wide-MLP program (tiled-GEMM pipeline) against the reference step program."""
import pytest
import torch

from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
from distributed_tensorflow_amd.models.wide_mlp import WideMLPProgram
from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine

HIDDEN = [64, 48]


def _run(cls, device, steps, optimizer="sgd", lr=0.1, hidden=HIDDEN, batch=64, **kw):
    ds = mnist.synthetic_data_sets(train_size=1024, validation_size=100, test_size=200, seed=3)
    st = build_mlp_state(784, hidden, 10, device, lr=lr, optimizer=optimizer)
    st.initialize(seed=7) if device == "cpu" else None
    if device != "cpu":
        ref = build_mlp_state(784, hidden, 10, "cpu", lr=lr, optimizer=optimizer)
        ref.initialize(seed=7)
        st.load_values({n: ref.var(n) for n in ref.names()})
    prog = cls(st, DeviceDataset(ds.train, device, shuffle_seed=3), batch, hidden, loss_mode=ops.LOSS_STANDARD, **kw)
    eng = SyncReplicasEngine(prog, st)
    eng.run(steps, 4)
    return ds, st, prog


def test_cpu_wide_pipeline_equals_reference_program():
    _, st_a, pa = _run(MLPProgram, "cpu", 10, fused_update=False, post_update_accuracy=False)
    _, st_b, pb = _run(WideMLPProgram, "cpu", 10, dtype=torch.float32)
    torch.testing.assert_close(st_a.params, st_b.params, rtol=1e-5, atol=1e-6)
    for a, b in zip(pa.read_log(0, 10), pb.read_log(0, 10)):
        assert abs(a.loss - b.loss) < 1e-5 and a.train_correct == b.train_correct


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,optimizer", [(torch.float32, "sgd"), (torch.float32, "adam"), (torch.bfloat16, "sgd")])
def test_gpu_wide_pipeline_tracks_cpu_reference(dtype, optimizer):
    lr = 0.1 if optimizer == "sgd" else 1e-3
    steps = 12
    ds, st_c, pc = _run(MLPProgram, "cpu", steps, optimizer, lr, fused_update=False, post_update_accuracy=False)
    _, st_g, pg = _run(WideMLPProgram, "cuda", steps, optimizer, lr, dtype=dtype)
    rc, rg = pc.read_log(0, steps), pg.read_log(0, steps)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(rc, rg):
        assert abs(a.loss - b.loss) < tol, (a.loss, b.loss)
        assert a.global_step == b.global_step
    ptol = dict(rtol=1e-3, atol=1e-5) if dtype == torch.float32 else dict(rtol=0.1, atol=5e-3)
    torch.testing.assert_close(st_g.params.cpu(), st_c.params, **ptol)
    xt, yt = torch.from_numpy(ds.test.images), torch.from_numpy(ds.test.labels)
    assert abs(pg.accuracy(xt.cuda(), yt.cuda()) - pc.accuracy(xt, yt)) < 0.03
    assert pg.use_graphs


@pytest.mark.gpu
def test_gpu_wide_shapes_4096():
    # one real-size hidden layer pair: exercises the full 128x128 tiling and the head at Hd = 4096
    ds, st, prog = _run(WideMLPProgram, "cuda", 6, "adam", 1e-3, hidden=[4096, 4096], batch=256, dtype=torch.bfloat16)
    recs = prog.read_log(0, 6)
    assert recs[-1].loss < recs[0].loss
    assert all(torch.isfinite(torch.tensor(r.loss)) for r in recs)


@pytest.mark.gpu
def test_gpu_wide_pipeline_is_identical_on_both_gemm_kernels():
    """The 8-wave ping-pong GEMM inside the training pipeline (forward with transposed/fp32 copies, masked backward
    reading the transposed activations, weight + bias gradients, ragged K = 784 of the first layer): the trajectory
    must equal the one of the 4-wave register-staged kernels bit for bit (same MFMA, same accumulation order)."""
    out = {}
    for mode in (1, 0):
        ops.gemm_set_modes(pingpong=mode)
        try:
            _, st, prog = _run(WideMLPProgram, "cuda", 8, "adam", 1e-3, hidden=[512, 256, 512], batch=256, dtype=torch.bfloat16)
            torch.cuda.synchronize()
            out[mode] = (st.params.clone(), [r.loss for r in prog.read_log(0, 8)])
        finally:
            ops.gemm_set_modes(pingpong=2)
    assert out[1][1] == out[0][1]
    assert torch.equal(out[1][0], out[0][0])
    assert out[1][1][-1] < out[1][1][0]


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_gpu_folded_optimizer_matches_the_separate_update(optimizer, monkeypatch):
    """One worker: the 4096x4096 weight-gradient product applies the optimizer in its epilogue (gradient never
    stored, master/moments touched once, compute copy refreshed, transposed copy rebuilt from it).  Same arithmetic
    element for element as the flat optimizer kernel: the trajectory must not differ by a bit."""
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("DTF_FOLD_OPTIMIZER", fold)
        _, st, prog = _run(WideMLPProgram, "cuda", 6, optimizer, 1e-3 if optimizer == "adam" else 0.05,
                           hidden=[4096, 4096], batch=256, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        assert any(prog.folded) == (fold == "1") and not prog.folded[0]
        out[fold] = (st.params.clone(), [r.loss for r in prog.read_log(0, 6)],
                     None if st.m is None else (st.m.clone(), st.v.clone()))
    assert out["1"][1] == out["0"][1]
    assert torch.equal(out["1"][0], out["0"][0])
    if optimizer == "adam":
        assert torch.equal(out["1"][2][0], out["0"][2][0]) and torch.equal(out["1"][2][1], out["0"][2][1])
    assert out["1"][1][-1] < out["1"][1][0]


@pytest.mark.gpu
def test_gpu_folded_layers_pick_up_restored_weights():
    """Layers whose weight-gradient epilogue keeps the bf16 copies current are not re-cast every step; when the host
    rewrites the masters between runs (restore, broadcast) the next run must rebuild the copies first."""
    _, st, prog = _run(WideMLPProgram, "cuda", 3, "adam", 1e-3, hidden=[4096, 4096], batch=256, dtype=torch.bfloat16)
    assert prog.folded[1]
    torch.cuda.synchronize()
    new = {n: torch.zeros_like(st.var(n)).cpu() for n in st.names()}
    st.load_values(new)                                      # every weight zero: logits are the bias, i.e. zero
    prog.run(2, 2)
    torch.cuda.synchronize()
    rec = prog.read_log(3, 1)[0]
    assert abs(rec.loss - 2.302585) < 1e-3                   # log(10): the stale bf16 copies would give another loss


@pytest.mark.gpu
@pytest.mark.parametrize("loss_mode", [0, 1])
@pytest.mark.parametrize("m,hd,c", [(64, 1024, 10), (40, 264, 7)])
def test_gpu_bf16_output_layer_kernels_against_fp32_reference(m, hd, c, loss_mode):
    """head_wide_bf16 / out_wgrad_wide (16-wave kernels reading the bf16 activation and its transpose) against the
    plain PyTorch mathematics of ops/reference.py on the same (bf16-rounded) activation, ragged shapes included."""
    from distributed_tensorflow_amd import ops
    from distributed_tensorflow_amd.ops import reference
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    h = torch.relu(torch.randn(m, hd, generator=g)).to(torch.bfloat16)
    w2, b2 = torch.randn(hd, c, generator=g) * 0.05, torch.randn(c, generator=g) * 0.1
    nlab = 2 * m
    labels = torch.zeros(nlab, c)
    labels[torch.arange(nlab), torch.randint(0, c, (nlab,), generator=g)] = 1.0
    rows = torch.randperm(nlab, generator=g)[:m].to(torch.int32)
    ref = reference.head(h.float(), w2, b2, labels, m, loss_mode, rows, want_dz1=True)
    hd_ = h.to(dev)
    ht = torch.zeros(hd, m + 8, dtype=torch.bfloat16, device=dev)          # wider than needed: ld != m
    ht[:, :m] = hd_.t()
    dz2 = torch.full((m, 16), 7.0, device=dev)
    dz1 = torch.zeros(m, hd, dtype=torch.bfloat16, device=dev)
    dz1_t = torch.zeros(hd, m + 8, dtype=torch.bfloat16, device=dev)
    nblk = (m + 15) // 16
    lp, cp, pp = torch.zeros(nblk, device=dev), torch.zeros(nblk, dtype=torch.int32, device=dev), torch.ones(nblk, dtype=torch.int32, device=dev)
    args = ops.build_head_wide_args(hd_, ht, w2.to(dev), b2.to(dev), labels.to(dev), m, loss_mode, rows=rows.to(dev), dz2=dz2,
                                    dz1_bf=dz1, dz1_bf_t=dz1_t, loss_part=lp, correct_part=cp, post_part=pp)
    ops.head_wide_bf16(args)
    torch.cuda.synchronize()
    assert abs(float(lp.sum()) - float(ref["loss_sum"])) <= 1e-4 * max(1.0, abs(float(ref["loss_sum"])))
    assert int(cp.sum()) == int(ref["correct"]) and int(pp.sum()) == 0
    assert torch.allclose(dz2[:, :c].cpu(), ref["dz2"], rtol=1e-4, atol=1e-7) and float(dz2[:, c:].abs().sum()) == 0.0
    want = ref["dz1"].to(torch.bfloat16).float()
    got = dz1.float().cpu()
    assert torch.allclose(got, want, rtol=2e-2, atol=1e-6)
    assert torch.equal((got != 0), (want != 0)) or float(((got != 0) != (want != 0)).float().mean()) < 1e-3
    assert torch.equal(dz1_t[:, :m].cpu(), dz1.t().cpu()) and float(dz1_t[:, m:].abs().sum()) == 0.0
    # output-layer weight gradient from the transposed activation
    dw, db = torch.zeros(hd, c, device=dev), torch.zeros(c, device=dev)
    ops.out_wgrad_wide(ops.build_out_wgrad_wide_args(ht, dz2, dw, db, m))
    torch.cuda.synchronize()
    dz2c = dz2[:, :c].cpu()
    assert torch.allclose(dw.cpu(), h.float().t() @ dz2c, rtol=1e-4, atol=1e-6)
    assert torch.allclose(db.cpu(), dz2c.sum(0), rtol=1e-4, atol=1e-7)

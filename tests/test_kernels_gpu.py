"""HIP kernels vs the plain PyTorch fp32 statement of the same op (CPU oracle).

Shapes follow the reference graph (B=100, D=784, H=100, C=10,
/root/reference/distributed.py:11-13,35) plus the evaluation batches (5000,
10000) and odd sizes that exercise every bounds guard.
"""
import pytest
import torch

from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.ops import reference as ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _close(a, b, rtol=2e-5, atol=2e-5):
    a, b = a.float().cpu(), b.float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref_mag = b.abs().max().item()
    assert err <= atol + rtol * ref_mag, f"max err {err} (ref magnitude {ref_mag})"


def test_native_library_is_loaded():
    from distributed_tensorflow_amd.ops import _lib
    assert _lib.kernels().dtf_kernels_abi_version() == 1


def test_mfma_layout_identity_asymmetric():
    # A = I against an asymmetric B catches a transposed C write or swapped fragment maps.
    n = 48
    a = torch.eye(n)
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n)
    out = ops.linear_fwd(a.to(DEV), b.to(DEV))
    assert torch.equal(out.cpu(), b)
    out = ops.linear_fwd(b.to(DEV), a.to(DEV))
    assert torch.equal(out.cpu(), b)


@pytest.mark.parametrize("M,K,N,relu", [(100, 784, 100, True), (1, 784, 100, True), (5000, 784, 100, True),
                                        (10000, 784, 100, False), (100, 784, 4096, True), (33, 100, 10, False),
                                        (17, 37, 29, True), (200, 784, 100, True), (64, 20, 48, False)])
def test_linear_fwd(M, K, N, relu):
    x, w, b = _rand(M, K, seed=1), _rand(K, N, seed=2, scale=0.1), _rand(N, seed=3)
    want = ref.linear_fwd(x, w, b, relu)
    got = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu)
    _close(got, want)


def test_linear_fwd_gather_rows():
    x, w, b = _rand(1000, 784, seed=1), _rand(784, 100, seed=2, scale=0.1), _rand(100, seed=3)
    rows = torch.randperm(1000, generator=torch.Generator().manual_seed(5))[:200].to(torch.int32)
    want = ref.linear_fwd(x, w, b, True, rows)
    out = torch.zeros(200, 100, device=DEV)
    ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), True, rows=rows.to(DEV), out=out)
    _close(out, want)


@pytest.mark.parametrize("M,K,N", [(100, 100, 100), (100, 4096, 4096), (37, 29, 52), (128, 784, 100)])
def test_linear_bwd_data(M, K, N):
    # forward weight [K, N]; dy [M, N]; result [M, K]
    dy, w, act = _rand(M, N, seed=4), _rand(K, N, seed=5, scale=0.1), _rand(M, K, seed=6)
    want = ref.linear_bwd_data(dy, w, act)
    got = ops.linear_bwd_data(dy.to(DEV), w.to(DEV), act.to(DEV))
    _close(got, want)
    want = ref.linear_bwd_data(dy, w, None)
    got = ops.linear_bwd_data(dy.to(DEV), w.to(DEV), None)
    _close(got, want)


# batches of 1024 rows and more take the variant that deals the batch rows of one tile to the four waves
@pytest.mark.parametrize("B,D,H,C", [(100, 784, 100, 10), (128, 784, 4096, 10), (7, 20, 12, 3), (1, 784, 100, 10),
                                     (4096, 40, 512, 10), (1031, 33, 20, 7)])
def test_wgrad_grouped_store(B, D, H, C):
    x, h = _rand(B, D, seed=1), _rand(B, H, seed=2)
    dz1, dz2p = _rand(B, H, seed=3), torch.zeros(B, 16)
    dz2p[:, :C] = _rand(B, C, seed=4)
    dw1, db1 = torch.zeros(D, H, device=DEV), torch.zeros(H, device=DEV)
    dw2, db2 = torch.zeros(H, C, device=DEV), torch.zeros(C, device=DEV)
    items = [ops.WgradItem(x.to(DEV), dz1.to(DEV), B, dw=dw1, db=db1),
             ops.WgradItem(h.to(DEV), dz2p.to(DEV), B, n=C, dw=dw2, db=db2)]
    ops.wgrad_grouped(items)
    w1, b1 = ref.wgrad(x, dz1)
    w2, b2 = ref.wgrad(h, dz2p[:, :C])
    _close(dw1, w1)
    _close(db1, b1)
    _close(dw2, w2)
    _close(db2, b2)


@pytest.mark.parametrize("B,N", [(100, 1000), (1300, 2000)])
def test_wgrad_grouped_fused_sgd_and_rows(B, N):
    D, H = 784, 100
    data = _rand(N, D, seed=1)
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(9))[:B].to(torch.int32)
    dz1 = _rand(B, H, seed=3)
    w, b = _rand(D, H, seed=7), _rand(H, seed=8)
    wd, bd = w.to(DEV), b.to(DEV)
    ops.wgrad_grouped([ops.WgradItem(data.to(DEV), dz1.to(DEV), B, rows=rows.to(DEV), w=wd, b=bd)], lr=0.5)
    gw, gb = ref.wgrad(data, dz1, rows)
    _close(wd, w - 0.5 * gw)
    _close(bd, b - 0.5 * gb)


@pytest.mark.parametrize("mode", [ops.LOSS_REFERENCE, ops.LOSS_STANDARD])
@pytest.mark.parametrize("M,Mt,H,C", [(100, 100, 100, 10), (200, 100, 100, 10), (5000, 0, 100, 10), (10000, 0, 100, 10), (37, 21, 52, 7),
                                      (64, 64, 4096, 10), (16, 16, 100, 16)])
def test_head(mode, M, Mt, H, C):
    hact = torch.relu(_rand(M, H, seed=1))
    w2, b2 = _rand(H, C, seed=2, scale=0.3), _rand(C, seed=3, scale=0.1)
    lab_idx = torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(4))
    labels = torch.nn.functional.one_hot(lab_idx, C).float()
    want = ref.head(hact, w2, b2, labels, Mt, mode)
    nblk = (M + 15) // 16
    dz2 = torch.full((max(Mt, 1), 16), 7.0, device=DEV)
    dz1 = torch.full((max(Mt, 1), H), 7.0, device=DEV)
    probs = torch.zeros(M, C, device=DEV)
    lp = torch.zeros(nblk, device=DEV)
    cp = torch.zeros(nblk, dtype=torch.int32, device=DEV)
    pp = torch.zeros(nblk, dtype=torch.int32, device=DEV)
    ops.head(hact.to(DEV), w2.to(DEV), b2.to(DEV), labels.to(DEV), M, Mt, mode, dz2=dz2, dz1=dz1, probs=probs,
             loss_part=lp, correct_part=cp, post_part=pp)
    _close(probs, want["probs"], rtol=1e-5, atol=1e-6)
    assert int(cp.sum()) == want["correct"]
    assert int(pp.sum()) == want["post_correct"]
    if Mt:
        assert abs(float(lp.sum()) - want["loss_sum"]) <= 2e-5 * max(1.0, abs(want["loss_sum"]))
        _close(dz2[:Mt, :C], want["dz2"], rtol=1e-4, atol=1e-8)
        assert float(dz2[:Mt, C:].abs().max()) == 0.0 if C < 16 else True
        _close(dz1[:Mt], want["dz1"], rtol=1e-4, atol=1e-8)


def test_head_label_gather_and_soft_labels():
    M, H, C, N = 48, 100, 10, 500
    hact = torch.relu(_rand(M, H, seed=1))
    w2, b2 = _rand(H, C, seed=2, scale=0.3), _rand(C, seed=3, scale=0.1)
    labels = torch.softmax(_rand(N, C, seed=5) * 3, 1)
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(6))[:M].to(torch.int32)
    want = ref.head(hact, w2, b2, labels, M, ops.LOSS_REFERENCE, rows)
    dz2 = torch.zeros(M, 16, device=DEV)
    dz1 = torch.zeros(M, H, device=DEV)
    lp = torch.zeros(3, device=DEV)
    cp = torch.zeros(3, dtype=torch.int32, device=DEV)
    pp = torch.zeros(3, dtype=torch.int32, device=DEV)
    ops.head(hact.to(DEV), w2.to(DEV), b2.to(DEV), labels.to(DEV), M, M, ops.LOSS_REFERENCE, rows=rows.to(DEV), dz2=dz2,
             dz1=dz1, loss_part=lp, correct_part=cp, post_part=pp)
    assert int(cp.sum()) == want["correct"]
    assert abs(float(lp.sum()) - want["loss_sum"]) <= 2e-5 * abs(want["loss_sum"])
    _close(dz1, want["dz1"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("n", [79510, 4, 3, 1 << 20, 1000003])
def test_sgd_and_accumulate(n):
    p, g = _rand(n, seed=1), _rand(n, seed=2)
    pd = p.to(DEV)
    ops.sgd_apply(pd, g.to(DEV), 0.01, 0.25)
    _close(pd, p - 0.01 * 0.25 * g, rtol=1e-6, atol=1e-7)
    d = p.to(DEV)
    ops.accumulate(d, g.to(DEV), 0.5, 2.0)
    _close(d, 0.5 * p + 2.0 * g, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n", [79510, 7, 1 << 18])
def test_adam_matches_torch_optim(n):
    p0 = _rand(n, seed=1)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([tp], lr=1e-3)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step_t = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in range(1, 6):
        g = _rand(n, seed=10 + t)
        tp.grad = g.clone()
        opt.step()
        if t % 2:
            ops.adam_apply(pd, g.to(DEV), m, v, 1e-3, step=t)
        else:
            step_t.fill_(t - 1)
            ops.adam_apply(pd, g.to(DEV), m, v, 1e-3, step_tensor=step_t, step_offset=1)
    _close(pd, tp.detach(), rtol=1e-5, atol=1e-6)


def test_truncated_normal_statistics():
    out = torch.empty(784 * 100, device=DEV)
    ops.truncated_normal_(out, 1.0 / 28, seed=1234, stream_id=1)
    o = out.cpu()
    assert o.abs().max() <= 2.0 / 28 + 1e-7
    assert abs(o.mean().item()) < 2e-3
    # std of a unit normal truncated at 2 sigma is 0.8796
    assert abs(o.std().item() * 28 - 0.8796) < 0.02
    out2 = torch.empty_like(out)
    ops.truncated_normal_(out2, 1.0 / 28, seed=1234, stream_id=1)
    assert torch.equal(out, out2)
    ops.truncated_normal_(out2, 1.0 / 28, seed=1234, stream_id=2)
    assert not torch.equal(out, out2)


def test_launchers_reject_bad_shapes():
    x, w = torch.zeros(4, 8, device=DEV), torch.zeros(9, 4, device=DEV)
    with pytest.raises(ValueError):
        ops.linear_fwd(x, w)
    with pytest.raises(ValueError):          # label rows narrower than the class count (20 classes: the wide-head path)
        ops.head(torch.zeros(4, 8, device=DEV), torch.zeros(8, 20, device=DEV), torch.zeros(20, device=DEV),
                 torch.zeros(4, 19, device=DEV), 4, 4)
    with pytest.raises(ValueError):          # and in the 16-class head kernel's range
        ops.head(torch.zeros(4, 8, device=DEV), torch.zeros(8, 10, device=DEV), torch.zeros(10, device=DEV),
                 torch.zeros(4, 9, device=DEV), 4, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("adam", [True, False])
def test_optimizer_over_ranges_matches_the_flat_kernels(adam):
    """One launch over several ranges of a flat buffer == the flat SGD/Adam kernels range by range, bit for bit;
    elements outside the ranges stay untouched."""
    import torch
    from distributed_tensorflow_amd import ops
    g = torch.Generator().manual_seed(3)
    n = 40000
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m0, v0 = torch.rand(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01
    ranges = [(0, 1024), (4096, 4100), (8192, 30000), (39996, 40000)]
    step = torch.tensor([4], dtype=torch.int32, device="cuda")
    a = [t.cuda() for t in (p0, gr, m0, v0)]
    b = [t.cuda() for t in (p0, gr, m0, v0)]
    ops.optimizer_ranges(a[0], a[1], ranges, 1e-3, a[2] if adam else None, a[3] if adam else None, grad_scale=0.5,
                         step_tensor=step, step_offset=1)
    for lo, hi in ranges:
        if adam:
            ops.adam_apply(b[0][lo:hi], b[1][lo:hi], b[2][lo:hi], b[3][lo:hi], 1e-3, grad_scale=0.5, step_tensor=step,
                           step_offset=1)
        else:
            ops.sgd_apply(b[0][lo:hi], b[1][lo:hi], 1e-3, 0.5)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(a[0][1024:4096].cpu(), p0[1024:4096])


def test_head_writes_bf16_operands_of_the_next_products():
    """dz1 straight into the bf16 operands (row-major + transposed) instead of fp32 + a cast pass."""
    M, H, C = 200, 1024, 10
    hact = torch.relu(_rand(M, H, seed=1)).to(DEV)
    w2, b2 = _rand(H, C, seed=2, scale=0.3).to(DEV), _rand(C, seed=3, scale=0.1).to(DEV)
    lab = torch.nn.functional.one_hot(torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(4)), C).float().to(DEV)
    dz2a, dz2b = torch.zeros(M, 16, device=DEV), torch.zeros(M, 16, device=DEV)
    dz1 = torch.zeros(M, H, device=DEV)
    ops.head(hact, w2, b2, lab, M, M, dz2=dz2a, dz1=dz1)
    bf = torch.zeros(M, H, dtype=torch.bfloat16, device=DEV)
    bft = torch.zeros(H, 208, dtype=torch.bfloat16, device=DEV)       # leading dimension a multiple of 4, > M
    ops.head(hact, w2, b2, lab, M, M, dz2=dz2b, dz1_bf=bf, dz1_bf_t=bft)
    assert torch.equal(dz2a, dz2b)
    assert torch.equal(bf, dz1.bfloat16())
    assert torch.equal(bft[:, :M], dz1.bfloat16().t())
    assert float(bft[:, M:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.head(hact, w2, b2, lab, M, M, dz2=dz2b, dz1_bf=bf, dz1_bf_t=torch.zeros(H, 202, dtype=torch.bfloat16, device=DEV))


def test_upload_from_pinned_is_a_plain_copy():
    """Kernel upload from page-locked host memory (the dataset's way onto the GPU): every word arrives; pageable or
    mis-sized sources are refused."""
    src = torch.arange(3 * 1000 * 7, dtype=torch.float32).reshape(3000, 7).pin_memory()
    dst = torch.zeros(3000, 7, device=DEV)
    ops.upload_from_pinned(dst, src)
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), src)
    isrc = torch.randint(0, 1 << 30, (12345,), dtype=torch.int32).pin_memory()
    idst = torch.zeros(12345, dtype=torch.int32, device=DEV)
    ops.upload_from_pinned(idst, isrc)
    torch.cuda.synchronize()
    assert torch.equal(idst.cpu(), isrc)
    with pytest.raises(ValueError):
        ops.upload_from_pinned(dst, torch.zeros(3000, 7))               # pageable
    with pytest.raises(ValueError):
        ops.upload_from_pinned(dst, torch.zeros(10).pin_memory())       # size


def test_h2d_verified_copies_views_and_converts():
    """The checked set-up upload: contiguous 4-byte data (kernel path), a strided view and a dtype conversion
    (PyTorch's copy), each compared with the source after the call."""
    src = torch.randn(37, 53)
    dst = torch.zeros(37, 53, device=DEV)
    assert ops.h2d_verified(dst, src, "t") is dst and torch.equal(dst.cpu(), src)
    flat = torch.zeros(4096, device=DEV)
    view = flat[128:128 + 37 * 53].view(37, 53)                   # a variable inside a flat parameter buffer
    ops.h2d_verified(view, src, "view")
    assert torch.equal(flat[128:128 + 37 * 53].cpu().view(37, 53), src) and float(flat[:128].abs().sum()) == 0.0
    wide = torch.zeros(37, 64, device=DEV)
    ops.h2d_verified(wide[:, :53], src, "strided")                # not contiguous
    assert torch.equal(wide[:, :53].cpu(), src) and float(wide[:, 53:].abs().sum()) == 0.0
    half = torch.zeros(37, 53, dtype=torch.bfloat16, device=DEV)
    ops.h2d_verified(half, src, "bf16")
    assert torch.equal(half.cpu(), src.to(torch.bfloat16))
    with pytest.raises(ValueError):
        ops.h2d_verified(dst, torch.zeros(3, 3), "shape")


def test_exported_regions_are_pooled_not_freed():
    """An exported (uncached, IPC-mapped) region that is released goes to a pool of the process and the next request
    that fits takes it again, zeroed: such memory is never handed back to the driver."""
    import ctypes as C
    from distributed_tensorflow_amd.ops import _lib
    from distributed_tensorflow_amd.parallel.async_peer import _DevicePtr
    lib = _lib.kernels()
    lib.dtf_ipc_alloc.restype = C.c_void_p
    lib.dtf_ipc_alloc.argtypes = [C.c_int64, C.c_char_p]
    lib.dtf_ipc_free.restype = C.c_int
    lib.dtf_ipc_free.argtypes = [C.c_void_p]
    torch.zeros(1, device=DEV)
    h = C.create_string_buffer(64)
    nbytes = 3 << 20
    p = lib.dtf_ipc_alloc(nbytes, h)
    assert p
    t = torch.as_tensor(_DevicePtr(p, nbytes // 4, "<f4"), device=DEV)
    t.fill_(7.0)
    torch.cuda.synchronize()
    del t
    assert lib.dtf_ipc_free(p) == 0 and lib.dtf_ipc_free(p) != 0           # the second release is refused
    q = lib.dtf_ipc_alloc(nbytes - 4096, h)                                   # fits: the pooled region comes back
    assert q == p
    t = torch.as_tensor(_DevicePtr(q, (nbytes - 4096) // 4, "<f4"), device=DEV)
    assert float(t.abs().sum()) == 0.0                                        # zeroed for its new user
    del t
    assert lib.dtf_ipc_free(q) == 0

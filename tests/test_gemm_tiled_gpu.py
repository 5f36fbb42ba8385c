"""Tiled MFMA GEMM (fp32 exact and bf16) vs a plain PyTorch fp32 statement of the same product."""
import pytest
import torch

from distributed_tensorflow_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _tol(dtype, K):
    return (2e-5, 2e-5) if dtype == torch.float32 else (2e-2, 2e-2 * (K ** 0.5) * 0.02)


def _check(got, want, rtol, atol):
    err = (got.float().cpu() - want).abs().max().item()
    assert err <= atol + rtol * want.abs().max().item(), f"max err {err} vs magnitude {want.abs().max().item()}"


def test_layout_identity_asymmetric():
    for dtype in (torch.float32, torch.bfloat16):
        n = 256
        eye = torch.eye(n)
        b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 125      # exact in bf16
        out = torch.zeros(n, n, dtype=dtype, device=DEV)
        ops.gemm_nt(eye.to(DEV, dtype), b.to(DEV, dtype), out)            # out = I @ b^T
        assert torch.equal(out.float().cpu(), b.t())
        ops.gemm_nt(b.to(DEV, dtype), eye.to(DEV, dtype), out)            # out = b @ I
        assert torch.equal(out.float().cpu(), b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 4096, 784), (1000, 520, 4096), (136, 72, 200), (8, 8, 8)])
def test_forward_epilogue(dtype, M, N, K):
    a, b, bias = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    ad, bd = a.to(DEV, dtype), b.to(DEV, dtype)
    want = torch.relu(ad.float().cpu() @ bd.float().cpu().t() + bias)
    out = torch.zeros(M, N, dtype=dtype, device=DEV)
    Mp = (M + 7) // 8 * 8
    out_t = torch.zeros(N, Mp, dtype=dtype, device=DEV)
    out32 = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(ad, bd, out, ops.EPI_FWD, bias=bias.to(DEV), relu=True, out_t=out_t, out_f32=out32)
    rtol, atol = _tol(dtype, K)
    _check(out32, want, 2e-5 if dtype == torch.float32 else 1e-3, 1e-4)
    _check(out, want, rtol, atol)
    assert torch.equal(out_t[:, :M].float().cpu(), out.float().cpu().t())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_and_wgrad_epilogues(dtype):
    M, IN, OUT = 264, 328, 200
    dy, w, act = _rand(M, OUT, seed=1), _rand(IN, OUT, seed=2, scale=0.1), _rand(M, IN, seed=3)
    dyd, wd, actd = dy.to(DEV, dtype), w.to(DEV, dtype), act.to(DEV, dtype)
    want = torch.where(actd.float().cpu() > 0, dyd.float().cpu() @ wd.float().cpu().t(), torch.zeros(M, IN))
    dx = torch.zeros(M, IN, dtype=dtype, device=DEV)
    dx_t = torch.zeros(IN, M, dtype=dtype, device=DEV)
    ops.gemm_nt(dyd, wd, dx, ops.EPI_BWD, mask=actd, out_t=dx_t)
    rtol, atol = _tol(dtype, OUT)
    _check(dx, want, rtol, atol)
    assert torch.equal(dx_t.float().cpu(), dx.float().cpu().t())
    # the same product with the mask also given transposed (wide loads in the epilogue)
    dx2 = torch.zeros(M, IN, dtype=dtype, device=DEV)
    ops.gemm_nt(dyd, wd, dx2, ops.EPI_BWD, mask=actd, mask_t=actd.t().contiguous())
    assert torch.equal(dx2.cpu(), dx.cpu())
    # dW[in,out] = X^T dY from the transposed copies; db = column sums of dY
    x_t, dy_t = _rand(IN, M, seed=4).to(DEV, dtype), dyd.t().contiguous()
    dw = torch.zeros(IN, OUT, device=DEV)
    db = torch.zeros(OUT, device=DEV)
    ops.gemm_nt(x_t, dy_t, dw, ops.EPI_WGRAD, bias_out=db)
    _check(dw, x_t.float().cpu() @ dyd.float().cpu(), 2e-5 if dtype == torch.float32 else 2e-3, 1e-4)
    _check(db, dyd.float().cpu().sum(0), 2e-5 if dtype == torch.float32 else 2e-3, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_cast_transpose(dtype):
    src = _rand(500, 784, seed=1)
    rows = torch.randperm(500, generator=torch.Generator().manual_seed(2))[:136].to(torch.int32)
    dst = torch.zeros(136, 784, dtype=dtype, device=DEV)
    dst_t = torch.zeros(784, 136, dtype=dtype, device=DEV)
    ops.gather_cast_transpose(src.to(DEV), rows.to(DEV), dst, dst_t)
    want = src[rows.long()].to(dtype)
    assert torch.equal(dst.cpu(), want) and torch.equal(dst_t.cpu(), want.t())


@pytest.mark.parametrize("M,N", [(37, 50), (64, 64), (130, 67), (4096, 784)])
def test_cast_transpose_odd_shapes(M, N):
    src = _rand(M, N, seed=3)
    for dtype in (torch.float32, torch.bfloat16):
        dst = torch.zeros(M, N, dtype=dtype, device=DEV)
        dst_t = torch.zeros(N, M, dtype=dtype, device=DEV)
        ops.gather_cast_transpose(src.to(DEV), None, dst, dst_t)
        assert torch.equal(dst.cpu(), src.to(dtype)) and torch.equal(dst_t.cpu(), src.to(dtype).t())
        only_t = torch.zeros(N, M, dtype=dtype, device=DEV)
        ops.gather_cast_transpose(src.to(DEV), None, None, only_t)
        assert torch.equal(only_t.cpu(), src.to(dtype).t())


def test_rejects_unaligned_k():
    a = torch.zeros(16, 10, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(a, a, torch.zeros(16, 16, dtype=torch.bfloat16, device=DEV))


@pytest.fixture
def pingpong_kernel():
    ops.gemm_set_modes(pingpong=1)          # take the 8-wave kernel whenever the shape allows
    yield
    ops.gemm_set_modes(pingpong=2)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 384), (768, 256, 1024), (256, 1024, 128 * 7),
                                   (512, 256, 784), (256, 512, 264), (256, 256, 448),
                                   (264, 520, 256), (1000, 776, 328), (8, 8, 256), (504, 248, 512)])
def test_pingpong_kernel_all_epilogues(pingpong_kernel, M, N, K):
    """The 8-wave direct-to-LDS kernel against fp32 statements; operands asymmetric so that a swapped tile, half
    or k-chunk anywhere in the staging/fragment maps shows."""
    dtype = torch.bfloat16
    a, b, bias = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    ad, bd = a.to(DEV, dtype), b.to(DEV, dtype)
    prod = ad.float().cpu() @ bd.float().cpu().t()
    rtol, atol = _tol(dtype, K)
    # forward
    out = torch.zeros(M, N, dtype=dtype, device=DEV)
    out_t = torch.zeros(N, M, dtype=dtype, device=DEV)
    out32 = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(ad, bd, out, ops.EPI_FWD, bias=bias.to(DEV), relu=True, out_t=out_t, out_f32=out32)
    _check(out32, torch.relu(prod + bias), 1e-3, 1e-4)
    _check(out, torch.relu(prod + bias), rtol, atol)
    assert torch.equal(out_t.float().cpu(), out.float().cpu().t())
    # the same product must come out of the 4-wave kernels bit for bit in fp32 (same MFMA, same k order per tile)
    ops.gemm_set_modes(pingpong=0)
    ref32 = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(ad, bd, torch.zeros_like(out), ops.EPI_FWD, bias=bias.to(DEV), relu=True, out_f32=ref32)
    ops.gemm_set_modes(pingpong=1)
    assert torch.equal(ref32.cpu(), out32.cpu())
    # backward with the transposed mask
    mask = _rand(M, N, seed=4).to(DEV, dtype)
    dx = torch.zeros(M, N, dtype=dtype, device=DEV)
    dx_t = torch.zeros(N, M, dtype=dtype, device=DEV)
    ops.gemm_nt(ad, bd, dx, ops.EPI_BWD, mask=mask, mask_t=mask.t().contiguous(), out_t=dx_t)
    _check(dx, torch.where(mask.float().cpu() > 0, prod, torch.zeros(M, N)), rtol, atol)
    assert torch.equal(dx_t.float().cpu(), dx.float().cpu().t())
    # weight gradient + bias gradient
    dw = torch.zeros(M, N, device=DEV)
    db = torch.zeros(N, device=DEV)
    ops.gemm_nt(ad, bd, dw, ops.EPI_WGRAD, bias_out=db)
    _check(dw, prod, 2e-3, 1e-4)
    _check(db, bd.float().cpu().sum(1), 2e-3, 1e-4)


def test_pingpong_identity_layout(pingpong_kernel):
    n = 512
    eye = torch.eye(n)
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 125      # exact in bf16
    out = torch.zeros(n, n, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(eye.to(DEV, torch.bfloat16), b.to(DEV, torch.bfloat16), out)
    assert torch.equal(out.float().cpu(), b.t())
    ops.gemm_nt(b.to(DEV, torch.bfloat16), eye.to(DEV, torch.bfloat16), out)
    assert torch.equal(out.float().cpu(), b)


def test_pingpong_full_chip_is_bit_identical_to_the_four_wave_kernel(pingpong_kernel):
    """One block on every CU, long K loop, repeated: a staging race (a fragment read before its half-tile has
    landed, or a half-tile overwritten while still being read) shows as a mismatch against the register-staged
    kernel, which accumulates in the same order."""
    M = N = 4096
    K = 2048 + 64 * 3 + 8                  # odd number of K-tiles plus a partial one
    a, b = _rand(M, K, seed=11).to(DEV, torch.bfloat16), _rand(N, K, seed=12, scale=0.1).to(DEV, torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_set_modes(pingpong=0)
    ref = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(a, b, out, ops.EPI_FWD, out_f32=ref)
    ops.gemm_set_modes(pingpong=1)
    for rep in range(6):
        got = torch.zeros(M, N, device=DEV)
        ops.gemm_nt(a, b, out, ops.EPI_FWD, out_f32=got)
        assert torch.equal(got, ref), f"repetition {rep}: {(got != ref).sum().item()} elements differ"


@pytest.mark.parametrize("M,N", [(64, 64), (72, 200), (4096, 784), (8, 8)])
def test_transpose16(M, N):
    src = _rand(M, N, seed=5).to(DEV, torch.bfloat16)
    dst = torch.zeros(N, M + 8, dtype=torch.bfloat16, device=DEV)           # wider than needed: ldt != M
    ops.transpose16(src, dst)
    assert torch.equal(dst[:, :M].cpu(), src.t().cpu()) and float(dst[:, M:].abs().sum()) == 0.0


def test_folded_optimizer_epilogue_against_separate_kernels(pingpong_kernel):
    """Weight-gradient product with the optimizer in the epilogue (ragged shape, Adam and SGD): masters, moments and
    the bf16 copy must equal gradient product + flat optimizer kernel + cast, bit for bit."""
    M, N, K = 520, 264, 512
    xt, dyt = _rand(M, K, seed=1).to(DEV, torch.bfloat16), _rand(N, K, seed=2, scale=0.1).to(DEV, torch.bfloat16)
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    for adam in (True, False):
        p0 = _rand(M, N, seed=3).to(DEV)
        m0, v0 = _rand(M, N, seed=4).abs().to(DEV) * 0.1, _rand(M, N, seed=5).abs().to(DEV) * 0.01
        grad, db = torch.zeros(M, N, device=DEV), torch.zeros(N, device=DEV)
        ops.gemm_nt(xt, dyt, grad, ops.EPI_WGRAD, bias_out=db)
        pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
        if adam:
            ops.adam_apply(pr, grad, mr, vr, 1e-3, step_tensor=step, step_offset=1)
        else:
            ops.sgd_apply(pr, grad, 0.05)
        pf, mf, vf = p0.clone(), m0.clone(), v0.clone()
        w = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        db2, unused = torch.zeros(N, device=DEV), torch.zeros(M, N, device=DEV)
        wt = torch.zeros(N, M + 8, dtype=torch.bfloat16, device=DEV)       # wider than needed: ldwt != M
        opt = ops.FoldedOptimizer(pf, mf if adam else None, vf if adam else None, w, 1e-3 if adam else 0.05,
                                  step_tensor=step, step_offset=1, wt=wt if adam else None)
        ops.gemm_nt(xt, dyt, unused, ops.EPI_WGRAD, bias_out=db2, opt=opt)
        assert torch.equal(pf, pr) and torch.equal(db2, db) and float(unused.abs().sum()) == 0.0
        assert torch.equal(w, pr.to(torch.bfloat16))
        if adam:        # the transposed compute copy comes out of the same epilogue; nothing outside [N, M] is written
            assert torch.equal(wt[:, :M], pr.to(torch.bfloat16).t()) and float(wt[:, M:].abs().sum()) == 0.0
        if adam:
            assert torch.equal(mf, mr) and torch.equal(vf, vr)


@pytest.mark.parametrize("M,N,K,S", [(784, 512, 4096, 4), (264, 520, 1024, 3)])
def test_split_k_weight_gradient_matches_the_single_pass(pingpong_kernel, M, N, K, S):
    """Split-K weight gradient (K ranges per tile + summing pass) against the one-pass product and the fp32 reference."""
    xt, dyt = _rand(M, K, seed=1).to(DEV, torch.bfloat16), _rand(N, K, seed=2, scale=0.1).to(DEV, torch.bfloat16)
    ref, db_ref = torch.zeros(M, N, device=DEV), torch.zeros(N, device=DEV)
    ops.gemm_nt(xt, dyt, ref, ops.EPI_WGRAD, bias_out=db_ref)
    out, db = torch.full((M, N), 3.0, device=DEV), torch.full((N,), 3.0, device=DEV)
    ops.gemm_nt(xt, dyt, out, ops.EPI_WGRAD, bias_out=db, split_k=ops.SplitK(S, M, N, DEV))
    torch.cuda.synchronize()
    want = xt.float() @ dyt.float().t()
    assert torch.allclose(out, want, rtol=1e-4, atol=1e-3) and torch.allclose(out, ref, rtol=1e-5, atol=1e-4)
    assert torch.allclose(db, dyt.float().sum(1), rtol=1e-4, atol=1e-3) and torch.allclose(db, db_ref, rtol=1e-5, atol=1e-4)
    again = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(xt, dyt, again, ops.EPI_WGRAD, split_k=ops.SplitK(S, M, N, DEV))
    assert torch.equal(again, out)              # fixed summation order: run-to-run identical
    assert ops.split_k_for(784, 4096, 4096) == 4 and ops.split_k_for(4096, 4096, 4096) == 1

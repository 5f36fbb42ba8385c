"""Fused batch norm + residual + ReLU kernels (csrc/kernels/batchnorm.hip) against the PyTorch operator sequence
in fp32 on the same bf16 inputs."""
import pytest
import torch

from distributed_tensorflow_amd.ops import batchnorm as bn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(n, c, h, w, seed, residual):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.4 * torch.randn(1, c, 1, 1, generator=g)).bfloat16()
    x = x.contiguous(memory_format=torch.channels_last)
    res = torch.randn(n, c, h, w, generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if residual else None
    dy = torch.randn(n, c, h, w, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    return x, res, dy, gamma, beta


@pytest.mark.parametrize("residual,relu", [(False, True), (True, True), (False, False), (True, False)])
@pytest.mark.parametrize("n,c,h,w", [(8, 64, 14, 14), (4, 256, 7, 9), (3, 2048, 5, 5), (2, 8, 3, 3), (16, 1024, 2, 2), (5, 512, 6, 6)])
def test_fused_bn_matches_operator_sequence(n, c, h, w, residual, relu):
    x, res, dy, gamma, beta = _inputs(n, c, h, w, seed=c + n, residual=residual)
    # oracle: fp32 arithmetic on the bf16 values
    xr = x.float().requires_grad_(True)
    rr = res.float().requires_grad_(True) if residual else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(c), torch.ones(c)
    want = bn.reference_bn_act(xr, gr, br, rm_ref, rv_ref, rr, relu, True, 0.1, 1e-5)
    want.backward(dy.float())

    xg = x.to(DEV).requires_grad_(True)
    rg = res.to(DEV).requires_grad_(True) if residual else None
    gg, bg = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    assert bn.kernel_path_takes(xg, rg, True)
    got = bn.bn_act(xg, gg, bg, rm, rv, rg, relu, True, 0.1, 1e-5)
    assert got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    got.backward(dy.to(DEV))

    def close(a, b, rtol, atol):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())

    close(got, want, 1e-2, 1e-2)                       # bf16 output rounding
    close(rm, rm_ref, 1e-4, 1e-5)
    close(rv, rv_ref, 1e-4, 1e-5)
    m = n * h * w
    close(gg.grad, gr.grad, 1e-3, 2e-3 * m ** 0.5)     # sums of m bf16-exact products in fp32: only ordering differs;
    close(bg.grad, br.grad, 1e-3, 2e-3 * m ** 0.5)     # the mask can flip where the pre-activation rounds to 0
    close(xg.grad, xr.grad, 2e-2, 2e-2)
    if residual:
        close(rg.grad, rr.grad, 1e-2, 1e-2)


def test_module_falls_back_outside_the_kernel_domain():
    mod = bn.FusedBatchNormAct(24).to(DEV)             # 24 channels: not 8 * 2^k
    x = torch.randn(4, 24, 5, 5, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    assert not bn.kernel_path_takes(x, None, True)
    y = mod(x)
    assert y.shape == x.shape and float(y.detach().min()) >= 0.0
    mod.eval()
    x64 = torch.randn(4, 64, 5, 5, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    assert not bn.kernel_path_takes(x64, None, False)


def test_large_activation_statistics():
    """3.2 M rows x 64 channels (the first ResNet layer at batch 256): the partial sums must not lose the variance."""
    n, c, h, w = 64, 64, 112, 112
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn(n, h, w, c, device=DEV, generator=g) * 0.5 + 3.0).bfloat16().permute(0, 3, 1, 2)
    assert x.is_contiguous(memory_format=torch.channels_last)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    y = bn.bn_act(x, gamma, beta, rm, rv, None, False, True, 1.0, 1e-5)
    xf = x.float()
    mean, var = xf.mean(dim=(0, 2, 3)), xf.var(dim=(0, 2, 3), unbiased=True)
    assert torch.allclose(rm, mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rv, var, rtol=1e-3, atol=1e-5)
    yf = y.float()
    assert float(yf.mean(dim=(0, 2, 3)).abs().max()) < 2e-2 and float((yf.var(dim=(0, 2, 3)) - 1).abs().max()) < 2e-2


@pytest.mark.parametrize("residual,relu,use", [(True, True, "both"), (True, True, "first"), (True, True, "second"),
                                               (False, True, "both"), (True, False, "both")])
def test_forked_output_adds_the_two_gradients_in_the_kernel(residual, relu, use):
    """``fork=True``: two handles of one output; their gradients reach the backward pass separately and the sum is
    formed inside the reduction kernel (residual + ReLU layers) or in front of it (other layers)."""
    n, c, h, w = 6, 128, 9, 7
    x, res, dy, gamma, beta = _inputs(n, c, h, w, seed=3, residual=residual)
    dy2 = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(9)).bfloat16().contiguous(memory_format=torch.channels_last)
    w1 = 1.0 if use in ("both", "first") else 0.0
    w2 = 1.0 if use in ("both", "second") else 0.0
    xr = x.float().requires_grad_(True)
    rr = res.float().requires_grad_(True) if residual else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    want = bn.reference_bn_act(xr, gr, br, torch.zeros(c), torch.ones(c), rr, relu, True, 0.1, 1e-5)
    want.backward(w1 * dy.float() + w2 * dy2.float())

    xg = x.to(DEV).requires_grad_(True)
    rg = res.to(DEV).requires_grad_(True) if residual else None
    gg, bg = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    a, b = bn.bn_act(xg, gg, bg, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), rg, relu, True, 0.1, 1e-5, fork=True)
    assert a.data_ptr() == b.data_ptr()
    outs, grads = [], []
    if w1:
        outs.append(a), grads.append(dy.to(DEV))
    if w2:
        outs.append(b), grads.append(dy2.to(DEV))
    torch.autograd.backward(outs, grads)

    def close(p, q, rtol, atol):
        p, q = p.detach().float().cpu(), q.detach().float().cpu()
        assert torch.allclose(p, q, rtol=rtol, atol=atol), float((p - q).abs().max())
    m = n * h * w
    close(a, want, 1e-2, 1e-2)
    # outside residual + ReLU layers the sum of the two gradients is rounded to bf16 before the kernels see it
    loose = 1.0 if (residual and relu) else 4.0
    close(gg.grad, gr.grad, 1e-3, loose * 3e-3 * m ** 0.5)
    close(bg.grad, br.grad, 1e-3, loose * 3e-3 * m ** 0.5)
    close(xg.grad, xr.grad, 2e-2, 3e-2)
    if residual:
        close(rg.grad, rr.grad, 1e-2, 2e-2)

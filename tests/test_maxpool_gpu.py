"""Max-pooling kernels (csrc/kernels/maxpool.hip) against F.max_pool2d in fp32 on the same bf16 values."""
import pytest
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.ops import maxpool as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,c,h,w,k,s,p", [(4, 64, 112, 112, 3, 2, 1), (3, 8, 7, 9, 3, 2, 1), (2, 16, 8, 8, 2, 2, 0), (2, 24, 13, 5, 3, 1, 1),
                                           (1, 8, 3, 3, 3, 3, 0), (2, 32, 10, 11, 5, 2, 2), (5, 128, 6, 6, 3, 2, 0)])
@pytest.mark.parametrize("relu_like", [False, True])
def test_maxpool_matches_operator(n, c, h, w, k, s, p, relu_like):
    g = torch.Generator().manual_seed(n * 100 + c)
    x = torch.randn(n, c, h, w, generator=g)
    if relu_like:
        x = x.clamp_min(0)            # many exact ties at zero: the first maximum in scan order takes the gradient
    x = x.bfloat16().contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True)
    want = F.max_pool2d(xr, k, stride=s, padding=p)
    dy = torch.randn(want.shape, generator=g).bfloat16()
    want.backward(dy.float())

    xg = x.to(DEV).requires_grad_(True)
    assert mp.kernel_path_takes(xg, k, s, p)
    got = mp.max_pool2d(xg, k, s, p)
    assert got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    got.backward(dy.to(DEV).contiguous(memory_format=torch.channels_last))
    assert torch.equal(got.float().cpu(), want.detach())                  # a maximum of bf16 values is one of them
    # every dx element is a sum of at most ceil(k/s)^2 bf16 values, rounded to bf16 once
    torch.testing.assert_close(xg.grad.float().cpu(), xr.grad, rtol=1e-2, atol=1e-2)
    # the gradient goes to the same elements
    assert torch.equal(xg.grad.float().cpu() != 0, xr.grad.bfloat16().float() != 0)


def test_operator_outside_the_kernel_domain():
    x = torch.randn(2, 12, 9, 9, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)   # 12 channels
    assert not mp.kernel_path_takes(x, 3, 2, 1)
    assert torch.equal(mp.max_pool2d(x, 3, 2, 1), F.max_pool2d(x, 3, stride=2, padding=1))
    xf = torch.randn(2, 16, 9, 9, device=DEV)                                                            # fp32
    assert not mp.kernel_path_takes(xf, 3, 2, 1)
    xc = torch.randn(2, 16, 9, 9)                                                                        # CPU
    assert not mp.kernel_path_takes(xc.bfloat16(), 3, 2, 1)

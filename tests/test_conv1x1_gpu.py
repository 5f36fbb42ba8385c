"""Weight gradient of stride-1 1x1 convolutions on the framework's own kernel (csrc/kernels/conv1x1.hip) against
PyTorch's fp32 convolution backward on the CPU, at the shapes ResNet-50 uses (smaller batch) and at ragged ones."""
import pytest
import torch

from distributed_tensorflow_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,hw,cin,cout", [(8, 56, 64, 64), (8, 56, 64, 256), (8, 56, 256, 64), (4, 28, 512, 128), (4, 28, 128, 512),
                                           (8, 14, 1024, 256), (8, 14, 256, 1024), (16, 7, 512, 2048), (16, 7, 2048, 512),
                                           (3, 5, 72, 200), (1, 3, 8, 8), (2, 9, 136, 24)])
def test_conv1x1_weight_gradient_against_fp32_convolution_backward(n, hw, cin, cout):
    g = torch.Generator().manual_seed(n * 1000 + cin + cout)
    x = torch.randn(n, hw, hw, cin, generator=g).bfloat16()             # channels-last storage
    dy = (torch.randn(n, hw, hw, cout, generator=g) * 0.1).bfloat16()
    M = n * hw * hw
    # reference: autograd of F.conv2d in fp32 on the CPU (on the bf16-rounded operands)
    xr = x.float().permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(cout, cin, 1, 1, requires_grad=True)
    torch.nn.functional.conv2d(xr, w).backward(dy.float().permute(0, 3, 1, 2).contiguous())
    want = w.grad.reshape(cout, cin)
    got = ops.conv1x1_wgrad(x.reshape(M, cin).cuda(), dy.reshape(M, cout).cuda())
    torch.cuda.synchronize()
    assert got.shape == (cout, cin) and got.dtype == torch.bfloat16
    scale = float(want.abs().max())
    # fp32 accumulation of bf16 products in another order, one rounding to bf16 at the end
    torch.testing.assert_close(got.float().cpu(), want, rtol=1e-2, atol=1e-2 * scale)
    # and twice the same bits: the partial sums are added in a fixed order
    again = ops.conv1x1_wgrad(x.reshape(M, cin).cuda(), dy.reshape(M, cout).cuda())
    assert (again == got).all()

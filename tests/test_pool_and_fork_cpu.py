"""Operator paths (CPU) of the pooling operator and of the forked batch-norm output: what runs wherever the kernels
do not take the call."""
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.models.resnet import ResNet
from distributed_tensorflow_amd.ops import batchnorm as bn
from distributed_tensorflow_amd.ops import maxpool as mp


def test_max_pool_runs_the_operator_on_the_cpu():
    x = torch.randn(2, 16, 9, 11).bfloat16().contiguous(memory_format=torch.channels_last)
    assert not mp.kernel_path_takes(x, 3, 2, 1)
    before = mp.calls["operators"]
    assert torch.equal(mp.max_pool2d(x, 3, 2, 1), F.max_pool2d(x, 3, stride=2, padding=1))
    assert mp.calls["operators"] == before + 1


def test_forked_output_gradients_add_up():
    torch.manual_seed(0)
    mod = bn.FusedBatchNormAct(8)
    x = torch.randn(4, 8, 5, 5, requires_grad=True)
    res = torch.randn(4, 8, 5, 5, requires_grad=True)
    a, b = mod(x, residual=res, fork=True)
    assert torch.equal(a, b)
    g1, g2 = torch.randn_like(a), torch.randn_like(a)
    torch.autograd.backward([a, b], [g1, g2])
    got = (x.grad.clone(), res.grad.clone(), mod.weight.grad.clone())
    x.grad = res.grad = mod.weight.grad = None
    mod.bias.grad = None
    y = mod(x, residual=res)
    y.backward(g1 + g2)
    for p, q in zip(got, (x.grad, res.grad, mod.weight.grad)):
        torch.testing.assert_close(p, q)


def test_blocks_hand_their_output_over_as_pairs():
    """Every bottleneck but the last forks its output; the network computes what the unforked network computes."""
    torch.manual_seed(1)
    net = ResNet(layers=(1, 2, 1, 1), num_classes=7, width=8)
    blocks = [b for layer in (net.layer1, net.layer2, net.layer3, net.layer4) for b in layer]
    assert [b.fork for b in blocks] == [True] * (len(blocks) - 1) + [False]
    x = torch.randn(2, 3, 32, 32)
    out = net(x)
    out.sum().backward()
    g_forked = [p.grad.clone() for p in net.parameters()]
    for b in blocks:
        b.fork = False
    net.zero_grad()
    out2 = net(x)
    out2.sum().backward()
    torch.testing.assert_close(out, out2)
    for p, g in zip(net.parameters(), g_forked):
        torch.testing.assert_close(p.grad, g)

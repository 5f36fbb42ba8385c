"""FusedBatchNormAct on the operator path (CPU): a drop-in for nn.BatchNorm2d (+ add + relu), same state dict."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from distributed_tensorflow_amd.ops.batchnorm import FusedBatchNormAct, bn_act, calls


def test_module_equals_batchnorm_add_relu_and_shares_state_dict():
    torch.manual_seed(0)
    ref, mod = nn.BatchNorm2d(12), FusedBatchNormAct(12)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.normal_()
    mod.load_state_dict(ref.state_dict())                      # same parameter and buffer names
    assert set(mod.state_dict()) == set(ref.state_dict())
    x = torch.randn(4, 12, 5, 5, requires_grad=True)
    r = torch.randn(4, 12, 5, 5, requires_grad=True)
    x2, r2 = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
    want = F.relu(ref(x) + r)
    got = mod(x2, residual=r2)
    torch.testing.assert_close(got, want)
    want.square().sum().backward()
    got.square().sum().backward()
    torch.testing.assert_close(x2.grad, x.grad)
    torch.testing.assert_close(r2.grad, r.grad)
    torch.testing.assert_close(mod.weight.grad, ref.weight.grad)
    torch.testing.assert_close(mod.running_var, ref.running_var)
    assert int(mod.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    # evaluation mode: running statistics, no update
    ref.eval(), mod.eval()
    torch.testing.assert_close(mod(x2.detach(), residual=r2.detach()), F.relu(ref(x.detach()) + r.detach()))
    assert int(mod.num_batches_tracked) == 1


def test_no_activation_variant_and_path_counter():
    before = dict(calls)
    x = torch.randn(3, 8, 4, 4)
    w, b = torch.ones(8), torch.zeros(8)
    y = bn_act(x, w, b, None, None, None, relu=False, training=True)
    torch.testing.assert_close(y, F.batch_norm(x, None, None, w, b, True))
    assert calls["operators"] == before["operators"] + 1 and calls["kernel"] == before["kernel"]

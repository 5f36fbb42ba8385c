"""Fused softmax cross-entropy for any class count (csrc/kernels/softmax_xent.hip) against a plain fp32 PyTorch
statement of the op on the CPU: both loss modes, one-hot rows and class indices, C in {10, 17, 1000, 5000},
B in {1, 256, 10000}."""
import pytest
import torch

from distributed_tensorflow_amd import ops


def _plain(z, lab_rows, mode):
    z = z.double().requires_grad_(True)
    y = torch.softmax(z, 1)
    logp = torch.log_softmax(y, 1) if mode == ops.LOSS_REFERENCE else torch.log_softmax(z, 1)
    rows = -(lab_rows.double() * logp).sum(1)
    rows.mean().backward()
    return rows.detach().float(), z.grad.float(), int((z.argmax(1) == lab_rows.argmax(1)).sum())


def _case(b, c, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(b, c, generator=g) * 3.0
    idx = torch.randint(0, c, (b,), generator=g)
    rows = torch.zeros(b, c)
    rows[torch.arange(b), idx] = 1.0
    return z, idx, rows


@pytest.mark.parametrize("mode", [ops.LOSS_STANDARD, ops.LOSS_REFERENCE])
def test_cpu_statement_matches_autograd(mode):
    z, idx, rows = _case(33, 17, 1)
    want_rows, want_g, want_hit = _plain(z, rows, mode)
    for labels in (idx, rows):
        r = ops.softmax_xent(z, labels, mode)
        torch.testing.assert_close(r.loss_rows, want_rows, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r.dlogits, want_g, rtol=1e-5, atol=1e-7)
        assert int(r.correct) == want_hit and abs(float(r.loss_sum) - float(want_rows.sum())) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [ops.LOSS_STANDARD, ops.LOSS_REFERENCE])
@pytest.mark.parametrize("b,c", [(1, 10), (256, 10), (10000, 10), (256, 17), (1, 1000), (256, 1000), (10000, 1000),
                                 (7, 64), (5, 65), (130, 2048), (9, 5000)])
def test_gpu_kernel_matches_fp32_cpu_statement(mode, b, c):
    z, idx, rows = _case(b, c, 7 * b + c)
    want_rows, want_g, want_hit = _plain(z, rows, mode)
    for labels in (idx, rows):
        loss_rows = torch.empty(b, device="cuda")
        gbf = torch.empty(b, c, dtype=torch.bfloat16, device="cuda")
        dl = torch.empty(b, c, device="cuda")
        r = ops.softmax_xent(z.cuda(), labels.cuda(), mode, loss_rows=loss_rows, dlogits=dl, dlogits_bf=gbf)
        torch.testing.assert_close(loss_rows.cpu(), want_rows, rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(dl.cpu(), want_g, rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(gbf.float().cpu(), want_g, rtol=1e-2, atol=1e-8)      # bf16 copy of the same gradient
        assert int(r.correct.item()) == want_hit
        assert abs(float(r.loss_sum.item()) - float(want_rows.double().sum())) <= 1e-5 * max(1.0, float(want_rows.sum()))


@pytest.mark.gpu
def test_gpu_argmax_takes_the_lowest_index_on_ties_and_soft_labels_work():
    z = torch.tensor([[1.0, 3.0, 3.0, 0.0], [2.0, 2.0, 2.0, 2.0]])
    lab = torch.tensor([[0.0, 0.5, 0.5, 0.0], [0.7, 0.1, 0.1, 0.1]])
    r = ops.softmax_xent(z.cuda(), lab.cuda(), ops.LOSS_STANDARD)
    assert int(r.correct.item()) == 2                       # argmax ties resolve to class 1 and class 0 on both sides
    want_rows, want_g, _ = _plain(z, lab, ops.LOSS_STANDARD)
    torch.testing.assert_close(r.dlogits.cpu(), want_g, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_gpu_linear_xent_node_matches_linear_plus_cross_entropy(dtype, tol):
    """The classifier + loss node of the ResNet-50 configuration (2048 features, 1000 classes) on the framework's
    GEMM and softmax cross-entropy kernels against nn.functional on the same operands."""
    from distributed_tensorflow_amd.ops import classifier
    g = torch.Generator().manual_seed(3)
    b, k, c = 64, 2048, 1000
    feat = (torch.randn(b, k, generator=g) * 0.5).cuda()
    w = (torch.randn(c, k, generator=g) * 0.02).cuda().requires_grad_(True)
    bias = (torch.randn(c, generator=g) * 0.1).cuda().requires_grad_(True)
    y = torch.randint(0, c, (b,), generator=g).cuda()
    f1 = feat.to(dtype).requires_grad_(True)
    before = classifier.calls["kernel"]
    loss, hits, logits = classifier.linear_xent(f1, w, bias, y)
    loss.backward()
    assert classifier.calls["kernel"] == before + 1
    f2 = feat.to(dtype).float().requires_grad_(True)
    w2 = w.detach().to(dtype).float().requires_grad_(True)
    b2 = bias.detach().clone().requires_grad_(True)
    ref_logits = torch.nn.functional.linear(f2, w2, b2)
    ref = torch.nn.functional.cross_entropy(ref_logits, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < tol
    assert abs(hits.item() - (ref_logits.argmax(1) == y).sum().item()) <= (0 if dtype == torch.float32 else 1)
    scale = lambda t: t.abs().max().item()
    assert (f1.grad.float() - f2.grad).abs().max().item() <= tol * scale(f2.grad) + 1e-7
    assert (w.grad - w2.grad).abs().max().item() <= tol * scale(w2.grad) + 1e-7
    assert (bias.grad - b2.grad).abs().max().item() <= tol * scale(b2.grad) + 1e-7

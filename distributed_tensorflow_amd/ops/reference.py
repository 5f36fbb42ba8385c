"""Plain PyTorch fp32 statements of every op in :mod:`distributed_tensorflow_amd.ops`.

These are (a) the CPU execution path (the framework runs end to end without a
GPU, e.g. the "1 ps + 2 workers on CPU" configuration) and (b) the oracle the
HIP kernels are tested against.  The math transcribes
/root/reference/distributed.py:78-89 and the gradients TensorFlow's autodiff
derives from it (``opt.minimize``, distributed.py:102).
"""
from __future__ import annotations

import math

import torch

LOSS_REFERENCE = 0   # softmax_cross_entropy_with_logits applied to an already-softmaxed y (distributed.py:81, :86)
LOSS_STANDARD = 1    # cross entropy on the logits


def linear_fwd(x, w, b=None, relu=False, rows=None):
    if rows is not None:
        x = x.index_select(0, rows.long())
    y = x @ w
    if b is not None:
        y = y + b
    return torch.relu(y) if relu else y


def linear_bwd_data(dy, w, act=None):
    """dX = dY @ W^T, masked where the forward activation was not positive (ReluGrad)."""
    dx = dy @ w.t()
    if act is not None:
        dx = torch.where(act > 0, dx, torch.zeros_like(dx))
    return dx


def wgrad(x, dy, rows=None):
    if rows is not None:
        x = x.index_select(0, rows.long())
    return x.t() @ dy, dy.sum(0)


def head(hact, w2, b2, labels, m_train, loss_mode=LOSS_REFERENCE, rows=None, want_dz1=True):
    """Last layer + loss + accuracy + gradients for the first ``m_train`` rows.

    Returns a dict with ``loss_sum`` (sum over training rows, not yet divided by
    the batch size), ``correct`` (training rows), ``post_correct`` (rows
    ``>= m_train``), ``probs``, ``dz2`` [m_train, C] and ``dz1`` [m_train, Hd].
    """
    m_total = hact.shape[0]
    if rows is not None:
        labels = labels.index_select(0, rows[:m_total].long())
    else:
        labels = labels[:m_total]
    z = hact @ w2 + b2
    y = torch.softmax(z, dim=1)
    hit = (y.argmax(1) == labels.argmax(1))
    out = {"probs": y,
           "correct": int(hit[:m_train].sum().item()),
           "post_correct": int(hit[m_train:].sum().item())}
    if m_train == 0:
        out["loss_sum"] = 0.0
        return out
    yt, lt, zt = y[:m_train], labels[:m_train], z[:m_train]
    inv_b = 1.0 / m_train
    if loss_mode == LOSS_REFERENCE:
        logp = torch.log_softmax(yt, dim=1)
        p = torch.softmax(yt, dim=1)
        dy = (p - lt) * inv_b
        dz2 = (dy - (dy * yt).sum(1, keepdim=True)) * yt
    else:
        logp = torch.log_softmax(zt, dim=1)
        dz2 = (yt - lt) * inv_b
    out["loss_sum"] = float(-(lt * logp).sum().item())
    out["loss_rows"] = -(lt * logp).sum(1)
    out["dz2"] = dz2
    if want_dz1:
        out["dz1"] = linear_bwd_data(dz2, w2, hact[:m_train])
    return out


def sgd_apply(param, grad, lr, grad_scale=1.0):
    param.sub_(grad * (lr * grad_scale))
    return param


def adam_apply(param, grad, m, v, lr, beta1, beta2, eps, grad_scale, step):
    """Textbook Adam (identical to torch.optim.Adam without weight decay); ``step`` counts from 1."""
    g = grad * grad_scale
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)
    return param


def truncated_normal(shape, stddev, generator=None, device="cpu"):
    """N(0, stddev) re-sampled until |v| <= 2 stddev (tf.truncated_normal, distributed.py:67-72)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    flat = out.view(-1)
    todo = torch.arange(flat.numel(), device=device)
    while todo.numel() > 0:
        cand = torch.randn(todo.numel(), generator=generator, device=device)
        ok = cand.abs() <= 2.0
        flat[todo[ok]] = cand[ok]
        todo = todo[~ok]
    return out * stddev


def softmax_xent(logits, labels, loss_mode=LOSS_STANDARD, grad_scale=None):
    """Loss rows, gradient wrt the logits and hit count of softmax cross-entropy over ``[B, C]`` logits; ``labels`` are
    one-hot / soft rows ``[B, C]`` or class indices ``[B]``.  Same two loss modes as :func:`head`."""
    b, c = logits.shape
    z = logits.float()
    if labels.dim() == 1:
        lab = torch.zeros_like(z)
        ok = (labels >= 0) & (labels < c)
        lab[ok, labels[ok].long()] = 1.0
        am_l = labels.long()
    else:
        lab = labels.float()
        am_l = lab.argmax(1)
    scale = (1.0 / b) if grad_scale is None else grad_scale
    y = torch.softmax(z, 1)
    if loss_mode == LOSS_REFERENCE:
        logp = torch.log_softmax(y, 1)
        p = torch.softmax(y, 1)
        dy = (p - lab) * scale
        dz = (dy - (dy * y).sum(1, keepdim=True)) * y
    else:
        logp = torch.log_softmax(z, 1)
        dz = (y - lab) * scale
    return {"loss_rows": -(lab * logp).sum(1), "dlogits": dz, "correct": int((z.argmax(1) == am_l).sum().item())}

"""Classifier + loss of a PyTorch module on the framework's own kernels.

``linear_xent(features, weight, bias, labels)`` is ``F.cross_entropy(F.linear(features, weight, bias), labels)``
with the mean reduction, as ONE autograd node whose forward and backward are the hand-written kernels:

    forward    logits = features W^T + b        tiled MFMA GEMM (``gemm_nt``, fp32 copy of the result)
               loss, dlogits, hit count         fused softmax cross-entropy (``softmax_xent``, any class count)
    backward   dfeatures = dlogits W            ``gemm_nt``
               dW = dlogits^T features          ``gemm_nt`` (weight-gradient epilogue, fp32 result)
               db = column sums of dlogits

The gradient of the logits is produced by the forward pass itself (the labels are known there), in fp32 and as
the bf16 operand of the two backward products.  On the GPU the products run in the dtype of ``features`` (bf16
under autocast, else fp32); on the CPU the same statements run in plain PyTorch.  Replaces the ``nn.Linear`` +
``F.cross_entropy`` pair of the ResNet-50 configuration (K2-K5 of the reference graph at 1000 classes,
/root/reference/distributed.py:81-87 with ``--loss standard``).
"""
from __future__ import annotations

import torch

from . import EPI_FWD, EPI_WGRAD, LOSS_STANDARD, gemm_nt, softmax_xent, transpose16

calls = {"kernel": 0, "operators": 0}      # which path ran (reported by the benchmark)


def _gpu_path_ok(feat: torch.Tensor, weight: torch.Tensor) -> bool:
    b, k = feat.shape
    c = weight.shape[0]
    return feat.is_cuda and feat.dtype in (torch.bfloat16, torch.float32) and b % 8 == 0 and k % 8 == 0 and c % 8 == 0


class _LinearXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, weight, bias, labels, loss_mode):
        b, k = feat.shape
        c = weight.shape[0]
        dev, cd = feat.device, feat.dtype
        calls["kernel"] += 1
        x = feat.contiguous()
        w = weight.detach().to(cd).contiguous()                     # compute-dtype copy of the fp32 master
        logits_c = torch.empty((b, c), dtype=cd, device=dev)
        logits = logits_c if cd == torch.float32 else torch.empty((b, c), dtype=torch.float32, device=dev)
        gemm_nt(x, w, logits_c, EPI_FWD, bias=bias.detach().float(), relu=False,
                out_f32=None if cd == torch.float32 else logits)
        dl32 = torch.empty((b, c), dtype=torch.float32, device=dev)
        dl_c = dl32 if cd == torch.float32 else torch.empty((b, c), dtype=cd, device=dev)
        r = softmax_xent(logits, labels, loss_mode, grad_scale=1.0 / b, dlogits=dl32,
                         dlogits_bf=None if cd == torch.float32 else dl_c)
        ctx.save_for_backward(x, w, dl32, dl_c)
        hits = r.correct.to(torch.float32)
        ctx.mark_non_differentiable(hits, logits)
        return r.loss_sum / b, hits, logits

    @staticmethod
    def backward(ctx, g_loss, _g_hits, _g_logits):
        x, w, dl32, dl_c = ctx.saved_tensors
        b, k = x.shape
        c = w.shape[0]
        dev, cd = x.device, x.dtype
        # transposed operand copies: every product is A[M,K] . B[N,K]^T with both operands k-contiguous
        if cd == torch.float32:
            w_t, x_t, dl_t = w.t().contiguous(), x.t().contiguous(), dl_c.t().contiguous()
        else:
            w_t = torch.empty((k, c), dtype=cd, device=dev)
            x_t = torch.empty((k, b), dtype=cd, device=dev)
            dl_t = torch.empty((c, b), dtype=cd, device=dev)
            transpose16(w, w_t)
            transpose16(x, x_t)
            transpose16(dl_c, dl_t)
        dfeat = torch.empty((b, k), dtype=cd, device=dev)
        gemm_nt(dl_c, w_t, dfeat, EPI_FWD)                             # [B,C] . [K,C]^T
        dw = torch.empty((c, k), dtype=torch.float32, device=dev)
        gemm_nt(dl_t, x_t, dw, EPI_WGRAD)                              # [C,B] . [K,B]^T
        db = dl32.sum(0)
        # the forward pass already scaled dlogits by 1/B; the incoming gradient of the mean loss is a scalar
        return dfeat * g_loss.to(cd), dw * g_loss, db * g_loss, None, None


def linear_xent(features: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, labels: torch.Tensor,
                loss_mode: int = LOSS_STANDARD):
    """Mean loss, hit count (float tensor) and fp32 logits of ``softmax_xent(features @ weight.T + bias, labels)``."""
    if _gpu_path_ok(features, weight):
        return _LinearXent.apply(features, weight, bias, labels, loss_mode)
    calls["operators"] += 1
    logits = torch.nn.functional.linear(features.float(), weight, bias)
    if loss_mode == LOSS_STANDARD:
        loss = torch.nn.functional.cross_entropy(logits, labels)
    else:
        y = torch.softmax(logits, 1)
        lab = labels if labels.dim() == 2 else torch.nn.functional.one_hot(labels, logits.shape[1]).float()
        loss = -(lab * torch.log_softmax(y, 1)).sum(1).mean()
    tgt = labels if labels.dim() == 1 else labels.argmax(1)
    return loss, (logits.argmax(1) == tgt).sum().float(), logits

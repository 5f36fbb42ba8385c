"""Batch normalisation fused with the ReLU and the residual addition that follow it (``csrc/kernels/batchnorm.hip``).

``FusedBatchNormAct`` is an ``nn.BatchNorm2d`` (same parameters, buffers and checkpoint names) whose forward is

    y = act(batch_norm(x) [+ residual])

On a GPU, in training mode, with bf16 channels-last activations of 8 * 2^k channels, forward and backward are
three launches each of the hand-written kernels: 8 tensor-sized memory passes per layer and training step (10
with a residual) instead of the 13 (16) of the operator sequence batch_norm / add / relu.  Everything else (CPU, fp32, evaluation mode,
other layouts) is that operator sequence, which is also the test oracle.  ``DTF_FUSED_BN=0`` forces the operator
sequence.

The reference has no convolutional model (/root/reference/distributed.py:78-81 is the whole network); this
operator belongs to the ResNet-50 configuration of the benchmark list.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

_workspaces = {}
_protos_done = False
calls = {"kernel": 0, "operators": 0}     # forward calls by path (benchmarks report which one ran)


def _kernels():
    global _protos_done
    lib = _lib.kernels()
    if not _protos_done:
        vp, fp = C.c_void_p, C.c_float
        lib.dtf_bn_workspace_floats.restype = C.c_int64
        lib.dtf_bn_workspace_floats.argtypes = [C.c_int]
        lib.dtf_bn_supported.restype = C.c_int
        lib.dtf_bn_supported.argtypes = [C.c_int64, C.c_int]
        lib.dtf_bn_fwd.restype = C.c_int
        lib.dtf_bn_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, fp, fp, vp]
        lib.dtf_bn_bwd.restype = C.c_int
        lib.dtf_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
        _protos_done = True
    return lib


def _workspace(device, channels: int) -> torch.Tensor:
    """Per-device scratch for the per-block partial sums; calls on one stream run one after the other, so one
    buffer serves every layer."""
    need = int(_kernels().dtf_bn_workspace_floats(channels))
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = _workspaces[key] = torch.empty(max(need, 2 * 1024 * 512), dtype=torch.float32, device=device)
    return ws


def _nhwc_rows(t: torch.Tensor) -> Optional[int]:
    """Rows of the [N*H*W, C] matrix a dense channels-last 4-D tensor is in memory, else None."""
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        return None
    return t.shape[0] * t.shape[2] * t.shape[3]


def kernel_path_takes(x: torch.Tensor, residual: Optional[torch.Tensor], training: bool) -> bool:
    if not (training and x.is_cuda and x.dtype == torch.bfloat16 and os.environ.get("DTF_FUSED_BN", "1") != "0"):
        return False
    m = _nhwc_rows(x)
    if m is None or not _kernels().dtf_bn_supported(m, x.shape[1]):
        return False
    return residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and residual.is_cuda)


def reference_bn_act(x, weight, bias, running_mean, running_var, residual, relu, training, momentum, eps):
    """The operator sequence the kernels replace."""
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, relu, momentum, eps, fork=False):
        lib = _kernels()
        m, ch = _nhwc_rows(x), x.shape[1]
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x)                                   # keeps the channels-last strides
        save = torch.empty(4 * ch, dtype=torch.float32, device=x.device)
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        # residual + ReLU: the mask of the backward pass as one bit per element (the kernels then never read y again)
        bits = torch.empty(m * ch // 8, dtype=torch.uint8, device=x.device) if (relu and residual is not None) else None
        _lib.check(lib.dtf_bn_fwd(x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                  bits.data_ptr() if bits is not None else None, w32.data_ptr(), b32.data_ptr(),
                                  running_mean.data_ptr() if running_mean is not None else None,
                                  running_var.data_ptr() if running_var is not None else None,
                                  save.data_ptr(), _workspace(x.device, ch).data_ptr(), m, ch, 1 if relu else 0,
                                  float(momentum), float(eps), _lib.stream_handle()), "bn_fwd")
        ctx.mask = 0 if not relu else (2 if residual is not None else 1)
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, bits, save)
        ctx.wdtype, ctx.bdtype = weight.dtype, bias.dtype
        if fork:
            # two handles of one output: the gradients of its two consumers then reach backward() separately, and
            # the reduction kernel adds them while it reads them (no addition kernel of autograd's in front)
            ctx.set_materialize_grads(False)
            return y, y.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        lib = _kernels()
        x, bits, save = ctx.saved_tensors
        m, ch = _nhwc_rows(x), x.shape[1]

        def dense(g):
            if g.dtype != x.dtype or not g.is_contiguous(memory_format=torch.channels_last):
                g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
            return g
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return (None,) * 10
        dy = dense(dy)
        if dy2 is not None:
            dy2 = dense(dy2)
            if ctx.mask != 2:
                dy, dy2 = dy + dy2, None                          # the kernels add only in front of the ReLU mask of a residual layer
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.mask == 2 else None
        dgb = torch.empty(2 * ch, dtype=torch.float32, device=x.device)
        coef = torch.empty(3 * ch, dtype=torch.float32, device=x.device)
        _lib.check(lib.dtf_bn_bwd(dy.data_ptr(), dy2.data_ptr() if dy2 is not None else None, x.data_ptr(),
                                  bits.data_ptr() if bits is not None else None, dx.data_ptr(),
                                  dres.data_ptr() if dres is not None else None, save.data_ptr(), dgb.data_ptr(),
                                  dgb.data_ptr() + 4 * ch, _workspace(x.device, ch).data_ptr(), coef.data_ptr(), m, ch,
                                  ctx.mask, _lib.stream_handle()), "bn_bwd")
        if ctx.has_res and dres is None:
            dres = dy                                             # no activation: the residual input sees dy itself
        return (dx, dgb[:ch].to(ctx.wdtype), dgb[ch:].to(ctx.bdtype), None, None, dres if ctx.has_res else None,
                None, None, None, None)


def bn_act(x, weight, bias, running_mean=None, running_var=None, residual=None, relu=True, training=True,
           momentum=0.1, eps=1e-5, fork=False):
    """``act(batch_norm(x) [+ residual])`` with the batch statistics of ``x`` (training) or the running ones.
    ``fork``: the result as a pair of handles for two consumers (a block's main branch and its skip connection)."""
    takes = kernel_path_takes(x, residual, training)
    calls["kernel" if takes else "operators"] += 1
    if takes:
        return _BnAct.apply(x, weight, bias, running_mean, running_var, residual, relu, momentum, eps, fork)
    y = reference_bn_act(x, weight, bias, running_mean, running_var, residual, relu, training, momentum, eps)
    return (y, y) if fork else y


class FusedBatchNormAct(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` + optional residual addition + optional ReLU in one operator."""

    def __init__(self, num_features, relu: bool = True, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__(num_features, eps=eps, momentum=momentum)
        self.relu = relu
        self.count_batches = True       # False: the owner of the model advances ``num_batches_tracked`` of all layers at once

    def forward(self, x, residual=None, fork=False):              # noqa: D102
        training = self.training or self.running_mean is None
        if training and self.count_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, residual, self.relu, training,
                      self.momentum, self.eps, fork)

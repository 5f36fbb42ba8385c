"""Max pooling for channels-last bf16 activations on the framework's kernels (``csrc/kernels/maxpool.hip``).

``max_pool2d(x, kernel, stride, padding)`` is ``F.max_pool2d`` (floor mode, no dilation).  On a GPU, with a dense
channels-last bf16 input of a multiple of 8 channels, forward and backward are one launch each: the forward pass keeps
one byte per output element (the position of the maximum inside its window), the backward pass gathers.  Everything
else runs the operator, which is also the test oracle.  ``DTF_OWN_MAXPOOL=0`` forces the operator.

The reference has no pooling layer (/root/reference/distributed.py:78-81 is the whole network); this operator
belongs to the ResNet-50 configuration of the benchmark list.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import _lib

_protos_done = False
calls = {"kernel": 0, "operators": 0}


def _kernels():
    global _protos_done
    lib = _lib.kernels()
    if not _protos_done:
        vp, i = C.c_void_p, C.c_int
        for fn in (lib.dtf_maxpool_fwd, lib.dtf_maxpool_bwd):
            fn.restype = C.c_int
            fn.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, vp]
        _protos_done = True
    return lib


def kernel_path_takes(x: torch.Tensor, kernel: int, stride: int, padding: int) -> bool:
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and os.environ.get("DTF_OWN_MAXPOOL", "1") != "0"):
        return False
    n, c, h, w = x.shape
    if c % 8 or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if not (1 <= kernel <= 15 and stride >= 1 and 0 <= 2 * padding <= kernel):
        return False
    return h + 2 * padding >= kernel and w + 2 * padding >= kernel


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        n, c, h, w = x.shape
        oh, ow = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
        y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        pos = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
        _lib.check(_kernels().dtf_maxpool_fwd(x.data_ptr(), y.data_ptr(), pos.data_ptr(), n, h, w, c, oh, ow, kernel, stride,
                                              padding, _lib.stream_handle()), "maxpool_fwd")
        ctx.save_for_backward(pos)
        ctx.dims = (n, h, w, c, oh, ow, kernel, stride, padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        (pos,) = ctx.saved_tensors
        n, h, w, c, oh, ow, kernel, stride, padding = ctx.dims
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        _lib.check(_kernels().dtf_maxpool_bwd(dy.data_ptr(), pos.data_ptr(), dx.data_ptr(), n, h, w, c, oh, ow, kernel, stride,
                                              padding, _lib.stream_handle()), "maxpool_bwd")
        return dx, None, None, None


def max_pool2d(x: torch.Tensor, kernel: int, stride: int, padding: int = 0) -> torch.Tensor:
    takes = kernel_path_takes(x, kernel, stride, padding)
    calls["kernel" if takes else "operators"] += 1
    if takes:
        return _MaxPool.apply(x, kernel, stride, padding)
    return F.max_pool2d(x, kernel, stride=stride, padding=padding)

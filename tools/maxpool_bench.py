#!/usr/bin/env python3
"""Max pooling 3x3 / stride 2 / padding 1 on [batch, 64, 112, 112] channels-last bf16 (the ResNet stem): the framework's
kernels (ops/maxpool.py) against F.max_pool2d, forward and backward, microseconds per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.ops import maxpool as mp

dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(batch, 64, 112, 112, device=dev).clamp_min(0).bfloat16().contiguous(memory_format=torch.channels_last)


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, op in (("own kernels", lambda t: mp.max_pool2d(t, 3, 2, 1)), ("F.max_pool2d", lambda t: F.max_pool2d(t, 3, stride=2, padding=1))):
    xr = x.detach().requires_grad_(True)
    y = op(xr)
    dy = torch.randn_like(y)
    fwd = timed(lambda: op(x))
    both = timed(lambda: op(xr).backward(dy))
    mb = x.numel() * 2 / 1e6
    print(f"{name:14s} forward {fwd:7.1f} us, forward + backward {both:7.1f} us   (input {mb:.0f} MB)")

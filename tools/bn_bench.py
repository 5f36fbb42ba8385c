#!/usr/bin/env python3
"""Fused batch norm + residual + ReLU (csrc/kernels/batchnorm.hip) against PyTorch's batch_norm / add / relu on the
activation shapes of ResNet-50 at batch 256 (channels-last bf16): forward + backward time per call and the
bandwidth the fused path reaches on its tensor passes (8 per call, 10 with a residual).

    python tools/bn_bench.py [--batch 256] [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd.ops import batchnorm as bn

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only_fused", action="store_true", help="run only the kernels (for counter collection)")
a = ap.parse_args()
dev = torch.device("cuda", 0)

# (channels, height = width, residual) of the batch norms of a ResNet-50
SHAPES = [(64, 112, False), (64, 56, False), (256, 56, True), (128, 28, False), (512, 28, True), (256, 14, False),
          (1024, 14, True), (512, 7, False), (2048, 7, True)]


def run(fn, x, res, w, b, rm, rv, dy, iters):
    def once():
        xx = x.detach().requires_grad_(True)
        rr = res.detach().requires_grad_(True) if res is not None else None
        y = fn(xx, w, b, rm, rv, rr, True, True, 0.1, 1e-5)
        y.backward(dy)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        once()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3     # us


print(f"{'C':>5} {'HxW':>8} {'res':>4} {'MB':>7} {'fused us':>9} {'ops us':>8} {'speedup':>7} {'fused TB/s':>10}")
for c, hw, residual in SHAPES:
    x = torch.randn(a.batch, hw, hw, c, device=dev).bfloat16().permute(0, 3, 1, 2)
    res = torch.randn(a.batch, hw, hw, c, device=dev).bfloat16().permute(0, 3, 1, 2) if residual else None
    dy = torch.randn(a.batch, hw, hw, c, device=dev).bfloat16().permute(0, 3, 1, 2)
    w = torch.ones(c, device=dev, requires_grad=True)
    b = torch.zeros(c, device=dev, requires_grad=True)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mb = x.numel() * 2 / 1e6
    # tensor passes of the fused path: forward stats 1, apply 2 (+1: residual); backward reduce 2 (+1: masked dy;
    # the mask bits are 1/16 of a pass each way), dx 3
    passes = 8 + (2.125 if residual else 0)
    tf = run(bn.bn_act, x, res, w, b, rm, rv, dy, a.iters)
    if a.only_fused:
        print(f"{c:5d} {hw:4d}x{hw:<3d} {str(residual)[0]:>4} {mb:7.1f} {tf:9.1f}")
        continue
    to = run(bn.reference_bn_act, x, res, w, b, rm, rv, dy, a.iters)
    print(f"{c:5d} {hw:4d}x{hw:<3d} {str(residual)[0]:>4} {mb:7.1f} {tf:9.1f} {to:8.1f} {to / tf:7.2f} {passes * mb / tf:10.2f}")

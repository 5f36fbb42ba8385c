#!/usr/bin/env python3
"""1x1 convolutions of ResNet-50 at batch 256 (channels-last bf16): MIOpen through PyTorch against the framework's
GEMM kernels on the same memory ([N*H*W, Cin] x [Cout, Cin]^T).  Prints microseconds per call: forward, data
gradient, weight gradient.

    python tools/conv1x1_bench.py [--batch 256]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True

# (Cin, Cout, H = W, count per network) of the stride-1 1x1 convolutions
SHAPES = [(64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1), (128, 512, 28, 4), (512, 128, 28, 3),
          (512, 256, 28, 1), (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1), (512, 2048, 7, 3), (2048, 512, 7, 2)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


print(f"{'Cin':>5} {'Cout':>5} {'HxW':>4} {'n':>2} | {'lib fwd':>8} {'lib dx':>8} {'lib dw':>8} | {'gemm fwd':>8} {'gemm dx':>8} {'gemm dw':>8} {'(+transp)':>9}")
tot = [0.0] * 8
for cin, cout, hw, cnt in SHAPES:
    M = a.batch * hw * hw
    x = torch.randn(a.batch, hw, hw, cin, device=dev).bfloat16().permute(0, 3, 1, 2)
    dy = torch.randn(a.batch, hw, hw, cout, device=dev).bfloat16().permute(0, 3, 1, 2)
    w = (torch.randn(cout, cin, 1, 1, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    cb = lambda mask: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    t_lf = timeit(lambda: F.conv2d(x, w), a.iters)
    t_ldx = timeit(lambda: cb([True, False, False]), a.iters)
    t_ldw = timeit(lambda: cb([False, True, False]), a.iters)
    x2, dy2, w2 = x.permute(0, 2, 3, 1).reshape(M, cin), dy.permute(0, 2, 3, 1).reshape(M, cout), w.reshape(cout, cin)
    wt = w2.t().contiguous()
    y2 = torch.empty(M, cout, dtype=torch.bfloat16, device=dev)
    dx2 = torch.empty(M, cin, dtype=torch.bfloat16, device=dev)
    dw = torch.empty(cout, cin, dtype=torch.float32, device=dev)
    t_gf = timeit(lambda: ops.gemm_nt(x2, w2, y2), a.iters)
    t_gdx = timeit(lambda: ops.gemm_nt(dy2, wt, dx2), a.iters)
    xt, dyt = x2.t().contiguous(), dy2.t().contiguous()
    t_gdw = timeit(lambda: ops.gemm_nt(dyt, xt, dw, epilogue=ops.EPI_WGRAD), a.iters)
    t_tr = timeit(lambda: (x2.t().contiguous(), dy2.t().contiguous()), a.iters)
    x2c, dy2c = x2.contiguous(), dy2.contiguous()
    t_own = timeit(lambda: ops.conv1x1_wgrad(x2c, dy2c), a.iters)
    dw_ref = cb([False, True, False])[1].reshape(cout, cin).float()
    err_dw = float((ops.conv1x1_wgrad(x2c, dy2c).float() - dw_ref).abs().max() / dw_ref.abs().max())
    ref = F.conv2d(x, w).permute(0, 2, 3, 1).reshape(M, cout).float()
    err = float((y2.float() - ref).abs().max())
    row = [t_lf, t_ldx, t_ldw, t_gf, t_gdx, t_gdw, t_tr, t_own]
    for i, v in enumerate(row):
        tot[i] += cnt * v
    print(f"{cin:5d} {cout:5d} {hw:4d} {cnt:2d} | {t_lf:8.1f} {t_ldx:8.1f} {t_ldw:8.1f} | {t_gf:8.1f} {t_gdx:8.1f} {t_gdw:8.1f} {t_tr:9.1f} | own dw {t_own:8.1f} ({2.0 * M * (cin + cout) / t_own / 1e6:5.2f} TB/s, rel err {err_dw:.4f})  err {err:.3f}")
print("per network (us):", " ".join(f"{v:9.0f}" for v in tot))

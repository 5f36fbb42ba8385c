#!/usr/bin/env python3
"""Plain bf16 products C = A . B^T at the wide model's shapes: the 8-wave kernel (forward epilogue without bias, ReLU or
transposed copy) against torch.matmul (hipBLASLt / rocBLAS), microseconds and TFLOP/s.

    python tools/gemm_vs_library.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd import ops

dev = torch.device("cuda", 0)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'M':>6} {'N':>6} {'K':>6} | {'own us':>8} {'TFLOP/s':>8} | {'library us':>10} {'TFLOP/s':>8} | {'library, B not transposed':>26}")
for M, N, K in [(4096, 4096, 4096), (8192, 4096, 4096), (4096, 4096, 8192), (4096, 4096, 1024), (8192, 8192, 8192)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    bt = b.t().contiguous()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_own = timeit(lambda: ops.gemm_nt(a, b, out, ops.EPI_FWD))
    t_lib = timeit(lambda: torch.matmul(a, b.t(), out=out))
    t_lib2 = timeit(lambda: torch.matmul(a, bt, out=out))
    fl = 2.0 * M * N * K
    err = float((out.float() - (a.float() @ bt.float())).abs().max())
    print(f"{M:6d} {N:6d} {K:6d} | {t_own:8.1f} {fl / t_own / 1e6:8.0f} | {t_lib:10.1f} {fl / t_lib / 1e6:8.0f} | {t_lib2:10.1f} {fl / t_lib2 / 1e6:8.0f}   (library result vs fp32: {err:.3f})")

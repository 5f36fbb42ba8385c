#!/usr/bin/env python3
"""Latency of the small GEMM kernel versus shape (graph-replayed, warm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops
from kbench import timeit

dev = torch.device("cuda")
for (M, K, N) in [(200, 784, 100), (200, 512, 100), (200, 256, 100), (200, 128, 100), (200, 64, 100), (200, 16, 100),
                  (16, 784, 16), (16, 128, 16), (200, 784, 16), (1600, 784, 100), (200, 784, 800)]:
    x = torch.rand(max(M, 1000), K, device=dev)
    w = torch.rand(K, N, device=dev) * 0.1
    b = torch.rand(N, device=dev)
    out = torch.zeros(M, N, device=dev)
    rows = torch.randperm(x.shape[0], device=dev)[:M].to(torch.int32)
    t_g = timeit(lambda: ops.linear_fwd(x, w, b, relu=True, rows=rows, out=out, m=M))
    t_n = timeit(lambda: ops.linear_fwd(x, w, b, relu=True, out=out, m=M))
    print(f"M={M:5d} K={K:4d} N={N:4d}: gather {t_g:6.2f} us   contiguous {t_n:6.2f} us")

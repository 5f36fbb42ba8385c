#!/usr/bin/env python3
"""Forward product 4096 x 4096 x 4096 (bf16, bias + ReLU, row-major and transposed output) on the 8-wave kernel, 200
launches in a hipGraph: microseconds per launch.  Run through tools/gemm_store_bound.sh."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd import ops

skip = sys.argv[1] if len(sys.argv) > 1 else "0"
what = {"0": "all stores", "1": "without the stores of C", "2": "without the stores of the transposed copy",
        "3": "without any store of the result"}[skip]
ops.gemm_set_modes(pingpong=1)
dev = torch.device("cuda")
for (M, N, K, with_t) in ((4096, 4096, 4096, True), (4096, 4096, 4096, False), (8192, 4096, 4096, True)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    out_t = torch.zeros(N, M, dtype=torch.bfloat16, device=dev) if with_t else None
    bias = torch.rand(N, device=dev)
    run = lambda: ops.gemm_nt(a, b, out, ops.EPI_FWD, bias=bias, relu=True, out_t=out_t)
    for _ in range(3):
        run()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            run()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
    us = sorted(ts)[len(ts) // 2]
    print(f"{M}x{N}x{K} {'C + C^T' if with_t else 'C only':8s} {what:45s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")

#!/usr/bin/env python3
"""TFLOP/s of the tiled GEMM on the wide-MLP shapes (random operands)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops

dev = torch.device("cuda")


def bench(M, N, K, dtype, epi, reps=20):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(dtype)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(dtype)
    out = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_WGRAD else dtype, device=dev)
    mask = (torch.rand(M, N, device=dev) - 0.5).to(dtype) if epi == ops.EPI_BWD else None
    mask_t = mask.t().contiguous() if mask is not None else None
    bias = torch.rand(N, device=dev) if epi == ops.EPI_FWD else None
    out_t = torch.zeros(N, M, dtype=dtype, device=dev) if epi != ops.EPI_WGRAD else None
    bias_out = torch.zeros(N, device=dev) if epi == ops.EPI_WGRAD else None
    f = lambda: ops.gemm_nt(a, b, out, epi, bias=bias, relu=True, mask=mask, mask_t=mask_t, out_t=out_t, bias_out=bias_out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = None
    if epi == ops.EPI_FWD:
        g = lambda: torch.relu(torch.addmm(bias.to(dtype), a, b.t()))
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            g()
        e1.record()
        torch.cuda.synchronize()
        ref = e0.elapsed_time(e1) / reps
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"{str(dtype)[6:]:9s} epi={epi} M={M:5d} N={N:5d} K={K:5d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s" +
          (f"   (torch addmm+relu {ref:8.3f} ms {2.0*M*N*K/ref/1e9:7.1f} TF)" if ref else ""))


if len(sys.argv) > 1:
    ops.gemm_set_modes(pingpong=int(sys.argv[1]))      # 0: 4-wave kernels only, 1/2: 8-wave ping-pong kernel
    print("pingpong mode", sys.argv[1])
for dtype in (torch.bfloat16, torch.float32):
    for (M, N, K) in [(8192, 4096, 4096), (1024, 4096, 4096), (8192, 4096, 784)]:
        bench(M, N, K, dtype, ops.EPI_FWD)
    bench(8192, 4096, 4096, dtype, ops.EPI_BWD)
    bench(4096, 4096, 8192, dtype, ops.EPI_WGRAD)

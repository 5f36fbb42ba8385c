#!/usr/bin/env python3
"""Where the 8-wave bf16 GEMM loses its time at the wide-MLP shape: a sweep over K at M = N = 4096 (256 blocks of
256x256, one per CU).  time(K) = fixed + K * per_k: `per_k` is the main loop's rate (what the two wave groups hide of
each other's fragment reads and staging), `fixed` is what happens once per block (launch, first tiles in flight
before the first MFMA, epilogue through LDS, stores).  Prints both, and the MFMA-pipe utilisation each implies.

    python tools/gemm_ksweep.py [epilogue: fwd|fwd_t|bwd|wgrad]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops

dev = torch.device("cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd_t"
M = N = 4096
PEAK = 2.5e15          # dense bf16 MFMA peak, FLOP/s


def run(K, reps=30):
    T = torch.bfloat16
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(T)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(T)
    if kind == "wgrad":
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        bo = torch.zeros(N, device=dev)
        f = lambda: ops.gemm_nt(a, b, out, ops.EPI_WGRAD, bias_out=bo)
    elif kind == "bwd":
        out, out_t = torch.zeros(M, N, dtype=T, device=dev), torch.zeros(N, M, dtype=T, device=dev)
        mask = (torch.rand(M, N, device=dev) - 0.5).to(T)
        mask_t = mask.t().contiguous()
        f = lambda: ops.gemm_nt(a, b, out, ops.EPI_BWD, mask=mask, mask_t=mask_t, out_t=out_t)
    else:
        out = torch.zeros(M, N, dtype=T, device=dev)
        out_t = torch.zeros(N, M, dtype=T, device=dev) if kind == "fwd_t" else None
        bias = torch.rand(N, device=dev)
        f = lambda: ops.gemm_nt(a, b, out, ops.EPI_FWD, bias=bias, relu=True, out_t=out_t)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3      # us


ks = [512, 1024, 2048, 4096, 8192, 16384]
ts = [run(k) for k in ks]
for k, t in zip(ks, ts):
    print(f"K={k:6d}: {t:8.2f} us  {2.0 * M * N * k / t / 1e6:7.1f} TFLOP/s")
mx, my = sum(ks) / len(ks), sum(ts) / len(ts)
slope = sum((x - mx) * (y - my) for x, y in zip(ks, ts)) / sum((x - mx) ** 2 for x in ks)
fixed = my - slope * mx
loop_rate = 2.0 * M * N / slope / 1e6
print(f"epilogue kind {kind}: fixed {fixed:.2f} us per launch, main loop {slope * 1e3:.3f} ns per unit of K = {loop_rate:.0f} TFLOP/s "
      f"= {loop_rate * 1e12 / PEAK * 100:.0f} % of the dense bf16 peak")
print(f"at K=4096: main loop {slope * 4096:.1f} us + fixed {fixed:.1f} us -> the fixed part is {fixed / (fixed + slope * 4096) * 100:.0f} % of the kernel")

#!/usr/bin/env python3
"""Time the output-layer kernels of the wide bf16 network (csrc/kernels/head_wide.hip) on the bench shape, each in a
loop of its own (operands warm in the caches) and behind a 64 MB write that pushes them out of the L2s.

    python tools/head_wide_probe.py [--batch 4096] [--hidden 4096] [--classes 10]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=30, flush=None):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for _ in range(3):
        fn()
    for a, b in ev:
        if flush is not None:
            flush.add_(1.0)
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--classes", type=int, default=10)
    a = ap.parse_args()
    from distributed_tensorflow_amd import ops
    dev = torch.device("cuda", 0)
    B, H, Cn = a.batch, a.hidden, a.classes
    h = torch.relu(torch.randn(B, H, device=dev)).to(torch.bfloat16)
    ht = h.t().contiguous()
    w2, b2 = torch.randn(H, Cn, device=dev) * 0.02, torch.zeros(Cn, device=dev)
    labels = torch.zeros(B, Cn, device=dev)
    labels[torch.arange(B), torch.randint(0, Cn, (B,), device=dev)] = 1.0
    dz2 = torch.zeros(B, 16, device=dev)
    dz1, dz1t = torch.zeros(B, H, dtype=torch.bfloat16, device=dev), torch.zeros(H, B, dtype=torch.bfloat16, device=dev)
    nb = (B + 15) // 16
    lp, cp = torch.zeros(nb, device=dev), torch.zeros(nb, dtype=torch.int32, device=dev)
    ha = ops.build_head_wide_args(h, ht, w2, b2, labels, B, 0, dz2=dz2, dz1_bf=dz1, dz1_bf_t=dz1t, loss_part=lp, correct_part=cp)
    dw, db = torch.zeros(H, Cn, device=dev), torch.zeros(Cn, device=dev)
    wa = ops.build_out_wgrad_wide_args(ht, dz2, dw, db, B)
    flush = torch.zeros(64 << 20, device=dev)           # 256 MB: larger than L2s + most of the memory-side cache
    res = {"head_wide_us_warm": timeit(lambda: ops.head_wide_bf16(ha)),
           "head_wide_us_cold": timeit(lambda: ops.head_wide_bf16(ha), flush=flush),
           "out_wgrad_wide_us_warm": timeit(lambda: ops.out_wgrad_wide(wa)),
           "out_wgrad_wide_us_cold": timeit(lambda: ops.out_wgrad_wide(wa), flush=flush)}
    hf = h.float()
    h32 = hf.clone()
    dz1f = torch.zeros(B, H, device=dev)
    oa = ops.build_head_args(h32, w2, b2, labels, B, B, 0, dz2=dz2, dz1=None, dz1_bf=dz1, dz1_bf_t=dz1t, loss_part=lp,
                             correct_part=cp, post_part=torch.zeros(nb, dtype=torch.int32, device=dev))
    res["head_fp32_4wave_us_warm"] = timeit(lambda: ops.head(h32, w2, b2, labels, B, B, args=oa))
    res["head_fp32_4wave_us_cold"] = timeit(lambda: ops.head(h32, w2, b2, labels, B, B, args=oa), flush=flush)
    print(res)


if __name__ == "__main__":
    main()

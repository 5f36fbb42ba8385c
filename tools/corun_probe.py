#!/usr/bin/env python3
"""Can a streaming kernel with few registers run BESIDE the 8-wave GEMM on the same CUs?  The GEMM workgroup takes 2 x 240
of the 512 vector registers per SIMD lane and 128 of 160 KB LDS; a kernel that needs <= 32 registers and no LDS still
fits next to it.  Times, on two streams: 20 forward products 4096^3 alone, a streaming pass (dst = a*dst + b*src,
18 registers) over 1 GiB alone, and both together.

    python tools/corun_probe.py [--epilogue fwd|bwd|wgrad]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epilogue", default="fwd", choices=["fwd", "wgrad"])
    ap.add_argument("--gemms", type=int, default=20)
    a = ap.parse_args()
    from distributed_tensorflow_amd import ops
    dev = torch.device("cuda", 0)
    M = N = K = 4096
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    out_t = torch.zeros(N, M, dtype=torch.bfloat16, device=dev)
    out32 = torch.zeros(M, N, device=dev)
    bias = torch.rand(N, device=dev)
    big_a = torch.zeros(128 << 20, device=dev)          # 512 MB each: one pass moves 1.5 GB
    big_b = torch.ones(128 << 20, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def gemms():
        with torch.cuda.stream(sa):
            for _ in range(a.gemms):
                if a.epilogue == "fwd":
                    ops.gemm_nt(x, w, out, ops.EPI_FWD, bias=bias, relu=True, out_t=out_t)
                else:
                    ops.gemm_nt(x, w, out32, ops.EPI_WGRAD)

    def stream_pass():
        with torch.cuda.stream(sb):
            ops.accumulate(big_a, big_b, 0.5, 0.25)

    def timed(fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    for f in (gemms, stream_pass):
        f()
    res = {}
    for name, fns in (("gemms_alone_us", [gemms]), ("stream_alone_us", [stream_pass]), ("both_us", [gemms, stream_pass]),
                      ("both_stream_first_us", [stream_pass, gemms])):
        res[name] = sorted(timed(fns) for _ in range(7))[3]
    res["sum_us"] = res["gemms_alone_us"] + res["stream_alone_us"]
    res["per_gemm_alone_us"] = res["gemms_alone_us"] / a.gemms
    res["epilogue"] = a.epilogue
    print(res)


if __name__ == "__main__":
    main()

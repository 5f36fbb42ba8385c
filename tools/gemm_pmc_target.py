#!/usr/bin/env python3
"""Small target for `rocprofv3 --pmc`: a few dispatches of the bf16 GEMM kernels on one shape.
    python tools/gemm_pmc_target.py [pingpong mode 0|1|2] [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 4096, 4096)
ops.gemm_set_modes(pingpong=mode)
dev = torch.device("cuda")
a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
out_t = torch.zeros(N, M, dtype=torch.bfloat16, device=dev)
bias = torch.rand(N, device=dev)
for _ in range(4):
    ops.gemm_nt(a, b, out, ops.EPI_FWD, bias=bias, relu=True, out_t=out_t)
torch.cuda.synchronize()

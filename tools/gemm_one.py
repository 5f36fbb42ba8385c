#!/usr/bin/env python3
"""One tiled-GEMM shape, a few launches (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda")
M, N, K = 8192, 4096, 4096
a = (torch.rand(M, K, device=dev) * 2 - 1).to(dtype)
b = (torch.rand(N, K, device=dev) * 2 - 1).to(dtype)
out = torch.zeros(M, N, dtype=dtype, device=dev)
bias = torch.rand(N, device=dev)
for _ in range(5):
    ops.gemm_nt(a, b, out, ops.EPI_FWD, bias=bias, relu=True)
torch.cuda.synchronize()

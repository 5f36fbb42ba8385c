#!/usr/bin/env python3
"""Bandwidth of the flat fused Adam kernel (28 bytes of traffic per parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 104_000_000
dev = "cuda"
p, g, m, v = (torch.rand(n, device=dev) for _ in range(4))
for _ in range(3):
    ops.adam_apply(p, g, m, v, 1e-3, step=5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.adam_apply(p, g, m, v, 1e-3, step=5)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"adam n={n}: {ms*1e3:.1f} us, {28*n/ms/1e9:.2f} TB/s")

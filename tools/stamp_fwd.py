#!/usr/bin/env python3
"""Diagnostic: where the forward GEMM of the reference MLP spends its time.
Needs the stamped build:  DTF_KERNELS_LIB=distributed_tensorflow_amd/lib/libdtf_kernels_diag.so"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.ops import _lib

dev = torch.device("cuda")
lib = _lib.kernels()
lib.dtf_set_stamp_buffer.argtypes = [C.c_void_p]
lib.dtf_stamp_now.argtypes = [C.c_void_p, C.c_void_p]
M, K, N = 200, 784, 100
x = torch.rand(55000, K, device=dev)
w = torch.rand(K, N, device=dev) * 0.1
b = torch.rand(N, device=dev)
rows = torch.randperm(55000, device=dev)[:M].to(torch.int32)
out = torch.zeros(M, N, device=dev)
nwg = ((M + 15) // 16) * ((N + 15) // 16)
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
t_ab = torch.zeros(4, dtype=torch.int64, device=dev)
lib.dtf_set_stamp_buffer(stamps.data_ptr())
s = torch.cuda.current_stream().cuda_stream
for it in range(5):
    rows = torch.randperm(55000, device=dev)[:M].to(torch.int32)
    torch.cuda.synchronize()
    lib.dtf_stamp_now(t_ab.data_ptr(), s)
    ops.linear_fwd(x, w, b, relu=True, rows=rows, out=out, m=M)
    lib.dtf_stamp_now(t_ab.data_ptr() + 8, s)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
    t0, t1 = int(t_ab[0]), int(t_ab[1])
    rel = (st[:, :4] - t0) * 10  # ns
    print(f"iter {it}: before->after stamp kernels {(t1 - t0) * 10} ns; WG start min/med/max {rel[:,0].min()}/{int(np.median(rel[:,0]))}/{rel[:,0].max()} ns;"
          f" loads+mfma med {int(np.median(rel[:,1]-rel[:,0]))} ns; reduce med {int(np.median(rel[:,2]-rel[:,1]))} ns;"
          f" epilogue med {int(np.median(rel[:,3]-rel[:,2]))} ns; last WG end {rel[:,3].max()} ns")

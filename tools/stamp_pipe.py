#!/usr/bin/env python3
"""Diagnostic: timeline inside one launch of the pipelined step (csrc/kernels/mlp_pipe.hip).
Needs the stamped build (tools/build_diag.sh):  DTF_KERNELS_LIB=distributed_tensorflow_amd/lib/libdtf_kernels_diag.so"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.ops import _lib
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state

dev = torch.device("cuda")
lib = _lib.kernels()
lib.dtf_set_stamp_buffer_pipe.argtypes = [C.c_void_p]
lib.dtf_stamp_now.argtypes = [C.c_void_p, C.c_void_p]
ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
st = build_mlp_state(784, [100], 10, dev, lr=0.01)
st.initialize(seed=1)
prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 100, [100])
assert prog.pipe
prog.run(16, 8)
torch.cuda.synchronize()
nwg = 512
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
t_ab = torch.zeros(4, dtype=torch.int64, device=dev)
lib.dtf_set_stamp_buffer_pipe(stamps.data_ptr())
s = torch.cuda.current_stream().cuda_stream
# role order (stamps are indexed by role index, not by block): L (14), G (98), Q (49), W (22), T (13)
ranges = [("L", 0, 14), ("G", 14, 112), ("Q", 112, 161), ("W", 161, 183), ("T", 183, 196)]
T_LO, T_HI = 183, 196


def show(tag, rel, lo, hi):
    r = rel[lo:hi]
    cols = " ".join(f"s{j}:{int(np.median(r[:, j])):5d}/{int(r[:, j].max()):5d}" for j in range(8) if (r[:, j] > 0).any())
    print(f"   {tag:3s} (n={hi - lo:3d}) median/max ns since pre-stamp: {cols}")


for it in range(6):
    q = it % prog.P
    stamps.zero_()
    torch.cuda.synchronize()
    lib.dtf_stamp_now(t_ab.data_ptr(), s)
    rc = lib.dtf_pipe_launch_roles(C.byref(prog._args[q]["pipe_step"]), s)
    assert rc == 0
    lib.dtf_stamp_now(t_ab.data_ptr() + 8, s)
    torch.cuda.synchronize()
    sv = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
    t0, t1 = int(t_ab[0]), int(t_ab[1])
    rel = np.where(sv > 0, (sv - t0) * 10, 0)
    rel[T_LO:T_HI, 2:4] = 0
    print(f"launch {it}: stamp->stamp {(t1 - t0) * 10} ns   (L: s0 start, s1 loads issued, s2 stage-1 MFMAs done, s3 barrier, s4 h tiles,"
          " s5 logits, s6 softmax, s7 end; Q/W: s1 MFMAs done; T: s1 row index known)")
    for tag, lo, hi in ranges:
        show(tag, rel, lo, hi)
    tw = sv[T_LO:T_HI]      # role T: slots 0/1 constant clock at start/end, slots 2/3 shader cycle counter at start/end
    ok = (tw[:, 1] > tw[:, 0]) & (tw[:, 3] > tw[:, 2])
    if ok.any():
        ghz = (tw[ok, 3] - tw[ok, 2]) / ((tw[ok, 1] - tw[ok, 0]) * 10.0)
        print(f"   shader clock during the launch (role T workgroups): {ghz.min():.2f}-{ghz.max():.2f} GHz")

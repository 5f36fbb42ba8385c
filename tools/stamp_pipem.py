#!/usr/bin/env python3
"""Diagnostic: timeline inside the row workgroups of one launch of the N-worker pipelined step
(csrc/kernels/mlp_pipe_multi.hip), one rank exchanging with itself through its own region.
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
from distributed_tensorflow_amd.parallel.pipe_exchange import PipeExchange

dev = torch.device("cuda")
lib = _lib.kernels()
lib.dtf_set_stamp_buffer_pipem.argtypes = [C.c_void_p]
lib.dtf_stamp_now.argtypes = [C.c_void_p, C.c_void_p]
ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
st = build_mlp_state(784, [100], 10, dev, lr=0.01)
st.initialize(seed=1)
os.environ["DTF_PIPELINED_STEP"] = "0"
data = DeviceDataset(ds.train, dev, shuffle_seed=0)
prog = MLPProgram(st, data, 100, [100], fused_update=False)
os.environ["DTF_PIPELINED_STEP"] = "1"
data.enable_peer_shuffles([0], 0)
x = PipeExchange(dev, 0, 1, timeout_s=2.0)
prog.use_pipe_exchange(x, 0.01)
prog.run(16, 8)
torch.cuda.synchronize()
nwg = 512
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
t_ab = torch.zeros(4, dtype=torch.int64, device=dev)
lib.dtf_set_stamp_buffer_pipem(stamps.data_ptr())
s = torch.cuda.current_stream().cuda_stream
names = ["start", "operands + strips in", "D(s-2) term", "boxes of s-1 checked, D(s-1) term", "U rows", "logits", "softmax", "boxes stored"]
for it in range(6):
    # whole steps, so that the counters advance and every launch finds the flags of the one before
    q = prog.steps_enqueued % prog.P
    stamps.zero_()
    torch.cuda.synchronize()
    lib.dtf_stamp_now(t_ab.data_ptr(), s)
    rc = lib.dtf_pipem_launch_roles(C.byref(prog._args[q]["pipe_step"]), s)
    assert rc == 0
    lib.dtf_stamp_now(t_ab.data_ptr() + 8, s)
    prog.steps_enqueued += 1
    torch.cuda.synchronize()
    sv = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
    t0, t1 = int(t_ab[0]), int(t_ab[1])
    rel = np.where(sv > 0, (sv - t0) * 10, 0)
    print(f"launch {it}: stamp->stamp {(t1 - t0) * 10} ns")
    for tag, lo, hi in (("L train", 0, 7), ("L post", 7, 14)):
        r = rel[lo:hi]
        print(f"   {tag:8s} median/max ns since pre-stamp: " +
              "  ".join(f"{names[j]}: {int(np.median(r[:, j]))}/{int(r[:, j].max())}" for j in range(8) if (r[:, j] > 0).any()))
print("timed out:", x.timed_out())

#!/usr/bin/env python3
"""Diagnostic: timeline inside the two launches of the fused reference step.
Needs the stamped build (tools/build_diag.sh):  DTF_KERNELS_LIB=distributed_tensorflow_amd/lib/libdtf_kernels_diag.so

    python tools/stamp_mlp2.py [update | gradients | async | sharded | replicated]

update (default): one worker, SGD in the weight-gradient epilogue (mode 1).  The others run the multi-worker code on
a one-rank group: gradients stored (mode 0), the asynchronous push / pull through the mapped master inside the backward
launch (mode 3), the sharded and the replicated exchange of the sync transports (modes 4 and 2)."""
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
lib.dtf_set_stamp_buffer_mlp2.argtypes = [C.c_void_p]
lib.dtf_stamp_now.argtypes = [C.c_void_p, C.c_void_p]
ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
variant = sys.argv[1] if len(sys.argv) > 1 else "update"
st = build_mlp_state(784, [100], 10, dev, lr=0.01)
st.initialize(seed=1)
engine = None
if variant == "update":
    os.environ["DTF_PIPELINED_STEP"] = "0"
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 100, [100])
    prog.run(6)
else:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=0, world_size=1)
    if variant == "async":
        from distributed_tensorflow_amd.parallel.async_peer import PeerMappedAsyncEngine
        os.environ["DTF_PEER_ASYNC_FUSED"] = "1"
        prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 100, [100], fused_update=False, gs_inc=0)
        engine = PeerMappedAsyncEngine(prog, st, 1, staleness=0, staleness_bound=-1)
        assert engine.fused
    else:
        from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
        os.environ["DTF_P2P_ALLREDUCE"] = "1"
        prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 100, [100], fused_update=False)
        coll = {"sharded": "p2p_sharded", "replicated": "p2p_replicated", "gradients": "all_reduce"}[variant]
        if variant == "gradients":
            os.environ["DTF_P2P_ALLREDUCE"] = "0"
        engine = SyncReplicasEngine(prog, st, collective=coll, force_collectives=True, capture_collectives=False)
        if variant != "gradients":
            assert engine.p2p is not None and engine.p2p_fused and engine.p2p_mode == variant, (engine.p2p_mode, engine.p2p_fused)
    engine.run(6, 2)
torch.cuda.synchronize()
print("variant", variant, "mode of the backward launch:", prog._args[0]["mlp2"].mode)
nwg = 256
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
t_ab = torch.zeros(4, dtype=torch.int64, device=dev)
lib.dtf_set_stamp_buffer_mlp2(stamps.data_ptr())
s = torch.cuda.current_stream().cuda_stream
args = prog._args[0]["mlp2"]


def show(tag, rel, sl):
    r = rel[sl]
    cols = " ".join(f"s{j}:{int(np.median(r[:, j])):5d}/{int(r[:, j].max()):5d}" for j in range(7) if (r[:, j] > 0).any())
    print(f"   {tag:6s} (n={r.shape[0]:3d}) median/max ns since pre-stamp: {cols}")


for which in ("fwd", "bwd"):
    for it in range(4):
        stamps.zero_()
        torch.cuda.synchronize()
        lib.dtf_stamp_now(t_ab.data_ptr(), s)
        (ops.mlp2_fwd if which == "fwd" else ops.mlp2_bwd)(args)
        lib.dtf_stamp_now(t_ab.data_ptr() + 8, s)
        torch.cuda.synchronize()
        sv = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
        t0, t1 = int(t_ab[0]), int(t_ab[1])
        rel = np.where(sv > 0, (sv - t0) * 10, 0)
        print(f"{which} iter {it}: stamp->stamp {(t1 - t0) * 10} ns; last workgroup stamp at {int(rel.max())} ns "
              f"(launch end + next kernel's start: {(t1 - t0) * 10 - int(rel.max())} ns)")
        if which == "fwd":
            show("tiles", rel, slice(0, 91))
        else:
            # stamps are indexed by role index: dW1 | dW2 | stat | tail
            show("dW1", rel, slice(0, 86))
            show("dW2", rel, slice(86, 88))
            show("stat", rel, slice(88, 101))
            show("tail", rel, slice(101, 114))

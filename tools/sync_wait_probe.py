#!/usr/bin/env python3
"""How a short timed run ends: wall time of a prepared 20-step run of the flagship program when the host waits for the
device with torch.cuda.synchronize() alone, or polls the stream first.  SYNC_PROBE_SPIN=1: hipSetDeviceFlags(spin) before
the runtime touches the device.

    python tools/sync_wait_probe.py"""
import ctypes
import os
import sys
import time

if os.environ.get("SYNC_PROBE_SPIN") == "1":
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))        # hipDeviceScheduleSpin
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", rc)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine

dev = torch.device("cuda", 0)
ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
st = build_mlp_state(784, [100], 10, dev)
st.initialize(seed=1234)
prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=1000), 100, [100], loss_mode=ops.LOSS_REFERENCE)
eng = SyncReplicasEngine(prog, st)
eng.run(200, 64)
torch.cuda.synchronize()
N = 20
stream = torch.cuda.current_stream()


def run_sync():
    t0 = time.perf_counter()
    eng.run(N, 64, defer_checks=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


def run_poll():
    t0 = time.perf_counter()
    eng.run(N, 64, defer_checks=True)
    while not stream.query():
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


ev = torch.cuda.Event()


def run_event_poll():
    t0 = time.perf_counter()
    eng.run(N, 64, defer_checks=True)
    ev.record(stream)
    while not ev.query():
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for name, fn in (("synchronize", run_sync), ("stream.query() spin, then synchronize", run_poll),
                 ("event.query() spin, then synchronize", run_event_poll), ("synchronize (again)", run_sync)):
    for _ in range(20):
        fn()
    ts = sorted(fn() for _ in range(300))
    print(f"{name:42s} median {ts[150]:7.1f} us  p10 {ts[30]:7.1f}  p90 {ts[270]:7.1f}   per step {ts[150] / N:5.2f} us")

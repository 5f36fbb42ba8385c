#!/usr/bin/env python3
"""Host microseconds of the pieces of one prepared ``run(20)`` of the flagship program (what happens between the
clock start of a short timed run and the graph launch)."""
import os
import sys
import time

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


def timed(fn, reps=300, sync=False):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
        if sync:
            torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2]


N = 20
print("launch_sizes      %.2f us" % timed(lambda: list(prog._launch_sizes(N, 64))))
print("data.prepare      %.2f us" % timed(lambda: prog.data.prepare(prog.steps_enqueued, N + 3, 100)))
print("pipe token        %.2f us" % timed(lambda: prog._pipe_token()) if prog.pipe else "")
print("prog.run(20)      %.2f us (host, until it returns)" % timed(lambda: prog.run(N, 64), sync=True))
print("engine.run(20)    %.2f us (host, until it returns)" % timed(lambda: eng.run(N, 64, defer_checks=True), sync=True))
g = prog._graph_for(N, last=True) if prog.pipe else prog._graph_for(N)
torch.cuda.synchronize()


def replay_only():
    g.replay()


print("graph.replay()    %.2f us (host)" % timed(replay_only, sync=True))
prog._pipe_live = None
prog.run(N, 64)
torch.cuda.synchronize()


def whole():
    t0 = time.perf_counter()
    eng.run(N, 64, defer_checks=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e6, (time.perf_counter() - t0) * 1e6


rs = sorted(whole() for _ in range(200))
print("engine.run(20) + synchronize: host %.1f us, total %.1f us (median of 200)" % (rs[100][0], sorted(r[1] for r in rs)[100]))
t = sorted(timed(lambda: torch.cuda.synchronize(), reps=1) for _ in range(200))[100]
print("synchronize on an idle stream %.2f us" % t)

# experiment: the first k steps of a run as direct kernel launches (the device starts ~4 us after the first call), the
# rest as ONE graph launched while they run (its ~12 us start-up latency hidden behind them)
for k in (0, 1, 2, 3, 4):
    def split_run():
        prog._pipe_open()
        prog.data.prepare(prog.steps_enqueued, N + 3, 100)
        for _ in range(k):
            prog._pipe_step()
        prog._graph_for(N - k, last=True).replay()
        prog.steps_enqueued += N - k
        prog._pipe_close(launched=True)
    for _ in range(3):
        split_run()
        torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        split_run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print("first %d steps direct + graph of %d: total %.1f us (median of 200), min %.1f" % (k, N - k, ts[100], ts[0]))

# experiment: the same graph launched through hipGraphLaunch directly (no torch bookkeeping around it), and after a
# hipGraphUpload
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipGraphLaunch.restype = ctypes.c_int
hip.hipGraphUpload.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipGraphUpload.restype = ctypes.c_int
stream = torch.cuda.current_stream().cuda_stream
for label, upload in (("torch replay", None), ("hipGraphLaunch", False), ("hipGraphUpload once + hipGraphLaunch", True)):
    keys = {}
    def one():
        prog._pipe_open()
        prog.data.prepare(prog.steps_enqueued, N + 3, 100)
        g = prog._graph_for(N, last=True)
        if upload is None:
            g.replay()
        else:
            ex = g.raw_cuda_graph_exec()
            if upload and ex not in keys:
                keys[ex] = hip.hipGraphUpload(ex, stream)
            rc = hip.hipGraphLaunch(ex, stream)
            assert rc == 0, rc
        prog.steps_enqueued += N
        prog._pipe_close(launched=True)
    for _ in range(8):
        one()
        torch.cuda.synchronize()
    ts, hs = [], []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
        hs.append((t1 - t0) * 1e6)
    ts.sort(), hs.sort()
    print("%-40s total %.1f us, host %.1f us (medians of 200)%s" % (label, ts[100], hs[100], "  upload rc %s" % set(keys.values()) if keys else ""))
st_check = prog.read_log(prog.steps_enqueued - 1, 1)[0]
print("last record after the experiment: local step %d global step %d loss %.4f" % (st_check.local_step, st_check.global_step, st_check.loss))

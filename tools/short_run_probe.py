#!/usr/bin/env python3
"""Where does the time of a SHORT timed run go?  bench.py at ``--steps 20`` reports more microseconds per step
than at ``--steps 2000``; this probe times prepared runs of several lengths the way bench.py does (synchronise,
clock, run, synchronise, clock) and separates the fixed cost per run from the cost per step, and the host time
until ``run`` returns from the wait for the device.

    python tools/short_run_probe.py [--lengths 5,10,20,40,80,160] [--repeats 15]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lengths", default="5,10,20,40,80,160")
    ap.add_argument("--repeats", type=int, default=15)
    ap.add_argument("--idle_ms", type=float, default=0.0, help="host sleep before each timed run (GPU clocks fall when idle)")
    ap.add_argument("--launch", default="graph", choices=["graph", "native"],
                    help="graph: replay prepared hipGraphs; native: one native call launches every kernel of the run")
    ap.add_argument("--first_launch", action="store_true",
                    help="capture the timed run's graph anew before every repeat: the timed replay is that graph's FIRST launch")
    a = ap.parse_args()
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
    lengths = [int(x) for x in a.lengths.split(",")]
    rows = []
    for n in lengths:
        tot, host = [], []
        for _ in range(a.repeats):
            if a.launch == "graph":
                if a.first_launch:
                    prog._graphs.clear()
                prog.prepare_graphs(5, 64)
                prog.prepare_graphs(n, 64, start_offset=5)
            eng.run(5, 64)
            torch.cuda.synchronize()
            if a.idle_ms:
                time.sleep(a.idle_ms * 1e-3)
            t0 = time.perf_counter()
            if a.launch == "graph":
                eng.run(n, 64)
            else:
                prog.run_native(n)
                eng.updates_applied += n
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            tot.append((t2 - t0) * 1e6)
            host.append((t1 - t0) * 1e6)
        tot.sort(), host.sort()
        rows.append({"steps": n, "total_us_median": tot[len(tot) // 2], "total_us_min": tot[0],
                     "host_until_run_returns_us_median": host[len(host) // 2]})
    # least squares total = fixed + per_step * n over the medians
    xs, ys = [r["steps"] for r in rows], [r["total_us_median"] for r in rows]
    mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    print(json.dumps({"rows": rows, "per_step_us": slope, "fixed_us": my - slope * mx, "idle_ms": a.idle_ms,
                      "launch": a.launch, "first_launch": a.first_launch}, indent=1))


if __name__ == "__main__":
    main()

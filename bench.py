#!/usr/bin/env python3
"""Headline benchmark: examples/sec of the reference MNIST-shape MLP
(784-100-10, batch 100 per worker, SGD lr 0.01, fp32; /root/reference/distributed.py:11-14)
trained synchronously with the parameter-server state sharded across the GPUs.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Every step does the full reference iteration: forward, (double-softmax) loss,
accuracy, backward, gradient aggregation over the workers, SGD update of the
parameter-server shards, parameter redistribution, global_step increment,
on-device selection of the next batch and the post-update training-accuracy
pass.  Data is a synthetic MNIST-shaped set resident in HBM; weights are
randomly initialised.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="mlp_sync",
                    choices=["mlp_sync", "mlp_async", "wide_mlp_sync", "wide_mlp_async", "resnet50_sync"])
    ap.add_argument("--batch_size", type=int, default=None, help="per-worker batch (default: the config's)")
    ap.add_argument("--hidden_units", type=int, default=None)
    ap.add_argument("--optimizer", default=None)
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--no_post_update_accuracy", action="store_true")
    ap.add_argument("--graph_steps", type=int, default=0)
    ap.add_argument("--collective", default="auto", choices=["auto", "reduce_scatter", "all_reduce", "bucketed"])
    ap.add_argument("--cpu", action="store_true", help="run on CPU with gloo (functional check only)")
    ap.add_argument("--force_collectives", action="store_true",
                    help="N=1 only: run the multi-rank code path (gradient store, RCCL collective on a one-rank "
                         "communicator, separate optimizer launch) to measure its overhead without peers")
    return ap.parse_args()


def main():
    args = parse_args()
    # Libraries (RCCL prints a version banner at communicator creation) must not write to the stream
    # that carries the result: everything except the final JSON line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # a long warm-up (library autotuning, first-touch compilation) must not look like a hang: one progress line
    # on stderr every 30 s
    import threading
    t_start, done = time.time(), threading.Event()

    def heartbeat():
        while not done.wait(30.0):
            sys.stderr.write(f"[bench] {args.config}: running, {time.time() - t_start:.0f} s\n")
            sys.stderr.flush()

    threading.Thread(target=heartbeat, daemon=True).start()
    try:
        from distributed_tensorflow_amd.benchmarks import run_benchmark
        result = run_benchmark(args)
    finally:
        done.set()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

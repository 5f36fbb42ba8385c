#!/usr/bin/env python3
"""Steps-to-target-loss of the reference MLP (sync, R ranks simulated by batch scaling is NOT done here:
one process, one device).  Prints one JSON line.  The reference loss is bounded below by 1.4612."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
from distributed_tensorflow_amd.utils.trace import steps_to_target

ap = argparse.ArgumentParser()
ap.add_argument("--target", type=float, default=1.60)
ap.add_argument("--max_steps", type=int, default=20000)
ap.add_argument("--lr", type=float, default=0.01)
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
a = ap.parse_args()
ds = mnist.synthetic_data_sets(seed=1)
st = build_mlp_state(784, [100], 10, a.device, lr=a.lr)
st.initialize(seed=1)
prog = MLPProgram(st, DeviceDataset(ds.train, a.device, shuffle_seed=1), a.batch, [100], log_capacity=2048)
losses, done = [], None
while len(losses) < a.max_steps and done is None:
    n = min(1024, a.max_steps - len(losses))
    first = len(losses)
    prog.run(n)
    losses += [r.loss for r in prog.read_log(first, n)]
    done = steps_to_target(losses, a.target, window=10)
print(json.dumps({"metric": "steps to target loss", "target": a.target, "steps": done, "lr": a.lr, "batch": a.batch,
                  "loss": "reference (softmax applied twice)", "final_loss": losses[-1], "device": a.device}))

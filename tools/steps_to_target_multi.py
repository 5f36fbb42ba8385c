#!/usr/bin/env python3
"""Steps-to-target-loss of the reference MLP with several workers, sync (SyncReplicas, R = N) or async
(Hogwild push/pull on one logical ps shard).  Launch one process per worker:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/steps_to_target_multi.py --mode sync

On a box with one GPU: DTF_DIST_BACKEND=gloo DTF_SHARE_GPU=1 DTF_P2P_ALLREDUCE=1 DTF_PEER_ASYNC=1 (the ranks then
share the GPU; the arithmetic is the same, the timing means nothing).  Rank 0 prints one JSON line: the value
of global_step (aggregated updates in sync mode, all workers' updates in async mode,
/root/reference/distributed.py:155) at which the trailing-10 mean of the training loss of EVERY worker was
below the target, and the examples consumed up to there."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
from distributed_tensorflow_amd.parallel import comm
from distributed_tensorflow_amd.parallel.async_peer import make_async_engine
from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["sync", "async"], default="sync")
ap.add_argument("--target", type=float, default=1.60)
ap.add_argument("--max_steps", type=int, default=20000, help="local steps per worker")
ap.add_argument("--lr", type=float, default=0.01)
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--chunk", type=int, default=16, help="local steps between checks")
ap.add_argument("--staleness", type=int, default=4, help="async: pull every (staleness+1)-th step")
a = ap.parse_args()

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = 0 if os.environ.get("DTF_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0))
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
comm.init_from_env(dev)
ds = mnist.synthetic_data_sets(seed=1)
st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=a.lr)
st.initialize(seed=1)                                        # same seed everywhere == chief init + broadcast
data = DeviceDataset(ds.train, dev, shuffle_seed=100 + rank)  # every worker draws its own batches (distributed.py:137)
if a.mode == "sync":
    prog = MLPProgram(st, data, a.batch, [100], fused_update=False, gs_inc=1, log_capacity=2048)
    eng = SyncReplicasEngine(prog, st)
else:
    prog = MLPProgram(st, data, a.batch, [100], fused_update=False, gs_inc=0, log_capacity=2048)
    eng = make_async_engine(prog, st, 1, a.staleness)
cpu = dist.get_backend() != "nccl"
losses, hit_gs, local_steps = [], None, 0
while local_steps < a.max_steps:
    eng.run(a.chunk, a.chunk)
    recs = prog.read_log(local_steps, a.chunk)
    local_steps += a.chunk
    losses += [r.loss for r in recs]
    mean10 = sum(losses[-10:]) / 10.0
    flag = torch.tensor([1 if mean10 > a.target else 0], dtype=torch.int32, device="cpu" if cpu else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)              # stop together, when every worker is below the target
    if int(flag.item()) == 0:
        if a.mode == "async":
            eng.sync_params()
        hit_gs = st.global_step()
        break
if hasattr(eng, "finish"):
    eng.finish()
if rank == 0:
    print(json.dumps({"metric": "steps to target loss", "mode": a.mode, "workers": world, "target": a.target,
                      "global_step": hit_gs, "local_steps_per_worker": local_steps, "examples": local_steps * a.batch * world,
                      "lr": a.lr, "batch_per_worker": a.batch, "loss": "reference (softmax applied twice)",
                      "final_loss_rank0": losses[-1], "engine": type(eng).__name__,
                      "transport": getattr(eng, "transport", None), "staleness": a.staleness if a.mode == "async" else None}),
          flush=True)
dist.barrier()
dist.destroy_process_group()

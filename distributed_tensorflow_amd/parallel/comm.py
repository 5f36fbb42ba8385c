"""Process-group bring-up and the few collectives the engines use.

One process per GPU; ``backend="nccl"`` is RCCL on ROCm and runs over xGMI
inside a node.  The CPU configuration uses gloo with the same call sites.
This replaces the in-process gRPC server of the reference
(``tf.train.Server``, /root/reference/distributed.py:54): rendezvous is a TCP
store at the chief's address, and tensors never travel through host memory on
the GPU path.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional

import torch
import torch.distributed as dist


def pick_backend(device: torch.device, requested: str = "auto") -> str:
    if requested not in ("auto", "", None):
        return requested
    if os.environ.get("DTF_DIST_BACKEND"):          # rehearsals: several ranks sharing one GPU cannot use RCCL
        return os.environ["DTF_DIST_BACKEND"]
    return "nccl" if device.type == "cuda" else "gloo"


def init_process_group(rank: int, world_size: int, master_addr: str, master_port: int, device: torch.device,
                       backend: str = "auto", timeout_s: float = 600.0) -> None:
    """Join the job-wide group.  Safe to call when a launcher (torchrun) already exported the env."""
    if dist.is_initialized():
        return
    backend = pick_backend(device, backend)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(device)
        kwargs["device_id"] = device
    dist.init_process_group(backend=backend, init_method=f"tcp://{master_addr}:{master_port}", rank=rank,
                            world_size=world_size, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)


def init_from_env(device: torch.device, backend: str = "auto", timeout_s: float = 600.0) -> None:
    """Join using RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run contract)."""
    init_process_group(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ.get("MASTER_ADDR", "127.0.0.1"),
                       int(os.environ.get("MASTER_PORT", "29500")), device, backend, timeout_s)


def backend_of(group=None) -> str:
    return dist.get_backend(group)


class Collectives:
    """reduce-scatter / all-gather / all-reduce on flat fp32 buffers, with gloo stand-ins
    where gloo lacks the primitive (results are identical, only the traffic differs)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = backend_of(group)
        self._native_rs = self.backend == "nccl"

    def _host(self, t: torch.Tensor) -> torch.Tensor:
        """gloo rehearsals with GPU-resident state (several ranks sharing one GPU): the collective runs on
        a host copy, so that gloo's own device-stream handling is not part of what is being rehearsed."""
        return t.cpu() if (t.is_cuda and self.backend != "nccl") else t

    def reduce_scatter_sum(self, out_shard: torch.Tensor, full: torch.Tensor) -> None:
        if self._native_rs:
            dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=self.group)
            return
        h = self._host(full)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        n = out_shard.numel()
        out_shard.copy_(h[self.rank * n:(self.rank + 1) * n])
        if h is not full:
            full.copy_(h)

    def all_gather(self, full: torch.Tensor, shard: torch.Tensor) -> None:
        if self.backend == "nccl":
            dist.all_gather_into_tensor(full, shard, group=self.group)
            return
        n = shard.numel()
        h = self._host(full)
        parts = [h[i * n:(i + 1) * n] for i in range(self.world)]
        dist.all_gather(parts, self._host(shard).clone(), group=self.group)
        if h is not full:
            full.copy_(h)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        h = self._host(t)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        if h is not t:
            t.copy_(h)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        h = self._host(t)
        dist.broadcast(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)

    def barrier(self) -> None:
        if self.backend == "nccl":
            dist.barrier(group=self.group, device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier(group=self.group)


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()

"""Peer-mapped exchange regions of the N-worker one-launch pipelined step (``csrc/kernels/mlp_pipe_multi.hip``).

Sync mode of the reference (``SyncReplicasOptimizer`` with ``replicas_to_aggregate == num_workers``,
/root/reference/distributed.py:91-102) moves every worker's whole gradient to the ps and the whole parameter set
back, 318 KB each way per step.  The pipelined step moves, per worker and step, ``dz1`` of its rows (40 KB) and its
share of the output layer's gradient (28 KB) straight into *boxes* in every rank's HBM - one hop over xGMI, stores
only - and every rank applies the same sums in the same order to its own replica.

A region (``DTF_PIPEM_REGION_BYTES``, ~4 MB) holds four generations of boxes for up to eight source ranks; every
word of a box carries its own arrival mark (lowest mantissa bit = phase of the writing step), so there are no flags
and a writer never waits for an acknowledgement.  It is allocated uncached, exported by IPC handle and mapped by
every peer; regions come from (and go back to) the pool of exported regions in ``p2p_allreduce.hip``.

One rank (``world == 1``) needs no process group: the step then exchanges with itself through its own region, which
is how the multi-worker launch is timed on a machine with one GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops import _lib
from . import async_peer as _ap

HANDLE_BYTES = 64


def default_timeout_s() -> float:
    return float(os.environ.get("DTF_P2P_TIMEOUT_S", "30"))


class PipeExchangeUnavailable(RuntimeError):
    """Raised on EVERY rank together when the regions cannot be allocated, exported or mapped."""


class PipeExchange:
    def __init__(self, device: torch.device, rank: int = 0, world: int = 1, group=None, timeout_s: Optional[float] = None):
        if device.type != "cuda":
            raise ValueError("the pipelined exchange runs on GPUs")
        if not 1 <= world <= 8:
            raise ValueError("the pipelined exchange covers 1..8 ranks of one node")
        self.device, self.rank, self.world, self.group = device, rank, world, group
        self.timeout_s = default_timeout_s() if timeout_s is None else float(timeout_s)
        lib = _ap._lib_k()
        self.bytes = int(_lib.kernels().dtf_pipem_region_bytes())
        self.err = torch.zeros(4, dtype=torch.int32, device=device)
        self._local = None
        self._mapped: List[int] = []
        handle = C.create_string_buffer(HANDLE_BYTES)
        with torch.cuda.device(device):
            self._local = lib.dtf_ipc_alloc(self.bytes, handle)
        good = bool(self._local)
        self.regions = [0] * world
        if world == 1:
            if not good:
                raise PipeExchangeUnavailable("the exchange region could not be allocated")
            self.regions[0] = self._local
            return
        # every rank walks through the same collectives, whatever failed locally
        handles: List[Optional[bytes]] = [None] * world
        dist.all_gather_object(handles, bytes(handle.raw) if good else b"", group=group)
        if good:
            for r in range(world):
                if r == rank:
                    self.regions[r] = self._local
                elif len(handles[r]) == HANDLE_BYTES:
                    with torch.cuda.device(device):
                        p = lib.dtf_ipc_open(handles[r])
                    if p:
                        self._mapped.append(p)
                        self.regions[r] = p
                    else:
                        good = False
                else:
                    good = False
        if not self._all_ok(good):
            self.close()
            raise PipeExchangeUnavailable("a rank could not allocate, export or map the exchange regions")

    def _all_ok(self, flag: bool) -> bool:
        cpu = dist.get_backend(self.group) != "nccl"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if cpu else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    def barrier(self) -> None:
        """Device idle here, then every rank: what the prime launch wrote is in place everywhere before a step runs."""
        torch.cuda.synchronize(self.device)
        if self.world > 1 and dist.is_initialized():
            if dist.get_backend(self.group) == "nccl":
                dist.barrier(group=self.group, device_ids=[self.device.index])
            else:
                dist.barrier(group=self.group)

    def timed_out(self) -> bool:
        """True once a wait for a peer's boxes ran into the time limit (synchronises with the device)."""
        return int(self.err[0].item()) != 0

    def close(self) -> None:
        lib = _ap._lib_k()
        for p in self._mapped:
            lib.dtf_ipc_close(p)
        self._mapped = []
        if self._local:
            lib.dtf_ipc_free(self._local)
            self._local = None
        self.regions = []

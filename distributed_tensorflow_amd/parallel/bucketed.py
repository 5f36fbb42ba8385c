"""Bucketed gradient aggregation overlapped with the backward pass (large models).

The reference's workers push each variable's gradient to its parameter server as soon as the
backward pass has produced it, while the rest of the backward pass is still running (the
per-variable accumulators of ``tf.train.SyncReplicasOptimizer``,
/root/reference/distributed.py:97-102).  The counterpart here, for models whose gradients are
hundreds of megabytes: the flat gradient buffer is cut into *buckets* (contiguous ranges that
cover whole variables, one per layer for the wide MLP, ~32 MB for a PyTorch module), and as soon
as the kernels that write a bucket have been enqueued the bucket goes through

    reduce-scatter (sum over ranks)  ->  fused optimizer on the owned 1/N of the bucket
                                     ->  all-gather of the updated bucket

on a second HIP stream, concurrently with the backward kernels of the layers below.  Ownership is
per bucket: rank r owns the r-th N-th of EVERY bucket (with its Adam moments), so each bucket is an
ordinary equal-chunk RCCL collective and the optimizer work stays balanced.  Only the buckets that
complete last (the first layer and the small output layer) are exposed.

xGMI sizing: an MI355X has 7 point-to-point links of ~153 GB/s.  A 4096x4096 fp32 layer is a
67 MB bucket; its reduce-scatter and all-gather each move 7/8 of that per GPU, well past the size
where RCCL's bandwidth plateaus, and the backward + weight-gradient products of the next layer
(~0.4 ms at batch 4096) cover them.

On ONE rank (no communicator) the same mechanism overlaps the optimizer with the backward pass: the
memory-bound update of layer l runs on the second stream beside the MFMA-bound products of the layers below.

The same object runs on CPU/gloo (no streams: the bucket is processed inline), which is how the
multi-rank arithmetic is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .. import ops
from .comm import Collectives
from .sharded_state import ALIGN, ShardedState

DEFAULT_BUCKET_BYTES = 32 << 20


def bucket_ranges(state: ShardedState, groups: Sequence[Sequence[str]]) -> List[Tuple[int, int]]:
    """Flat ranges [lo, hi) of the variable groups.  Every group must be consecutive variables of the layout,
    the groups must tile the whole buffer, and every range must split evenly over the ranks."""
    lay = state.layout
    order = [v.name for v in lay.variables]
    ends = {}
    for i, name in enumerate(order):
        ends[name] = lay.offsets[order[i + 1]] if i + 1 < len(order) else lay.total
    seen = set()
    out = []
    for g in groups:
        if not g:
            raise ValueError("empty gradient bucket")
        idx = [order.index(n) for n in g]
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError(f"bucket {list(g)} is not a run of consecutive variables")
        if seen & set(g):
            raise ValueError(f"variable in two buckets: {sorted(seen & set(g))}")
        seen |= set(g)
        lo, hi = lay.offsets[g[0]], ends[g[-1]]
        if (hi - lo) % state.world_size:
            raise ValueError(f"bucket {list(g)}: {hi - lo} elements do not split over {state.world_size} ranks")
        out.append((lo, hi))
    if seen != set(order):
        raise ValueError(f"variables without a bucket: {sorted(set(order) - seen)}")
    return out


def group_by_bytes(state: ShardedState, bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> List[List[str]]:
    """Consecutive variables grouped until a bucket holds at least ``bucket_bytes`` (layout order)."""
    groups, cur, size = [], [], 0
    for v in state.layout.variables:
        cur.append(v.name)
        size += (v.numel + ALIGN - 1) // ALIGN * ALIGN * 4
        if size >= bucket_bytes:
            groups.append(cur)
            cur, size = [], 0
    if cur:
        if groups and size < bucket_bytes // 4:
            groups[-1].extend(cur)          # no tiny trailing bucket
        else:
            groups.append(cur)
    return groups


class BucketedUpdate:
    """Per-bucket reduce-scatter -> optimizer -> all-gather on a side stream.

    ``ready(i)`` is called by the step program right after the kernels that complete bucket ``i`` were
    enqueued on the compute stream; ``finish()`` once per step after the last bucket: the compute stream
    then waits for the side stream, so the next step reads updated parameters and may overwrite gradients.
    """

    def __init__(self, state: ShardedState, groups: Sequence[Sequence[str]], coll: Optional[Collectives],
                 grad_scale: float, native=None):
        self.state, self.coll, self.native, self.scale = state, coll, native, grad_scale
        self.groups = [list(g) for g in groups]
        self.ranges = bucket_ranges(state, self.groups)
        W, r = state.world_size, state.rank
        self.own: List[Tuple[int, int]] = []        # owned slice of each bucket (flat coordinates)
        self.local: List[Tuple[int, int]] = []      # where that slice's optimizer state / reduced gradient lives
        off = 0
        for lo, hi in self.ranges:
            n = (hi - lo) // W
            self.own.append((lo + r * n, lo + (r + 1) * n))
            self.local.append((off, off + n))
            off += n
        assert off == state.layout.shard_size
        self.reduced = torch.empty(off, dtype=torch.float32, device=state.device) if coll is not None else None
        self.cuda = state.device.type == "cuda"
        if self.cuda:
            self.side = torch.cuda.Stream(device=state.device)
            self._ev_ready = [torch.cuda.Event() for _ in self.ranges]
            self._ev_done = torch.cuda.Event()
        self._signalled = [False] * len(self.ranges)
        self.steps = 0

    # ---- collectives on the current stream
    def _reduce_scatter(self, out: torch.Tensor, full: torch.Tensor) -> None:
        if self.native is not None:
            self.native.reduce_scatter_sum(out, full)
        else:
            self.coll.reduce_scatter_sum(out, full)

    def _all_gather(self, full: torch.Tensor, shard: torch.Tensor) -> None:
        if self.coll is None:
            return
        if self.native is not None:
            self.native.all_gather(full, shard)
        else:
            self.coll.all_gather(full, shard)

    def _process(self, i: int) -> None:
        st = self.state
        lo, hi = self.ranges[i]
        olo, ohi = self.own[i]
        llo, lhi = self.local[i]
        if self.coll is None:
            red = st.grads[lo:hi]                   # one rank without a communicator: the bucket is its own sum
        else:
            red = self.reduced[llo:lhi]
            self._reduce_scatter(red, st.grads[lo:hi])
        p = st.params[olo:ohi]
        if st.optimizer == "sgd":
            ops.sgd_apply(p, red, st.lr, self.scale)
        else:
            # every bucket of a step uses the same step number; finish() advances it
            ops.adam_apply(p, red, st.m[llo:lhi], st.v[llo:lhi], st.lr, st.beta1, st.beta2, st.eps, self.scale,
                           step_tensor=st.state[2:3], step_offset=1)
        self._all_gather(st.params[lo:hi], p)

    def ready(self, i: int) -> None:
        if self._signalled[i]:
            raise RuntimeError(f"gradient bucket {i} signalled twice in one step")
        self._signalled[i] = True
        if not self.cuda:
            self._process(i)
            return
        ev = self._ev_ready[i]
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self._process(i)

    def run_behind(self, fn) -> None:
        """Enqueue ``fn`` on the side stream, behind the buckets processed so far and behind everything the current
        stream has enqueued so far (a consumer of a bucket's all-gather that must not start before a kernel of the
        compute stream has read what it overwrites).  ``finish`` covers it like a bucket."""
        if not self.cuda:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()

    def finish(self) -> None:
        if not all(self._signalled):
            missing = [i for i, s in enumerate(self._signalled) if not s]
            raise RuntimeError(f"gradient buckets never signalled this step: {missing}")
        self._signalled = [False] * len(self.ranges)
        if self.cuda:
            self._ev_done.record(self.side)
            torch.cuda.current_stream().wait_event(self._ev_done)
        if self.state.optimizer != "sgd":
            self.state.state[2:3].add_(1)
        self.steps += 1

"""Parameter-server state resident in device memory.

The reference places its variables on ``/job:ps`` processes
(``tf.train.replica_device_setter``, /root/reference/distributed.py:60-73) and
every worker reads them over the network each step.  Here all trainable
variables live in ONE flat fp32 buffer per rank (a full replica that the compute
kernels read), and "owning" a variable — the ps role — means owning a contiguous
``1/world`` slice of that buffer together with the optimizer state of that
slice.  With 288 GB of HBM per MI355X the slices can be very large, so a sync
step is exactly one reduce-scatter, one fused shard update and one all-gather.

``global_step`` is an int32 scalar that starts at 1 (distributed.py:65) and is
kept in ``state[0]`` on the device; ``state[1]`` is the worker's local step.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops

ALIGN = 64  # elements; every variable starts on a 256-byte boundary of the flat buffer
LOCAL_STEP_KEY = "dtf/local_step"   # checkpoint entry beyond the reference's variables: the saving worker's local step


@dataclass
class VariableSpec:
    name: str
    shape: Tuple[int, ...]
    init: str = "zeros"            # "zeros" | "truncated_normal" | "he_normal"
    stddev: float = 0.0

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def mlp_variables(input_dim: int, hidden: Sequence[int], classes: int) -> List[VariableSpec]:
    """Variables of the reference network in creation order (distributed.py:67-73).

    One hidden layer gives exactly ``hid_w, hid_b, sm_w, sm_b`` with
    stddev 1/sqrt(fan_in) (1/28 for 784 inputs).  Deeper variants name the extra
    hidden layers ``hid{i}_w`` / ``hid{i}_b``.
    """
    specs: List[VariableSpec] = []
    fan_in = input_dim
    for i, h in enumerate(hidden):
        prefix = "hid" if i == 0 else f"hid{i + 1}"
        specs.append(VariableSpec(f"{prefix}_w", (fan_in, h), "truncated_normal", 1.0 / math.sqrt(fan_in)))
        specs.append(VariableSpec(f"{prefix}_b", (h,), "zeros"))
        fan_in = h
    specs.append(VariableSpec("sm_w", (fan_in, classes), "truncated_normal", 1.0 / math.sqrt(fan_in)))
    specs.append(VariableSpec("sm_b", (classes,), "zeros"))
    return specs


@dataclass
class FlatLayout:
    """Offsets of the variables inside the flat buffer and the shard boundaries."""
    variables: List[VariableSpec]
    world_size: int = 1
    offsets: Dict[str, int] = field(default_factory=dict)
    total: int = 0          # padded length, a multiple of world_size * ALIGN
    used: int = 0           # elements that belong to some variable

    def __post_init__(self):
        off = 0
        used = 0
        for v in self.variables:
            if v.name in self.offsets:
                raise ValueError(f"duplicate variable {v.name}")
            self.offsets[v.name] = off
            used += v.numel
            off += (v.numel + ALIGN - 1) // ALIGN * ALIGN
        quantum = self.world_size * ALIGN
        self.total = max(quantum, (off + quantum - 1) // quantum * quantum)
        self.used = used

    @property
    def shard_size(self) -> int:
        return self.total // self.world_size

    def shard_range(self, rank: int) -> Tuple[int, int]:
        s = self.shard_size
        return rank * s, (rank + 1) * s

    def owner_of(self, name: str) -> List[int]:
        """Ranks whose shard overlaps the variable (the hbm-mode counterpart of ps placement)."""
        lo = self.offsets[name]
        hi = lo + next(v.numel for v in self.variables if v.name == name)
        s = self.shard_size
        return list(range(lo // s, (hi - 1) // s + 1))


class ShardedState:
    """Flat parameters + gradients (full replicas) and the optimizer state of the local shard."""

    def __init__(self, variables: Sequence[VariableSpec], device, world_size: int = 1, rank: int = 0,
                 optimizer: str = "sgd", lr: float = 0.01, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self.layout = FlatLayout(list(variables), world_size)
        self.device = torch.device(device)
        self.world_size, self.rank = world_size, rank
        self.optimizer, self.lr, self.beta1, self.beta2, self.eps = optimizer, lr, beta1, beta2, eps
        if optimizer not in ("sgd", "adam"):
            raise ValueError(f"unknown optimizer {optimizer!r}")
        n = self.layout.total
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        # state[0] = global_step (starts at 1), state[1] = local_step, state[2] = updates applied to the local shard
        self.state = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.state[0] = 1
        lo, hi = self.layout.shard_range(rank)
        self.shard = (lo, hi)
        if optimizer == "adam":
            self.m = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
        else:
            self.m = self.v = None
        self.initialized = False
        self.values_version = 0       # bumped whenever the host rewrites the parameters (initialise, restore, broadcast)

    # ---- views
    def _view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        spec = next(v for v in self.layout.variables if v.name == name)
        off = self.layout.offsets[name]
        return flat[off:off + spec.numel].view(spec.shape)

    def var(self, name: str) -> torch.Tensor:
        return self._view(self.params, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.grads, name)

    def names(self) -> List[str]:
        return [v.name for v in self.layout.variables]

    @property
    def param_shard(self) -> torch.Tensor:
        return self.params[self.shard[0]:self.shard[1]]

    @property
    def grad_shard(self) -> torch.Tensor:
        return self.grads[self.shard[0]:self.shard[1]]

    # ---- init (chief runs this once: global_variables_initializer, distributed.py:108)
    def initialize(self, seed: int) -> None:
        self.params.zero_()
        for i, v in enumerate(self.layout.variables):
            if v.init == "truncated_normal":
                ops.truncated_normal_(self.var(v.name), v.stddev, seed, stream_id=i + 1)
            elif v.init == "he_normal":
                ops.truncated_normal_(self.var(v.name), v.stddev, seed, stream_id=i + 1)
            elif v.init != "zeros":
                raise ValueError(f"unknown initializer {v.init}")
        self.state.zero_()
        self.state[0] = 1
        if self.m is not None:
            self.m.zero_()
            self.v.zero_()
        self.initialized = True
        self.values_version += 1

    def load_values(self, values: Dict[str, torch.Tensor], global_step: Optional[int] = None) -> None:
        """Inject explicit values (parity tests, checkpoint restore)."""
        names = set(self.names())
        for name, val in values.items():
            if name == "global_step":
                global_step = int(val)
                continue
            if name == LOCAL_STEP_KEY:
                self.state[1] = int(val)
                continue
            if name not in names:
                continue                    # optimizer slots (load_moments) and whatever else the bundle holds
            dst = self.var(name)
            val = torch.as_tensor(val, dtype=torch.float32)
            if tuple(val.shape) != tuple(dst.shape):
                raise ValueError(f"{name}: shape {tuple(val.shape)} != {tuple(dst.shape)}")
            ops.h2d_verified(dst, val, name)
        if global_step is not None:
            self.state[0] = int(global_step)
        self.initialized = True
        self.values_version += 1

    def named_tensors(self, moments: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Everything the reference's default Saver would write: global_step + trainable variables (distributed.py:
        65-73, :110-111), plus this worker's local step (``dtf/local_step``: where its data pipeline stands, so that a
        restored run continues the same trajectory).  With Adam and ``moments`` = the full flat (m, v) buffers gathered
        from their owners, also the optimizer's slot variables under the names a TF ``AdamOptimizer`` gives them:
        ``<var>/Adam``, ``<var>/Adam_1``, ``beta1_power``, ``beta2_power``."""
        out = {"global_step": self.state[0:1].cpu().reshape(())}
        for v in self.layout.variables:
            out[v.name] = self.var(v.name).detach().cpu().clone()
        out[LOCAL_STEP_KEY] = self.state[1:2].cpu().reshape(())
        if self.optimizer == "adam" and moments is not None:
            full_m, full_v = moments
            if full_m.numel() != self.layout.total or full_v.numel() != self.layout.total:
                raise ValueError("moments must be the full flat buffers (gather them from their owners first)")
            for v in self.layout.variables:
                out[f"{v.name}/Adam"] = self._view(full_m, v.name).detach().cpu().clone()
                out[f"{v.name}/Adam_1"] = self._view(full_v, v.name).detach().cpu().clone()
            t = int(self.state[2].item())
            # TF keeps beta^(t+1) after t updates (the accumulators start at beta and are multiplied once per update)
            out["beta1_power"] = torch.tensor(self.beta1 ** (t + 1), dtype=torch.float32)
            out["beta2_power"] = torch.tensor(self.beta2 ** (t + 1), dtype=torch.float32)
        return out

    def gather_moments(self, coll=None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        """Full flat Adam moments assembled from the ranks' slices (X10, the checkpoint gather).  Collective when the
        state is sharded over several ranks: every rank calls it with the job's ``Collectives``."""
        if self.optimizer != "adam":
            return None
        if self.world_size == 1:
            return self.m, self.v
        if coll is None:
            raise ValueError("gathering the moments of a sharded state needs the job's collectives")
        full_m = torch.empty(self.layout.total, dtype=torch.float32, device=self.device)
        full_v = torch.empty_like(full_m)
        coll.all_gather(full_m, self.m)
        coll.all_gather(full_v, self.v)
        return full_m, full_v

    def load_moments(self, values: Dict[str, torch.Tensor]) -> bool:
        """Restore Adam's slot variables from checkpoint tensors (every rank keeps the slice it owns).  Returns False
        (state untouched: moments start at zero) when the checkpoint holds none."""
        got = self.moments_from_values(values)
        if got is None:
            return False
        full_m, full_v, t = got
        lo, hi = self.shard
        ops.h2d_verified(self.m, full_m[lo:hi], "Adam m")
        ops.h2d_verified(self.v, full_v[lo:hi], "Adam v")
        self.state[2] = t
        return True

    def moments_from_values(self, values: Dict[str, torch.Tensor]):
        """(full flat m, full flat v, updates applied) on the host from checkpoint tensors, or None without Adam slots."""
        if self.optimizer != "adam" or "beta1_power" not in values:
            return None
        import math
        full_m = torch.zeros(self.layout.total, dtype=torch.float32)
        full_v = torch.zeros(self.layout.total, dtype=torch.float32)
        for v in self.layout.variables:
            for key, full in ((f"{v.name}/Adam", full_m), (f"{v.name}/Adam_1", full_v)):
                if key not in values:
                    raise ValueError(f"checkpoint has beta1_power but no {key}")
                self._view(full, v.name).copy_(torch.as_tensor(values[key], dtype=torch.float32))
        t = int(round(math.log(float(values["beta1_power"])) / math.log(self.beta1))) - 1
        return full_m, full_v, max(0, t)

    # ---- optimizer on the owned slice
    def apply_shard(self, grad_shard: Optional[torch.Tensor] = None, grad_scale: float = 1.0) -> None:
        g = self.grad_shard if grad_shard is None else grad_shard
        if self.optimizer == "sgd":
            ops.sgd_apply(self.param_shard, g, self.lr, grad_scale)
        else:
            # step number = updates applied so far + 1, read on the device (graph replay safe)
            ops.adam_apply(self.param_shard, g, self.m, self.v, self.lr, self.beta1, self.beta2, self.eps, grad_scale,
                           step_tensor=self.state[2:3], step_offset=1)
            self.state[2:3].add_(1)

    def apply_ranges(self, ranges, grad_scale: float = 1.0) -> None:
        """One worker: the optimizer over the given flat ranges only (the rest of the buffer was updated elsewhere,
        e.g. inside the weight-gradient epilogues); advances the update counter once."""
        if self.world_size != 1:
            raise ValueError("apply_ranges is the one-worker path")
        ops.optimizer_ranges(self.params, self.grads, list(ranges), self.lr, self.m, self.v, self.beta1, self.beta2, self.eps,
                             grad_scale, step_tensor=self.state[2:3], step_offset=1)
        if self.optimizer != "sgd":
            self.state[2:3].add_(1)

    def apply_full(self, grad_scale: float = 1.0) -> None:
        """Replicated update of the whole buffer (latency path: one all-reduce instead of RS + AG)."""
        if self.optimizer != "sgd":
            raise ValueError("apply_full is only defined for stateless SGD")
        ops.sgd_apply(self.params, self.grads, self.lr, grad_scale)

    def global_step(self) -> int:
        return int(self.state[0].item())

    def local_step(self) -> int:
        return int(self.state[1].item())

    def nbytes(self) -> int:
        return self.layout.used * 4

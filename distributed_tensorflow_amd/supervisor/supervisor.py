"""Session lifecycle: chief init-or-restore, wait-for-ready, periodic checkpoints, step rate
(``tf.train.Supervisor`` as used at /root/reference/distributed.py:108-111, :121-126).

* The chief restores the newest checkpoint of ``logdir`` if there is one, otherwise runs the
  init function (``global_variables_initializer``), then publishes readiness.
* Every other worker polls for readiness once per ``recovery_wait_secs`` (1 s in the reference)
  and then receives the chief's values.
* The chief saves ``model.ckpt-<global_step>`` every ``save_model_secs`` (600 s, the Supervisor
  default) and once more at ``stop()``; it also reports ``global_step/sec`` every 120 s.
* With Adam the optimizer's slot variables are saved and restored too (a default TF Saver covers
  them: ``<var>/Adam``, ``<var>/Adam_1``, ``beta1_power``, ``beta2_power``).  They live with the
  owners of the parameter slices: in sync mode every worker joins the gather (the chief's decision
  to save travels in a one-word broadcast per call), in async mode the chief reads the owners'
  mapped regions by itself.

Checkpoints are taken by the training loop between hipGraph launches (``maybe_checkpoint``),
not by a timer thread racing the update kernels: the snapshot is always a consistent parameter
version.
"""
from __future__ import annotations

import json
import os
import tempfile
import time
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from .. import ops

from ..checkpoint.saver import Saver, latest_checkpoint


class Supervisor:
    def __init__(self, is_chief: bool, logdir: Optional[str], init_fn: Callable[[], None], state,
                 recovery_wait_secs: float = 1.0, save_model_secs: float = 600.0, save_summaries_secs: float = 120.0,
                 store=None, group_size: int = 1, saver: Optional[Saver] = None, log: Callable[[str], None] = print,
                 examples_per_global_step: Optional[int] = None):
        self.is_chief = is_chief
        self.logdir = logdir if logdir is not None else tempfile.mkdtemp()   # distributed.py:109
        self.init_fn = init_fn
        self.state = state
        self.recovery_wait_secs = recovery_wait_secs
        self.save_model_secs = save_model_secs
        self.save_summaries_secs = save_summaries_secs
        # batch x R in sync mode, batch in async mode: turns global_step/sec into the examples/sec of the whole job
        self.examples_per_global_step = examples_per_global_step
        self.store = store
        self.group_size = group_size
        self.saver = saver or Saver()
        self.log = log
        self.restored_from: Optional[str] = None
        self._last_save = time.time()
        self._last_rate_time = time.time()
        self._last_rate_step: Optional[int] = None
        self.waited_polls = 0
        self._events = None
        self._tb = None                  # TensorBoard event file (events.out.tfevents.*), chief only
        self.engine = None               # set by the training script: async engines hand out the owner-resident moments
        self.restored_values: Optional[Dict] = None

    # ------------------------------------------------------------------ who takes part in a save
    def _collective_save(self) -> bool:
        """Sync mode, Adam, several ranks: the moments are sharded, every rank contributes its slice."""
        st = self.state
        return (getattr(st, "optimizer", "sgd") == "adam" and self.group_size > 1 and dist.is_initialized()
                and not hasattr(self.engine, "gather_moments"))

    def _moments(self):
        st = self.state
        if getattr(st, "optimizer", "sgd") != "adam":
            return None
        if self.engine is not None and hasattr(self.engine, "gather_moments"):
            return self.engine.gather_moments()              # async: one-sided, chief only
        if self.engine is not None and hasattr(self.engine, "moments_to_state"):
            self.engine.moments_to_state()                   # sync, owner-resident moments: back into the slices (collective)
        if self.group_size > 1 and dist.is_initialized():
            from ..parallel.comm import Collectives
            return st.gather_moments(Collectives())
        return st.gather_moments()

    # ------------------------------------------------------------------ session bring-up
    def prepare_or_wait_for_session(self) -> None:
        st = self.state
        if self.is_chief:
            ckpt = latest_checkpoint(self.logdir)
            if ckpt is not None:
                values = Saver.restore(ckpt)
                st.load_values({k: torch.from_numpy(v) for k, v in values.items() if k != "global_step"},
                               global_step=int(values["global_step"]))
                self.restored_from = ckpt
                self.restored_values = values
            else:
                self.init_fn()
            if self.store is not None:
                self.store.set("dtf/ready", "1")
        else:
            if self.store is None:
                raise RuntimeError("a non-chief worker needs a rendezvous store to wait for the chief")
            while not self.store.check(["dtf/ready"]):
                self.waited_polls += 1
                time.sleep(self.recovery_wait_secs)
        if self.group_size > 1 and dist.is_initialized():
            # every worker receives the chief's variables (the reference's workers read them from the ps)
            from ..parallel.comm import Collectives
            coll = Collectives()          # stages GPU tensors through the host on gloo rehearsal groups
            coll.broadcast(st.params, 0)
            coll.broadcast(st.state, 0)
            st.values_version += 1
            st.initialized = True
            if getattr(st, "optimizer", "sgd") == "adam":
                # the slot variables: the chief broadcasts the full buffers, every rank keeps the slice it owns
                have = torch.tensor([1 if (self.is_chief and self.restored_values is not None
                                           and "beta1_power" in self.restored_values) else 0], dtype=torch.int32,
                                    device=st.device if coll.backend == "nccl" else "cpu")
                coll.broadcast(have, 0)
                if int(have.item()):
                    self._broadcast_moments(coll)
        elif self.is_chief and self.restored_values is not None and getattr(st, "optimizer", "sgd") == "adam":
            st.load_moments({k: torch.from_numpy(v) for k, v in self.restored_values.items()})
        self._last_save = time.time()
        self._last_rate_time = time.time()
        self._last_rate_step = None

    def _broadcast_moments(self, coll) -> None:
        st = self.state
        n = st.layout.total
        lo, hi = st.shard
        full = torch.zeros((2, n), dtype=torch.float32, device=st.device)
        t = torch.zeros(1, dtype=torch.int32, device=st.device)
        if self.is_chief:
            full_m, full_v, steps = st.moments_from_values({k: torch.from_numpy(v) for k, v in self.restored_values.items()})
            ops.h2d_verified(full[0], full_m, "Adam m")
            ops.h2d_verified(full[1], full_v, "Adam v")
            t[0] = steps
        coll.broadcast(full, 0)
        coll.broadcast(t, 0)
        st.m.copy_(full[0, lo:hi])
        st.v.copy_(full[1, lo:hi])
        st.state[2] = int(t.item())

    # ------------------------------------------------------------------ services driven by the training loop
    def save(self, global_step: Optional[int] = None) -> Optional[str]:
        """Write ``model.ckpt-<global_step>``.  Chief only - except that with sharded Adam moments (sync mode, several
        ranks) every rank must call it: the gather of the slot variables is a collective."""
        collective = self._collective_save()
        if not self.is_chief and not collective:
            return None
        moments = self._moments()
        if not self.is_chief:
            return None
        tensors = self.state.named_tensors(moments)
        gs = int(tensors["global_step"]) if global_step is None else global_step
        path = self.saver.save(tensors, self.logdir, gs)
        self._last_save = time.time()
        return path

    def maybe_checkpoint(self) -> Optional[str]:
        if not self.save_model_secs:
            return None
        if self._collective_save():
            # every rank is here once per launch; the chief's clock decides for all
            due = torch.tensor([1 if (self.is_chief and time.time() - self._last_save >= self.save_model_secs) else 0],
                               dtype=torch.int32)
            from ..parallel.comm import Collectives
            coll = Collectives()
            if coll.backend == "nccl":
                due = due.to(self.state.device)
            coll.broadcast(due, 0)
            return self.save() if int(due.item()) else None
        if not self.is_chief:
            return None
        if time.time() - self._last_save >= self.save_model_secs:
            return self.save()
        return None

    def report_step_rate(self, global_step: int, force: bool = False) -> Optional[float]:
        """``global_step/sec`` summary (Supervisor writes it because ``global_step=`` was passed, :111)."""
        now = time.time()
        if self._last_rate_step is None:
            self._last_rate_step, self._last_rate_time = global_step, now
            return None
        if not force and now - self._last_rate_time < self.save_summaries_secs:
            return None
        dt = max(now - self._last_rate_time, 1e-9)
        rate = (global_step - self._last_rate_step) / dt
        self._last_rate_step, self._last_rate_time = global_step, now
        if self.is_chief:
            if self._events is None:
                os.makedirs(self.logdir, exist_ok=True)
                self._events = open(os.path.join(self.logdir, "events.jsonl"), "a")
                try:        # the TensorBoard file needs the native runtime; the JSON lines are always written
                    from ..utils.tfevents import EventFileWriter
                    self._tb = EventFileWriter(self.logdir, now)
                except (OSError, ImportError, AttributeError):
                    self._tb = None
            rec = {"wall_time": now, "step": global_step, "global_step/sec": rate}
            if self.examples_per_global_step:
                rec["examples/sec"] = rate * self.examples_per_global_step
            self._events.write(json.dumps(rec) + "\n")
            self._events.flush()
            if self._tb is not None:
                self._tb.add_scalar("global_step/sec", rate, global_step, now)
                if self.examples_per_global_step:
                    self._tb.add_scalar("examples/sec", rec["examples/sec"], global_step, now)
        return rate

    def stop(self, save: bool = True) -> None:
        if save and self.save_model_secs and (self.is_chief or self._collective_save()):
            self.save()
        if self._events is not None:
            self._events.close()
            self._events = None
        if self._tb is not None:
            self._tb.close()
            self._tb = None

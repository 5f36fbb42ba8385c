"""Per-process communication endpoint (``tf.train.Server``, /root/reference/distributed.py:54-56).

The reference starts an in-process gRPC server in every task and lets ps tasks
block in ``server.join()`` forever.  Here:

* ``ps_backend='hbm'``: parameter-server state lives in the worker GPUs, so the
  only service a task needs is rendezvous.  The chief (worker 0) hosts a TCP
  key-value store at its ``--worker_hosts`` address; workers form the RCCL/gloo
  process group through it.  A ps task registers itself and ``join()`` blocks
  until the workers announce the end of training (a clean shutdown instead of
  the reference's never-returning ps).
* ``ps_backend='host'``: a ps task runs the native parameter service on its
  ``--ps_hosts`` address (``parallel/host_ps.py``); ``join()`` serves until the
  chief sends the shutdown request.
"""
from __future__ import annotations

import datetime
import time
from typing import Optional

import torch
import torch.distributed as dist

from .spec import ClusterSpec, Role, parse_host_port

_STORE_TIMEOUT = datetime.timedelta(seconds=1800)


class Server:
    def __init__(self, cluster: ClusterSpec, job_name: str, task_index: int, ps_backend: str = "hbm",
                 device: Optional[torch.device] = None, dist_backend: str = "auto", start: bool = True):
        self.role = Role(job_name, task_index, cluster)
        self.cluster = cluster
        self.ps_backend = ps_backend
        self.device = device or torch.device("cpu")
        self.dist_backend = dist_backend
        self.store = None
        self.ps_service = None
        self._joined = False
        # validates the index against the cluster (raises ValueError when out of range)
        self.address = self.role.address
        if start:
            self.start()

    @property
    def target(self) -> str:
        """What a session would connect to (``server.target``, distributed.py:125)."""
        return f"dtf://{self.address}"

    # ------------------------------------------------------------------ bring-up
    def start(self) -> None:
        if self.ps_backend == "host":
            self._start_host()
        else:
            self._start_hbm()

    def _start_hbm(self) -> None:
        role = self.role
        n_workers = role.num_workers
        if n_workers == 1 and role.is_worker and role.num_ps == 0:
            return  # nothing to rendezvous with
        host, port = role.chief_address
        is_master = role.is_chief
        self.store = dist.TCPStore(host, port, world_size=None, is_master=is_master, timeout=_STORE_TIMEOUT,
                                   wait_for_workers=False)
        if role.is_ps:
            self.store.set(f"dtf/ps/{role.task_index}/alive", str(time.time()))
            return
        if n_workers > 1 and not dist.is_initialized():
            backend = self.dist_backend
            if backend in ("auto", "", None):
                backend = "nccl" if self.device.type == "cuda" else "gloo"
            kwargs = {}
            if backend == "nccl":
                torch.cuda.set_device(self.device)
                kwargs["device_id"] = self.device
            dist.init_process_group(backend=backend, store=dist.PrefixStore("dtf/pg", self.store), rank=role.task_index,
                                    world_size=n_workers, timeout=_STORE_TIMEOUT, **kwargs)

    def _start_host(self) -> None:
        from ..parallel import host_ps
        if self.role.is_ps:
            host, port = parse_host_port(self.address)
            self.ps_service = host_ps.ParameterService(port, self.role.num_workers, host)
            self.ps_service.start()

    # ------------------------------------------------------------------ ps role
    def join(self, poll_secs: float = 0.5) -> None:
        """Block while the job runs (``server.join()``, distributed.py:56)."""
        if self.ps_backend == "host":
            if self.ps_service is not None:
                self.ps_service.wait()
            return
        if self.store is None:
            return
        while True:
            try:
                if self.store.check(["dtf/shutdown"]):
                    break
            except (RuntimeError, dist.DistNetworkError, dist.DistStoreError):
                break  # the chief (store host) is gone: the job is over
            time.sleep(poll_secs)
        self._joined = True

    # ------------------------------------------------------------------ teardown (workers)
    def shutdown(self) -> None:
        """Collective end of training: workers synchronise, the chief releases the ps tasks."""
        role = self.role
        if self.ps_backend == "host":
            return
        if dist.is_initialized():
            try:
                if dist.get_backend() == "nccl":
                    dist.barrier(device_ids=[self.device.index or 0])
                else:
                    dist.barrier()
            except RuntimeError:
                pass
        if self.store is not None and role.is_chief:
            self.store.set("dtf/shutdown", "1")
            # give ps tasks one poll period to notice before the store host exits
            time.sleep(1.0 if role.num_ps else 0.0)
        if dist.is_initialized():
            dist.destroy_process_group()

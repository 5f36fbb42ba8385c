"""Cluster description, roles and variable placement.

Counterparts of ``tf.train.ClusterSpec`` (/root/reference/distributed.py:49-53),
the static chief election (:58) and ``tf.train.replica_device_setter`` (:59-64).
"""
from __future__ import annotations

import socket
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple


def split_hosts(text: str) -> List[str]:
    """``FLAGS.ps_hosts.split(',')`` (distributed.py:49-50); surrounding blanks are tolerated."""
    return [h.strip() for h in text.split(",") if h.strip()]


def parse_host_port(hp: str) -> Tuple[str, int]:
    host, sep, port = hp.rpartition(":")
    if not sep or not port.isdigit():
        raise ValueError(f"expected host:port, got {hp!r}")
    return host, int(port)


class ClusterSpec:
    """Jobs -> ordered task addresses, e.g. ``{'ps': [...], 'worker': [...]}``."""

    def __init__(self, jobs: Dict[str, Sequence[str]]):
        self._jobs = {name: list(tasks) for name, tasks in jobs.items()}
        for name, tasks in self._jobs.items():
            for t in tasks:
                parse_host_port(t)
            if len(set(tasks)) != len(tasks):
                raise ValueError(f"job {name!r} lists the same address twice")

    @property
    def jobs(self) -> List[str]:
        return list(self._jobs)

    def job_tasks(self, job: str) -> List[str]:
        if job not in self._jobs:
            raise ValueError(f"no such job in the cluster: {job!r}")
        return list(self._jobs[job])

    def num_tasks(self, job: str) -> int:
        return len(self._jobs.get(job, []))

    def task_address(self, job: str, index: int) -> str:
        tasks = self.job_tasks(job)
        if not 0 <= index < len(tasks):
            raise ValueError(f"task index {index} out of range for job {job!r} with {len(tasks)} tasks")
        return tasks[index]

    def as_dict(self) -> Dict[str, List[str]]:
        return {k: list(v) for k, v in self._jobs.items()}

    def __eq__(self, other):
        return isinstance(other, ClusterSpec) and self._jobs == other._jobs

    def __repr__(self):
        return f"ClusterSpec({self._jobs!r})"


@dataclass(frozen=True)
class Role:
    """Who this process is (distributed.py:40-58)."""
    job_name: str
    task_index: int
    cluster: ClusterSpec

    @property
    def is_ps(self) -> bool:
        return self.job_name == "ps"

    @property
    def is_worker(self) -> bool:
        return self.job_name == "worker"

    @property
    def is_chief(self) -> bool:
        """Worker 0 is the chief (``is_chief = (FLAGS.task_index == 0)``, distributed.py:58)."""
        return self.is_worker and self.task_index == 0

    @property
    def num_workers(self) -> int:
        return self.cluster.num_tasks("worker")

    @property
    def num_ps(self) -> int:
        return self.cluster.num_tasks("ps")

    @property
    def address(self) -> str:
        return self.cluster.task_address(self.job_name, self.task_index)

    @property
    def worker_device(self) -> str:
        """The TF device string the reference builds (distributed.py:59); kept for logs."""
        return "/job:worker/task:%d/gpu:0" % self.task_index

    @property
    def chief_address(self) -> Tuple[str, int]:
        return parse_host_port(self.cluster.task_address("worker", 0))

    def local_rank(self) -> int:
        """Index of this worker among the workers that share its host: which GPU of the node it
        drives when every GPU is visible to every process."""
        host = parse_host_port(self.address)[0]
        same = [i for i, a in enumerate(self.cluster.job_tasks(self.job_name)) if parse_host_port(a)[0] == host]
        return same.index(self.task_index)


def validate_role(job_name, task_index, echo=None) -> None:
    """The two checks of distributed.py:40-47, same messages, in the same order: with ``echo`` (a print function)
    the accepted job name is printed before the task index is examined, as in the reference."""
    if job_name is None or job_name == '':
        raise ValueError('Must specify an explicit job_name !')
    if echo is not None:
        echo('job_name : %s' % job_name)
    if task_index is None or task_index == '':
        raise ValueError('Must specify an explicit task_index!')
    if echo is not None:
        echo('task_index : %d' % task_index)


def round_robin_placement(variable_names: Sequence[str], num_ps: int) -> Dict[str, int]:
    """``replica_device_setter`` default strategy: variables go to ps tasks round-robin in
    creation order (``global_step, hid_w, hid_b, sm_w, sm_b`` for the reference graph,
    distributed.py:65-73).  Whole variables, no partitioning."""
    if num_ps <= 0:
        raise ValueError("need at least one ps task")
    return {name: i % num_ps for i, name in enumerate(variable_names)}


def is_local_address(host: str) -> bool:
    if host in ("localhost", "127.0.0.1", "0.0.0.0", "::1", ""):
        return True
    try:
        return host in (socket.gethostname(), socket.getfqdn()) or \
            socket.gethostbyname(host) in ("127.0.0.1", socket.gethostbyname(socket.gethostname()))
    except OSError:
        return False

"""Spawn helpers for multi-process CPU (gloo) tests."""
import os
import socket
import sys
import traceback

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_one_rank_group(backend: str, **kwargs) -> None:
    """A one-rank process group inside the calling process, on a free local port; another port is tried when somebody
    else bound the chosen one between picking and listening (EADDRINUSE)."""
    import torch.distributed as dist
    last = None
    for _ in range(5):
        try:
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, **kwargs)
            return
        except Exception as e:                     # DistNetworkError: the port was taken in the meantime
            if "EADDRINUSE" not in str(e) and "address already in use" not in str(e):
                raise
            last = e
    raise last


def _entry(rank, world, port, fn, args, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    import torch
    torch.set_num_threads(1)
    try:
        q.put((rank, "ok", fn(rank, world, port, *args)))
    except BaseException:  # noqa
        q.put((rank, "err", traceback.format_exc()))


def run_ranks(fn, world, *args, timeout=100):
    """Run ``fn(rank, world, port, *args)`` in ``world`` processes; returns results by rank.  The rendezvous port is
    picked a moment before the ranks bind it; if somebody else took it in between, one more try on another port."""
    try:
        return _run_ranks_once(fn, world, args, timeout)
    except AssertionError as e:
        if "EADDRINUSE" not in str(e) and "address already in use" not in str(e):
            raise
    return _run_ranks_once(fn, world, args, timeout)


def _run_ranks_once(fn, world, args, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, args, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    results = {}
    deadline = time.time() + timeout
    try:
        while len(results) < world:
            try:
                rank, status, payload = q.get(timeout=1.0)
            except queue.Empty:
                # a rank that died without raising (signal, abort) leaves its peers blocked in a collective
                dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0) and i not in results]
                if dead:
                    raise AssertionError(f"rank(s) exited without a result: {dead}")
                if time.time() > deadline:
                    raise AssertionError(f"ranks {sorted(set(range(world)) - set(results))} gave no result in {timeout}s")
                continue
            if status != "ok":
                raise AssertionError(f"rank {rank} failed:\n{payload}")
            results[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return [results[r] for r in range(world)]

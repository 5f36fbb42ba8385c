"""Engines with several ranks sharing the one GPU of the test box (gloo carries the tensors, so this
checks the GPU-side logic of the engines — streams, service threads, captured steps — not RCCL)."""
import os

import pytest
import torch
import torch.distributed as dist

from dist_utils import free_port, run_ranks
from test_cli import COMMON, _finish, _launch, _steps

pytestmark = pytest.mark.gpu


def _sync_gpu(rank, world, port, collective, p2p=False, fused="1"):
    if p2p:
        os.environ["DTF_P2P_ALLREDUCE"] = "1"
        os.environ["DTF_P2P_FUSED"] = fused
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False, use_graphs=False)
    eng = SyncReplicasEngine(prog, st, collective=collective)
    eng.run(20)
    torch.cuda.synchronize()
    recs = prog.read_log(0, 20)
    out = {"params": st.params.cpu().numpy().copy(), "gs": st.global_step(), "first": recs[0].loss, "last": sum(r.loss for r in recs[-5:]) / 5.0,
           "p2p": eng.p2p is not None, "fused": eng.p2p_fused, "mode": eng.p2p_mode, "transport": eng.transport}
    dist.barrier()
    dist.destroy_process_group()
    return out


@pytest.mark.parametrize("collective", ["all_reduce", "reduce_scatter"])
def test_two_ranks_sync_on_one_gpu(collective):
    a, b = run_ranks(_sync_gpu, 2, collective)
    assert (a["params"] == b["params"]).all()
    assert a["gs"] == b["gs"] == 21
    assert a["last"] < a["first"] - 0.3


def _async_gpu(rank, world, port):
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterServerEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = AsyncParameterServerEngine(prog, st, num_shards=1, staleness=2)
    eng.run(30)
    eng.sync_params()
    recs = prog.read_log(0, 30)
    eng.finish()
    dist.barrier()
    out = {"gs": st.global_step(), "first": recs[0].loss, "last": sum(r.loss for r in recs[-5:]) / 5.0, "applied": eng.updates_applied_here}
    dist.destroy_process_group()
    return out


def test_three_ranks_async_on_one_gpu():
    res = run_ranks(_async_gpu, 3)
    assert res[0]["gs"] == 1 + 3 * 30 and res[0]["applied"] == 3 * 30
    assert all(r["last"] < r["first"] - 0.15 for r in res)      # mean of the last five batches


def test_host_backend_with_gpu_worker():
    ps_port, w_port = free_port(), free_port()
    args = ["--ps_backend=host", "--ps_hosts=127.0.0.1:%d" % ps_port, "--worker_hosts=127.0.0.1:%d" % w_port,
            "--train_steps=25"] + [a for a in COMMON if not a.startswith("--device")] + ["--device=gpu"]
    ps = _launch(["--job_name=ps", "--task_index=0"] + args)
    w = _launch(["--job_name=worker", "--task_index=0"] + args)
    lines = _finish(w)
    _finish(ps)
    st = _steps(lines)
    assert [s[1] for s in st] == list(range(1, 26))
    assert st[-1][2] < st[0][2] - 0.3


def _p2p_rank(rank, world, port):
    from distributed_tensorflow_amd.parallel.p2p_allreduce import P2PAllReduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ok = P2PAllReduce.self_check(79616, dev, rounds=9)
    dist.barrier()
    dist.destroy_process_group()
    return {"ok": ok}


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_allreduce_between_processes_sharing_the_gpu(world):
    """IPC mapping, flag protocol and reduction order of the one-shot all-reduce, rehearsed with several
    processes on the one GPU of the test box (the peers' regions are then IPC-mapped memory of the same device)."""
    res = run_ranks(_p2p_rank, world)
    assert all(r["ok"] for r in res)


@pytest.mark.parametrize("world,fused", [(2, "1"), (2, "0"), (3, "1")])
def test_ranks_sync_with_peer_mapped_allreduce(world, fused):
    """The whole engine path of the one-shot all-reduce: self-check, gradients written into the exported slots by
    the backward launch, then either the same launch exchanges tiles with the peers and updates (fused) or one
    update kernel per step does; against the same run over the process group."""
    res = run_ranks(_sync_gpu, world, "all_reduce", True, fused)
    a, b = res[0], res[-1]
    c = run_ranks(_sync_gpu, world, "all_reduce", False)[0]
    assert a["p2p"] and b["p2p"] and not c["p2p"] and a["fused"] == (fused == "1")
    assert (a["params"] == b["params"]).all()              # replicas stay bit-identical
    assert a["gs"] == b["gs"] == 21
    import numpy as np
    np.testing.assert_allclose(a["params"], c["params"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sync_with_sharded_peer_exchange(world):
    """Parameter-server state sharded over the ranks inside the backward launch: every workgroup's tiles travel to
    their owner rank, which sums them in rank order, updates and stores the new tiles into every replica; against
    the same run over the process group's reduce-scatter / all-gather."""
    res = run_ranks(_sync_gpu, world, "p2p_sharded", True, "1")
    c = run_ranks(_sync_gpu, world, "reduce_scatter", False)[0]
    assert all(r["p2p"] and r["fused"] and r["mode"] == "sharded" for r in res) and not c["p2p"]
    assert "sharded" in res[0]["transport"]
    for r in res[1:]:
        assert (r["params"] == res[0]["params"]).all()          # every replica holds the owners' bits
    assert all(r["gs"] == 21 for r in res)
    import numpy as np
    np.testing.assert_allclose(res[0]["params"], c["params"], rtol=1e-4, atol=1e-6)
    assert res[0]["last"] < res[0]["first"] - 0.3


def _peer_async_gpu(rank, world, port, shards, fused="0"):
    os.environ["DTF_PEER_ASYNC"] = "1"
    os.environ["DTF_PEER_ASYNC_FUSED"] = fused
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_peer import PeerMappedAsyncEngine, make_async_engine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = make_async_engine(prog, st, shards, 2)
    kind = type(eng).__name__
    eng.run(24, 8)                      # graph-replayed launches of steps
    for _ in range(6):
        eng.step()
    torch.cuda.synchronize()
    dist.barrier()                      # everybody has pushed everything
    gs = eng.sync_params()
    recs = prog.read_log(0, 30)
    params = st.params.cpu().numpy().copy()
    eng.finish()
    out = {"kind": kind, "gs": gs, "first": recs[0].loss, "last": sum(r.loss for r in recs[-5:]) / 5.0, "applied": eng.updates_applied_here,
           "params": params, "gs_seen": [r.global_step for r in recs]}
    dist.barrier()
    dist.destroy_process_group()
    return out


@pytest.mark.parametrize("world,shards,fused", [(3, 1, "0"), (4, 2, "0"), (3, 2, "1")])
def test_peer_mapped_async_engine_on_one_gpu(world, shards, fused):
    """Device-side Hogwild: every worker applies its gradients to the owners' mapped slices with one kernel per
    step (or, fused, inside the backward launch); global_step counts all updates; after the last push everybody
    pulls the same parameters."""
    res = run_ranks(_peer_async_gpu, world, shards, fused)
    assert all(r["kind"] == "PeerMappedAsyncEngine" for r in res)
    assert all(r["gs"] == 1 + world * 30 for r in res)
    for r in range(shards):
        assert res[r]["applied"] == world * 30
    for r in res[1:]:
        assert (r["params"] == res[0]["params"]).all()
    for r in res:
        assert r["last"] < r["first"] - 0.15                    # mean of the last five batches
        assert all(b >= a for a, b in zip(r["gs_seen"], r["gs_seen"][1:]))      # the counter a worker sees never goes back


def test_cli_async_workers_use_the_device_side_engine(monkeypatch):
    """Reference launch recipe, async mode, two workers sharing the GPU: the script picks the peer-mapped engine,
    republishes the ps-resident copies after the chief's initialisation, and global_step counts both workers."""
    monkeypatch.setenv("DTF_PEER_ASYNC", "1")
    ps_port, w0, w1 = free_port(), free_port(), free_port()
    args = ["--ps_hosts=127.0.0.1:%d" % ps_port, "--worker_hosts=127.0.0.1:%d,127.0.0.1:%d" % (w0, w1), "--train_steps=60",
            "--dist_backend=gloo"] + [a for a in COMMON if not a.startswith("--device")] + ["--device=gpu"]
    ps = _launch(["--job_name=ps", "--task_index=0"] + args)
    p1 = _launch(["--job_name=worker", "--task_index=1"] + args)
    p0 = _launch(["--job_name=worker", "--task_index=0"] + args)
    (o0, e0), (o1, e1) = p0.communicate(timeout=300), p1.communicate(timeout=300)
    _finish(ps)
    assert p0.returncode == 0 and p1.returncode == 0, e0[-2000:] + e1[-2000:]
    assert "update engine PeerMappedAsyncEngine" in e0 and "update engine PeerMappedAsyncEngine" in e1
    s0, s1 = _steps(o0.splitlines()), _steps(o1.splitlines())
    assert s0 and s1
    gs = sorted([s[1] for s in s0] + [s[1] for s in s1])
    assert gs[-1] >= 60 and len(s0) + len(s1) >= 59            # the two workers share the 60 updates
    assert s0[-1][2] < s0[0][2] - 0.2 and s1[-1][2] < s1[0][2] - 0.2


def _bench_two_ranks(config, extra, steps, warmup, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTF_DIST_BACKEND="gloo", DTF_SHARE_GPU="1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup",
           str(warmup), "--config", config] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 prints ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("config,extra,collective", [
    ("wide_mlp_sync", ["--dtype", "bf16", "--hidden_units", "1024", "--batch_size", "256"], "bucketed"),
    ("wide_mlp_async", ["--dtype", "bf16", "--hidden_units", "512", "--batch_size", "256"], None),
    ("resnet50_sync", ["--batch_size", "8"], "bucketed")])
def test_bench_launch_line_of_the_large_configs_with_two_ranks(config, extra, collective):
    """The N > 1 code paths of the large configurations end to end (bucketed aggregation behind the backward pass,
    owner-resident Adam for the async wide model), small sizes, two ranks sharing the GPU over gloo."""
    d = _bench_two_ranks(config, extra, 4, 2, {"DTF_MIOPEN_FIND": "0"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0
    assert d.get("final_loss", 0.0) == d.get("final_loss", 0.0)          # not NaN where reported
    if collective is not None:
        assert d["config"]["collective"] == collective


def test_bench_falls_back_to_the_process_group_when_the_peer_transport_gives_up():
    """A peer wait that runs into its limit during the warm-up (here: a limit of 100 ns) makes every rank rebuild
    the job without the peer-mapped transport; the JSON line says so."""
    d = _bench_two_ranks("mlp_sync", [], 60, 20, {"DTF_P2P_ALLREDUCE": "1", "DTF_P2P_TIMEOUT_S": "1e-7"})
    assert d["config"]["peer_mapped_allreduce"] is False and d["n_gpus"] == 2 and d["value"] > 0


@pytest.mark.parametrize("config,expect", [("mlp_sync", ("peer_mapped_allreduce", True)),
                                           ("mlp_async", ("engine", "PeerMappedAsyncEngine"))])
def test_bench_launch_line_with_two_ranks_sharing_the_gpu(config, expect):
    """The multi-GPU launch line of bench.py (torch.distributed.run, one rank per GPU) rehearsed with two ranks on
    the one GPU of the test box: gloo carries the process group, the peer-mapped kernels carry the updates."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTF_DIST_BACKEND="gloo", DTF_SHARE_GPU="1", DTF_P2P_ALLREDUCE="1", DTF_PEER_ASYNC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "120", "--warmup",
           "20", "--config", config]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 prints ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 120 and d["warmup"] == 20 and d["value"] > 0
    assert d["config"]["global_batch"] == 200 and d["config"][expect[0]] == expect[1]
    assert abs(d["value"] - 200 * 120 / (d["ms_per_step"] * 120 / 1e3)) < 1e-6 * d["value"]


def _sync_adam_gpu(rank, world, port, p2p, steps, resume_at):
    os.environ["DTF_P2P_ALLREDUCE"] = "1" if p2p else "0"
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.comm import Collectives
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=2000, validation_size=100, test_size=300, seed=rank)

    def build():
        st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, optimizer="adam", lr=2e-3)
        st.initialize(seed=3)
        prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False)
        return st, prog, SyncReplicasEngine(prog, st)

    st, prog, eng = build()
    assert (eng.p2p is not None and eng.p2p_mode == "sharded" and eng.p2p_fused) == p2p, eng.transport
    eng.run(resume_at, 8)
    saved = None
    if p2p:
        # checkpoint + restore in the middle: moments home to the slices, a fresh job, moments out to the owners again
        eng.moments_to_state()
        torch.cuda.synchronize()
        full = st.gather_moments(Collectives())
        saved = {n: st.var(n).cpu().clone() for n in st.names()}
        saved.update(st.named_tensors(full))
        eng.p2p.close()
        st, prog, eng = build()
        st.load_values(saved)
        st.load_moments(saved)
        eng.publish_state()
    eng.run(steps - resume_at, 8)
    eng.moments_to_state() if p2p else None
    torch.cuda.synchronize()
    full = st.gather_moments(Collectives())
    out = {"params": st.params.cpu().numpy().copy(), "m": full[0].cpu().numpy().copy(), "v": full[1].cpu().numpy().copy(),
           "gs": st.global_step(), "t": int(st.state[2].item()), "transport": eng.transport}
    dist.barrier()
    dist.destroy_process_group()
    return out


def test_sync_adam_on_the_sharded_exchange_matches_the_collective_path():
    """`--sync_replicas --optimizer adam` with two workers: the tile owners keep Adam's moments inside the backward
    launch (sharded peer exchange).  Same trajectory as reduce-scatter -> shard Adam -> all-gather through the process
    group, identical replicas, the same moments at the end, and a checkpoint / restore in the middle changes nothing."""
    steps = 24
    a = run_ranks(_sync_adam_gpu, 2, True, steps, 10, timeout=150)
    b = run_ranks(_sync_adam_gpu, 2, False, steps, 10, timeout=150)
    assert "sharded" in a[0]["transport"] and "sharded" not in b[0]["transport"]
    assert (a[0]["params"] == a[1]["params"]).all()                       # replicas identical, bit for bit
    assert a[0]["gs"] == b[0]["gs"] == steps + 1 and a[0]["t"] == b[0]["t"] == steps
    import numpy as np
    np.testing.assert_allclose(a[0]["params"], b[0]["params"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(a[0]["m"], b[0]["m"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(a[0]["v"], b[0]["v"], rtol=2e-3, atol=1e-9)


def _supervised_adam_resume(rank, world, port, logdir):
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    from distributed_tensorflow_amd.supervisor.supervisor import Supervisor
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=700, validation_size=50, test_size=50, seed=40 + rank)

    def build(sub):
        st = build_mlp_state(784, [64], 10, dev, world_size=world, rank=rank, optimizer="adam", lr=0.01)
        sv = Supervisor(rank == 0, os.path.join(logdir, sub), lambda: st.initialize(9), st, recovery_wait_secs=0.05,
                        save_model_secs=600, store=store, group_size=world)
        sv.prepare_or_wait_for_session()
        prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=40 + rank), 50, [64], fused_update=False,
                          post_update_accuracy=False)
        eng = SyncReplicasEngine(prog, st)
        sv.engine = eng
        eng.publish_state()                       # what train.py does after the session is there
        return st, prog, eng, sv

    st_a, prog_a, eng_a, _ = build("a")
    assert eng_a.p2p is not None and eng_a.p2p_mode == "sharded", eng_a.transport
    eng_a.run(40, 8)
    st_b, prog_b, eng_b, sv_b = build("b")
    eng_b.run(20, 8)
    sv_b.save()                                   # every rank: the moments come home through a collective
    dist.barrier()
    st_c, prog_c, eng_c, sv_c = build("b")        # a new job on the same directory
    gs_restored = st_c.global_step()
    eng_c.run(20, 8)
    torch.cuda.synchronize()
    out = {"a": st_a.params.cpu().numpy().copy(), "c": st_c.params.cpu().numpy().copy(), "gs_restored": gs_restored,
           "gs": st_c.global_step(), "loss_a": [r.loss for r in prog_a.read_log(20, 20)],
           "loss_c": [r.loss for r in prog_c.read_log(20, 20)]}
    dist.barrier()
    dist.destroy_process_group()
    return out


def test_supervised_resume_with_adam_on_the_sharded_exchange(tmp_path):
    """Two GPU workers, sync mode, Adam with the moments on the tile owners: 20 steps, Supervisor.save (moments gathered
    from the owners under TF's slot names), a new job restores and publishes them, 20 more steps = 40 uninterrupted
    ones bit for bit."""
    res = run_ranks(_supervised_adam_resume, 2, str(tmp_path), timeout=200)
    for r in res:
        assert r["gs_restored"] == 21 and r["gs"] == 41
        assert (r["a"] == r["c"]).all() and r["loss_a"] == r["loss_c"]
    assert (res[0]["c"] == res[1]["c"]).all()


def _late_peer(rank, world, port, mode):
    os.environ.update(DTF_P2P_ALLREDUCE="1", DTF_P2P_FUSED="1", DTF_P2P_MODE=mode, DTF_P2P_TIMEOUT_S="0.25")
    import time
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False)
    eng = SyncReplicasEngine(prog, st, collective="all_reduce")
    prog.prepare_graphs(32, 16)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 1:
        time.sleep(3.0)                     # this rank's first step comes long after the other's time limit
    t0 = time.time()
    prog.run(32, 16)                        # 2 graphs of 16 steps = 64 launches that wait for a peer
    late = eng.p2p.timed_out()              # synchronises
    waited = time.time() - t0
    dist.barrier()
    dist.destroy_process_group()
    return {"late": bool(late), "waited": waited, "mode": eng.p2p_mode}


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
def test_a_dead_link_costs_one_time_limit_not_one_per_step(mode):
    """Rank 1 starts 3 s late, the limit of a peer wait is 0.25 s: rank 0's first wait runs into the limit, and the other
    31 steps of its two graphs must give up at once (the kernels look at the error word first) instead of waiting
    0.25 s each."""
    res = run_ranks(_late_peer, 2, mode, timeout=120)
    assert res[0]["mode"] == mode and res[0]["late"]
    assert res[0]["waited"] < 2.0, res[0]


def test_bench_wide_async_with_adam_on_the_device_side_engine():
    """`wide_mlp_async` (Adam) on the peer-mapped push/pull engine: the moments live next to the master slice in the
    owner's exported region, the bf16 products run on the pulled replica; two ranks sharing the GPU."""
    d = _bench_two_ranks("wide_mlp_async", ["--dtype", "bf16", "--hidden_units", "512", "--batch_size", "256"], 6, 2,
                         {"DTF_PEER_ASYNC": "1"})
    c = d["config"]
    assert c["engine"] == "PeerMappedAsyncEngine" and c["optimizer"] == "adam" and c["mode"] == "async"
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    assert c["staleness_configured"] == 0 and d["staleness_max"] >= 0 and d["global_step"] >= 1 + 2 * 8


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_ranks_bucketed_overlap_on_one_gpu(dtype):
    # buckets are processed on a second stream while the backward kernels of the layers below run;
    # result must equal the single reduce-scatter after the backward pass, bit for bit
    from test_bucketed import _wide_rank
    a = run_ranks(_wide_rank, 2, "bucketed", "adam", "cuda", dtype)
    b = run_ranks(_wide_rank, 2, "reduce_scatter", "adam", "cuda", dtype)
    assert (a[0][0] == a[1][0]).all()
    assert (a[0][0] == b[0][0]).all()
    assert a[0][1] == 6 and a[0][2][-1] < a[0][2][0]


def _bucketed_native_one_rank(rank, world, port, collective):
    # one-rank RCCL communicator: the native collectives issued on the side stream, events in both directions
    from distributed_tensorflow_amd import ops
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import build_mlp_state
    from distributed_tensorflow_amd.models.wide_mlp import WideMLPProgram
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    hidden = [512, 512, 256]
    ds = mnist.synthetic_data_sets(train_size=2048, validation_size=10, test_size=10, seed=0)
    st = build_mlp_state(784, hidden, 10, dev, optimizer="adam", lr=1e-3)
    st.initialize(seed=5)
    prog = WideMLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 256, hidden, loss_mode=ops.LOSS_STANDARD,
                          dtype=torch.bfloat16)
    eng = SyncReplicasEngine(prog, st, collective=collective, force_collectives=True, capture_collectives=False)
    eng.run(8, 4)
    torch.cuda.synchronize()
    out = (st.params.cpu().numpy().copy(), eng.native_comm is not None, eng.bucketed is not None,
           [r.loss for r in prog.read_log(0, 8)], prog._run_behind is not None and all(prog._side_fresh))
    dist.destroy_process_group()
    return out


def test_bucketed_update_through_native_rccl_entry_points(monkeypatch):
    (a,) = run_ranks(_bucketed_native_one_rank, 1, "bucketed")
    (b,) = run_ranks(_bucketed_native_one_rank, 1, "reduce_scatter")
    assert a[1] and a[2] and not b[2]
    assert a[4] and not b[4]           # the compute copies of the weights were rebuilt behind each bucket's all-gather
    assert (a[0] == b[0]).all()
    assert a[3][-1] < a[3][0]
    monkeypatch.setenv("DTF_SIDE_REFRESH", "0")
    (c,) = run_ranks(_bucketed_native_one_rank, 1, "bucketed")
    assert c[2] and not c[4] and (c[0] == a[0]).all()


def _module_bucketed_one_rank(rank, world, port, collective, steps, autocast=False):
    # buckets released from autograd's device thread: stream hand-over must hold there too
    from distributed_tensorflow_amd.models.resnet import ResNet, TorchModuleProgram
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    model = ResNet(layers=(1, 1, 1, 1), num_classes=10, width=16)
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(32, 3, 32, 32, generator=g), torch.randint(0, 10, (32,), generator=g)
    prog = TorchModuleProgram(model, dev, x, y, batch=16, optimizer="sgd", lr=0.05, bucket_bytes=128 << 10,
                              autocast=torch.bfloat16 if autocast else None)
    eng = SyncReplicasEngine(prog, prog.state, collective=collective, force_collectives=True, capture_collectives=False)
    eng.run(steps)
    torch.cuda.synchronize()
    nb = len(eng.bucketed.ranges) if eng.bucketed is not None else 0
    out = (prog.state.params.cpu().numpy().copy(), nb, [r.loss for r in prog.read_log(0, steps)], prog._direct_now,
           len(prog._copies), prog._side is not None)
    dist.destroy_process_group()
    return out


def test_module_buckets_released_from_autograd_thread_on_gpu():
    # MIOpen's convolution gradients are not run-to-run reproducible and this small network amplifies the last-bit
    # differences within a few steps (two identical runs differ by 4e-2 after 6 steps), so parameters are compared
    # after ONE update and the longer run only has to train
    import numpy as np
    (a,) = run_ranks(_module_bucketed_one_rank, 1, "bucketed", 1)
    (b,) = run_ranks(_module_bucketed_one_rank, 1, "reduce_scatter", 1)
    assert a[1] > 2 and b[1] == 0
    np.testing.assert_allclose(a[0], b[0], rtol=1e-3, atol=1e-4)
    (c,) = run_ranks(_module_bucketed_one_rank, 1, "bucketed", 8)
    assert abs(c[2][0] - a[2][0]) < 1e-4 and c[2][-1] < c[2][0]


@pytest.mark.parametrize("side", [False, True])
def test_module_buckets_with_bf16_weight_copies(side, monkeypatch):
    """The single-GPU arrangement of the convolutional step - gradients handed over by autograd, persistent bf16 copies
    of the convolution weights, optionally weight gradients on a second stream - under the bucketed aggregation: a
    bucket is copied into the flat buffer and released when its last gradient exists."""
    import numpy as np
    monkeypatch.setenv("DTF_WGRAD_STREAM_BUCKETED", "1" if side else "0")
    (a,) = run_ranks(_module_bucketed_one_rank, 1, "bucketed", 1, True)
    (b,) = run_ranks(_module_bucketed_one_rank, 1, "reduce_scatter", 1, True)
    assert a[1] > 2 and b[1] == 0
    assert a[3] and a[4] > 0 and a[5] == side              # direct gradients, weight copies (side stream as asked)
    np.testing.assert_allclose(a[0], b[0], rtol=2e-2, atol=2e-3)          # bf16 convolutions, not run-to-run reproducible
    (c,) = run_ranks(_module_bucketed_one_rank, 1, "bucketed", 8, True)
    assert abs(c[2][0] - a[2][0]) < 2e-2 and c[2][-1] < c[2][0]


def _replica_fallback(rank, world, port):
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    os.environ["DTF_P2P_FUSED"] = "1"
    import warnings
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False)
    eng = SyncReplicasEngine(prog, st, collective="p2p_replicated")     # the exchange of gradient tiles inside the backward launch
    assert eng.p2p is not None and eng.p2p_fused and eng.p2p_mode == "replicated"
    eng.run(6, 4)                                   # first run: replicas compared, equal
    torch.cuda.synchronize()
    # what a machine with a different ordering of peer traffic would produce: one replica off
    if rank == 1:
        st.params[5] += 1.0
    eng._replica_checks_left = 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.run(6, 4)
    torch.cuda.synchronize()
    warned = any("separate update launch" in str(x.message) for x in w)
    eng.run(8, 4)                                   # continues on the separate update launch, compared once more
    torch.cuda.synchronize()
    out = (st.params.cpu().numpy().copy(), warned, eng.p2p_fused, st.global_step())
    dist.barrier()
    dist.destroy_process_group()
    return out


def test_differing_replicas_fall_back_to_the_update_launch():
    a, b = run_ranks(_replica_fallback, 2)
    assert a[1] and b[1] and not a[2] and not b[2]
    assert (a[0] == b[0]).all() and a[3] == b[3] == 21


def _sync_gpu_long(rank, world, port, steps, collective="all_reduce"):
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    os.environ["DTF_P2P_FUSED"] = "1"
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=5000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.05)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False)
    eng = SyncReplicasEngine(prog, st, collective=collective)
    assert eng.p2p_fused and prog.use_graphs
    done = 0
    while done < steps:                            # graph replays of 64 steps, ranks free-running between host checks
        n = min(1024, steps - done)
        eng.run(n, 64)
        done += n
    torch.cuda.synchronize()
    out = (st.params.cpu().numpy().copy(), st.global_step())
    dist.barrier()
    dist.destroy_process_group()
    return out


@pytest.mark.parametrize("world,collective", [(2, "all_reduce"), (3, "all_reduce"), (2, "p2p_sharded"), (3, "p2p_sharded")])
def test_peer_exchange_keeps_replicas_identical_over_thousands_of_graph_replayed_steps(world, collective):
    """Slots, boxes and flags are reused step after step and the ranks run ahead of each other by up to a step inside
    the replayed graphs: after 4000 steps every replica must still hold the same bits (both transports)."""
    res = run_ranks(_sync_gpu_long, world, 4000, collective, timeout=150)
    assert all(r[1] == 4001 for r in res), [r[1] for r in res]
    for i, r in enumerate(res[1:], 1):
        differ = int((r[0] != res[0][0]).sum())
        assert differ == 0, f"rank {i}: {differ} of {r[0].size} parameters differ from rank 0"


# ------------------------------------------------------------------------------------------------------------------
# SyncReplicas with replicas_to_aggregate != workers on the device (stamped contributions, first-R selection by the
# chief, multi-round accumulation; csrc/kernels/p2p_allreduce.hip: sync_general_kernel), against the NumPy oracle of
# tests/test_sync_engine.py.  Ranks share the one GPU of the test box; an artificial DEVICE-side delay makes one rank late.
# ------------------------------------------------------------------------------------------------------------------
def _general_gpu(rank, world, port, R, steps, slow_rank, stale_rank=-1, graph_steps=4):
    os.environ["DTF_P2P_ALLREDUCE"] = "1"
    import oracle
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=600, validation_size=50, test_size=50, seed=100 + rank)
    st = build_mlp_state(784, [16], 10, dev, world_size=world, rank=rank, lr=0.5)
    init = oracle.init_params(16, seed=3)
    st.load_values({k: torch.from_numpy(v).float() for k, v in init.items()})
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=100 + rank), 50, [16], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = SyncReplicasEngine(prog, st, replicas_to_aggregate=R, collective="all_reduce",
                             device_delay_s=0.03 if rank == slow_rank else 0.0, stamp_offset=-1 if rank == stale_rank else 0)
    assert eng.general_on_device and prog.use_graphs
    dist.barrier()
    eng.run(steps, graph_steps)                 # graph-replayed: no host between the rounds
    torch.cuda.synchronize()
    masks = prog.wlog.cpu()[:steps * eng.rounds, 2].tolist()
    out = {"params": {n: st.var(n).cpu().numpy().copy() for n in st.names()}, "gs": st.global_step(), "ls": st.local_step(),
           "dropped": eng.dropped, "masks": masks, "recs": [r.global_step for r in prog.read_log(0, steps * eng.rounds)]}
    dist.barrier()
    dist.destroy_process_group()
    return out


def _oracle_run(contributions, R, steps):
    """contributions(s) -> list of (rank, batch index) whose gradients update s averages."""
    import numpy as np
    import oracle
    from distributed_tensorflow_amd.data import mnist
    params = oracle.init_params(16, seed=3)
    sets = {}
    for s in range(steps):
        grads = []
        for rank, b in contributions(s):
            if rank not in sets:
                sets[rank] = mnist.synthetic_data_sets(train_size=600, validation_size=50, test_size=50, seed=100 + rank).train
            rows = sets[rank].batch_rows(b * 50, 50)
            x, y = sets[rank].images[rows].astype(np.float64), sets[rank].labels[rows].astype(np.float64)
            grads.append(oracle.loss_and_grads(params, x, y)[2])
        params = oracle.sgd(params, {k: sum(g[k] for g in grads) / R for k in params}, 0.5)
    return params


def _assert_params(res, want):
    import numpy as np
    for r in res[1:]:
        for k in res[0]["params"]:
            np.testing.assert_array_equal(r["params"][k], res[0]["params"][k])       # identical replicas
    for k, v in want.items():
        np.testing.assert_allclose(res[0]["params"][k], v, rtol=5e-4, atol=5e-6)


def test_device_side_backup_worker_gradient_is_dropped_when_r_less_than_n():
    world, R, steps, slow = 3, 2, 4, 2
    res = run_ranks(_general_gpu, world, R, steps, slow)
    assert res[slow]["dropped"] == steps and res[0]["dropped"] == 0 and res[1]["dropped"] == 0
    assert all(m == 0b011 for m in res[0]["masks"])                 # the chief took ranks 0 and 1 every time
    _assert_params(res, _oracle_run(lambda s: [(0, s), (1, s)], R, steps))
    assert all(r["gs"] == 1 + steps and r["ls"] == steps for r in res)
    assert res[0]["recs"] == list(range(1, steps + 1))              # global_step fetched with each step


def test_device_side_r_greater_than_n_accumulates_several_rounds():
    world, R, steps = 2, 3, 3          # every update: both workers once, then the faster one again
    res = run_ranks(_general_gpu, world, R, steps, 1)
    _assert_params(res, _oracle_run(lambda s: [(0, 2 * s), (1, 2 * s), (0, 2 * s + 1)], R, steps))
    assert all(r["gs"] == 1 + steps and r["ls"] == 2 * steps for r in res)     # one increment per aggregate
    assert res[1]["dropped"] == steps and res[0]["dropped"] == 0
    assert res[0]["masks"] == [0b11, 0b01] * steps


def test_device_side_stale_stamp_is_never_aggregated():
    """A contribution that carries the stamp of an older round (a worker that computed on an old token) is dropped
    by the accumulator, however early it arrives (ConditionalAccumulator semantics, distributed.py:97-100)."""
    world, R, steps, stale = 3, 2, 3, 1
    res = run_ranks(_general_gpu, world, R, steps, -1, stale)
    assert res[stale]["dropped"] == steps and res[0]["dropped"] == 0 and res[2]["dropped"] == 0
    assert all(m == 0b101 for m in res[0]["masks"])
    _assert_params(res, _oracle_run(lambda s: [(0, s), (2, s)], R, steps))


def test_device_side_general_path_with_r_equal_to_n_is_the_mean():
    world, steps = 2, 5
    # R == N goes through the fast path by default; the aggregation kernel must give the same mean when asked to
    res = run_ranks(_general_forced, world, steps)
    _assert_params(res, _oracle_run(lambda s: [(0, s), (1, s)], world, steps))


def _general_forced(rank, world, port, steps):
    os.environ["DTF_FORCE_GENERAL_SYNC"] = "1"
    return _general_gpu(rank, world, port, world, steps, -1)


# ------------------------------------------------------------------------------------------------------------------
# Peer-mapped async engine: measured staleness, owner-resident Adam, no job-wide coupling in steady state
# ------------------------------------------------------------------------------------------------------------------
def _peer_async_v2(rank, world, port, shards, staleness, optimizer, steps, leaver=-1):
    os.environ["DTF_PEER_ASYNC"] = "1"
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_peer import make_async_engine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.5 if optimizer == "sgd" else 0.01,
                         optimizer=optimizer)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = make_async_engine(prog, st, shards, staleness)
    kind = type(eng).__name__
    if rank == leaver:
        eng.run(5, 5)
        torch.cuda.synchronize()
        return {"kind": kind, "left_after": 5}       # no finish(), no barrier: this worker is simply gone
    eng.run(steps, 8)
    torch.cuda.synchronize()
    if leaver < 0:
        dist.barrier()                               # everybody has pushed everything
    else:
        import time
        time.sleep(1.0)                              # the leaver's five steps are long done
    gs = eng.sync_params()
    torch.cuda.synchronize()
    recs = prog.read_log(0, steps)
    smax, smean = eng.staleness_stats(0, steps)
    pulls = eng.alog.cpu()[:steps]
    everyone = eng.finish(timeout_s=3.0 if leaver >= 0 else 60.0)
    out = {"kind": kind, "gs": gs, "first": recs[0].loss, "last": sum(r.loss for r in recs[-5:]) / 5.0,
           "applied": eng.updates_applied_here, "params": st.params.cpu().numpy().copy(), "smax": smax, "smean": smean,
           "gs_at_push": pulls[:, 0].tolist(), "everyone": everyone,
           "m_norm": float(st.m.abs().sum().item()) if st.m is not None else 0.0, "adam_t": int(st.state[2].item())}
    if leaver < 0:
        dist.barrier()
        dist.destroy_process_group()
    return out


def _one_rank_async(rank, world, port, optimizer, staleness, fused):
    """One worker and its own ps shard through the mapped region: the arithmetic of a push is the same whether the
    backward launch or the separate kernel applies it."""
    os.environ["DTF_PEER_ASYNC_FUSED"] = fused
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_peer import PeerMappedAsyncEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=0)
    st = build_mlp_state(784, [100], 10, dev, lr=0.5 if optimizer == "sgd" else 0.01, optimizer=optimizer)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), 100, [100], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = PeerMappedAsyncEngine(prog, st, 1, staleness=staleness)
    eng.run(21, 8)
    eng.step()
    torch.cuda.synchronize()
    before_pull = st.params.cpu().numpy().copy()
    gs = eng.sync_params()
    torch.cuda.synchronize()
    out = {"fused": eng.fused, "gs": gs, "replica": before_pull, "params": st.params.cpu().numpy().copy(),
           "loss": [r.loss for r in prog.read_log(0, 22)], "alog": eng.alog.cpu()[:22].tolist()}
    eng.finish()
    out["adam_t"] = int(st.state[2].item())
    dist.destroy_process_group()
    return out


@pytest.mark.parametrize("optimizer,staleness", [("sgd", 0), ("sgd", 2), ("adam", 0), ("adam", 3)])
def test_push_inside_the_backward_launch_equals_the_separate_kernel(optimizer, staleness):
    (a,) = run_ranks(_one_rank_async, 1, optimizer, staleness, "1")
    (b,) = run_ranks(_one_rank_async, 1, optimizer, staleness, "0")
    assert a["fused"] and not b["fused"] and a["gs"] == b["gs"] == 23
    assert a["alog"] == b["alog"] and a["adam_t"] == b["adam_t"]
    assert a["loss"] == b["loss"]
    assert (a["params"] == b["params"]).all()            # the master after 22 pushes
    import numpy as np
    # between pulls the replica follows the worker's own updates: the same values up to the rounding of w + (new - old)
    np.testing.assert_allclose(a["replica"], b["replica"], rtol=1e-5, atol=1e-6)
    assert a["loss"][-1] < a["loss"][0]


@pytest.mark.parametrize("fused", ["1", "0"])
def test_peer_async_staleness_is_measured_on_the_device(fused, monkeypatch):
    """staleness 2: a worker pushes every step and pulls every third; the kernel logs how many updates of the OTHER
    workers its replica had missed at each push."""
    monkeypatch.setenv("DTF_PEER_ASYNC_FUSED", fused)
    world, steps = 3, 30
    res = run_ranks(_peer_async_v2, world, 1, 2, "sgd", steps)
    assert all(r["kind"] == "PeerMappedAsyncEngine" for r in res)
    assert all(r["gs"] == 1 + world * steps for r in res)           # global_step counts every worker's update
    assert res[0]["applied"] == world * steps
    for r in res[1:]:
        assert (r["params"] == res[0]["params"]).all()              # after the last pull everybody holds the master
    for r in res:
        assert r["last"] < r["first"] - 0.1
        assert 0 <= r["smean"] <= r["smax"] <= world * steps
        assert all(b > a for a, b in zip(r["gs_at_push"], r["gs_at_push"][1:]))    # the counter only grows
    assert max(r["smax"] for r in res) > 0                          # three workers do miss each other's updates


@pytest.mark.parametrize("fused", ["1", "0"])
def test_peer_async_adam_keeps_its_moments_with_the_owner(fused, monkeypatch):
    monkeypatch.setenv("DTF_PEER_ASYNC_FUSED", fused)
    world, steps = 2, 24
    res = run_ranks(_peer_async_v2, world, 2, 0, "adam", steps)
    assert all(r["kind"] == "PeerMappedAsyncEngine" for r in res)
    assert all(r["gs"] == 1 + world * steps for r in res)
    assert all(r["applied"] == world * steps for r in res)          # both own a slice; each slice saw every update
    assert all(r["m_norm"] > 0 and r["adam_t"] == world * steps for r in res)     # slot variables came home at finish()
    assert (res[0]["params"] == res[1]["params"]).all()
    assert all(r["last"] < r["first"] - 0.05 for r in res)


def test_peer_async_worker_that_leaves_does_not_block_the_others():
    """Async workers only talk to the ps shards (device_filters, distributed.py:116-117): one of three workers runs
    five steps and disappears; the other two finish their 30 steps, pull, and their finish() gives up waiting for
    the missing worker after its time limit."""
    world, steps, leaver = 3, 30, 2
    res = run_ranks(_peer_async_v2, world, 1, 0, "sgd", steps, leaver, timeout=120)
    assert res[leaver]["left_after"] == 5
    assert res[0]["gs"] == res[1]["gs"] == 1 + 2 * steps + 5
    assert res[0]["applied"] == 2 * steps + 5
    assert res[0]["everyone"] is False and res[1]["everyone"] is False
    assert (res[0]["params"] == res[1]["params"]).all()


def _peer_async_bound(rank, world, port, bound, steps):
    """Rank 1 is the slow worker (it sleeps on the host between launches of four steps), rank 0 runs flat out."""
    os.environ["DTF_PEER_ASYNC"] = "1"
    import time
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
    from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state
    from distributed_tensorflow_amd.parallel.async_peer import make_async_engine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ds = mnist.synthetic_data_sets(train_size=1000, validation_size=100, test_size=300, seed=rank)
    st = build_mlp_state(784, [100], 10, dev, world_size=world, rank=rank, lr=0.1)
    st.initialize(seed=3)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=rank), 100, [100], fused_update=False, gs_inc=0,
                      post_update_accuracy=False)
    eng = make_async_engine(prog, st, 1, 3, staleness_bound=bound)      # pull every fourth step unless forced
    dist.barrier()
    if rank == 1:
        for _ in range(steps // 4):
            eng.run(4, 4)
            torch.cuda.synchronize()
            time.sleep(0.01)
    else:
        eng.run(steps, 8)
    torch.cuda.synchronize()
    smax, smean = eng.staleness_stats(0, steps)
    out = {"kind": type(eng).__name__, "smax": smax, "smean": smean, "forced": eng.forced_pulls(), "waits": eng.bound_waits(),
           "gave_up": eng.bound_given_up(), "bound": eng.staleness_bound}
    dist.barrier()
    out["gs"] = eng.sync_params()
    eng.finish(timeout_s=30.0)
    dist.barrier()
    dist.destroy_process_group()
    return out


def test_peer_async_staleness_bound_is_enforced_on_the_device():
    """Bound 3 with two workers, one of them slow: the fast worker waits on the device (its push kernel) whenever it is
    more than three steps ahead, replicas that fell behind are refreshed ahead of their pull period, and no gradient is
    applied more than bound + (workers - 1) x (2 x bound + 1) updates late.  Without a bound the slow worker's
    gradients are as late as the fast worker is fast."""
    world, steps, bound = 2, 64, 3
    res = run_ranks(_peer_async_bound, world, bound, steps, timeout=200)
    assert all(r["kind"] == "PeerMappedAsyncEngine" and r["bound"] == bound and not r["gave_up"] for r in res)
    assert all(r["gs"] == 1 + world * steps for r in res)
    limit = bound + (world - 1) * (2 * bound + 1)
    assert all(r["smax"] <= limit for r in res), [r["smax"] for r in res]
    assert res[0]["waits"] > 0                                          # the fast worker did wait for the slow one
    assert res[0]["forced"] + res[1]["forced"] > 0
    free = run_ranks(_peer_async_bound, world, -1, steps, timeout=200)
    assert all(r["bound"] is None and r["waits"] == 0 and r["forced"] == 0 for r in free)
    assert max(r["smax"] for r in free) > limit                         # plain Hogwild: nothing holds the fast worker back

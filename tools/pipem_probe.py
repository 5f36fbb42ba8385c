"""Where the time of the N-worker pipelined step goes on ONE rank that exchanges with itself through its own region:
graphs of launches of `pipem_step_kernel` restricted to subsets of its workgroup roles (results of such launches are
meaningless; only the time is read).  The probe launches carry a time limit of zero: a row workgroup that finds a box
word without the mark it expects (role T absent: the step counter stands still; virtual peers: nobody writes their
boxes) goes on at once, so the launches cost what a step costs whose boxes have all arrived.

    python tools/pipem_probe.py [launches per graph] [repeats] [virtual world]

`virtual world` > 1: the launch shape of that many workers (Gram tiles, K split, strips) with every "peer" region
being this rank's own - the compute growth with the world size, without any link.
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd import ops                                   # noqa: E402
from distributed_tensorflow_amd.data import mnist                            # noqa: E402
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset     # noqa: E402
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state    # noqa: E402
from distributed_tensorflow_amd.parallel.pipe_exchange import PipeExchange   # noqa: E402

L = ops._lib
ROLES = {"L": L.PIPE_ROLE_L, "Q": L.PIPE_ROLE_Q, "G": L.PIPE_ROLE_G, "W": L.PIPE_ROLE_W, "T": L.PIPE_ROLE_T, "F": L.PIPE_ROLE_F}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    vworld = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
    st = build_mlp_state(784, [100], 10, dev, lr=0.01)
    st.initialize(seed=1)
    os.environ["DTF_PIPELINED_STEP"] = "0"
    data = DeviceDataset(ds.train, dev, shuffle_seed=0)
    prog = MLPProgram(st, data, 100, [100], fused_update=False)
    os.environ["DTF_PIPELINED_STEP"] = "1"
    x = PipeExchange(dev, 0, 1, timeout_s=2.0 if vworld == 1 else 0.0)
    if vworld > 1:
        # every virtual peer is this rank: same region, same shuffle (rank 0 stays the "own" rank)
        x.world = vworld
        x.regions = [x.regions[0]] * vworld
        data.enable_peer_shuffles([0] * vworld, 0)
        data.peer_perm = [data.perm] * vworld
    else:
        data.enable_peer_shuffles([0], 0)
    prog.use_pipe_exchange(x, 0.01 / vworld)
    prog.run(16, 8)
    torch.cuda.synchronize()
    assert vworld > 1 or not x.timed_out()
    lib = L.kernels()
    subsets = (os.environ.get("PIPE_PROBE_SUBSETS") or "L,Q,G,W,T,LT,LW,LQ,LG,QGW,LQGWT").split(",")
    print(f"world {vworld}: {n} launches per graph, median of {reps} replays, us per launch")
    for names in subsets:
        mask = 0
        for c in names:
            mask |= ROLES[c]
        packs = []
        for q in range(prog.P):
            a = L.PipeMultiArgs.from_buffer_copy(prog._args[q]["pipe_step"])
            a.b.roles = mask
            a.timeout_ticks = 0
            packs.append(a)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                rc = lib.dtf_pipem_launch_roles(C.byref(packs[i % prog.P]), L.stream_handle())
                assert rc == 0, rc
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"  roles {names:6s} {ts[len(ts) // 2] / n * 1e6:7.2f}", flush=True)
    print("timed out:", x.timed_out())


if __name__ == "__main__":
    main()

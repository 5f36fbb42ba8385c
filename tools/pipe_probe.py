"""Where the time of the one-launch pipelined step goes: graphs of 200 launches of `pipe_step_kernel` restricted to
subsets of its workgroup roles (results of such launches are meaningless; only the time is read).

    python tools/pipe_probe.py [launches per graph] [repeats]

PIPE_PROBE_HIDDEN / PIPE_PROBE_BATCH: another shape than the reference's 100 / 100.
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

L = ops._lib
ROLES = {"L": L.PIPE_ROLE_L, "Q": L.PIPE_ROLE_Q, "G": L.PIPE_ROLE_G, "W": L.PIPE_ROLE_W, "T": L.PIPE_ROLE_T, "F": L.PIPE_ROLE_F}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
    hidden, batch = int(os.environ.get("PIPE_PROBE_HIDDEN", 100)), int(os.environ.get("PIPE_PROBE_BATCH", 100))
    st = build_mlp_state(784, [hidden], 10, dev, lr=0.01)
    st.initialize(seed=1)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), batch, [hidden])
    print(f"hidden {hidden}, batch {batch}")
    assert prog.pipe
    prog.run(16, 8)
    torch.cuda.synchronize()
    lib = L.kernels()
    subsets = (os.environ.get("PIPE_PROBE_SUBSETS") or "L,Q,G,W,T,LT,LW,LQ,LG,LWQ,QGW,LQGWT").split(",")
    print(f"{n} launches per graph, median of {reps} replays, us per launch")
    for names in subsets:
        mask = 0
        for c in names:
            mask |= ROLES[c]
        packs = []
        for q in range(prog.P):
            a = L.PipeArgs.from_buffer_copy(prog._args[q]["pipe_step"])
            a.roles = mask
            packs.append(a)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                rc = lib.dtf_pipe_launch_roles(C.byref(packs[i % prog.P]), L.stream_handle())
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
        print(f"  roles {names:6s} {ts[len(ts) // 2] / n * 1e6:7.2f}")


if __name__ == "__main__":
    main()

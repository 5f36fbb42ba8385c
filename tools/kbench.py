#!/usr/bin/env python3
"""Per-kernel latency of the reference MLP step (each kernel replayed from its own hipGraph)."""
import argparse
import sys, os
os.environ.setdefault("DTF_FUSED_STEP", "0")      # this tool times the kernels of the general three-launch step
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import ops
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.device_dataset import DeviceDataset
from distributed_tensorflow_amd.models.mlp import MLPProgram, build_mlp_state


def timeit(fn, reps=200, inner=20):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=100)
    ap.add_argument("--post", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda")
    ds = mnist.synthetic_data_sets(train_size=55000, validation_size=100, test_size=100, seed=0)
    st = build_mlp_state(784, [a.hidden], 10, dev, lr=0.01)
    st.initialize(seed=1)
    prog = MLPProgram(st, DeviceDataset(ds.train, dev, shuffle_seed=0), a.batch, [a.hidden], post_update_accuracy=bool(a.post))
    prog.start()
    prog.run(4)
    torch.cuda.synchronize()
    q = 0
    rows = prog.ridx[q]
    R, B = prog.R, prog.B

    def fwd():
        ops.linear_fwd(prog.data.images, st.var("hid_w"), st.var("hid_b"), relu=True, rows=rows, out=prog.hact[0], m=R)

    def head():
        ops.head(prog.hact[0], st.var("sm_w"), st.var("sm_b"), prog.data.labels, R, B, args=prog._args[q]["head"])

    def wgrad():
        ops.wgrad_grouped(prog._args[q]["items"], args=prog._args[q]["wgrad"])

    notail = ops.build_wgrad_args(prog._args[q]["items"], lr=st.lr, tail=None)

    def wgrad_notail():
        ops.wgrad_grouped(prog._args[q]["items"], args=notail)

    store = ops.build_wgrad_args(prog._args[q]["items"], lr=None, tail=None)

    def wgrad_store():
        ops.wgrad_grouped(prog._args[q]["items"], args=store)

    tail = prog._tail(q)

    def tail_only():
        ops.step_tail(tail, init_only=False)

    def sgd():
        ops.sgd_apply(st.params, st.grads, 0.01, 1.0)

    def empty():
        ops.accumulate(st.grads[:4], st.grads[4:8], 1.0, 0.0)

    def step():
        fwd(); head(); wgrad()

    import time
    prog.use_graphs = False
    prog.run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); prog.run(2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"eager step: host enqueue {(t1 - t0) / 2000 * 1e6:.1f} us/step, end-to-end {(t2 - t0) / 2000 * 1e6:.1f} us/step")
    for name, fn in [("empty(4 elems)", empty), ("fwd", fwd), ("head", head), ("wgrad+sgd+tail", wgrad),
                     ("wgrad+sgd", wgrad_notail), ("wgrad store", wgrad_store), ("tail only", tail_only),
                     ("sgd flat 79.5k", sgd), ("full step", step)]:
        print(f"{name:20s} {timeit(fn):8.2f} us")


if __name__ == "__main__":
    main()

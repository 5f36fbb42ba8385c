"""The algebra of the one-launch pipelined step for N synchronous workers (csrc/kernels/mlp_pipe_multi.hip), restated in
NumPy with the kernel's buffers, generations and exchange, against plain mean-gradient SGD - what
``SyncReplicasOptimizer`` with ``replicas_to_aggregate == num_workers`` applies (/root/reference/distributed.py:91-102).

Worker r trains on its own batch X_r(s) (no data sharding, :137); an update applies the mean over the workers:

    W(s+1) = W(s) - lr/N sum_r X'_r(s)^T dz1_r(s)                       (X' = [X, 1]: b1 is the last row of W)

A worker never waits for W to be materialised.  What it exchanges with its peers is dz1 of its rows (and its share of
the output layer's gradient), and the products with the updates it has not folded into its working copy yet are
Gram-matrix corrections over every worker's batch (every worker can gather every batch: same data set, known shuffles):

    z1_r(s) = X'_r(s) W(s-2)  -  lr/N sum_r' G_rr'(s,s-2) dz1_r'(s-2)  -  lr/N sum_r' G_rr'(s,s-1) dz1_r'(s-1)
              `---- Q_r(s): role Q of launch s-1 ----'   `-- complete when launch s starts --'   `-- arrives during launch s --'

Launch s only ever reads dz1(s-1) of the PEERS in the row workgroups (which wait for it); the weight update and the Q
products of the same launch use dz1(s-2), which every peer delivered before launch s-1 ended.  The test runs the N
workers in lockstep, in float64, over several runs (prime ... flush) and worlds 1, 2, 3, 8."""
import numpy as np
import pytest


def _aug(x):
    return np.concatenate([x, np.ones((x.shape[0], 1))], 1)


def _softmax(z):
    e = np.exp(z - z.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True)


def _head(h, w2a, y_, batch):
    z2 = _aug(h) @ w2a
    y = _softmax(z2)
    p = _softmax(y)                                    # the reference applies softmax twice
    loss = -(y_ * np.log(p)).sum(1).mean()
    dy = (p - y_) / batch
    dz2 = (dy - (dy * y).sum(1, keepdims=True)) * y
    return loss, dz2


K0, K1, K2, K3 = range(4)        # Gram kinds written by launch s: G(s,s), G(s+1,s), G(s+1,s-1), G(s+2,s-1)


@pytest.mark.parametrize("N", [1, 2, 3, 8])
def test_multi_worker_pipelined_schedule_reproduces_mean_gradient_sgd(N):
    rng = np.random.default_rng(N)
    B, In, Hd, C, lr, steps = 12, 20, 9, 10, 0.8, 13
    X = rng.random((N, steps + 8, B, In))
    Y = np.eye(C)[rng.integers(0, C, (N, steps + 8, B))]
    w1a0 = np.vstack([rng.normal(0, .2, (In, Hd)), np.zeros(Hd)])
    w2a0 = np.vstack([rng.normal(0, .3, (Hd, C)), np.zeros(C)])
    xs = lambda r, s: _aug(X[r, s + 3])                # batch of worker r at step s (steps -3.. exist)
    ys = lambda r, s: Y[r, s + 3]
    lrn = lr / N

    # ---- plain synchronous SGD: every worker's gradient on the same parameters, mean applied once ----
    wa, w2a, plain = w1a0, w2a0, []
    for s in range(steps):
        gw, gw2, rec = 0.0, 0.0, []
        for r in range(N):
            z1 = xs(r, s) @ wa
            h = np.maximum(z1, 0)
            loss, dz2 = _head(h, w2a, ys(r, s), B)
            post = (np.argmax(_aug(np.maximum(xs(r, s - 1) @ wa, 0)) @ w2a, 1) == np.argmax(ys(r, s - 1), 1)).sum()
            dz1 = (dz2 @ w2a[:Hd].T) * (z1 > 0)
            gw, gw2 = gw + xs(r, s).T @ dz1, gw2 + _aug(h).T @ dz2
            rec.append((loss, post))
        wa, w2a = wa - lrn * gw, w2a - lrn * gw2
        plain.append(rec)

    class Worker:
        """Buffers of one rank.  `box_d[g][src]`, `box_p[g][src]`: what rank `src` delivered for the step with
        ring index g (= step & 3)."""

        def __init__(self):
            self.box_d = [[np.zeros((B, Hd)) for _ in range(N)] for _ in range(4)]
            self.box_p = [[np.zeros((Hd + 1, C)) for _ in range(N)] for _ in range(4)]
            self.gg = [[[np.full((B, B), 5.0) for _ in range(N)] for _ in range(4)] for _ in range(2)]    # finite garbage
            self.q, self.z1k, self.wm, self.w2m = [None, None], [None, None], [None, None], [None, None]

    def run(f, n, wa_flat, w2_flat, workers, out):
        """One run(): prime on every rank (+ barrier), n step launches, flush.  The flat parameters are identical on
        every rank before and after."""
        for w in workers:
            w.wm = [wa_flat.copy(), wa_flat.copy()]
            w.w2m = [w2_flat.copy(), w2_flat.copy()]
            for g in range(4):
                for src in range(N):
                    w.box_d[g][src][:] = 0.0
                    w.box_p[g][src][:] = 0.0
        for r, w in enumerate(workers):
            w.q[f & 1], w.z1k[(f - 1) & 1] = xs(r, f) @ wa_flat, xs(r, f - 1) @ wa_flat
        for s in range(f, f + n):
            a, p, g0, g1, g2 = s & 1, (s - 1) & 1, s & 3, (s - 1) & 3, (s - 2) & 3
            sends, rec = [], []
            for r, w in enumerate(workers):
                # role L: waits until every rank's dz1 / output-layer share of step s-1 sits in its boxes
                w2s = w.w2m[p] - lrn * sum(w.box_p[g1][src] for src in range(N))
                z1 = (w.q[a] - lrn * sum(w.gg[p][K2][src] @ w.box_d[g2][src] for src in range(N))
                      - lrn * sum(w.gg[p][K1][src] @ w.box_d[g1][src] for src in range(N)))
                h = np.maximum(z1, 0)
                loss, dz2 = _head(h, w2s, ys(r, s), B)
                zp = w.z1k[p] - lrn * sum(w.gg[p][K0][src] @ w.box_d[g1][src] for src in range(N))
                post = (np.argmax(_aug(np.maximum(zp, 0)) @ w2s, 1) == np.argmax(ys(r, s - 1), 1)).sum()
                sends.append(((dz2 @ w2s[:Hd].T) * (z1 > 0), _aug(h).T @ dz2))
                # roles Q, W, G: nothing newer than step s-2 from the peers
                new_q = xs(r, s + 1) @ w.wm[p] - lrn * sum(w.gg[p][K3][src] @ w.box_d[g2][src] for src in range(N))
                new_w = w.wm[p] - lrn * sum(xs(src, s - 2).T @ w.box_d[g2][src] for src in range(N))
                new_g = [[xs(r, s) @ xs(src, s).T for src in range(N)],
                         [xs(r, s + 1) @ xs(src, s).T for src in range(N)],
                         [xs(r, s + 1) @ xs(src, s - 1).T for src in range(N)],
                         [xs(r, s + 2) @ xs(src, s - 1).T for src in range(N)]]
                w.q[p], w.wm[a], w.gg[a], w.w2m[a], w.z1k[a] = new_q, new_w, new_g, w2s, z1
                rec.append((loss, post))
            # the exchange: every rank's results into every rank's boxes of ring index s & 3 (its own included)
            for src, (d1, p2) in enumerate(sends):
                for w in workers:
                    w.box_d[g0][src], w.box_p[g0][src] = d1.copy(), p2.copy()
            out.append(rec)
        e = f + n - 1
        flat = []
        for w in workers:                              # flush: the two updates the working copy has not seen
            wa_new = w.wm[e & 1] - lrn * sum(xs(src, e - 1).T @ w.box_d[(e - 1) & 3][src] for src in range(N)) \
                - lrn * sum(xs(src, e).T @ w.box_d[e & 3][src] for src in range(N))
            flat.append((wa_new, w.w2m[e & 1] - lrn * sum(w.box_p[e & 3][src] for src in range(N))))
        for wa_new, w2_new in flat[1:]:
            assert (wa_new == flat[0][0]).all() and (w2_new == flat[0][1]).all()       # replicas: the same sums, bit for bit
        return flat[0]

    workers = [Worker() for _ in range(N)]
    pipe, wa_p, w2_p, f = [], w1a0, w2a0, 0
    for n in (1, 4, 2, 6):
        wa_p, w2_p = run(f, n, wa_p, w2_p, workers, pipe)
        f += n
    assert f == steps
    assert np.abs(wa_p - wa).max() < 1e-12 and np.abs(w2_p - w2a).max() < 1e-12
    for s in range(steps):
        for r in range(N):
            assert abs(plain[s][r][0] - pipe[s][r][0]) < 1e-12
            assert plain[s][r][1] == pipe[s][r][1]


def test_a_run_that_continues_the_pipeline_equals_one_longer_run():
    """Flush leaves working copies, boxes and pending products alone: two runs without a prime in between are one run."""
    rng = np.random.default_rng(5)
    N, B, In, Hd, C, lr = 2, 8, 10, 5, 4, 0.5
    X = rng.random((N, 20, B, In))
    Y = np.eye(C)[rng.integers(0, C, (N, 20, B))]
    lrn = lr / N
    xs = lambda r, s: _aug(X[r, s + 3])

    def simulate(cuts):
        wa = np.vstack([rng0.normal(0, .2, (In, Hd)), np.zeros(Hd)])
        w2 = np.vstack([rng0.normal(0, .3, (Hd, C)), np.zeros(C)])
        wm, box = [wa.copy(), wa.copy()], [[np.zeros((B, Hd)) for _ in range(N)] for _ in range(4)]
        flats = []
        s = 0
        for n in cuts:
            for _ in range(n):
                # only the weight path matters here: dz1 stands in as a fixed function of the step
                wm[s & 1] = wm[(s - 1) & 1] - lrn * sum(xs(src, s - 2).T @ box[(s - 2) & 3][src] for src in range(N))
                for src in range(N):
                    box[s & 3][src] = np.sin(np.arange(B * Hd).reshape(B, Hd) * (1 + s + src))
                s += 1
            e = s - 1
            flats.append(wm[e & 1] - lrn * sum(xs(src, e - 1).T @ box[(e - 1) & 3][src] for src in range(N))
                         - lrn * sum(xs(src, e).T @ box[e & 3][src] for src in range(N)))
        return flats[-1]

    rng0 = np.random.default_rng(9)
    one = simulate([9])
    rng0 = np.random.default_rng(9)
    cut = simulate([2, 3, 4])
    assert (one == cut).all()

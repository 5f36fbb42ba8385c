"""The algebra of the one-launch pipelined step (csrc/kernels/mlp_pipe.hip), restated in NumPy with the kernel's
buffers and generations, against plain SGD on the reference network (/root/reference/distributed.py:78-89).

    W1'(s) = W1'(s-2) - lr X'(s-2)^T dz1(s-2) - lr X'(s-1)^T dz1(s-1)
    z1(s)  = [X'(s) W1'(s-2) - lr G(s,s-2) dz1(s-2)] - lr G(s,s-1) dz1(s-1),   G(a,b) = X'(a) X'(b)^T

In float64 the two must agree to rounding of the last bits, over several runs (prime ... flush) of different lengths,
with arbitrary finite garbage in the Gram buffers at the start of a run."""
import numpy as np


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


def test_pipelined_schedule_reproduces_plain_sgd():
    rng = np.random.default_rng(0)
    B, In, Hd, C, lr, N = 20, 30, 12, 10, 0.5, 14
    X = rng.random((N + 6, B, In))
    Y = np.eye(C)[rng.integers(0, C, (N + 6, B))]
    w1a0 = np.vstack([rng.normal(0, .2, (In, Hd)), np.zeros(Hd)])
    w2a0 = np.vstack([rng.normal(0, .3, (Hd, C)), np.zeros(C)])
    xs = lambda s: _aug(X[s + 1])                      # batch of step s (index 0: "previous batch" of step 0)

    # plain SGD, with the reference's post-update accuracy of the previous batch
    wa, w2a, plain = w1a0, w2a0, []
    for s in range(N):
        z1 = xs(s) @ wa
        h = np.maximum(z1, 0)
        loss, dz2 = _head(h, w2a, Y[s + 1], B)
        post = (np.argmax(_aug(np.maximum(xs(s - 1) @ wa, 0)) @ w2a, 1) == np.argmax(Y[s], 1)).sum()
        dz1 = (dz2 @ w2a[:Hd].T) * (z1 > 0)
        wa, w2a = wa - lr * xs(s).T @ dz1, w2a - lr * _aug(h).T @ dz2
        plain.append((loss, post))

    def run(f, n, wa_flat, w2_flat, out):
        """One run(): prime, n step launches, flush.  Index [s & 1] = generation of step s."""
        wm = [wa_flat.copy(), wa_flat.copy()]
        w2m = [w2_flat.copy(), w2_flat.copy()]
        d1 = [np.zeros((B, Hd)), np.zeros((B, Hd))]
        p2 = [np.zeros((Hd + 1, C)), np.zeros((Hd + 1, C))]
        gg = [[np.full((B, B), 7.0)] * 3, [np.full((B, B), -3.0)] * 3]      # finite garbage
        q, z1k = [None, None], [None, None]
        # prime: Q of the first step and z1 of the batch before it, from the flat parameters
        q[f & 1], z1k[(f - 1) & 1] = xs(f) @ wa_flat, xs(f - 1) @ wa_flat
        for s in range(f, f + n):
            a, p = s & 1, (s - 1) & 1
            # role L (reads generation p, writes generation a)
            w2s = w2m[p] - lr * p2[p]
            z1 = q[a] - lr * gg[p][1] @ d1[p]
            h = np.maximum(z1, 0)
            loss, dz2 = _head(h, w2s, Y[s + 1], B)
            zp = z1k[p] - lr * gg[p][0] @ d1[p]
            post = (np.argmax(_aug(np.maximum(zp, 0)) @ w2s, 1) == np.argmax(Y[s], 1)).sum()
            new_d1, new_p2 = (dz2 @ w2s[:Hd].T) * (z1 > 0), _aug(h).T @ dz2
            # roles Q, W, G (read generation p)
            q[p] = xs(s + 1) @ wm[p] - lr * gg[p][2] @ d1[p]
            new_w = wm[p] - lr * xs(s - 1).T @ d1[p]
            new_g = [xs(s) @ xs(s).T, xs(s + 1) @ xs(s).T, xs(s + 2) @ xs(s).T]
            wm[a], gg[a], d1[a], p2[a], w2m[a], z1k[a] = new_w, new_g, new_d1, new_p2, w2s, z1
            out.append((loss, post))
        p = (f + n - 1) & 1
        return wm[p] - lr * xs(f + n - 1).T @ d1[p], w2m[p] - lr * p2[p]      # flush

    pipe, wa_p, w2_p, f = [], w1a0, w2a0, 0
    for n in (1, 5, 2, 6):
        wa_p, w2_p = run(f, n, wa_p, w2_p, pipe)
        f += n
    assert f == N
    assert np.abs(wa_p - wa).max() < 1e-13 and np.abs(w2_p - w2a).max() < 1e-13
    assert max(abs(x[0] - y[0]) for x, y in zip(plain, pipe)) < 1e-13
    assert [x[1] for x in plain] == [y[1] for y in pipe]

"""NumPy oracle: the math of /root/reference/distributed.py:78-89 written out by hand.

Forward: xw_plus_b -> relu -> xw_plus_b -> softmax (:78-81); loss =
mean(softmax_cross_entropy_with_logits(labels, logits=y)) where y is ALREADY a
softmax output (:86-87), i.e. softmax is applied twice; accuracy (:83-84);
plain SGD (:89).  Backward is the chain rule through exactly those ops.
Everything is float64 so the oracle is strictly more accurate than any fp32
implementation under test.
"""
import numpy as np


def softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def forward(params, x):
    z1 = x @ params["hid_w"] + params["hid_b"]
    h = np.maximum(z1, 0.0)
    z2 = h @ params["sm_w"] + params["sm_b"]
    y = softmax(z2)
    return z1, h, z2, y


def loss_and_grads(params, x, labels, loss="reference"):
    B = x.shape[0]
    z1, h, z2, y = forward(params, x)
    if loss == "reference":
        p = softmax(y)
        logp = np.log(p)
        per_example = -(labels * logp).sum(axis=1)
        dy = (p - labels) / B
        dz2 = (dy - (dy * y).sum(axis=1, keepdims=True)) * y
    else:
        logp = np.log(y)
        per_example = -(labels * logp).sum(axis=1)
        dz2 = (y - labels) / B
    grads = {"sm_b": dz2.sum(axis=0), "sm_w": h.T @ dz2}
    dh = dz2 @ params["sm_w"].T
    dz1 = dh * (z1 > 0)
    grads["hid_b"] = dz1.sum(axis=0)
    grads["hid_w"] = x.T @ dz1
    acc = float((y.argmax(1) == labels.argmax(1)).mean())
    return float(per_example.mean()), acc, grads


def accuracy(params, x, labels):
    return float((forward(params, x)[3].argmax(1) == labels.argmax(1)).mean())


def sgd(params, grads, lr):
    return {k: params[k] - lr * grads[k] for k in params}


def init_params(hidden=100, seed=0, d=784, c=10):
    rng = np.random.default_rng(seed)

    def tn(shape, std):
        v = rng.standard_normal(shape)
        bad = np.abs(v) > 2
        while bad.any():
            v[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2
        return v * std

    return {"hid_w": tn((d, hidden), 1.0 / 28), "hid_b": np.zeros(hidden), "sm_w": tn((hidden, c), 1.0 / np.sqrt(hidden)),
            "sm_b": np.zeros(c)}

"""MNIST input pipeline with the interface of
``tensorflow.examples.tutorials.mnist.input_data`` (used at
/root/reference/distributed.py:6, :38, :137, :141-142, :163-164).

``read_data_sets(data_dir, one_hot=True)`` returns ``.train`` / ``.validation``
/ ``.test`` splits of 55000 / 5000 / 10000 examples; images are float32 in
[0, 1] flattened to 784 columns; ``next_batch`` walks a fresh permutation every
epoch and, when an epoch ends inside a batch, completes the batch from the next
permutation.  The idx-gz files are decoded by the native runtime when it is
built (``csrc/runtime/idx_reader.cpp``), otherwise by NumPy.

``synthetic_data_sets`` builds a dataset of the same shapes from class
templates plus noise: learnable, so convergence tests mean something, and
needs no files (there is no network to download MNIST from).
"""
from __future__ import annotations

import gzip
import os
import struct
from typing import Optional

import numpy as np

TRAIN_IMAGES = "train-images-idx3-ubyte.gz"
TRAIN_LABELS = "train-labels-idx1-ubyte.gz"
TEST_IMAGES = "t10k-images-idx3-ubyte.gz"
TEST_LABELS = "t10k-labels-idx1-ubyte.gz"
VALIDATION_SIZE = 5000
NUM_CLASSES = 10
IMAGE_PIXELS = 28  # distributed.py:35


def epoch_permutation(seed: int, epoch: int, n: int) -> np.ndarray:
    """Row order of one epoch.  Host ``next_batch`` and the device-resident
    batch selector both use this, so they walk identical sequences."""
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, int(epoch)])
    return rng.permutation(n).astype(np.int32)


def dense_to_one_hot(labels: np.ndarray, num_classes: int = NUM_CLASSES) -> np.ndarray:
    out = np.zeros((labels.shape[0], num_classes), dtype=np.float32)
    out[np.arange(labels.shape[0]), labels.astype(np.int64)] = 1.0
    return out


def _open_maybe_gz(path: str):
    if os.path.exists(path):
        return gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")
    alt = path[:-3] if path.endswith(".gz") else path + ".gz"
    if os.path.exists(alt):
        return gzip.open(alt, "rb") if alt.endswith(".gz") else open(alt, "rb")
    raise FileNotFoundError(f"MNIST file not found: {path} (there is no download fallback; pass --synthetic_data)")


def read_idx(path: str) -> np.ndarray:
    """Decode one idx file (optionally gzip-compressed) into a uint8 array."""
    from ..utils import native
    if native.available():
        real = path if os.path.exists(path) else (path[:-3] if path.endswith(".gz") else path + ".gz")
        if not os.path.exists(real):
            raise FileNotFoundError(f"MNIST file not found: {path} (there is no download fallback; pass --synthetic_data)")
        return native.read_idx(real)
    with _open_maybe_gz(path) as f:
        data = f.read()
    zero, dtype_code, ndim = struct.unpack(">HBB", data[:4])
    if zero != 0 or dtype_code != 0x08:
        raise ValueError(f"{path}: not an unsigned-byte idx file (magic {data[:4]!r})")
    dims = struct.unpack(">" + "I" * ndim, data[4:4 + 4 * ndim])
    arr = np.frombuffer(data, dtype=np.uint8, offset=4 + 4 * ndim)
    if arr.size != int(np.prod(dims)):
        raise ValueError(f"{path}: payload size {arr.size} does not match dims {dims}")
    return arr.reshape(dims)


class DataSet:
    """One split; ``next_batch`` has the TF tutorial semantics."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, seed: int = 0):
        assert images.shape[0] == labels.shape[0], (images.shape, labels.shape)
        self._images = np.ascontiguousarray(images, dtype=np.float32)
        self._labels = np.ascontiguousarray(labels, dtype=np.float32)
        self._n = images.shape[0]
        self._seed = seed
        self._position = 0            # absolute sample counter = epochs_completed * n + index_in_epoch
        self._perm_cache = {}

    @property
    def images(self):
        return self._images

    @property
    def labels(self):
        return self._labels

    @property
    def num_examples(self):
        return self._n

    @property
    def epochs_completed(self):
        return self._position // self._n

    @property
    def seed(self):
        return self._seed

    def _perm(self, epoch: int) -> np.ndarray:
        p = self._perm_cache.get(epoch)
        if p is None:
            p = epoch_permutation(self._seed, epoch, self._n)
            self._perm_cache = {k: v for k, v in self._perm_cache.items() if k >= epoch - 1}
            self._perm_cache[epoch] = p
        return p

    def batch_rows(self, start: int, batch_size: int) -> np.ndarray:
        pos = np.arange(start, start + batch_size, dtype=np.int64)
        epochs, offs = pos // self._n, pos % self._n
        rows = np.empty(batch_size, dtype=np.int32)
        for e in np.unique(epochs):
            m = epochs == e
            rows[m] = self._perm(int(e))[offs[m]]
        return rows

    def next_batch(self, batch_size: int):
        rows = self.batch_rows(self._position, batch_size)
        self._position += batch_size
        return self._images[rows], self._labels[rows]


class Datasets:
    def __init__(self, train: DataSet, validation: DataSet, test: DataSet):
        self.train, self.validation, self.test = train, validation, test


def read_data_sets(data_dir: str, one_hot: bool = True, validation_size: int = VALIDATION_SIZE, seed: int = 0) -> Datasets:
    def load(images_name, labels_name):
        imgs = read_idx(os.path.join(data_dir, images_name))
        labs = read_idx(os.path.join(data_dir, labels_name))
        imgs = imgs.reshape(imgs.shape[0], -1).astype(np.float32) * (1.0 / 255.0)
        labs = dense_to_one_hot(labs) if one_hot else labs.astype(np.float32)[:, None]
        return imgs, labs

    tr_x, tr_y = load(TRAIN_IMAGES, TRAIN_LABELS)
    te_x, te_y = load(TEST_IMAGES, TEST_LABELS)
    if not 0 <= validation_size <= tr_x.shape[0]:
        raise ValueError(f"validation_size must be in [0, {tr_x.shape[0]}]")
    return Datasets(DataSet(tr_x[validation_size:], tr_y[validation_size:], seed),
                    DataSet(tr_x[:validation_size], tr_y[:validation_size], seed),
                    DataSet(te_x, te_y, seed))


def synthetic_data_sets(train_size: int = 55000, validation_size: int = 5000, test_size: int = 10000,
                        num_classes: int = NUM_CLASSES, pixels: int = IMAGE_PIXELS * IMAGE_PIXELS, seed: int = 0,
                        noise: float = 0.35, data_seed: int = 1234) -> Datasets:
    """MNIST-shaped synthetic data: per-class smooth templates + uniform noise, clipped to [0, 1]."""
    rng = np.random.default_rng(data_seed)
    templates = (rng.random((num_classes, pixels), dtype=np.float32) < 0.25).astype(np.float32) * 0.9

    def make(n):
        cls = rng.integers(0, num_classes, size=n)
        x = templates[cls] + (rng.random((n, pixels), dtype=np.float32) - 0.5) * (2 * noise)
        np.clip(x, 0.0, 1.0, out=x)
        return x.astype(np.float32), dense_to_one_hot(cls, num_classes)

    tr, va, te = make(train_size), make(validation_size), make(test_size)
    return Datasets(DataSet(*tr, seed=seed), DataSet(*va, seed=seed), DataSet(*te, seed=seed))


def load(data_dir: Optional[str], synthetic: bool, seed: int = 0, synthetic_train_size: int = 55000) -> Datasets:
    if synthetic:
        return synthetic_data_sets(train_size=synthetic_train_size, seed=seed)
    return read_data_sets(data_dir, one_hot=True, seed=seed)

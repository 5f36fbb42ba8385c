"""MNIST idx pipeline (native reader vs NumPy), split sizes, next_batch semantics, synthetic data."""
import gzip
import os
import struct

import numpy as np
import pytest

from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.utils import native


def _write_idx(path, arr, compress=True):
    header = struct.pack(">HBB", 0, 0x08, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
    data = header + arr.astype(np.uint8).tobytes()
    if compress:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        open(path, "wb").write(data)


@pytest.fixture
def fake_mnist(tmp_path):
    rng = np.random.default_rng(0)
    tr_x = rng.integers(0, 256, size=(6000, 28, 28), dtype=np.uint8)
    tr_y = rng.integers(0, 10, size=(6000,), dtype=np.uint8)
    te_x = rng.integers(0, 256, size=(1000, 28, 28), dtype=np.uint8)
    te_y = rng.integers(0, 10, size=(1000,), dtype=np.uint8)
    _write_idx(tmp_path / mnist.TRAIN_IMAGES, tr_x)
    _write_idx(tmp_path / mnist.TRAIN_LABELS, tr_y)
    _write_idx(tmp_path / mnist.TEST_IMAGES, te_x)
    _write_idx(tmp_path / mnist.TEST_LABELS[:-3], te_y, compress=False)      # also accepts un-gzipped files
    return tmp_path, tr_x, tr_y, te_x, te_y


def test_native_idx_reader_matches_numpy(fake_mnist):
    d, tr_x, tr_y, _, _ = fake_mnist
    assert native.available()
    np.testing.assert_array_equal(native.read_idx(str(d / mnist.TRAIN_IMAGES)), tr_x)
    np.testing.assert_allclose(native.read_images_f32(str(d / mnist.TRAIN_IMAGES)), tr_x.reshape(6000, 784) / 255.0, rtol=1e-6)
    np.testing.assert_array_equal(native.read_labels_one_hot(str(d / mnist.TRAIN_LABELS)), mnist.dense_to_one_hot(tr_y))
    rows = np.array([5, 0, 5999, 17], dtype=np.int32)
    x = tr_x.reshape(6000, 784).astype(np.float32)
    np.testing.assert_array_equal(native.gather_rows(x, rows), x[rows])
    with pytest.raises(IndexError):
        native.gather_rows(x, np.array([6000], dtype=np.int32))
    bad = d / "bad.idx"
    bad.write_bytes(b"\x00\x00\x0d\x01\x00\x00\x00\x01\x00")
    with pytest.raises(ValueError):
        native.read_idx(str(bad))


def test_read_data_sets_splits_and_scaling(fake_mnist):
    d, tr_x, tr_y, te_x, te_y = fake_mnist
    ds = mnist.read_data_sets(str(d), one_hot=True, validation_size=500)
    assert ds.train.num_examples == 5500 and ds.validation.num_examples == 500 and ds.test.num_examples == 1000
    assert ds.train.images.dtype == np.float32 and ds.train.images.shape == (5500, 784)
    assert 0.0 <= ds.train.images.min() and ds.train.images.max() <= 1.0
    np.testing.assert_allclose(ds.validation.images, tr_x[:500].reshape(500, 784) / 255.0, rtol=1e-6)   # first rows = validation
    np.testing.assert_array_equal(ds.test.labels.argmax(1), te_y)
    assert ds.train.labels.shape == (5500, 10) and np.all(ds.train.labels.sum(1) == 1)
    with pytest.raises(FileNotFoundError):
        mnist.read_data_sets(str(d / "missing"))


def test_next_batch_epoch_semantics():
    ds = mnist.synthetic_data_sets(train_size=250, validation_size=10, test_size=10, seed=4)
    seen = []
    for _ in range(5):                                  # 5 x 100 = two full epochs of 250
        x, y = ds.train.next_batch(100)
        assert x.shape == (100, 784) and y.shape == (100, 10)
        seen.append(ds.train.batch_rows(len(seen) * 100, 100))
    rows = np.concatenate(seen)
    assert sorted(rows[:250]) == list(range(250)) and sorted(rows[250:]) == list(range(250))   # each epoch is a permutation
    assert not np.array_equal(rows[:250], rows[250:])                                          # reshuffled every epoch
    assert ds.train.epochs_completed == 2
    other = mnist.synthetic_data_sets(train_size=250, validation_size=10, test_size=10, seed=5)
    assert not np.array_equal(other.train.batch_rows(0, 100), seen[0])                         # per-worker shuffle


def test_synthetic_is_learnable_shape():
    ds = mnist.synthetic_data_sets(train_size=100, validation_size=20, test_size=30)
    assert ds.train.images.shape == (100, 784) and ds.validation.labels.shape == (20, 10) and ds.test.images.shape == (30, 784)
    assert ds.train.images.min() >= 0 and ds.train.images.max() <= 1

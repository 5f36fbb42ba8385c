"""TF-Saver-V2 bundle layout: native C++ writer/reader cross-checked against an independent
pure-Python reader/writer of the same format.  No TensorFlow is installed and the reference
tree ships no checkpoint file, so byte-parity with a real TF Saver is *unpinned*; what is
pinned: CRC32C test vectors, table structure (footer magic, block trailers, restart arrays,
prefix compression), proto field numbers, and the key order / data-file order."""
import os
import struct

import numpy as np
import pytest

from distributed_tensorflow_amd.checkpoint import bundle, pyformat, saver


def _tensors(hidden=100):
    rng = np.random.default_rng(0)
    return {"global_step": np.array(1234, dtype=np.int32),
            "hid_w": rng.standard_normal((784, hidden)).astype(np.float32),
            "hid_b": rng.standard_normal(hidden).astype(np.float32),
            "sm_w": rng.standard_normal((hidden, 10)).astype(np.float32),
            "sm_b": rng.standard_normal(10).astype(np.float32)}


def test_crc32c_known_vectors():
    # RFC 3720 B.4 test vectors
    assert bundle.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert bundle.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert bundle.crc32c(bytes(range(32))) == 0x46DD794E
    assert bundle.crc32c(b"123456789") == 0xE3069283
    assert pyformat.crc32c(b"123456789") == 0xE3069283
    for data in (b"", b"a", b"hello world" * 37):
        assert bundle.crc32c(data) == pyformat.crc32c(data)
        assert bundle.crc32c_masked(data) == pyformat.mask(pyformat.crc32c(data))


def test_native_writer_read_by_python_reader(tmp_path):
    t = _tensors()
    prefix = str(tmp_path / "model.ckpt-1234")
    bundle.write_bundle(prefix, t)
    assert sorted(os.listdir(tmp_path)) == ["model.ckpt-1234.data-00000-of-00001", "model.ckpt-1234.index"]
    header, entries = pyformat.read_index(prefix + ".index")
    assert header == {"num_shards": 1, "endianness": 0, "producer": 1}
    assert list(entries) == ["global_step", "hid_b", "hid_w", "sm_b", "sm_w"]          # sorted keys
    assert entries["global_step"]["dtype"] == 3 and entries["global_step"]["shape"] == () and entries["global_step"]["size"] == 4
    assert entries["hid_w"]["dtype"] == 1 and entries["hid_w"]["shape"] == (784, 100)
    off = 0
    for name in entries:                                                               # data file = key order, no padding
        assert entries[name]["offset"] == off
        off += entries[name]["size"]
    assert os.path.getsize(prefix + ".data-00000-of-00001") == off == 4 + 4 * 79510
    got = pyformat.read_bundle(prefix)
    for k in t:
        np.testing.assert_array_equal(got[k], t[k])
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57


def test_python_writer_read_by_native_reader_and_bytes_agree(tmp_path):
    t = _tensors(hidden=7)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    pyformat.write_bundle(a, t)
    bundle.write_bundle(b, t)
    got = bundle.read_bundle(a)
    for k in t:
        np.testing.assert_array_equal(got[k], t[k])
    # two independent writers of the same format description produce identical files
    assert open(a + ".index", "rb").read() == open(b + ".index", "rb").read()
    assert open(a + ".data-00000-of-00001", "rb").read() == open(b + ".data-00000-of-00001", "rb").read()


def test_many_entries_span_restart_points_and_blocks(tmp_path):
    rng = np.random.default_rng(1)
    t = {f"layer_{i:04d}/weights": rng.standard_normal((3, i % 5 + 1)).astype(np.float32) for i in range(300)}
    t["big"] = rng.standard_normal(100000).astype(np.float32)
    prefix = str(tmp_path / "many")
    bundle.write_bundle(prefix, t)
    for reader in (bundle.read_bundle, pyformat.read_bundle):
        got = reader(prefix)
        assert set(got) == set(t)
        for k in t:
            np.testing.assert_array_equal(got[k], t[k])


def test_corruption_is_detected(tmp_path):
    t = _tensors(hidden=5)
    prefix = str(tmp_path / "c")
    bundle.write_bundle(prefix, t)
    path = prefix + ".data-00000-of-00001"
    raw = bytearray(open(path, "rb").read())
    raw[100] ^= 0x40
    open(path, "wb").write(bytes(raw))
    with pytest.raises(OSError, match="checksum|read failed"):
        bundle.read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(OSError):
        bundle.read_bundle(prefix)


def test_saver_state_file_and_retention(tmp_path):
    s = saver.Saver(max_to_keep=5)
    d = str(tmp_path)
    for step in range(1, 8):
        t = _tensors(hidden=3)
        t["global_step"] = np.array(step * 10, dtype=np.int32)
        s.save(t, d, step * 10)
    st = saver.read_checkpoint_state(d)
    assert st["model_checkpoint_path"] == "model.ckpt-70"
    assert st["all_model_checkpoint_paths"] == [f"model.ckpt-{i}" for i in (30, 40, 50, 60, 70)]
    text = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert text[0] == 'model_checkpoint_path: "model.ckpt-70"'
    assert text[1] == 'all_model_checkpoint_paths: "model.ckpt-30"'
    assert not os.path.exists(os.path.join(d, "model.ckpt-10.index"))
    latest = saver.latest_checkpoint(d)
    assert latest.endswith("model.ckpt-70")
    got = saver.Saver.restore(latest)
    assert int(got["global_step"]) == 70 and got["global_step"].dtype == np.int32
    assert saver.latest_checkpoint(str(tmp_path / "nothing")) is None

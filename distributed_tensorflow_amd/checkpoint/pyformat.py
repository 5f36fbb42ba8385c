"""Pure-Python statement of the tensor-bundle / sorted-string-table format.

Independent of the C++ implementation: it exists so tests can cross-check the
native writer against a second reader (and the native reader against a second
writer) written separately from the same format description.  Not used on any
production path.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 9: np.dtype("<i8")}
_DT_REV = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("uint8"): 4, np.dtype("int64"): 9}


def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    r = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _block(buf: bytes, off: int, size: int) -> List[Tuple[bytes, bytes]]:
    body = buf[off:off + size]
    trailer = buf[off + size:off + size + 5]
    assert trailer[0] == 0, "compressed block"
    assert struct.unpack("<I", trailer[1:])[0] == mask(crc32c(body + trailer[:1])), "block crc mismatch"
    nrestart = struct.unpack("<I", body[-4:])[0]
    end = len(body) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(body, pos)
        non_shared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        out.append((key, body[pos:pos + vlen]))
        pos += vlen
    return out


def _parse_proto(buf: bytes) -> Dict[int, list]:
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
        elif wire == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack("<Q", buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError(f"wire type {wire}")
        out.setdefault(field, []).append(v)
    return out


def read_index(path: str) -> Tuple[dict, Dict[str, dict]]:
    buf = open(path, "rb").read()
    footer = buf[-48:]
    assert struct.unpack("<Q", footer[40:])[0] == MAGIC, "bad magic"
    pos = 0
    _, pos = _varint(footer, pos)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    header, entries = None, {}
    for _, handle in _block(buf, ioff, isize):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, val in _block(buf, off, size):
            msg = _parse_proto(val)
            if key == b"":
                header = {"num_shards": msg.get(1, [0])[0], "endianness": msg.get(2, [0])[0],
                          "producer": _parse_proto(msg[3][0]).get(1, [0])[0] if 3 in msg else None}
            else:
                dims = []
                for d in _parse_proto(msg[2][0]).get(2, []) if 2 in msg else []:
                    dims.append(_parse_proto(d).get(1, [0])[0])
                entries[key.decode()] = {"dtype": msg.get(1, [0])[0], "shape": tuple(dims), "shard_id": msg.get(3, [0])[0],
                                         "offset": msg.get(4, [0])[0], "size": msg.get(5, [0])[0], "crc32c": msg[6][0]}
    return header, entries


def read_bundle(prefix: str) -> Dict[str, np.ndarray]:
    header, entries = read_index(prefix + ".index")
    assert header is not None and header["num_shards"] == 1 and header["endianness"] == 0
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out = {}
    for name, e in entries.items():
        raw = data[e["offset"]:e["offset"] + e["size"]]
        assert mask(crc32c(raw)) == e["crc32c"], f"{name}: tensor crc mismatch"
        out[name] = np.frombuffer(raw, dtype=_DT[e["dtype"]]).reshape(e["shape"]).copy()
    return out


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Second, independent writer (single data block, which is all a small model needs)."""
    def field_varint(f, v):
        return _enc_varint(f << 3) + _enc_varint(v)

    def field_bytes(f, b):
        return _enc_varint(f << 3 | 2) + _enc_varint(len(b)) + b

    data = b""
    kv = [(b"", field_varint(1, 1) + field_bytes(3, field_varint(1, 1)))]
    for name in sorted(tensors):
        arr = np.asarray(tensors[name])
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        shape = b"".join(field_bytes(2, field_varint(1, d) if d else b"") for d in arr.shape)
        msg = field_varint(1, _DT_REV[arr.dtype]) + field_bytes(2, shape)
        if len(data):
            msg += field_varint(4, len(data))
        if len(raw):
            msg += field_varint(5, len(raw))
        msg += _enc_varint(6 << 3 | 5) + struct.pack("<I", mask(crc32c(raw)))
        kv.append((name.encode(), msg))
        data += raw

    def build_block(items, interval):
        body, restarts, last, count = b"", [0], b"", 0
        for k, v in items:
            shared = 0
            if count < interval:
                while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                    shared += 1
            else:
                restarts.append(len(body))
                count = 0
            body += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
            last, count = k, count + 1
        return body + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def emit(file, body):
        off = len(file)
        file += body + b"\x00" + struct.pack("<I", mask(crc32c(body + b"\x00")))
        return file, (off, len(body))

    file = b""
    file, dh = emit(file, build_block(kv, 16))
    file, mh = emit(file, build_block([], 16))
    last = bytearray(kv[-1][0])
    for i, b in enumerate(last):
        if b != 0xFF:
            last[i] = b + 1
            last = last[:i + 1]
            break
    file, ih = emit(file, build_block([(bytes(last), _enc_varint(dh[0]) + _enc_varint(dh[1]))], 1))
    footer = _enc_varint(mh[0]) + _enc_varint(mh[1]) + _enc_varint(ih[0]) + _enc_varint(ih[1])
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    open(prefix + ".index", "wb").write(file + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(data)

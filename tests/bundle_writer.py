"""Test-side WRITER of TensorFlow's TensorBundle format, written independently of nerf_tex_amd/checkpoint.py
(the reader under test) from the same published format descriptions.  Not product code."""

import struct

import numpy as np

from nerf_tex_amd.checkpoint import crc32c, mask_crc   # only the checksum primitive is shared


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F; n >>= 7
        if n: out.append(b | 0x80)
        else:
            out.append(b); return bytes(out)


def _field(num, wt, payload):
    return _vi((num << 3) | wt) + payload


def _entry_proto(dtype, shape, shard, offset, size, crc):
    dims = b"".join(_field(2, 2, (lambda d: _vi(len(d)) + d)(_field(1, 0, _vi(s)))) for s in shape)
    msg = _field(1, 0, _vi(dtype)) + _field(2, 2, _vi(len(dims)) + dims)
    if shard: msg += _field(3, 0, _vi(shard))
    msg += _field(4, 0, _vi(offset)) + _field(5, 0, _vi(size)) + _field(6, 5, struct.pack("<I", crc))
    return msg


def _block(items, restart_interval=16):
    buf, restarts, prev = bytearray(), [], b""
    for n, (k, v) in enumerate(items):
        if n % restart_interval == 0:
            restarts.append(len(buf)); shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]: shared += 1
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts: buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, tensors, block_bytes=512):
    """tensors: {name: np.ndarray}; small `block_bytes` forces several data blocks."""
    data, entries = bytearray(), {}
    for name in sorted(tensors, key=lambda s: s.encode()):
        a = np.ascontiguousarray(tensors[name])
        dt = {np.dtype("float32"): 1, np.dtype("int64"): 9, np.dtype("int32"): 3}[a.dtype]
        raw = a.tobytes()
        entries[name] = _entry_proto(dt, a.shape, 0, len(data), len(raw), mask_crc(crc32c(raw)))
        data += raw
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    header = _field(1, 0, _vi(1)) + _field(2, 0, _vi(0)) + _field(3, 2, (lambda d: _vi(len(d)) + d)(_field(1, 0, _vi(1))))
    items = [(b"", header)] + [(k.encode(), entries[k]) for k in sorted(entries, key=lambda s: s.encode())]
    out, index_items, cur, cur_size = bytearray(), [], [], 0

    def flush():
        nonlocal cur, cur_size
        if not cur: return
        blk = _block(cur)
        off = len(out)
        out.extend(blk + b"\x00" + struct.pack("<I", mask_crc(crc32c(blk + b"\x00"))))
        index_items.append((cur[-1][0] + b"\x00", _vi(off) + _vi(len(blk))))    # separator >= last key of the block
        cur, cur_size = [], 0

    for kv in items:
        cur.append(kv); cur_size += len(kv[0]) + len(kv[1])
        if cur_size >= block_bytes: flush()
    flush()
    meta = _block([])
    moff = len(out); out += meta + b"\x00" + struct.pack("<I", mask_crc(crc32c(meta + b"\x00")))
    idx = _block(index_items, restart_interval=1)
    ioff = len(out); out += idx + b"\x00" + struct.pack("<I", mask_crc(crc32c(idx + b"\x00")))
    footer = _vi(moff) + _vi(len(meta)) + _vi(ioff) + _vi(len(idx))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    out += footer
    open(prefix + ".index", "wb").write(bytes(out))

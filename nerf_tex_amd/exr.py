"""OpenEXR scanline images, the part of the format the Logger's `write_exr` option needs (`logger.py:139-142`: `pyexr.write(path, img)` of a
float32 [H, W, C] array; `dataset.py:99-101` / `data/nerf2tfr.py:47-49` read them back through pyexr) -- written and read without OpenEXR.

The format as published (OpenEXR "File Layout" / "Technical Introduction"; no OpenEXR library here to check a byte against: unpinned):
  magic 20000630, version 2 (single-part scanline, no flags) | header: attributes `name\\0 type\\0 size:int32 value`, ended by one \\0 |
  offset table: one uint64 per chunk | chunks: `y:int32 size:int32 data`.  A chunk holds 1 scanline (NO / ZIPS) or 16 (ZIP); its pixel data
  are, scanline by scanline, the channels in ALPHABETICAL order, each a row of `width` values (HALF 2 bytes, FLOAT 4, UINT 4).  ZIP / ZIPS:
  the chunk's bytes are split into even-indexed then odd-indexed bytes, delta-coded (`d[i] = b[i] - b[i-1] + 128`), deflated; a chunk that
  does not shrink is stored raw.
Written: float32 (or float16) channels, NO_COMPRESSION or ZIP, increasing y -- files any EXR reader opens; the pixels are pyexr's, the bytes
are not (pyexr's default compression is PIZ, which is not built here: reading a PIZ file is refused by name)."""

from __future__ import annotations

import struct
import zlib
from typing import Dict, Sequence, Tuple

import numpy as np

MAGIC = 20000630
_COMPRESSION = {0: ("NO", 1), 2: ("ZIPS", 1), 3: ("ZIP", 16)}
_NAMES = {0: "NO", 1: "RLE", 2: "ZIPS", 3: "ZIP", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


def _attr(name: str, kind: str, value: bytes) -> bytes:
    return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(value)) + value


def _zip_pack(raw: bytes, level: int) -> bytes:
    b = np.frombuffer(raw, np.uint8)
    t = np.concatenate([b[0::2], b[1::2]]).astype(np.int16)
    t[1:] = t[1:] - t[:-1] + 128
    packed = zlib.compress((t & 255).astype(np.uint8).tobytes(), level)
    return packed if len(packed) < len(raw) else raw


def _zip_unpack(data: bytes, size: int) -> bytes:
    if len(data) == size:
        return data                                                # stored raw
    t = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int64)
    if t.size != size:
        raise ValueError(f"EXR: a chunk inflates to {t.size} bytes, {size} expected")
    t[1:] -= 128
    t = (np.cumsum(t) & 255).astype(np.uint8)
    half = (size + 1) // 2
    out = np.empty(size, np.uint8)
    out[0::2], out[1::2] = t[:half], t[half:]
    return out.tobytes()


def write_exr(path: str, img, channel_names: Sequence[str] = None, compression: str = "ZIP", level: int = 4) -> None:
    """`pyexr.write(path, img)`: img [H, W, C] (or [H, W]) float32 / float16; the channels named as pyexr names them by count -- Z; X, Y;
    R, G, B; R, G, B, A -- unless `channel_names` says otherwise."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.dtype not in (np.float32, np.float16):
        raise ValueError(f"write_exr takes float32 / float16 [H, W, C], got {a.dtype} {a.shape}")
    h, w, c = a.shape
    names = list(channel_names) if channel_names is not None else ({1: ["Z"], 2: ["X", "Y"], 3: list("RGB"), 4: list("RGBA")}.get(c) or [f"C{k:02d}" for k in range(c)])
    if len(names) != c or len(set(names)) != c:
        raise ValueError(f"{c} channels, names {names}")
    comp = {"NO": 0, "NONE": 0, "ZIPS": 2, "ZIP": 3}.get(str(compression).upper())
    if comp is None:
        raise ValueError(f"compression {compression!r}: NO, ZIPS or ZIP")
    ptype = 2 if a.dtype == np.float32 else 1
    order = sorted(range(c), key=lambda k: names[k])               # alphabetical in the file
    chlist = b"".join(names[k].encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for k in order) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([comp])) + _attr("dataWindow", "box2i", box) +
              _attr("displayWindow", "box2i", box) + _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines = _COMPRESSION[comp][1]
    planar = np.ascontiguousarray(a[:, :, order].transpose(0, 2, 1)).astype(a.dtype.newbyteorder("<"))      # [H, C (file order), W]
    chunks = []
    for y in range(0, h, lines):
        raw = planar[y:y + lines].tobytes()
        data = raw if comp == 0 else _zip_pack(raw, level)
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    start = 8 + len(header) + 8 * len(chunks)
    offsets, at = [], start
    for ch in chunks:
        offsets.append(at); at += len(ch)
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", MAGIC, 2) + header + struct.pack(f"<{len(offsets)}Q", *offsets) + b"".join(chunks))


def read_header(buf: bytes) -> Tuple[Dict[str, Tuple[str, bytes]], int]:
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ValueError("not an OpenEXR file")
    if (version & 0xFF) != 2 or (version & 0x1A00):                    # tiled (0x200), deep (0x800), multi-part (0x1000)
        raise NotImplementedError(f"EXR version field {version:#x}: only single-part scanline files are read")
    at, attrs = 8, {}
    while buf[at] != 0:
        e = buf.index(b"\0", at); name = buf[at:e].decode(); at = e + 1
        e = buf.index(b"\0", at); kind = buf[at:e].decode(); at = e + 1
        size, = struct.unpack_from("<i", buf, at); at += 4
        attrs[name] = (kind, buf[at:at + size]); at += size
    return attrs, at + 1


def read_exr(path: str, with_names: bool = False):
    """`pyexr.read(path)`: float32 [H, W, C], the channels in the order R, G, B, A when those are their names (alphabetical otherwise);
    HALF and UINT channels are converted.  `with_names`: also the channel names."""
    with open(path, "rb") as f:
        buf = f.read()
    attrs, at = read_header(buf)
    comp = attrs["compression"][1][0]
    if comp not in _COMPRESSION:
        raise NotImplementedError(f"{path}: EXR compression {_NAMES.get(comp, comp)} is not read here (NO, ZIPS, ZIP are)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, p, cl = [], 0, attrs["channels"][1]
    while cl[p] != 0:
        e = cl.index(b"\0", p); name = cl[p:e].decode(); p = e + 1
        ptype, _, xs, ys = struct.unpack_from("<iB3xii", cl, p); p += 16
        if (xs, ys) != (1, 1):
            raise NotImplementedError(f"{path}: subsampled channel {name}")
        chans.append((name, _PIXEL[ptype]))
    lines = _COMPRESSION[comp][1]
    n_chunks = -(-h // lines)
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, at)
    row_bytes = sum(dt.itemsize for _, dt in chans) * w
    out = np.zeros((h, w, len(chans)), np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        n = min(lines, y0 + h - y)
        raw = buf[off + 8:off + 8 + size]
        if comp != 0:
            raw = _zip_unpack(raw, n * row_bytes)
        if len(raw) != n * row_bytes:
            raise ValueError(f"{path}: a chunk of {len(raw)} bytes, {n * row_bytes} expected")
        q = 0
        for r in range(n):
            for k, (_, dt) in enumerate(chans):
                out[y - y0 + r, :, k] = np.frombuffer(raw, dt, w, q); q += w * dt.itemsize
    names = [n for n, _ in chans]
    if set(names) <= set("RGBA") and len(names) > 1:
        order = [names.index(n) for n in "RGBA" if n in names]
        out, names = out[:, :, order], [names[k] for k in order]
    return (out, names) if with_names else out

"""A PNG reader on zlib: the pixels stb_image hands `loadTexture` (instancer/src/instancer.cpp:38, `stbi_load(path, &w, &h, &channels, 0)`).

[height, width, channels] uint8, rows top-down, with stb's channel count: greyscale 1, greyscale + alpha 2, RGB 3, RGBA 4; a palette
image expanded to RGB (RGBA when it has a tRNS chunk); a tRNS colour key on a greyscale / RGB image adds an alpha channel; samples of
1 / 2 / 4 bits scaled to 0 .. 255, of 16 bits reduced to their high byte; Adam7 interlacing undone.  Nothing else (no gamma, no colour
management): that is what stb does.  Other formats the reference could read through stb (JPEG, BMP ...) are not built; its texture files
are PNGs (configs/config_carpet_render.py:86, config_plush_render.py:100).
"""

from __future__ import annotations

import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}         # samples per pixel by colour type


def _unfilter(raw: np.ndarray, rows: int, stride: int, bpp: int) -> np.ndarray:
    """Undo the per-row filters (PNG spec 9): raw = rows x (1 + stride) bytes -> rows x stride.  None / Up are whole-row numpy operations,
    Sub a running sum modulo 256 per byte lane; Average and Paeth depend on the byte to the left AND the row above, so they run pixel by
    pixel -- over plain Python ints of a bytearray (a few hundred ns a byte: seconds for a 2k texture, where numpy scalars took minutes)."""
    raw = np.asarray(raw, np.uint8).ravel()
    if raw.size != rows * (1 + stride):
        raise ValueError(f"PNG: {raw.size} bytes of image data, {rows} rows of 1 + {stride} bytes need {rows * (1 + stride)} (truncated or corrupt file)")
    table = raw.reshape(rows, 1 + stride)
    if int(table[:, 0].max(initial=0)) > 4:
        raise ValueError(f"PNG: filter type {int(table[:, 0].max())}")
    if int(np.count_nonzero(table[:, 0] >= 3)) * stride > 1 << 16:             # many Average / Paeth bytes: by anti-diagonals instead
        return _unfilter_wavefront(table, rows, stride, bpp)
    out = np.zeros((rows, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    for r in range(rows):
        ft = int(table[r, 0]); line = table[r, 1:]
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = line + prev                                                  # uint8 arithmetic wraps modulo 256
        elif ft == 1:
            pad = (-stride) % bpp
            lanes = np.concatenate([line, np.zeros(pad, np.uint8)]).reshape(-1, bpp).astype(np.uint32)
            cur = (np.cumsum(lanes, axis=0) & 255).astype(np.uint8).reshape(-1)[:stride]
        elif ft in (3, 4):
            ln, pv, cu = bytearray(line.tobytes()), bytearray(prev.tobytes()), bytearray(stride)
            if ft == 3:
                for i in range(stride):
                    left = cu[i - bpp] if i >= bpp else 0
                    cu[i] = (ln[i] + ((left + pv[i]) >> 1)) & 255
            else:
                for i in range(stride):
                    if i >= bpp:
                        a_, c_ = cu[i - bpp], pv[i - bpp]
                    else:
                        a_ = c_ = 0
                    b_ = pv[i]
                    pa, pb, pc = abs(b_ - c_), abs(a_ - c_), abs(a_ + b_ - 2 * c_)
                    pred = a_ if (pa <= pb and pa <= pc) else (b_ if pb <= pc else c_)
                    cu[i] = (ln[i] + pred) & 255
            cur = np.frombuffer(bytes(cu), np.uint8)
        out[r] = cur
        prev = out[r]
    return out


def _unfilter_wavefront(table: np.ndarray, rows: int, stride: int, bpp: int) -> np.ndarray:
    """The same, for images with many Average / Paeth rows: every filter predicts a byte from its left, upper and upper-left neighbours
    (one pixel away), so the pixels of one anti-diagonal (row + column = d) depend only on diagonals d - 1 and d - 2 and are undone
    together, whatever mix of filter types the rows have -- rows + columns numpy steps instead of a Python step per byte (an 800 x 800
    RGBA image of Paeth rows: 0.85 s -> 0.25 s)."""
    npx = -(-stride // bpp)
    filt = np.zeros((rows, npx * bpp), np.int32)
    filt[:, :stride] = table[:, 1:]
    filt = filt.reshape(rows, npx, bpp)
    kind = table[:, 0].astype(np.int32)
    data = np.zeros((rows + 1, npx + 1, bpp), np.int32)                       # a border of zeros above and to the left
    for d in range(rows + npx - 1):
        r = np.arange(max(0, d - npx + 1), min(rows - 1, d) + 1)
        x = d - r
        a, b, c = data[r + 1, x], data[r, x + 1], data[r, x]                  # left, up, up-left
        f = kind[r][:, None]
        pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        pred = np.where(f == 4, paeth, np.where(f == 3, (a + b) >> 1, np.where(f == 2, b, np.where(f == 1, a, 0))))
        data[r + 1, x + 1] = (filt[r, x] + pred) & 255
    return np.ascontiguousarray(data[1:, 1:].reshape(rows, npx * bpp)[:, :stride].astype(np.uint8))


def _samples(lines: np.ndarray, width: int, depth: int, n: int) -> np.ndarray:
    """Unfiltered rows -> [rows, width, n] raw sample values (not yet scaled)."""
    rows = lines.shape[0]
    if depth == 8:
        return lines[:, :width * n].reshape(rows, width, n).astype(np.uint16)
    if depth == 16:
        b = lines[:, :width * n * 2].reshape(rows, width, n, 2).astype(np.uint16)
        return (b[..., 0] << 8) | b[..., 1]
    bits = np.unpackbits(lines, axis=1)[:, :width * n * depth].reshape(rows, width * n, depth)
    vals = np.zeros((rows, width * n), np.uint16)
    for k in range(depth):
        vals = (vals << 1) | bits[..., k]
    return vals.reshape(rows, width, n)


def read_png(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_png(f.read(), path)


def with_channels(img: np.ndarray, channels: int = 4) -> np.ndarray:
    """`tf.image.decode_image(..., channels=4)` / `stbi_load(..., 4)` of a decoded [H, W, 1..4] image: grey becomes R = G = B, a missing
    alpha 255; `channels=3` drops the alpha."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    n = a.shape[2]
    top = np.iinfo(a.dtype).max if a.dtype.kind in "iu" else 1
    rgb = a[..., :3] if n >= 3 else np.repeat(a[..., :1], 3, axis=2)
    if channels == 3:
        return np.ascontiguousarray(rgb)
    if channels != 4:
        raise ValueError("channels 3 or 4")
    alpha = a[..., n - 1:n] if n in (2, 4) else np.full(a.shape[:2] + (1,), top, a.dtype)
    return np.ascontiguousarray(np.concatenate([rgb, alpha], axis=2))


def decode_png(data: bytes, path: str = "<bytes>") -> np.ndarray:
    """The uint8 [H, W, C] image of an encoded PNG (every colour type and bit depth, Adam7; 16-bit samples keep their high byte)."""
    if data[:8] != _SIGNATURE:
        raise ValueError(f"{path}: not a PNG file")
    at = 8
    ihdr = None; idat = []; palette = None; trns = None
    while at + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        at += 12 + n
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if ihdr is None or not idat:
        raise ValueError(f"{path}: PNG without IHDR / IDAT")
    width, height, depth, ctype, _, _, interlace = ihdr
    if ctype not in _CHANNELS or depth not in (1, 2, 4, 8, 16):
        raise ValueError(f"{path}: PNG colour type {ctype} / bit depth {depth}")
    n = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    bpp = max(1, n * depth // 8)
    if interlace == 0:
        stride = (width * n * depth + 7) // 8
        vals = _samples(_unfilter(raw, height, stride, bpp), width, depth, n)
    else:                                                                                    # Adam7
        vals = np.zeros((height, width, n), np.uint16)
        at = 0
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            w = (width - x0 + dx - 1) // dx; h = (height - y0 + dy - 1) // dy
            if w <= 0 or h <= 0:
                continue
            stride = (w * n * depth + 7) // 8
            size = h * (1 + stride)
            vals[y0::dy, x0::dx] = _samples(_unfilter(raw[at:at + size], h, stride, bpp), w, depth, n)
            at += size
    if ctype == 3:                                                                           # palette -> RGB(A)
        if palette is None:
            raise ValueError(f"{path}: palette image without PLTE")
        idx = vals[..., 0].astype(np.int64)
        out = palette[np.minimum(idx, len(palette) - 1)]
        if trns is not None:
            alpha = np.full(len(palette), 255, np.uint8); alpha[:len(trns)] = np.frombuffer(trns, np.uint8)[:len(palette)]
            out = np.concatenate([out, alpha[np.minimum(idx, len(palette) - 1)][..., None]], -1)
        return np.ascontiguousarray(out.astype(np.uint8))
    key = None
    if trns is not None and ctype in (0, 2):                                                 # a colour key: stb adds an alpha channel
        key = np.asarray(struct.unpack(">" + "H" * n, trns[:2 * n]), np.uint16)
    if depth == 16:
        out = (vals >> 8).astype(np.uint8)
    elif depth == 8:
        out = vals.astype(np.uint8)
    else:
        out = (vals * (255 // ((1 << depth) - 1))).astype(np.uint8)
    if key is not None:
        alpha = np.where(np.all(vals == key, axis=-1), 0, 255).astype(np.uint8)
        out = np.concatenate([out, alpha[..., None]], -1)
    return np.ascontiguousarray(out)


def write_png(path: str, img: np.ndarray, level: int = 6) -> None:
    with open(path, "wb") as f:
        f.write(encode_png(img, level))


def encode_png(img: np.ndarray, level: int = 6) -> bytes:
    """An 8-bit non-interlaced PNG of img [H, W, C] (or [H, W]: grey), C in 1 (grey), 2 (grey + alpha), 3 (RGB), 4 (RGBA) -- what
    `tf.io.encode_png` makes of the uint8 image in `Logger.write_image` (logger.py:139-144); also the tests' stand-in textures.  Rows
    unfiltered (type 0), one IDAT; the pixels decode to `img`, the bytes are not TensorFlow's."""
    a = np.ascontiguousarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] not in (1, 2, 3, 4):
        raise ValueError(f"write_png takes uint8 [H, W, 1..4], got {a.dtype} {a.shape}")
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], axis=1).tobytes()
    chunk = lambda kind, body: struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return _SIGNATURE + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b"")

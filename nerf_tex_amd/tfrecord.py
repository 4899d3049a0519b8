"""TFRecord files of `tf.train.Example`s, read and written without TensorFlow -- the training sets of the reference
(`network/dataset.py:77-129` reads them, `data/nerf2tfr.py` makes them: features `image` (an encoded PNG, or a serialized float32 tensor
when `read_exr`), `pose` and `parameters` (serialized float32 tensors: `tf.io.serialize_tensor`), `angle` (float)).

Formats, as published with TensorFlow (no TensorFlow here to check a byte against: unpinned, like the checkpoint reader):
  * record framing (`tensorflow/core/lib/io/record_writer.cc`): uint64 length | uint32 masked crc32c(length) | data | uint32 masked
    crc32c(data), little endian, the mask of `tensorflow/core/lib/hash/crc32c.h` (rotate right 15, + 0xa282ead8); `compression_type`
    "GZIP" / "ZLIB" wrap the whole file;
  * `Example { Features features = 1 }`, `Features { map<string, Feature> feature = 1 }`, `Feature { oneof: BytesList bytes_list = 1,
    FloatList float_list = 2, Int64List int64_list = 3 }`, each list's `value = 1` (floats / ints packed or not);
  * `TensorProto { dtype = 1, tensor_shape = 2, version_number = 3, tensor_content = 4, float_val = 5, double_val = 6, int_val = 7,
    int64_val = 10 }` -- `serialize_tensor` writes `tensor_content`; the typed `*_val` fields are read too, a single value of them
    filling the shape as `tf.make_ndarray` does.
"""

from __future__ import annotations

import gzip
import struct
import zlib
from typing import Dict, Iterator, List, Union

import numpy as np

from .checkpoint import DT_FLOAT, DT_INT32, DT_INT64, _NP, _pb, _pb_bytes, _proto_fields, _vi, crc32c, mask_crc

DT_DOUBLE, DT_UINT8 = 2, 4
_DTYPES = dict(_NP); _DTYPES[DT_UINT8] = np.dtype("u1")
_CODES = {np.dtype("<f4"): DT_FLOAT, np.dtype("<f8"): DT_DOUBLE, np.dtype("<i4"): DT_INT32, np.dtype("<i8"): DT_INT64, np.dtype("u1"): DT_UINT8}


# ---- framing ---------------------------------------------------------------------------------
def _open(path: str, compression_type: str = None, mode: str = "rb"):
    c = (compression_type or "").upper()
    if c == "GZIP":
        return gzip.open(path, mode)
    if c == "ZLIB":
        if "r" not in mode:
            raise ValueError("ZLIB files are read, not written, here")
        import io
        with open(path, "rb") as f:
            return io.BytesIO(zlib.decompress(f.read()))
    if c:
        raise ValueError(f"compression_type {compression_type!r} (None, '', 'GZIP' or 'ZLIB')")
    return open(path, mode)


def read_records(path: str, compression_type: str = None, verify: bool = True) -> Iterator[bytes]:
    """The records of one file, in order.  `verify`: both checksums of every record (what TensorFlow's reader does); False skips the
    data's (pure-Python crc32c: ~20 MB/s)."""
    with _open(path, compression_type) as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            n, = struct.unpack("<Q", head[:8])
            if struct.unpack("<I", head[8:])[0] != mask_crc(crc32c(head[:8])):
                raise ValueError(f"{path}: corrupted record length")
            body = f.read(n + 4)
            if len(body) < n + 4:
                raise ValueError(f"{path}: truncated record ({len(body)} of {n + 4} bytes)")
            if verify and struct.unpack("<I", body[n:])[0] != mask_crc(crc32c(body[:n])):
                raise ValueError(f"{path}: corrupted record data")
            yield body[:n]


def write_records(path: str, records, compression_type: str = None) -> int:
    n = 0
    with _open(path, compression_type, "wb") as f:
        for r in records:
            head = struct.pack("<Q", len(r))
            f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + r + struct.pack("<I", mask_crc(crc32c(r))))
            n += 1
    return n


# ---- tf.train.Example ------------------------------------------------------------------------
def _packed(wt: int, v, fmt: str, size: int) -> list:
    if wt == 2:
        return list(struct.unpack(f"<{len(v) // size}{fmt}", v))
    return [struct.unpack(f"<{fmt}", struct.pack("<I" if size == 4 else "<Q", v))[0]]


def parse_example(buf: bytes) -> Dict[str, Union[List[bytes], np.ndarray]]:
    """{feature name: [bytes, ...] | float32 array | int64 array} of one serialized `tf.train.Example`."""
    out: Dict[str, Union[List[bytes], np.ndarray]] = {}
    for f, _, features in _proto_fields(buf):
        if f != 1:
            continue
        for f1, _, entry in _proto_fields(features):
            if f1 != 1:
                continue
            key, feature = None, b""
            for f2, _, v in _proto_fields(entry):
                if f2 == 1: key = v.decode("utf-8")
                elif f2 == 2: feature = v
            value: Union[List[bytes], np.ndarray] = []
            for kind, _, lst in _proto_fields(feature):
                items = _proto_fields(lst)
                if kind == 1:
                    value = [v for f3, _, v in items if f3 == 1]
                elif kind == 2:
                    value = np.asarray([x for f3, wt, v in items if f3 == 1 for x in _packed(wt, v, "f", 4)], np.float32)
                elif kind == 3:
                    vals = []
                    for f3, wt, v in items:
                        if f3 != 1:
                            continue
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _varint_signed(v, p); vals.append(x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
                    value = np.asarray(vals, np.int64)
            if key is not None:
                out[key] = value
    return out


def _varint_signed(buf: bytes, p: int):
    from .checkpoint import _varint
    v, p = _varint(buf, p)
    return (v - (1 << 64) if v >> 63 else v), p


def make_example(features: Dict[str, object]) -> bytes:
    """A serialized `tf.train.Example`: bytes (or a list of bytes) -> bytes_list, float(s) -> float_list, int(s) -> int64_list.  Keys in
    sorted order, as protobuf's deterministic map serialization writes them."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], (bytes, bytearray)):
            feature = _pb_bytes(1, b"".join(_pb_bytes(1, bytes(b)) for b in v))
        else:
            a = np.atleast_1d(np.asarray(v))
            if a.dtype.kind == "f":
                feature = _pb_bytes(2, _pb_bytes(1, a.astype("<f4").tobytes()))
            elif a.dtype.kind in "iub":
                feature = _pb_bytes(3, _pb_bytes(1, b"".join(_vi(int(x) & ((1 << 64) - 1)) for x in a)))
            else:
                raise TypeError(f"feature {key!r}: {a.dtype}")
        entries += _pb_bytes(1, _pb_bytes(1, key.encode("utf-8")) + _pb_bytes(2, feature))
    return _pb_bytes(1, entries)


# ---- TensorProto (tf.io.serialize_tensor / tf.io.parse_tensor) ---------------------------------
def parse_tensor(buf: bytes, dtype=None) -> np.ndarray:
    """`tf.io.parse_tensor(buf, dtype)`: the array of a serialized TensorProto; a `dtype` that is not the stored one is an error, as there."""
    code, shape, content, vals = 0, [], None, []
    for f, wt, v in _proto_fields(buf):
        if f == 1: code = v
        elif f == 2:
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _proto_fields(dim):
                        if f3 == 1: size = s
                    shape.append(size)
        elif f == 4: content = v
        elif f == 5: vals += _packed(wt, v, "f", 4)
        elif f == 6: vals += _packed(wt, v, "d", 8)
        elif f in (7, 10):
            if wt == 2:
                p = 0
                while p < len(v):
                    x, p = _varint_signed(v, p); vals.append(x)
            else:
                vals.append(v - (1 << 64) if v >> 63 else v)
    if code not in _DTYPES:
        raise ValueError(f"TensorProto dtype {code} is not one this reader knows (float32/64, int32/64, uint8)")
    dt = _DTYPES[code]
    if dtype is not None and np.dtype(dtype) != dt:
        raise ValueError(f"the tensor holds {dt}, {np.dtype(dtype)} was asked for")
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        a = np.frombuffer(content, dt)
        if a.size != n:
            raise ValueError(f"tensor_content holds {a.size} values, the shape {shape} needs {n}")
        return a.reshape(shape).copy()
    if len(vals) == n:
        return np.asarray(vals, dt).reshape(shape)
    if len(vals) == 1:
        return np.full(shape, vals[0], dt)
    if not vals and n == 0:
        return np.zeros(shape, dt)
    raise ValueError(f"{len(vals)} values for shape {shape}")


def serialize_tensor(a) -> bytes:
    """`tf.io.serialize_tensor(a)`: dtype, shape, tensor_content."""
    a = np.ascontiguousarray(a)
    if a.dtype not in _CODES:
        raise TypeError(f"serialize_tensor: {a.dtype}")
    shape = b"".join(_pb_bytes(2, _pb(1, 0, _vi(int(d)))) for d in a.shape)
    return _pb(1, 0, _vi(_CODES[a.dtype])) + _pb_bytes(2, shape) + _pb_bytes(4, a.astype(a.dtype.newbyteorder("<")).tobytes())


# ---- data/nerf2tfr.py: a NeRF (Blender layout) folder as TFRecord shards ---------------------------
def convert_folder(path_in: str, path_out: str, subsets=("train",), skip_params: bool = False, imgs_per_shard: int = -1, compression_type: str = "") -> List[str]:
    """`data/nerf2tfr.py:65-112`: for every subset the sorted files of `<path_in>/<subset>/` with the frames of `transforms_<subset>.json`
    in order -- `image` the `.png` file's bytes as they are, `pose` the frame's `transform_matrix`, `parameters` the values of its
    `driver_parameters` in the file's order (none with `skip_params`), `angle` = `camera_angle_x` -- into `<path_out>/<subset>[_<shard>].tfr`.
    An `.exr` image goes in as the serialized float32 tensor of its pixels (:47-49; `nerf_tex_amd/exr.py` reads it), to be loaded with
    `TFRecord(read_exr=True)`.  Returns the files."""
    import json
    import math
    import os
    os.makedirs(path_out)                                             # the reference refuses an existing target the same way
    written = []
    for subset in subsets:
        folder = os.path.join(path_in, subset)
        names = sorted(os.listdir(folder))
        with open(os.path.join(path_in, "transforms_" + subset + ".json")) as f:
            d = json.load(f)
        frames, angle = d["frames"], float(d["camera_angle_x"])
        if len(frames) < len(names):
            raise ValueError(f"{subset}: {len(names)} images, {len(frames)} frames")
        per = len(names) if imgs_per_shard < 0 else imgs_per_shard
        n_shards = math.ceil(len(names) / per) if names else 0
        for shard in range(n_shards):
            def records():
                for i in range(shard * per, min((shard + 1) * per, len(names))):
                    ext = os.path.splitext(names[i])[1]
                    if ext == ".png":
                        with open(os.path.join(folder, names[i]), "rb") as g:
                            img = g.read()
                    elif ext == ".exr":
                        from . import exr
                        img = serialize_tensor(exr.read_exr(os.path.join(folder, names[i])))
                    else:
                        raise ValueError(f"{names[i]}: unknown filetype")              # nerf2tfr.py:51
                    fr = frames[i]
                    par = [] if skip_params or "driver_parameters" not in fr else list(fr["driver_parameters"].values())
                    yield make_example({"image": img, "pose": serialize_tensor(np.asarray(fr["transform_matrix"], np.float32)), "angle": angle,
                                        "parameters": serialize_tensor(np.asarray(par, np.float32))})
            out = os.path.join(path_out, subset + ("" if n_shards == 1 else "_" + str(shard)) + ".tfr")
            write_records(out, records(), compression_type)
            written.append(out)
    return written


if __name__ == "__main__":                                            # python -m nerf_tex_amd.tfrecord <path_in> <path_out> ...: data/nerf2tfr.py's command line
    import argparse
    ap = argparse.ArgumentParser(description="Converts NeRF dataset to TFR dataset.")
    ap.add_argument("path_in", help="Path to NeRF dataset.")
    ap.add_argument("path_out", help="Path to save TFR dataset to.")
    ap.add_argument("--subsets", nargs="+", default=["train"], help="Subsets to process.")
    ap.add_argument("--skip_params", action="store_true", help="Do not include the driver parameters in the tfr file.")
    ap.add_argument("--imgs_per_shard", type=int, default=-1, help="Number of images per shard.")
    ap.add_argument("--compression_type", type=str, default="", help="Compression used for the tfrecords ('', GZIP).")
    a = ap.parse_args()
    print("wrote", ", ".join(convert_folder(a.path_in, a.path_out, a.subsets, a.skip_params, a.imgs_per_shard, a.compression_type)))

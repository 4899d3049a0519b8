"""Reader for the checkpoints the reference writes.

The reference saves `tf.train.Checkpoint(**{<model name>: keras_model, 'step': ..., 'optimizer': ...})` through a
`CheckpointManager` (`network/logger.py:30-39, 84-86`): `checkpoints/ckpt-<step>.index` + `.data-00000-of-00001`,
TensorFlow's TensorBundle format.  This module parses it WITHOUT TensorFlow and returns the weights in the
order `NerfModel.set_weights` / `ntx_create` want (Keras `get_weights()` order).

PARITY UNPINNED: the reference ships no checkpoint, and TensorFlow cannot run here, so the parser follows the
published formats (LevelDB table format for `.index`; `BundleHeaderProto` / `BundleEntryProto` of
tensorflow/core/protobuf/tensor_bundle.proto; the object-graph key naming of TF2 checkpoints) and is tested by
round trip against an independently written writer in `tests/`.

Formats, as implemented:
  .index  = LevelDB table: data blocks of prefix-compressed (key, value) entries
            [varint shared][varint non_shared][varint value_len][key suffix][value], a restart array, a 5-byte
            trailer (compression type, masked crc32c); index block of (separator key -> BlockHandle); 48-byte
            footer ending in the magic 0xdb4775248b80fb57.  TF writes the table uncompressed.
            key ""   -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
            key name -> BundleEntryProto  {1: dtype, 2: shape{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c}
  .data-SSSSS-of-NNNNN = tensor bytes at [offset, offset+size)
  variable keys: <root attr>/layer_with_weights-<i>/{kernel,bias}/.ATTRIBUTES/VARIABLE_VALUE
"""

from __future__ import annotations

import os
import re
import struct
from typing import Dict, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32, DT_INT64, DT_STRING = 1, 3, 9, 7
_NP = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8"), 2: np.dtype("<f8")}


# ---- crc32c (Castagnoli), software table -------------------------------------------------
def _make_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _make_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    """leveldb/TF crc masking: rotate right by 15 and add a constant."""
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / protobuf ------------------------------------------------------------------
def _varint(buf: bytes, p: int) -> Tuple[int, int]:
    shift = val = 0
    while True:
        b = buf[p]; p += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, p
        shift += 7


def _proto_fields(buf: bytes) -> List[Tuple[int, int, object]]:
    """[(field number, wire type, value)] of one message; nested messages stay bytes."""
    out, p = [], 0
    while p < len(buf):
        tag, p = _varint(buf, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, p)[0]; p += 8
        elif wt == 2:
            n, p = _varint(buf, p)
            v = buf[p:p + n]; p += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, p)[0]; p += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((f, wt, v))
    return out


def _parse_entry(buf: bytes) -> dict:
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for f, _, v in _proto_fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                                   # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1: size = v3
                    e["shape"].append(size)
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
    return e


# ---- table ---------------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    block = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype != 0:
        raise ValueError(f"compressed table block (type {ctype}); TensorFlow writes bundle indices uncompressed")
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if mask_crc(crc32c(buf[offset:offset + size + 1])) != stored:
            raise ValueError("table block checksum mismatch")
    return block


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    out, p, key = [], 0, b""
    while p < end:
        shared, p = _varint(block, p)
        non_shared, p = _varint(block, p)
        vlen, p = _varint(block, p)
        key = key[:shared] + block[p:p + non_shared]; p += non_shared
        out.append((key, block[p:p + vlen])); p += vlen
    return out


def read_bundle_index(index_path: str, verify: bool = True) -> Dict[str, dict]:
    """`<prefix>.index` -> {tensor name: entry dict}; the header is returned under the key ''."""
    buf = open(index_path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{index_path}: not a TensorBundle index (bad table magic)")
    footer = buf[-48:]
    _, p = _varint(footer, 0); _, p = _varint(footer, p)           # metaindex handle
    ioff, p = _varint(footer, p); isize, p = _varint(footer, p)    # index handle
    entries: Dict[str, dict] = {}
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, q = _varint(handle, 0); bsize, q = _varint(handle, q)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            name = key.decode("utf-8")
            if name == "":
                hdr = {f: v for f, _, v in _proto_fields(value)}
                entries[""] = {"num_shards": hdr.get(1, 1), "endianness": hdr.get(2, 0)}
            else:
                entries[name] = _parse_entry(value)
    return entries


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """All numeric tensors of the bundle `<prefix>.index` / `<prefix>.data-*`."""
    entries = read_bundle_index(prefix + ".index", verify)
    hdr = entries.pop("", {"num_shards": 1, "endianness": 0})
    if hdr.get("endianness", 0) != 0:
        raise ValueError("big-endian bundles are not supported")
    shards: Dict[int, bytes] = {}
    out = {}
    for name, e in entries.items():
        if e["dtype"] not in _NP:
            continue                                            # strings (the object graph) etc.
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = open(f"{prefix}.data-{sid:05d}-of-{hdr['num_shards']:05d}", "rb").read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError(f"{name}: data shard too short")
        if verify and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError(f"{name}: tensor checksum mismatch")
        out[name] = np.frombuffer(raw, dtype=_NP[e["dtype"]]).reshape(e["shape"]).copy()
    return out


def latest_checkpoint(checkpoint_dir: str) -> str:
    """Prefix of the newest `ckpt-<n>` in a directory written by tf.train.CheckpointManager (logger.py:33-39)."""
    best, best_n = None, -1
    for f in os.listdir(checkpoint_dir):
        m = re.fullmatch(r"(ckpt-(\d+))\.index", f)
        if m and int(m.group(2)) > best_n:
            best, best_n = os.path.join(checkpoint_dir, m.group(1)), int(m.group(2))
    if best is None:
        raise FileNotFoundError(f"no ckpt-*.index in {checkpoint_dir}")
    return best


_VAR = re.compile(r"^(?P<root>.+?)/layer_with_weights-(?P<i>\d+)/(?P<kind>kernel|bias)/\.ATTRIBUTES/VARIABLE_VALUE$")


def model_weights_from_bundle(tensors: Dict[str, np.ndarray], layer_table, root: str = "model") -> List[np.ndarray]:
    """Pick the Dense kernels/biases of the Keras model saved under checkpoint attribute `root` (the model's
    name: `render.py:28` passes the `{name: model}` dict as checkpoint_variables) and return them in
    `layer_table` order = [kernel, bias, kernel, bias, ...].

    `layer_with_weights-<i>` follows `model.layers`, whose order for a functional model is by graph depth, not
    by creation: the same order as `get_weights()` / `layer_table` (alpha head last).  Layers are nevertheless
    matched by kernel SHAPE, so that a bundle written by a differently ordered Keras version still restores; the
    256x256 group (trunk 1-4, 6, 7 and the feature layer, in this order under either ordering) is matched by
    ascending index."""
    layers: Dict[int, dict] = {}
    for name, arr in tensors.items():
        m = _VAR.match(name)
        if m and m.group("root") == root:
            layers.setdefault(int(m.group("i")), {})[m.group("kind")] = arr
    if not layers:
        roots = sorted({_VAR.match(n).group("root") for n in tensors if _VAR.match(n)})
        raise KeyError(f"no variables under '{root}/layer_with_weights-*' (roots present: {roots})")
    pool = [layers[i] for i in sorted(layers)]
    for d in pool:
        if "kernel" not in d or "bias" not in d:
            raise KeyError("a layer_with_weights entry lacks kernel or bias")
    out: List[np.ndarray] = []
    used = [False] * len(pool)
    for lname, i, o in layer_table:
        for k, d in enumerate(pool):                             # first unused layer of that shape, ascending index
            if not used[k] and d["kernel"].shape == (i, o) and d["bias"].shape == (o,):
                used[k] = True
                out += [np.asarray(d["kernel"], np.float32), np.asarray(d["bias"], np.float32)]
                break
        else:
            raise KeyError(f"checkpoint has no unused Dense layer of shape ({i},{o}) for '{lname}'")
    if not all(used):
        raise KeyError(f"{used.count(False)} Dense layer(s) of the checkpoint do not belong to this architecture")
    return out


def load_checkpoint(model, path: str, root: str = None, verify: bool = True) -> str:
    """Restore `model` (a `nerf_tex_amd.model.NerfModel`) from a checkpoint prefix, or from the newest
    checkpoint of a directory (what `checkpoint.restore(manager.latest_checkpoint)` does, logger.py:39)."""
    prefix = latest_checkpoint(path) if os.path.isdir(path) else path
    tensors = read_bundle(prefix, verify)
    model.set_weights(model_weights_from_bundle(tensors, model.layer_table(), root or model.name))
    return prefix

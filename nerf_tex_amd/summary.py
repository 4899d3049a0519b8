"""TensorBoard event files, written without TensorFlow -- what the reference's Logger leaves in the run's folder (`logger.py:41-44, 60-64,
79-81`: `tf.summary.create_file_writer(target_path)`, `tf.summary.scalar('Loss', ...)` every `i_summary` steps, `tf.summary.image('Validation
Rendering', imgs, step)` every `i_img`).

An event file is a TFRecord file (`nerf_tex_amd/tfrecord.py`) of `Event` messages: `wall_time = 1` (double), `step = 2`, the first record
`file_version = 3: "brain.Event:2"`, the others `summary = 5 { value = 1 { tag = 1, metadata = 9, tensor = 8 } }` as TF 2.x's summary ops
write them: a scalar is a float32 `TensorProto` of rank 0 under plugin "scalars" (data class 1), an image summary a string tensor
[width, height, PNG, PNG, ...] under plugin "images" (data class 3; at most `max_outputs` = 3 images).  As published in
tensorflow/core/util/event.proto, framework/summary.proto, tensorboard/plugins/{scalar,image}: no TensorBoard here to open one (unpinned)."""

from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, List

import numpy as np

from .checkpoint import DT_FLOAT, DT_STRING, _pb, _pb_bytes, _proto_fields, _vi, crc32c, mask_crc


def _event(wall_time: float, step: int = 0, file_version: str = None, summary: bytes = None) -> bytes:
    e = _pb(1, 1, struct.pack("<d", wall_time))
    if step:
        e += _pb(2, 0, _vi(int(step)))
    if file_version is not None:
        e += _pb_bytes(3, file_version.encode())
    if summary is not None:
        e += _pb_bytes(5, summary)
    return e


def _value(tag: str, plugin: str, data_class: int, tensor: bytes, content: bytes = b"") -> bytes:
    plugin_data = _pb_bytes(1, plugin.encode()) + (_pb_bytes(2, content) if content else b"")
    meta = _pb_bytes(1, plugin_data) + _pb(4, 0, _vi(data_class))
    return _pb_bytes(1, _pb_bytes(1, tag.encode()) + _pb_bytes(9, meta) + _pb_bytes(8, tensor))


class FileWriter:
    """`tf.summary.create_file_writer(logdir)`: `events.out.tfevents.<seconds>.<host>.<pid>.v2` in `logdir`, appended to record by record."""

    def __init__(self, logdir: str) -> None:
        os.makedirs(logdir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(now)}.{socket.gethostname()}.{os.getpid()}.v2")
        self._f = open(self.path, "ab")
        self._write(_event(now, file_version="brain.Event:2"))

    def _write(self, record: bytes) -> None:
        head = struct.pack("<Q", len(record))
        self._f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + record + struct.pack("<I", mask_crc(crc32c(record))))

    def scalar(self, tag: str, value: float, step: int, wall_time: float = None) -> None:
        """`tf.summary.scalar(tag, value, step=step)`"""
        tensor = _pb(1, 0, _vi(DT_FLOAT)) + _pb_bytes(2, b"") + _pb_bytes(5, struct.pack("<f", float(value)))      # dtype, shape {}, float_val (packed)
        self._write(_event(time.time() if wall_time is None else wall_time, step, summary=_value(tag, "scalars", 1, tensor)))

    def image(self, tag: str, images_u8, step: int, max_outputs: int = 3, wall_time: float = None) -> None:
        """`tf.summary.image(tag, images, step=step)` of uint8 images [k, H, W, C] (the float -> uint8 conversion is the caller's, as
        `image_epilogue` does it): the first `max_outputs` as PNGs."""
        from . import png
        a = np.asarray(images_u8)
        if a.ndim != 4 or a.dtype != np.uint8:
            raise ValueError(f"image summary takes uint8 [k, H, W, C], got {a.dtype} {a.shape}")
        strings = [str(a.shape[2]).encode(), str(a.shape[1]).encode()] + [png.encode_png(im) for im in a[:max_outputs]]
        shape = _pb_bytes(2, _pb(1, 0, _vi(len(strings))))
        tensor = _pb(1, 0, _vi(DT_STRING)) + _pb_bytes(2, shape) + b"".join(_pb_bytes(8, s) for s in strings)
        content = _pb(1, 0, _vi(max_outputs))                                                                   # ImagePluginData.max_images_requested
        self._write(_event(time.time() if wall_time is None else wall_time, step, summary=_value(tag, "images", 3, tensor, content)))

    def flush(self) -> None:
        self._f.flush()

    def close(self) -> None:
        if not self._f.closed:
            self._f.close()


def read_events(path: str) -> List[dict]:
    """The events of a file back as dicts {'wall_time', 'step', ['file_version'], ['tag', 'plugin', 'value' | 'images']} (tests, and a quick
    look at a run without TensorBoard)."""
    from . import png, tfrecord
    out = []
    for rec in tfrecord.read_records(path):
        e: Dict[str, object] = {"step": 0}
        for f, wt, v in _proto_fields(rec):
            if f == 1: e["wall_time"] = struct.unpack("<d", struct.pack("<Q", v))[0]
            elif f == 2: e["step"] = v
            elif f == 3: e["file_version"] = v.decode()
            elif f == 5:
                for _, _, val in _proto_fields(v):
                    fields = {ff: vv for ff, _, vv in _proto_fields(val)}
                    e["tag"] = fields[1].decode()
                    meta = {ff: vv for ff, _, vv in _proto_fields(fields[9])}
                    e["plugin"] = {ff: vv for ff, _, vv in _proto_fields(meta[1])}[1].decode()
                    e["data_class"] = meta.get(4)
                    tensor = _proto_fields(fields[8])
                    if e["plugin"] == "scalars":
                        raw = [vv for ff, _, vv in tensor if ff == 5][0]
                        e["value"] = struct.unpack("<f", raw if isinstance(raw, bytes) else struct.pack("<I", raw))[0]
                    else:
                        strings = [vv for ff, _, vv in tensor if ff == 8]
                        e["size"] = (int(strings[0]), int(strings[1]))
                        e["images"] = [png.decode_png(s) for s in strings[2:]]
        out.append(e)
    return out

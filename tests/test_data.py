"""The data side of training (network/dataset.py:10-196, data/nerf2tfr.py): TFRecord framing, tf.train.Example, TensorProto, the loaders and
the order `Dataset` hands batches out in.  No TensorFlow here: the reader is checked against bytes assembled by hand in this file from the
published formats (independent of the product's writer), the published crc32c check value, and the product's writer against its reader."""

import gzip
import json
import os
import struct

import numpy as np
import pytest
import torch

from nerf_tex_amd import dataset as D
from nerf_tex_amd import png, tfrecord as T


# ---- an independent, bitwise crc32c and protobuf assembly (nothing shared with the product) ----
def crc32c_bitwise(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def masked(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n: int) -> bytes:
    out = b""
    while n >= 0x80:
        out += bytes([n & 0x7F | 0x80]); n >>= 7
    return out + bytes([n])


def ld(field: int, payload: bytes) -> bytes:                       # a length-delimited field
    return varint(field << 3 | 2) + varint(len(payload)) + payload


def frame(record: bytes) -> bytes:
    head = struct.pack("<Q", len(record))
    return head + struct.pack("<I", masked(crc32c_bitwise(head))) + record + struct.pack("<I", masked(crc32c_bitwise(record)))


def test_crc32c_and_the_record_framing(tmp_path):
    from nerf_tex_amd.checkpoint import crc32c, mask_crc
    assert crc32c(b"123456789") == 0xE3069283 == crc32c_bitwise(b"123456789")            # the check value of CRC-32C (Castagnoli)
    assert crc32c(b"\x00" * 32) == 0x8A9136AA                                              # RFC 3720 B.4: 32 bytes of zeros
    assert mask_crc(0xE3069283) == masked(0xE3069283)
    big = np.random.default_rng(0).integers(0, 256, 70001, dtype=np.uint8).tobytes()       # long buffers go through the stretch-parallel pass
    assert crc32c(big) == crc32c_bitwise(big) and crc32c(big[30000:], crc32c(big[:30000])) == crc32c(big) and crc32c(big[:65536]) == crc32c_bitwise(big[:65536])
    recs = [b"", b"a", bytes(range(256)) * 5]
    path = str(tmp_path / "x.tfr")
    with open(path, "wb") as f:
        f.write(b"".join(frame(r) for r in recs))
    assert list(T.read_records(path)) == recs
    with gzip.open(path + ".gz", "wb") as f:
        f.write(b"".join(frame(r) for r in recs))
    assert list(T.read_records(path + ".gz", "GZIP")) == recs
    import zlib
    with open(path + ".z", "wb") as f:
        f.write(zlib.compress(b"".join(frame(r) for r in recs)))
    assert list(T.read_records(path + ".z", "ZLIB")) == recs
    T.write_records(path + ".w", recs)                                                     # the product's writer: the same bytes
    assert open(path + ".w", "rb").read() == open(path, "rb").read()
    raw = bytearray(open(path, "rb").read())
    raw[-6] ^= 1                                                                           # a bit of the last record's data
    open(path + ".bad", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupted record data"):
        list(T.read_records(path + ".bad"))
    assert list(T.read_records(path + ".bad", verify=False))[-1] != recs[-1]
    raw = bytearray(open(path, "rb").read()); raw[0] ^= 1                                  # the first record's length
    open(path + ".bad", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupted record length"):
        list(T.read_records(path + ".bad"))
    open(path + ".bad", "wb").write(open(path, "rb").read()[:-3])
    with pytest.raises(ValueError, match="truncated"):
        list(T.read_records(path + ".bad"))
    with pytest.raises(ValueError, match="compression_type"):
        list(T.read_records(path, "LZ4"))


def test_examples_and_tensors_from_hand_assembled_bytes():
    # Feature { bytes_list = 1 | float_list = 2 | int64_list = 3 }, each { value = 1 }
    f_bytes = ld(1, ld(1, b"\x89PNG...") + ld(1, b"second"))
    f_float_packed = ld(2, ld(1, struct.pack("<2f", 0.6, -1.5)))
    f_float_single = ld(2, varint(1 << 3 | 5) + struct.pack("<f", 0.25))                   # an unpacked float: wire type 5
    f_ints = ld(3, ld(1, varint(7) + varint((1 << 64) - 3)) + varint(1 << 3 | 0) + varint(9))
    entry = lambda k, f: ld(1, ld(1, k.encode()) + ld(2, f))
    example = ld(1, entry("image", f_bytes) + entry("angle", f_float_packed) + entry("x", f_float_single) + entry("n", f_ints))
    got = T.parse_example(example)
    assert got["image"] == [b"\x89PNG...", b"second"]
    assert np.array_equal(got["angle"], np.asarray([0.6, -1.5], np.float32)) and got["angle"].dtype == np.float32
    assert np.array_equal(got["x"], np.asarray([0.25], np.float32))
    assert np.array_equal(got["n"], [7, -3, 9]) and got["n"].dtype == np.int64
    again = T.parse_example(T.make_example({"image": [b"\x89PNG...", b"second"], "angle": [0.6, -1.5], "n": [7, -3, 9]}))
    assert again["image"] == got["image"] and np.array_equal(again["angle"], got["angle"]) and np.array_equal(again["n"], got["n"])
    # TensorProto { dtype = 1, tensor_shape = 2 { dim = 2 { size = 1 } }, tensor_content = 4, float_val = 5 }
    a = np.arange(16, dtype=np.float32).reshape(4, 4) / 3
    shape = ld(2, ld(2, varint(1 << 3) + varint(4)) + ld(2, varint(1 << 3) + varint(4)))
    proto = varint(1 << 3) + varint(1) + shape + ld(4, a.tobytes())
    assert np.array_equal(T.parse_tensor(proto, np.float32), a)
    assert T.serialize_tensor(a) == proto                                                   # the writer: the same bytes
    as_vals = varint(1 << 3) + varint(1) + shape + ld(5, a.tobytes())                     # packed float_val instead of tensor_content
    assert np.array_equal(T.parse_tensor(as_vals), a)
    fill = varint(1 << 3) + varint(1) + shape + varint(5 << 3 | 5) + struct.pack("<f", 2.5)   # one value fills the shape
    assert np.array_equal(T.parse_tensor(fill), np.full((4, 4), 2.5, np.float32))
    empty = varint(1 << 3) + varint(1) + ld(2, ld(2, varint(1 << 3) + varint(0)))          # tf.constant([]): shape [0], no content
    assert T.parse_tensor(empty, np.float32).shape == (0,)
    assert T.parse_tensor(T.serialize_tensor(np.zeros(0, np.float32))).shape == (0,)
    ints = varint(1 << 3) + varint(9) + ld(2, ld(2, varint(1 << 3) + varint(2))) + ld(10, varint(5) + varint((1 << 64) - 1))
    assert np.array_equal(T.parse_tensor(ints), np.asarray([5, -1], np.int64))
    with pytest.raises(ValueError, match="was asked for"):
        T.parse_tensor(proto, np.float64)
    with pytest.raises(ValueError, match="dtype"):
        T.parse_tensor(varint(1 << 3) + varint(7))                                         # DT_STRING


def nerf_folder(root, n=5, h=12, w=16, seed=0, parameters=True):
    """A NeRF (Blender layout) folder: <root>/train/r_<i>.png (RGBA; one grey + alpha, one RGB) and transforms_train.json."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "train"))
    frames, imgs = [], []
    for i in range(n):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        stored = img[..., [0, 3]] if i == 1 else img[..., :3] if i == 2 else img
        png.write_png(os.path.join(root, "train", f"r_{i:02d}.png"), stored)
        imgs.append(png.with_channels(stored, 4))
        c2w = np.eye(4); c2w[:3, 3] = rng.normal(size=3)
        fr = {"file_path": f"./train/r_{i:02d}", "transform_matrix": c2w.tolist()}
        if parameters:
            fr["driver_parameters"] = {"zeta": float(i), "alpha": 0.5, "len": float(rng.normal())}   # NOT sorted: the file's order counts
        frames.append(fr)
    with open(os.path.join(root, "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": 0.6, "frames": frames}, f)
    return imgs, frames


def test_with_channels_is_decode_image_channels_4():
    g = np.asarray([[[10], [200]]], np.uint8)
    assert np.array_equal(png.with_channels(g, 4), [[[10, 10, 10, 255], [200, 200, 200, 255]]])
    ga = np.asarray([[[10, 7]]], np.uint8)
    assert np.array_equal(png.with_channels(ga, 4), [[[10, 10, 10, 7]]])
    rgb = np.asarray([[[1, 2, 3]]], np.uint8)
    assert np.array_equal(png.with_channels(rgb, 4), [[[1, 2, 3, 255]]]) and np.array_equal(png.with_channels(rgb, 3), rgb)
    rgba = np.asarray([[[1, 2, 3, 4]]], np.uint8)
    assert np.array_equal(png.with_channels(rgba, 4), rgba) and np.array_equal(png.with_channels(rgba, 3), rgb)


@pytest.mark.parametrize("compression", [None, "GZIP"])
def test_folder_to_tfrecord_to_views(tmp_path, compression):
    """data/nerf2tfr.py then dataset.TFRecord give the views dataset.FileFolder reads from the folder itself: poses, parameters in the
    file's order, the images as RGBA, height / width / focal from the first record."""
    imgs, frames = nerf_folder(str(tmp_path / "nerf"))
    files = T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "tfr"), imgs_per_shard=2, compression_type=compression or "")
    assert [os.path.basename(f) for f in files] == ["train_0.tfr", "train_1.tfr", "train_2.tfr"]
    with pytest.raises(FileExistsError):
        T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "tfr"))
    views, h, w, focal, cb, bc = D.TFRecord(files[0], compression_type=compression)
    assert len(views) == 2 and (h, w) == (12, 16) and not cb
    assert focal == 16 / np.tan(float(np.float32(0.6)) / 2) / 2                          # the angle went through a float32 feature
    one = T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "one"), compression_type=compression or "")
    assert [os.path.basename(f) for f in one] == ["train.tfr"]
    views, h, w, focal, cb, bc = D.TFRecord(one[0], composite_bkgd=True, bkgd_color=[0, 1, 0.5], compression_type=compression)
    assert len(views) == 5 and cb and bc == [0, 1, 0.5]
    folder, h2, w2, focal2, _, _ = D.FileFolder(str(tmp_path / "nerf" / "train"), str(tmp_path / "nerf" / "transforms_train.json"), idxs=list(range(5)))
    assert (h2, w2) == (h, w) and abs(focal2 - focal) < 1e-5 * focal
    for k, (a, b) in enumerate(zip(views, folder)):
        assert np.array_equal(a["rgba"], imgs[k]) and np.array_equal(b["rgba"], imgs[k]) and a["rgba"].dtype == np.uint8
        assert np.array_equal(a["pose"], np.asarray(frames[k]["transform_matrix"], np.float32)) and np.array_equal(a["pose"], b["pose"])
        want = np.asarray([frames[k]["driver_parameters"][n] for n in ("zeta", "alpha", "len")], np.float32)
        assert np.array_equal(a["parameters"], want) and np.array_equal(b["parameters"], want)
    sub, _, _, _, _, _ = D.FileFolder(str(tmp_path / "nerf" / "train"), str(tmp_path / "nerf" / "transforms_train.json"), idxs=[0, 3])
    assert len(sub) == 2 and np.array_equal(sub[1]["rgba"], imgs[3]) and sub[1]["parameters"][0] == 3.0
    skipped = T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "skip"), skip_params=True)
    assert D.TFRecord(skipped[0])[0][0]["parameters"].shape == (0,)
    as_dir = D.TFRecord(str(tmp_path / "tfr"), compression_type=compression)[0]            # a directory: every file of it (os.listdir order)
    assert len(as_dir) == 5


def test_exr_tensors_in_a_tfrecord(tmp_path):
    """read_exr (dataset.py:99-101, 125-126): the image is a serialized float32 [H, W, 4] tensor, its colours are taken as they are (no
    premultiplication), the background compositing is switched off."""
    rng = np.random.default_rng(3)
    imgs = [rng.random((6, 5, 4), dtype=np.float32) * 3 for _ in range(2)]
    recs = [T.make_example({"image": T.serialize_tensor(im), "pose": T.serialize_tensor(np.eye(4, dtype=np.float32)), "angle": 0.7,
                            "parameters": T.serialize_tensor(np.asarray([1.0, 2.0], np.float32))}) for im in imgs]
    path = str(tmp_path / "exr.tfr")
    T.write_records(path, recs)
    ds = D.Dataset({"module": "nerf_tex_amd.dataset.TFRecord", "tfr_path": path, "read_exr": True, "composite_bkgd": True},
                   {"module": "nerf_tex_amd.pixel_sampler.Full"}, n_epochs=1, device="cpu")
    assert not ds.composite_bkgd and ds.n_parameters == 2 and ds.n_samples == 30
    for k, b in enumerate(ds):
        assert torch.equal(b["color"][0], torch.from_numpy(imgs[k][..., :3].reshape(-1, 3))) and torch.equal(b["alpha"][0], torch.from_numpy(imgs[k][..., 3].reshape(-1)))
    with pytest.raises(ValueError):
        D.TFRecord(path)                                                                   # not PNG bytes


def image_dataset(tmp_path, **kw):
    imgs, frames = nerf_folder(str(tmp_path / "nerf"), n=kw.pop("n", 5))
    cfg = {"module": "network.dataset.FileFolder", "imgs_path": str(tmp_path / "nerf" / "train"), "poses_path": str(tmp_path / "nerf" / "transforms_train.json"),
           "idxs": list(range(len(imgs))), **kw.pop("loader", {})}
    from nerf_tex_amd import util
    cfg = util.remap_reference_config(cfg)
    return D.Dataset(cfg, {"module": "nerf_tex_amd.pixel_sampler.Full"}, device="cpu", **kw), imgs, frames


def test_colours_of_a_batch_are_the_loaders_maps_at_the_sampled_pixels(tmp_path):
    """dataset.py:104-112 + :49, :57: convert_image_dtype (x * (1 / 255) in float32), colour times alpha, over the background colour when
    asked -- evaluated here on the gathered pixels of the resident uint8 image, bit for bit what the whole-image map gives."""
    for cb in (False, True):
        ds, imgs, _ = image_dataset(tmp_path / str(cb), n_epochs=1, batchsize=2, loader={"composite_bkgd": cb, "bkgd_color": [0.25, 1.0, 0.5]})
        assert ds.has_images and ds.ray_sampler is None and ds.n_parameters == 3 and ds.n_samples == 12 * 16 and len(ds) == 3
        batches = list(ds)
        assert [b["color"].shape[0] for b in batches] == [2, 2, 1] and batches[0]["color"].shape == (2, 192, 3) and batches[0]["alpha"].shape == (2, 192)
        assert batches[0]["parameters"].shape == (2, 3) and set(batches[0]) == {"parameters", "color", "alpha"}
        for k in range(5):
            f = imgs[k].astype(np.float32) * np.float32(1.0 / 255)
            color = f[..., :3] * f[..., 3:]
            if cb:
                color = color + (1 - f[..., 3:]) * np.asarray([0.25, 1.0, 0.5], np.float32)
            b = batches[k // 2]
            assert np.array_equal(b["color"][k % 2].numpy(), color.reshape(-1, 3)) and np.array_equal(b["alpha"][k % 2].numpy(), f[..., 3].reshape(-1))
        loc = torch.tensor([[0, 0], [11, 15], [3, 7]], dtype=torch.int32)                  # an [n, 2] tensor of (row, col): tf.gather_nd
        c, a = ds.colors_at(4, loc, torch.device("cpu"))
        assert torch.equal(c, batches[2]["color"][0].reshape(12, 16, 3)[[0, 11, 3], [0, 15, 7]]) and torch.equal(a, batches[2]["alpha"][0].reshape(12, 16)[[0, 11, 3], [0, 15, 7]])
        with pytest.raises(NotImplementedError):
            ds.colors_at(0, loc.float(), torch.device("cpu"))


def test_the_order_of_batches_is_shuffle_repeat_batch(tmp_path):
    """dataset.py:62: `.shuffle(buffer, reshuffle_each_iteration=True).repeat(n_epochs).batch(batchsize)`."""
    ds, imgs, _ = image_dataset(tmp_path / "a", n_epochs=2, batchsize=3)
    which = lambda b: [int(p[0]) for p in b["parameters"]]                                # 'zeta' = the view's index
    assert [which(b) for b in ds] == [[0, 1, 2], [3, 4, 0], [1, 2, 3], [4]]              # batches across the seam of two epochs, the last short
    assert [which(b) for b in ds.take(2)] == [[0, 1, 2], [3, 4, 0]] and list(ds.take(0)) == []
    ds, _, _ = image_dataset(tmp_path / "b", batchsize=2)                                  # n_epochs=None: for ever
    assert [which(b) for b in ds.take(6)] == [[0, 1], [2, 3], [4, 0], [1, 2], [3, 4], [0, 1]]
    ds, _, _ = image_dataset(tmp_path / "c", n=7, n_epochs=3, batchsize=7, shuffle_buffer_size=100, seed=5)
    epochs = [which(b) for b in ds]
    assert all(sorted(e) == list(range(7)) for e in epochs) and len(epochs) == 3           # a buffer that holds the epoch: a permutation of it
    assert epochs[0] != list(range(7)) and len({tuple(e) for e in epochs}) > 1             # reshuffled each epoch
    again, _, _ = image_dataset(tmp_path / "d", n=7, n_epochs=3, batchsize=7, shuffle_buffer_size=100, seed=5)
    assert [which(b) for b in again] == epochs                                             # the seed decides
    ds, _, _ = image_dataset(tmp_path / "e", n=7, n_epochs=1, batchsize=1, shuffle_buffer_size=3, seed=1)
    order = [which(b)[0] for b in ds]
    assert sorted(order) == list(range(7)) and all(order[i] <= i + 2 for i in range(7))    # element i leaves a buffer of 3 no earlier than position i - 2
    with pytest.raises(ValueError):
        D.Dataset({"module": "nerf_tex_amd.dataset.FileFolder", "poses_path": str(tmp_path / "a" / "nerf" / "transforms_train.json"), "idxs": [0]},
                  {"module": "nerf_tex_amd.pixel_sampler.Full"}, device="cpu")               # neither rays nor images


def test_reference_module_names_resolve_to_the_data_side():
    from nerf_tex_amd import util
    cfg = util.remap_reference_config({"module": "network.train.Train", "train_dataset_config": {"module": "network.dataset.Dataset",
                                      "data_loader_config": {"module": "network.dataset.TFRecord"}, "pixel_sampler_config": {"module": "network.pixel_sampler.Proxy"}}})
    assert cfg.module == "nerf_tex_amd.train.Train" and cfg.train_dataset_config.data_loader_config.module == "nerf_tex_amd.dataset.TFRecord"
    assert cfg.train_dataset_config.pixel_sampler_config.module == "nerf_tex_amd.pixel_sampler.Proxy"
    for m in ("nerf_tex_amd.dataset.TFRecord", "nerf_tex_amd.dataset.FileFolder", "nerf_tex_amd.pixel_sampler.Independent", "nerf_tex_amd.train.Train"):
        assert callable(util.get_attr_from_path(m))


# ---- pose / parameter generators: PINNED against the reference's own data/distribution.py + data/sampler.py -------------------
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_distributions_are_the_reference_s_bit_for_bit():
    """tests/golden/distributions.json: 15 blocks (random spheres, hemispheres on grids, boxes, ranges, constants, concatenations of
    distributions and of samplers) run through the reference's own modules by oracle/gen_golden.py, seven draws each under a seeded numpy
    stream; the package's generators give the same float64 values, the same `sampler.n`, the same count and `done()`."""
    from nerf_tex_amd import util
    cases = json.load(open(os.path.join(GOLDEN, "distributions.json")))["cases"]
    assert len(cases) == 15
    for c in cases:
        np.random.seed(c["seed"])
        dist = util.instantiate(c["config"])                       # reference module names resolve to nerf_tex_amd.distributions
        assert type(dist).__module__ == "nerf_tex_amd.distributions"
        got = [np.asarray(dist(), np.float64).tolist() for _ in range(7)]
        assert got == c["samples"], c["config"]
        assert (dist.sampler.n, dist.sampler.idx, dist.sampler.done()) == (c["n"], c["idx_after"], c["done_after"]), c["config"]


def test_file_names_are_the_reference_s():
    """util.format_name (util.py:56-62; the Logger's image and step-folder names), run by oracle/gen_golden.py: `render.util_format`."""
    from nerf_tex_amd.render import util_format
    names = json.load(open(os.path.join(GOLDEN, "distributions.json")))["format_name"]
    assert len(names) == 9 and all(util_format(idx, mx)[:-4] + ".png" == name for idx, mx, name in names), names


def test_jittered_grid_points_stay_in_their_cells():
    """data.sampler.Stratified cannot run in the reference (it calls a method its parent lacks): no vector to pin; what it describes is checked."""
    from nerf_tex_amd.distributions import GridPoints, JitteredGridPoints
    np.random.seed(0)
    g, j = GridPoints(d=2, n=9), JitteredGridPoints(d=2, n=9)
    for _ in range(9):
        corner, p = g(), j()
        assert np.all(p >= corner) and np.all(p < corner + 1 / 3)
    assert j.done() and j.cells_per_d == 3


@pytest.mark.parametrize("family", ["carpet", "grass", "grass_filtered", "plush"])
def test_generated_views_of_the_shipped_render_configs(family):
    """dataset.GenerateData on the shipped render configs' own `data_loader_config` blocks (committed with the fixture) under `main.py`'s seed:
    the poses are `look_at` of the reference modules' samples times the radius, the parameters the reference modules' vectors."""
    doc = json.load(open(os.path.join(GOLDEN, f"cameras_{family}.json")))
    np.random.seed(doc["seed"])
    block = dict(doc["data_loader_config"]); block.pop("module")
    views, h, w, focal, cb, bc = D.GenerateData(**block)
    assert (h, w) == (doc["height"], doc["width"]) and focal == doc["focal"] and len(views) >= len(doc["views"]) > 0
    for mine, ref in zip(views, doc["views"]):
        assert np.array_equal(mine["parameters"], np.asarray(ref["parameters"], np.float32))
        want = D.look_at(np.asarray(ref["pose_dist_sample"]) * ref["radius"], offset=block.get("offset", (0., 0., 0.)))
        assert np.array_equal(mine["pose"], want)
        assert np.allclose(mine["pose"], np.asarray(ref["c2w_oracle_look_at_f32"], np.float32), rtol=0, atol=1e-6)


@pytest.mark.parametrize("family", ["carpet", "fur", "grass", "grass_filtered", "plush"])
def test_validation_views_of_the_shipped_training_configs(family):
    doc = json.load(open(os.path.join(GOLDEN, "train_configs.json")))[family]
    np.random.seed(doc["seed"])
    block = dict(doc["val_dataset_config"]["data_loader_config"]); block.pop("module")
    views = D.GenerateData(**block)[0]
    ref = doc["val_views_reference"]
    assert len(views) == ref["n"]
    for mine, r in zip(views, ref["views"]):
        assert np.array_equal(mine["parameters"], np.asarray(r["parameters"], np.float32))
        assert np.array_equal(mine["pose"], D.look_at(np.asarray(r["pose_dist_sample"]) * r["radius"]))


def test_generate_data_quirks():
    c = {"module": "data.distribution.Constant", "constants": [[0.0, -1.0, 0.5]]}
    off = D.GenerateData(pose_dist_config=c, parameter_dist_config=c, offset=[0.0, 0.0, 1.0])[0]
    assert len(off) == 1 and np.array_equal(off[0]["pose"][:3, 3], np.asarray([0.0, -5.0, 3.5], np.float32))     # radius 5, then the offset
    many = D.GenerateData(pose_dist_config=c, parameter_dist_config=c, offset=[0.0, 0.0, 1.0], dataset_size=300)[0]
    assert len(many) == 300 and np.array_equal(many[0]["pose"][:3, 3], np.asarray([0.0, -5.0, 2.5], np.float32))  # the generator branch drops it
    endless = {"module": "data.distribution.Sphere"}
    assert D.GenerateData(pose_dist_config=endless, parameter_dist_config=endless)[0] == []
    radius = {"module": "data.distribution.Range", "n": 3, "b_0": 2.0, "b_1": 8.0}
    r = D.GenerateData(pose_dist_config=c, parameter_dist_config=c, radius=radius)[0]
    assert len(r) == 1 and np.allclose(np.linalg.norm(r[0]["pose"][:3, 3]), 2.0 * np.linalg.norm([0.0, -1.0, 0.5]))
    with pytest.raises(ValueError):
        D.GenerateData()


# ---- OpenEXR (the Logger's write_exr, the EXR branch of nerf2tfr / TFRecord) ---------------------------------------------------------
def test_exr_files_round_trip_and_a_hand_assembled_one(tmp_path):
    from nerf_tex_amd import exr
    rng = np.random.default_rng(0)
    for shape in ((37, 29, 4), (16, 16, 3), (5, 7, 1), (33, 10, 2), (20, 21)):
        for comp in ("NO", "ZIPS", "ZIP"):
            for dt in (np.float32, np.float16):
                a = (rng.standard_normal(shape) * 3).astype(dt)
                exr.write_exr(str(tmp_path / "a.exr"), a, compression=comp)
                b, names = exr.read_exr(str(tmp_path / "a.exr"), with_names=True)
                assert np.array_equal(b[..., 0] if a.ndim == 2 else b, a.astype(np.float32)) and b.dtype == np.float32
                assert names == {1: ["Z"], 2: ["X", "Y"], 3: list("RGB"), 4: list("RGBA")}[1 if a.ndim == 2 else a.shape[2]]
    smooth = np.linspace(0, 1, 64 * 64 * 4, dtype=np.float32).reshape(64, 64, 4)
    exr.write_exr(str(tmp_path / "s.exr"), smooth)
    assert os.path.getsize(tmp_path / "s.exr") < smooth.nbytes / 4 and np.array_equal(exr.read_exr(str(tmp_path / "s.exr")), smooth)
    # a 2 x 3 file assembled here from the published layout: channels B (HALF) and R (FLOAT), uncompressed, data window starting at (5, 7)
    attr = lambda n, t, v: n.encode() + b"\0" + t.encode() + b"\0" + struct.pack("<i", len(v)) + v
    ch = b"B\0" + struct.pack("<iB3xii", 1, 0, 1, 1) + b"R\0" + struct.pack("<iB3xii", 2, 0, 1, 1) + b"\0"
    box = struct.pack("<4i", 5, 7, 7, 8)
    header = attr("channels", "chlist", ch) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + \
        attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    B = np.asarray([[0.5, 1.0, -2.0], [4.0, 0.25, 8.0]], "<f2"); R = np.asarray([[1.5, 2.5, 3.5], [-1.0, 0.0, 1e-3]], "<f4")
    rows = [struct.pack("<ii", 7 + r, 3 * 2 + 3 * 4) + B[r].tobytes() + R[r].tobytes() for r in range(2)]
    start = 8 + len(header) + 16
    blob = struct.pack("<ii", 20000630, 2) + header + struct.pack("<2Q", start, start + len(rows[0])) + rows[0] + rows[1]
    open(tmp_path / "h.exr", "wb").write(blob)
    img, names = exr.read_exr(str(tmp_path / "h.exr"), with_names=True)
    assert names == ["R", "B"] and np.array_equal(img[..., 0], R) and np.array_equal(img[..., 1], B.astype(np.float32))
    piz = bytearray(blob); piz[blob.index(b"compression\0compression\0") + 28] = 4
    open(tmp_path / "p.exr", "wb").write(bytes(piz))
    with pytest.raises(NotImplementedError, match="PIZ"):
        exr.read_exr(str(tmp_path / "p.exr"))
    with pytest.raises(ValueError):
        exr.write_exr(str(tmp_path / "x.exr"), np.zeros((2, 2, 3), np.uint8))


def test_exr_images_through_the_folder_converter(tmp_path):
    """nerf2tfr.py:47-49 + dataset.py:99-101: `.exr` images become serialized float32 tensors in the TFRecord; `read_exr=True` gives their
    first three channels as the colours, the fourth as alpha."""
    from nerf_tex_amd import exr
    rng = np.random.default_rng(1)
    os.makedirs(tmp_path / "nerf" / "train")
    imgs = [rng.random((6, 8, 4), dtype=np.float32) * 2 for _ in range(3)]
    for k, im in enumerate(imgs):
        exr.write_exr(str(tmp_path / "nerf" / "train" / f"r_{k}.exr"), im)
    json.dump({"camera_angle_x": 0.5, "frames": [{"transform_matrix": np.eye(4).tolist(), "driver_parameters": {"a": float(k)}} for k in range(3)]},
              open(tmp_path / "nerf" / "transforms_train.json", "w"))
    files = T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "tfr"))
    views, h, w, _, cb, _ = D.TFRecord(files[0], read_exr=True, composite_bkgd=True)
    assert (h, w) == (6, 8) and not cb and all(np.array_equal(v["rgba"], im) and v["premultiplied"] for v, im in zip(views, imgs))
    os.rename(tmp_path / "nerf" / "train" / "r_2.exr", tmp_path / "nerf" / "train" / "r_2.txt")
    with pytest.raises(ValueError, match="unknown filetype"):
        T.convert_folder(str(tmp_path / "nerf"), str(tmp_path / "tfr2"))


def test_tensorboard_event_files(tmp_path):
    """nerf_tex_amd/summary.py (logger.py:41-44, 60-64, 79-81): the writer's file read back, and one scalar event assembled here from the
    published messages (event.proto, summary.proto, tensor.proto) record for record."""
    from nerf_tex_amd import summary
    w = summary.FileWriter(str(tmp_path))
    w.scalar("Loss", 0.125, 10, wall_time=1234.5)
    imgs = np.random.default_rng(0).integers(0, 256, (4, 6, 5, 4), dtype=np.uint8)
    w.image("Validation Rendering", imgs, 20)
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.") and w.path.endswith(".v2")
    ev = summary.read_events(w.path)
    assert ev[0]["file_version"] == "brain.Event:2" and (ev[1]["tag"], ev[1]["step"], ev[1]["value"], ev[1]["wall_time"], ev[1]["data_class"]) == ("Loss", 10, 0.125, 1234.5, 1)
    assert ev[2]["plugin"] == "images" and ev[2]["size"] == (5, 6) and len(ev[2]["images"]) == 3 and np.array_equal(ev[2]["images"][1], imgs[1])
    # Event { wall_time = 1 (double), step = 2, summary = 5 { value = 1 { tag = 1, tensor = 8 { dtype = 1: DT_FLOAT, tensor_shape = 2 {}, float_val = 5 }, metadata = 9 {
    #   plugin_data = 1 { plugin_name = 1 }, data_class = 4: DATA_CLASS_SCALAR } } } }
    tensor = varint(1 << 3) + varint(1) + ld(2, b"") + ld(5, struct.pack("<f", 0.125))
    meta = ld(1, ld(1, b"scalars")) + varint(4 << 3) + varint(1)
    event = varint(1 << 3 | 1) + struct.pack("<d", 1234.5) + varint(2 << 3) + varint(10) + ld(5, ld(1, ld(1, b"Loss") + ld(9, meta) + ld(8, tensor)))
    recs = list(T.read_records(w.path))
    assert recs[1] == event
    open(tmp_path / "hand.tfevents", "wb").write(frame(varint(1 << 3 | 1) + struct.pack("<d", 1.0) + ld(3, b"brain.Event:2")) + frame(event))
    back = summary.read_events(str(tmp_path / "hand.tfevents"))
    assert back[1]["value"] == 0.125 and back[1]["step"] == 10 and back[0]["file_version"] == "brain.Event:2"
    with pytest.raises(ValueError):
        summary.FileWriter(str(tmp_path)).image("x", np.zeros((2, 2, 3), np.float32), 1)


def test_blur_augmentation_of_a_folder(tmp_path):
    """nerf_tex_amd/augment.py (data/blur.py): names, the pose file with `Blur` as the FIRST driver parameter, the sigmas of the seeded
    draw, and what the blur does -- nothing at sigma 0 but the gamma round trip, a premultiplied gaussian otherwise (energy of alpha kept away
    from the border, the kernel's taps on an impulse)."""
    from scipy import ndimage
    from nerf_tex_amd import augment, exr
    imgs, frames = nerf_folder(str(tmp_path / "nerf"), n=3, h=24, w=20)
    for fr in json.load(open(tmp_path / "nerf" / "transforms_train.json"))["frames"]:
        assert "file_path" in fr
    out = augment.blur_folder(str(tmp_path / "nerf"), str(tmp_path / "blur"), max_sigma=2.0, dataset_size_increase=2)
    names = sorted(os.listdir(tmp_path / "blur" / "train"))
    assert names == [f"r_{k}.png" for k in range(6)] and out == [str(tmp_path / "blur" / "transforms_train.json")]
    d = json.load(open(out[0]))
    np.random.seed(0); u = np.random.rand(6)
    want = (-np.log(1 - u * (1 - np.exp(-3.0))) / 3.0 * 2.0).tolist()
    assert [fr["driver_parameters"]["Blur"] for fr in d["frames"]] == want and all(0 <= s <= 2.0 for s in want)
    assert list(d["frames"][4]["driver_parameters"]) == ["Blur", "zeta", "alpha", "len"] and d["frames"][4]["driver_parameters"]["zeta"] == 1.0
    assert d["frames"][4]["file_path"] == "./train/r_4" and d["camera_angle_x"] == 0.6
    with pytest.raises(FileExistsError):
        augment.blur_folder(str(tmp_path / "nerf"), str(tmp_path / "blur"))
    same = augment.blur_png(imgs[0], 0.0)                                                  # sigma 0: premultiply, un-premultiply
    opaque = imgs[0][..., 3] > 32
    assert np.abs(same[opaque].astype(int) - imgs[0][opaque].astype(int)).max() <= 1 and np.array_equal(same[..., 3], imgs[0][..., 3])
    dot = np.zeros((21, 21, 4), np.uint8); dot[10, 10] = 255
    b = augment.blur_png(dot, 1.5)
    g = ndimage.gaussian_filter(np.eye(1, 21, 10)[0], 1.5, mode="constant", truncate=4.0)  # the 1-d taps of the same filter
    assert np.array_equal(b[..., 3], np.rint(np.outer(g, g) * 255).astype(np.uint8)) and b[10, 10, 3] > b[10, 12, 3] > b[10, 14, 3] > 0
    assert (b[..., :3][b[..., 3] > 8] >= 250).all()                                        # white stays white where there is alpha to divide by
    e = np.random.default_rng(0).random((12, 10, 4), dtype=np.float32)
    assert np.array_equal(augment.blur_exr(e, 0.1), e)                                      # int(0.6) = 0 taps
    be = augment.blur_exr(np.ones((12, 10, 4), np.float32), 1.0)                           # 6 taps, SAME padding: 1 inside, less at the border
    assert np.allclose(be[4:8, 4:6], 1.0, atol=1e-6) and be[0, 0, 0] < 0.5 and be.dtype == np.float32
    os.makedirs(tmp_path / "nerf2" / "train")
    exr.write_exr(str(tmp_path / "nerf2" / "train" / "r_0.exr"), e)
    json.dump({"camera_angle_x": 0.5, "frames": [{"file_path": "./train/r_0", "transform_matrix": np.eye(4).tolist(), "driver_parameters": {"a": 1.0}}]},
              open(tmp_path / "nerf2" / "transforms_train.json", "w"))
    augment.blur_folder(str(tmp_path / "nerf2"), str(tmp_path / "blur2"), max_sigma=1.0)
    assert exr.read_exr(str(tmp_path / "blur2" / "train" / "r_0.exr")).shape == (12, 10, 4)

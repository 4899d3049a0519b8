"""CPU tests of the host side: plugin mechanism, model container, the C ABI's exported symbols and
the host-only weight packer, ray sharding and the world_size-2 gather (gloo)."""

import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from nerf_tex_amd import _lib
    header = open(os.path.join(ROOT, "include", "nerftex.h")).read()
    declared = set(re.findall(r"\b(ntx_[a-z0-9_]+)\s*\(", header))
    declared -= {"ntx_ctx", "ntx_stream", "ntx_comm"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib.ntx_abi_version() == _lib.ABI_VERSION == 2


def test_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the CPU-container check")
    from nerf_tex_amd import _lib
    d = _lib.ModelDesc(0, 1, 6, 3, 10, 4, 4, 8, 256, 4, 1, 0)
    h = C.c_void_p()
    rc = _lib.lib.ntx_create(C.byref(d), None, 0, 0, C.byref(h))
    assert rc in (_lib.NTX_E_NODEVICE, _lib.NTX_E_HIP) and _lib.lib.ntx_last_error()


def test_unsupported_desc_is_rejected_on_host():
    from nerf_tex_amd import _lib
    for bad in (_lib.ModelDesc(0, 5, 3, 3, 10, 4, 4, 8, 256, 4, 1, 0), _lib.ModelDesc(0, 1, 9, 3, 10, 4, 4, 8, 256, 4, 1, 0),
                _lib.ModelDesc(0, 1, 6, 3, 10, 4, 4, 8, 256, 4, 2, 0), _lib.ModelDesc(0, 1, 6, 3, 8, 4, 4, 8, 256, 4, 1, 0),
                _lib.ModelDesc(0, 1, 6, 3, 10, 4, 4, 6, 256, 4, 1, 0), _lib.ModelDesc(0, 1, 6, 3, 10, 4, 4, 8, 128, 4, 1, 0),
                _lib.ModelDesc(0, 1, 3, 3, 10, 4, 4, 8, 256, 4, 1, 1), _lib.ModelDesc(0, 1, 6, 6, 10, 4, 4, 8, 256, 4, 1, 1)):
        assert _lib.lib.ntx_weight_count(C.byref(bad)) == 0
        assert b"unsupported" in _lib.lib.ntx_last_error()


@pytest.mark.parametrize("desc,count", [((0, 1, 6, 3, 10, 4, 4, 8, 256, 4, 1, 0), 683524), ((0, 1, 4, 3, 10, 4, 4, 8, 256, 4, 1, 0), 678916),
                                        ((0, 3, 3, 3, 10, 4, 4, 8, 256, 4, 1, 0), None), ((0, 0, 2, 3, 10, 4, 4, 8, 256, 4, 1, 0), None),
                                        ((0, 4, 8, 3, 10, 4, 4, 8, 256, 4, 1, 0), None),
                                        ((0, 2, 3, 3, 10, 4, 4, 8, 256, 4, 1, 0), 681220), ((1, 0, 0, 3, 10, 4, 0, 8, 256, 4, 0, 0), 593408 + 8 * 256 + 1 + 256 + 128 + 3),
                                        ((0, 1, 3, 6, 10, 4, 4, 8, 256, 4, 1, 1), 675076)])
def test_pack_weights_is_a_permutation_with_wraparound_tail(desc, count):
    """Host-only packer: every reference weight lands in the stream exactly once (rest is zero pad),
    the tail repeats the first 8 records, biases/heads land in the aux block."""
    from nerf_tex_amd import _lib
    d = _lib.ModelDesc(*desc)
    n = _lib.lib.ntx_weight_count(C.byref(d))
    if count is None:        # the generic family (any [g <= 4, a <= 8]): the count follows the model's own dimensions
        g, a = desc[1], desc[2]
        pm, dm = 63 + 9 * g, 27 + 9 * a
        count = pm * 256 + 256 + 4 * (256 * 256 + 256) + (256 + pm) * 256 + 256 + 2 * (256 * 256 + 256) + (256 * 256 + 256) \
            + (256 + dm) * 256 + 256 + 256 * 128 + 128 + 128 * 3 + 3 + 256 + 1
    assert n == count
    npk = _lib.lib.ntx_packed_count(C.byref(d))
    rng = np.random.default_rng(0)
    blob = (rng.permutation(n) + 1).astype(np.float32)            # distinct, non-zero, exactly representable
    out = np.empty(npk, np.float32)
    fp = C.POINTER(C.c_float)
    assert _lib.lib.ntx_pack_weights(C.byref(d), blob.ctypes.data_as(fp), n, out.ctypes.data_as(fp), npk) == 0
    aux_floats = 3776                                             # ntx_layout.h aux_total(): 12*256 bias + 260 alpha + 388 rgb, rounded to 64
    stream = out[:npk - aux_floats]
    head, tail = stream[:8 * 256], stream[-8 * 256:]
    np.testing.assert_array_equal(head, tail)
    body = stream[:-8 * 256]
    vals = np.concatenate([body[body != 0], out[npk - aux_floats:][out[npk - aux_floats:] != 0]])
    assert vals.size == n and np.array_equal(np.sort(vals), np.sort(blob))
    assert _lib.lib.ntx_pack_weights(C.byref(d), blob.ctypes.data_as(fp), n - 1, out.ctypes.data_as(fp), npk) == _lib.NTX_E_INVALID


def test_pack_weights_fp16x3_splits_every_weight_once():
    """Host-only packer of the fp16x3 stream: each matrix weight appears exactly once as a (hi, lo) pair of IEEE halves with
    hi = float16(w) and lo = float16(w - hi) exactly as numpy rounds them (round to nearest even, subnormals kept, tiny and
    huge weights included); the stream is a whole number of LDS ring turns (64 records)."""
    from nerf_tex_amd import _lib
    d = _lib.ModelDesc(0, 1, 6, 3, 10, 4, 4, 8, 256, 4, 1, 0)
    n = _lib.lib.ntx_weight_count(C.byref(d))
    nb = _lib.lib.ntx_packed_fp16x3_bytes(C.byref(d))
    assert nb % 1024 == 0 and (nb // 1024) % 64 == 0
    rng = np.random.default_rng(1)
    mag = np.exp2(rng.uniform(-30, 4, size=n))                  # spans half subnormals (lo parts well below 2^-24 too)
    blob = (mag * rng.choice([-1.0, 1.0], size=n)).astype(np.float32)
    blob[:8] = [65504.0, -65504.0, 65519.0, 65520.0, 1e9, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25]   # L0 kernel, first rows
    out = np.zeros(nb // 2, np.uint16)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint16)
    assert _lib.lib.ntx_pack_weights_fp16x3(C.byref(d), blob.ctypes.data_as(fp), n, out.ctypes.data_as(up), nb) == 0
    body = out.reshape(-1, 2, 512)                               # (hi record, lo record) per (k16-step, tile)
    hi, lo = body[:, 0].ravel(), body[:, 1].ravel()
    # the float32 packer puts the same weights in a stream of its own: every (hi, lo) must be the numpy split of one of them
    f32pk = np.empty(_lib.lib.ntx_packed_count(C.byref(d)), np.float32)
    assert _lib.lib.ntx_pack_weights(C.byref(d), blob.ctypes.data_as(fp), n, f32pk.ctypes.data_as(fp), f32pk.size) == 0
    stream = f32pk[:f32pk.size - 3776 - 8 * 256]
    w = stream[stream != 0]
    with np.errstate(over="ignore", invalid="ignore"):
        wh = w.astype(np.float16)
        wl = (w - wh.astype(np.float32)).astype(np.float16)
    want = set(zip(wh.view(np.uint16).tolist(), wl.view(np.uint16).tolist()))
    used = (hi != 0) | (lo != 0)
    got = set(zip(hi[used].tolist(), lo[used].tolist()))
    assert got <= want
    # (the 81 direction rows of C1 are not in this stream: they are applied per ray in float32 by dir_block)
    n_matrix = 72 * 256 + 4 * 256 * 256 + 328 * 256 + 2 * 256 * 256 + 256 * 256 + 256 * 256 + 256 * 128
    underflow = int(np.sum((wh.view(np.uint16) & 0x7fff) == 0))          # |w| < 2^-25 rounds to (0, 0): not counted as used
    assert n_matrix - underflow - 300 <= int(used.sum()) <= n_matrix
    mip = _lib.ModelDesc(0, 1, 3, 6, 10, 4, 4, 8, 256, 4, 1, 1)          # the IPE family has its own (shorter) position segment
    nb_mip = _lib.lib.ntx_packed_fp16x3_bytes(C.byref(mip))
    assert nb_mip > 0 and nb_mip % (64 * 1024) == 0


def test_instantiate_and_reference_config_remap():
    from nerf_tex_amd import util
    cfg = {"module": "network.model.ParamNerf",
           "pos_embedding": {"module": "network.model.FourierFeatures", "n_freq_bands": 10},
           "dir_embedding": {"module": "network.model.FourierFeatures", "n_freq_bands": 4},
           "param_embedding": {"module": "network.model.FourierFeatures", "n_freq_bands": 4},
           "n_parameters": [1, 6]}
    mapped = util.remap_reference_config(cfg)
    assert mapped.module == "nerf_tex_amd.model.ParamNerf" and cfg["module"] == "network.model.ParamNerf"
    assert mapped.pos_embedding.module == "nerf_tex_amd.layer.FourierFeatures"
    np.random.seed(0)
    model = util.instantiate(mapped)
    assert list(model) == ["model"]
    m = model["model"]
    assert m.pos_map_dim == 72 and m.dir_map_dim == 81 and m.macs_per_sample() == 680832
    assert util.instantiate(None) is None
    ws = m.get_weights()
    # Keras get_weights() order: trunk 0-7, feature, colour layer, colour half, color, and the alpha head LAST
    assert len(ws) == 26 and ws[0].shape == (72, 256) and ws[10].shape == (328, 256) and ws[16].shape == (256, 256)
    assert ws[18].shape == (337, 256) and ws[20].shape == (256, 128) and ws[22].shape == (128, 3) and ws[24].shape == (256, 1)
    assert np.all(ws[1] == 0) and abs(float(ws[0].max())) <= np.sqrt(6 / (72 + 256))      # glorot_uniform / zeros
    ws[3] = ws[3] + 1
    m.set_weights(ws)
    np.testing.assert_array_equal(m.get_weights()[3], ws[3])
    with pytest.raises(ValueError):
        m.set_weights(ws[:-1])
    # a list in CREATION order (alpha before feature, model.py:111-114) has the right total size, so only the shapes can tell
    creation = ws[:16] + ws[24:26] + ws[16:24]
    assert sum(a.size for a in creation) == m.n_weight_floats()
    with pytest.raises(ValueError, match="feature"):
        m.set_weights(creation)


def test_weight_blob_is_keras_get_weights_order():
    """The blob of the C ABI == np.concatenate(keras_model.get_weights()): a functional tf.keras.Model sorts its layers by
    graph depth, ties by traversal from outputs=[color, alpha] (model.py:125), so alpha comes last.  The host-side packer
    must pick each layer from that position: mark every layer's kernel with a constant and look where it lands."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.model import ParamNerf, Nerf
    from tests.common import EMB
    np.random.seed(0)
    for model in (ParamNerf(EMB(10), EMB(4), EMB(4), [1, 6])["model"], Nerf(EMB(10), EMB(4))["model"]):
        names = [n for n, _, _ in model.layer_table()]
        assert names[:8] == [f"trunk{i}" for i in range(8)] and names[8] == "feature" and names[-2:] == ["color", "alpha"]
        ws = [np.full_like(w, float(k // 2 + 1)) if k % 2 == 0 else np.full_like(w, 100.0 + k // 2) for k, w in enumerate(model.get_weights())]
        model.set_weights(ws)
        blob = model.get_blob()
        d = model.desc()
        npk = _lib.lib.ntx_packed_count(C.byref(d))
        out = np.empty(npk, np.float32)
        fp = C.POINTER(C.c_float)
        assert _lib.lib.ntx_pack_weights(C.byref(d), blob.ctypes.data_as(fp), blob.size, out.ctypes.data_as(fp), npk) == 0
        aux = out[npk - 3776:]
        k_alpha, k_rgb = names.index("alpha"), names.index("color")
        # aux block (ntx_layout.h): 12 x 256 biases | alpha head [2][128] + bias | rgb head [3][2][64] + bias[3]
        assert np.all(aux[12 * 256:12 * 256 + 256] == k_alpha + 1) and aux[12 * 256 + 256] == 100.0 + k_alpha
        rgb_off = 12 * 256 + 2 * 128 + 4
        assert np.all(aux[rgb_off:rgb_off + 384] == k_rgb + 1) and np.all(aux[rgb_off + 384:rgb_off + 387] == 100.0 + k_rgb)
        assert np.all(aux[8 * 256:9 * 256] == 100.0 + names.index("feature"))            # bias slot 8 = the feature layer
        stream = out[:npk - 3776]
        first = stream[:2 * 256]                                                          # L0's first records: trunk0's kernel
        assert set(np.unique(first[first != 0]).tolist()) == {1.0}


def test_reference_render_config_runs_through_remap():
    """A reference config file (if the reference tree is present) maps onto this package's modules."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree only exists in the build container")
    sys.path.insert(0, ref)
    try:
        import importlib
        cfg = importlib.import_module("configs.config_carpet_render").config
    finally:
        sys.path.remove(ref)
    from nerf_tex_amd import util
    m = util.remap_reference_config(cfg)
    assert m.module == "nerf_tex_amd.render.Render"
    assert m.test_dataset_config.module == "nerf_tex_amd.dataset.Dataset"
    assert m.test_dataset_config.proxy_config.module == "nerf_tex_amd.proxy.AABB"
    assert m.renderer_config.module == "nerf_tex_amd.renderer.InstanceRenderer"
    assert m.renderer_config.instancer_config.module == "instancer.instancer.Instancer"   # the Embree instancer stays the reference's
    assert m.logger_config.module == "network.logger.Logger"                               # out of scope: left untouched


def test_renderer_kwargs_mirror_reference():
    import inspect
    from nerf_tex_amd.renderer import Renderer
    sig = inspect.signature(Renderer.__init__)
    for k, dflt in [("model_fine", None), ("n_samples", 64), ("n_importance", 0), ("perturb", True), ("raw_noise_std", 0),
                    ("render_chunk", 32768), ("net_chunk", 65536), ("downsampling_factor", 1), ("blur_idx", None), ("map_exr", False)]:
        assert sig.parameters[k].default == dflt                               # renderer.py:34
    call = inspect.signature(Renderer.__call__)
    assert list(call.parameters)[1:6] == ["rays_o", "rays_d", "t", "parameters", "cone_scale"]   # renderer.py:47
    assert call.parameters["composite_bkgd"].default is False and call.parameters["training"].default is True
    with pytest.raises(NotImplementedError):
        Renderer(model=None, raw_noise_std=1.0)


def test_shard_map_partitions_and_matches_the_c_abi():
    """ShardMap (python) == ntx_shard_count (C ABI); every pixel belongs to exactly one rank; bands are contiguous."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.dist import ShardMap, shard_range
    for n in (0, 1, 7, 800 * 800, 640001, 1600 * 1600):
        for world in (1, 2, 3, 8):
            for run in (None, 1, 5, 800, 1600):
                m = ShardMap(n, world, run)
                counts = [m.count(r) for r in range(world)]
                assert sum(counts) == n and counts[0] == m.capacity == max(counts)
                for r in range(world):
                    assert _lib.lib.ntx_shard_count(n, m.run, world, r) == counts[r]
                if n <= 10000 or run in (800, 1600):
                    seen = np.concatenate([m.local_pixels(r) for r in range(world)]) if n else np.zeros(0, np.int64)
                    assert np.array_equal(np.sort(seen), np.arange(n))
                    for r in range(world):
                        p0, cnt, rl, rs = m.pixel_set(r)
                        k = np.arange(cnt)
                        assert np.array_equal(m.local_pixels(r), p0 + (k // rl) * rs + k % rl)
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, c1) in zip(spans[:-1], spans[1:]):
                assert f0 + c0 == f1 or c1 == 0
    assert ShardMap(640000, 8).count(3) == 80000 and ShardMap(1600 * 1600, 8, 1600).count(7) == 320000   # BASELINE configs[3] / [4]
    assert _lib.lib.ntx_shard_count(10, 0, 2, 0) == -1 and _lib.lib.ntx_shard_count(10, 4, 2, 2) == -1


def test_unshard_formula_of_the_gather_kernel():
    """ntx_unshard_kernel (ntx_comm.hip) on the root: image[p] = staging[(q % R) * cap + (q / R) * L + p % L], q = p / L, with
    every rank's shard at staging[rank * cap ...] in its local ray order.  Emulated here on the shard maps of BASELINE
    configs[3] / [4] and on ragged ones (no multi-GPU box is available to run the kernel with R > 1)."""
    from nerf_tex_amd.dist import ShardMap
    for n, R, L in [(800 * 800, 8, 800), (1600 * 1600, 8, 1600), (1000, 3, 7), (37, 4, 5), (640000, 8, None), (101, 2, None)]:
        m = ShardMap(n, R, L)
        cap = m.capacity
        full = np.arange(n, dtype=np.int64) * 3 + 1
        staging = np.full(R * cap, -1, np.int64)
        for r in range(R):
            staging[r * cap: r * cap + m.count(r)] = full[m.local_pixels(r)]
        p = np.arange(n)
        q = p // m.run
        assert np.array_equal(staging[(q % R) * cap + (q // R) * m.run + p % m.run], full)


def _gather_worker(rank, world, port, n_total, run, q):
    import torch
    import torch.distributed as dist
    from nerf_tex_amd.dist import ShardMap, gather_image
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = ShardMap(n_total, world, run)
    full = torch.arange(n_total * 4, dtype=torch.float32).reshape(n_total, 4)
    img = gather_image(full[torch.as_tensor(shard.local_pixels(rank))].clone(), shard)
    ok = (img is None) if rank != 0 else bool(torch.equal(img, full))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,run", [(64, None), (101, None), (96, 8), (101, 7)])
def test_gather_image_world2_gloo(n_total, run):
    """N > 1 path on CPU: bands or interleaved runs, even or uneven, one gather, image on rank 0 equals the unsharded one."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_total, run, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_bench_self_launch_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (one per GPU, rendezvous on 127.0.0.1)."""
    import importlib
    bench = importlib.import_module("bench")
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env, stdout))
        def wait(self):
            return 0
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    args = type("A", (), {"gpus": 4})()
    assert bench.self_launch(args) == 0 and len(started) == 4
    for r, (cmd, env, out) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "2"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4" and env["MASTER_ADDR"] == "127.0.0.1"
        assert env["MASTER_PORT"] == started[0][1]["MASTER_PORT"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert (out is None) == (r == 0)


def test_checkpoint_reader_round_trip(tmp_path):
    """nerf_tex_amd/checkpoint.py against the independent writer in tests/bundle_writer.py:
    multi-block table, prefix-compressed keys, Keras object-graph names `layer_with_weights-k` in `model.layers` order."""
    from nerf_tex_amd import checkpoint as ck
    from nerf_tex_amd.model import ParamNerf
    from tests.bundle_writer import write_bundle
    from tests.common import EMB
    assert ck.crc32c(b"123456789") == 0xE3069283                      # the standard CRC-32C check value
    np.random.seed(3)
    src = ParamNerf(EMB(10), EMB(4), EMB(4), [1, 6])["model"]
    ws = src.get_weights()
    names = [n for n, _, _ in src.layer_table()]
    # model.layers order of a functional Keras model is by depth: trunk0-7, feature, colour layers, then the two heads
    order = [n for n in names if n.startswith("trunk")] + ["feature", "color_hidden0", "color_half", "color", "alpha"]
    assert order == names
    tensors = {"save_counter/.ATTRIBUTES/VARIABLE_VALUE": np.asarray(7, np.int64), "step/.ATTRIBUTES/VARIABLE_VALUE": np.asarray(5000, np.int64)}
    for i, n in enumerate(order):
        k = names.index(n)
        tensors[f"model/layer_with_weights-{i}/kernel/.ATTRIBUTES/VARIABLE_VALUE"] = ws[2 * k]
        tensors[f"model/layer_with_weights-{i}/bias/.ATTRIBUTES/VARIABLE_VALUE"] = ws[2 * k + 1] + np.float32(0.01 * i)
        tensors[f"optimizer/iter/{i}"] = np.asarray([i], np.int32)
    d = tmp_path / "checkpoints"; d.mkdir()
    write_bundle(str(d / "ckpt-1000"), {"model/layer_with_weights-0/bias/.ATTRIBUTES/VARIABLE_VALUE": ws[1]})
    write_bundle(str(d / "ckpt-5000"), tensors)
    assert ck.latest_checkpoint(str(d)).endswith("ckpt-5000")
    got = ck.read_bundle(str(d / "ckpt-5000"))
    assert set(got) == set(tensors) and int(got["step/.ATTRIBUTES/VARIABLE_VALUE"]) == 5000
    dst = ParamNerf(EMB(10), EMB(4), EMB(4), [1, 6])["model"]
    ck.load_checkpoint(dst, str(d))
    for k, (a, b) in enumerate(zip(dst.get_weights(), ws)):
        if k % 2 == 0:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_array_equal(a, b + np.float32(0.01 * order.index(names[k // 2])))
    # integrity: a flipped data byte is caught by the tensor checksum, a flipped index byte by the block checksum
    p = str(d / "ckpt-5000.data-00000-of-00001"); raw = bytearray(open(p, "rb").read()); raw[100] ^= 1; open(p, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        ck.read_bundle(str(d / "ckpt-5000"))
    with pytest.raises(KeyError):
        ck.model_weights_from_bundle(got, dst.layer_table(), root="model_fine")
    other = ParamNerf(EMB(10), EMB(4), EMB(4), [1, 4])["model"]
    with pytest.raises(KeyError):
        ck.model_weights_from_bundle(got, other.layer_table())


def test_main_entry_point_prepares_reference_configs():
    """nerf_tex_amd.main (reference: main.py): config file -> remapped config; `--volumetric` swaps the Embree-backed
    InstanceRenderer for the volumetric Renderer; training configs are refused."""
    from nerf_tex_amd import main as m
    cfg = m.prepare(m.load_config(os.path.join(ROOT, "configs", "example_carpet_render.py")))
    assert cfg.module == "nerf_tex_amd.render.Render" and cfg.renderer_config.module == "nerf_tex_amd.renderer.Renderer"
    assert "seed" not in cfg and len(cfg.test_dataset_config.data_loader_config.views) == 2
    with pytest.raises(NotImplementedError):
        m.prepare({"module": "network.train.Train"})
    ref = "/root/reference/configs/config_carpet_render.py"
    if os.path.exists(ref):
        sys.path.insert(0, "/root/reference")
        try:
            raw = m.load_config(ref)
        finally:
            sys.path.remove("/root/reference")
        assert raw.renderer_config.module == "network.renderer.InstanceRenderer"
        vol = m.prepare(raw, volumetric=True)
        assert vol.renderer_config.module == "nerf_tex_amd.renderer.Renderer" and "instancer_config" not in vol.renderer_config
        assert m.prepare(raw).renderer_config.module == "nerf_tex_amd.renderer.InstanceRenderer"


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/nerftex.h is a C99 header (no C++ / torch types) and a C program links against libnerftex_hip.so: the drop-in
    boundary is a C ABI, ctypes is only one of its clients."""
    import shutil
    import subprocess
    from nerf_tex_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "nerftex.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    ntx_model_desc d = {NTX_MODEL_PARAMNERF, 1, 6, 3, 10, 4, 4, 8, 256, 4, 1, NTX_POS_FOURIER};\n'
                   '    printf("%d %zu %lld\\n", ntx_abi_version(), ntx_weight_count(&d), (long long)ntx_shard_count(640000, 800, 8, 3));\n'
                   '    return ntx_abi_version() == NTX_ABI_VERSION ? 0 : 1;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["2", "683524", "80000"]

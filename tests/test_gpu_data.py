"""The training loop fed the way the reference feeds it (train.py:20-26, 60): `network.dataset.Dataset` over a TFRecord file, `pixel_sampler.Proxy`,
`ray_sampler.Proxy`, the proxy box -- the shipped carpet training config's own blocks (tests/golden/train_configs.json, made from the reference's
config module), pointed at a small file made here from a teacher network's renders."""

import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
H = W = 64


def dev():
    return torch.device("cuda", 0)


def teacher_set(root, n_views=6, S=48):
    """A NeRF-layout folder of a teacher network's renders (uint8 RGBA PNGs, un-premultiplied as logger.py:133-144 writes them) from cameras
    on a ring, the seven carpet parameters as `driver_parameters`; converted to TFRecord shards.  Returns (tfr dir, the uint8 images, poses)."""
    from nerf_tex_amd import dataset as D, png, synthetic, tfrecord
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.layer import FourierFeatures
    from nerf_tex_amd.render import image_epilogue, render_image
    from nerf_tex_amd.renderer import Renderer
    cfg = json.load(open(os.path.join(GOLDEN, "train_configs.json")))["carpet"]
    box = cfg["train_dataset_config"]["proxy_config"]
    teacher = ParamNerf(FourierFeatures(10), FourierFeatures(4), FourierFeatures(4), [1, 6])["model"]
    teacher.set_blob(synthetic.synthetic_weights(teacher.layer_table(), seed=2, dense_media=True))
    params = [1.0, 1.0, 1.0, 0.1, 0.0, -0.707, 0.707]
    views = [{"pose": D.look_at(np.asarray([np.cos(a), np.sin(a), 0.6]) * 5), "parameters": params} for a in np.linspace(0, 2 * np.pi, n_views, endpoint=False)]
    ds = D.Dataset({"module": "nerf_tex_amd.dataset.FromViews", "views": views, "height": H, "width": W, "angle": 0.63},
                   {"module": "network.pixel_sampler.Full"}, {"module": "network.ray_sampler.Proxy"}, dict(box), n_epochs=1, device=dev())
    renderer = Renderer(model=teacher, n_samples=S, perturb=False)
    os.makedirs(os.path.join(root, "nerf", "train"))
    imgs, frames = [], []
    for k, data in enumerate(ds):
        rgba = render_image(renderer, ds, data)[0]
        u8 = image_epilogue(rgba, uint8=True)[1].cpu().numpy()
        png.write_png(os.path.join(root, "nerf", "train", f"r_{k:03d}.png"), u8)
        imgs.append(u8)
        frames.append({"transform_matrix": views[k]["pose"].tolist(), "driver_parameters": {f"p{i}": v for i, v in enumerate(params)}})
    with open(os.path.join(root, "nerf", "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": 0.63, "frames": frames}, f)
    tfrecord.convert_folder(os.path.join(root, "nerf"), os.path.join(root, "tfr"), imgs_per_shard=4)
    return os.path.join(root, "tfr"), imgs, [v["pose"] for v in views], teacher


def carpet_blocks(tfr_path, **over):
    cfg = json.load(open(os.path.join(GOLDEN, "train_configs.json")))["carpet"]
    train = json.loads(json.dumps(cfg["train_dataset_config"]))
    assert train["data_loader_config"] == {"module": "network.dataset.TFRecord", "tfr_path": "datasets/materials/carpet/tfr/train.tfr"}
    train["data_loader_config"]["tfr_path"] = tfr_path
    train.update(over)
    val = json.loads(json.dumps(cfg["val_dataset_config"]))
    val["data_loader_config"].update(height=H, width=W)                               # 256 x 256 in the config: smaller here
    return cfg, train, val


def test_batches_of_a_tfrecord_dataset(tmp_path):
    """What `Dataset` hands the step: per view `n_samples` pixels among those whose rays hit the proxy (pixel_sampler.Proxy), their rays, and
    the image's premultiplied colour / alpha AT those pixels -- recovered here from the rays themselves (a ray's direction names its pixel)."""
    from nerf_tex_amd import util
    tfr, imgs, poses, _ = teacher_set(str(tmp_path))
    _, train, _ = carpet_blocks(tfr, seed=3)
    ds = util.instantiate(dict(train, device=dev()))
    assert type(ds).__module__ == "nerf_tex_amd.dataset" and ds.has_images and len(ds.views) == 6
    assert (ds.height, ds.width, ds.n_samples, ds.n_parameters, ds.batchsize) == (H, W, 256, 7, 4)
    assert abs(ds.focal - W / np.tan(float(np.float32(0.63)) / 2) / 2) < 1e-9
    seen = []
    for b in ds.take(5):                                                               # endless (n_epochs None): the loop takes what it needs
        assert {k: tuple(v.shape) for k, v in b.items()} == {"parameters": (4, 7), "rays_o": (4, 256, 3), "rays_d": (4, 256, 3), "t": (4, 256, 2),
                                                             "cone_scale": (4, 256, 1), "color": (4, 256, 3), "alpha": (4, 256)}
        assert all(v.is_cuda and v.dtype == torch.float32 for v in b.values())
        assert torch.isfinite(b["t"]).float().mean() > 0.9                             # chosen among the pixels that hit ON THE 8x COARSER GRID (pixel_sampler.py:40-59): a few at the rim miss
        for e in range(4):
            o = b["rays_o"][e, 0].cpu().numpy()
            k = int(np.argmin([np.abs(p[:3, 3] - o).max() for p in poses]))            # which view: its camera position
            seen.append(k)
            c2w = torch.as_tensor(poses[k], device=dev())
            d_cam = b["rays_d"][e] @ c2w[:3, :3]                                       # back to camera space: (j + .5 - W/2) / f, -(i + .5 - H/2) / f, -1, normalised
            d_cam = d_cam / -d_cam[:, 2:3]
            j = torch.round(d_cam[:, 0] * ds.focal + W / 2 - 0.5).long()
            i = torch.round(-d_cam[:, 1] * ds.focal + H / 2 - 0.5).long()
            assert len({(int(a), int(c)) for a, c in zip(i.tolist(), j.tolist())}) == 256   # a permutation's head: no pixel twice
            px = torch.as_tensor(imgs[k], device=dev())[i, j].float() * torch.tensor(1.0 / 255)
            assert torch.equal(b["color"][e], px[:, :3] * px[:, 3:]) and torch.equal(b["alpha"][e], px[:, 3])
    assert len(set(seen[:6])) == 6 and sorted(seen[:6]) == list(range(6))             # an epoch holds every view once (buffer 100 > 6), batches cut across
    one = list(util.instantiate(dict(train, device=dev())).take(4))                     # a batch made in one piece is the batch made view by view
    per_view = list(util.instantiate(dict(train, device=dev(), fused_batches=False)).take(4))
    assert all(torch.equal(a[k], b[k]) for a, b in zip(one, per_view) for k in a) and set(one[0]) == set(per_view[0])
    again = [float(b["color"].sum()) for b in util.instantiate(dict(train, device=dev())).take(3)]
    assert again == [float(b["color"].sum()) for b in util.instantiate(dict(train, device=dev())).take(3)]   # the seed decides pixels and order


def test_train_from_the_config_s_own_dataset_blocks(tmp_path):
    """`Train(**config)` with the shipped carpet config's blocks as written -- TFRecord dataset (pointed at the file made here), Proxy pixel
    and ray samplers in its box, batches of 4 x 256 rays, GenerateData validation views from `data.distribution.Constant`, ParamNerf,
    AlphaLoss(smape, mse), the renderer block (fewer samples), Adam + decay: the loss falls, the validation view moves towards the teacher's
    render of that camera, checkpoints and validation PNGs are where the Logger puts them, and the run resumes."""
    from nerf_tex_amd import dataset as D, png
    from nerf_tex_amd.render import render_image
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Train
    tfr, imgs, poses, teacher = teacher_set(str(tmp_path))
    cfg, train, val = carpet_blocks(tfr, seed=1)
    rcfg = dict(cfg["renderer_config"], n_samples=48)
    model_config = {k: v for k, v in cfg["model_config"].items()}
    torch.manual_seed(0); np.random.seed(0)                                          # the initial weights are numpy draws (main.py:30 seeds them from the config)
    kw = dict(train_dataset_config=train, val_dataset_config=val, model_config=model_config, loss_config=cfg["loss_config"], lrate=cfg["lrate"],
              lrate_decay=cfg["lrate_decay"], renderer_config=rcfg)
    out = Train(str(tmp_path / "run"), n_iters=400, logger_config={"module": "network.logger.Logger", "i_print": 20, "i_img": 200, "i_checkpoint": 200}, **kw)
    losses = [v for _, v in out["loss"]]
    assert out["step"] == 400 and np.isfinite(losses).all() and np.mean(losses[-3:]) < 0.7 * np.mean(losses[:3]), losses
    assert sorted(out["images"]) == [200, 400] and len(out["images"][400]) == 2        # the config's two validation parameter sets
    assert sorted(os.listdir(tmp_path / "run" / "media" / "validation")) == ["200", "400"]
    assert np.asarray(png.read_png(str(tmp_path / "run" / "media" / "validation" / "400" / "1.png"))).shape[:2] == (H, W)
    assert sorted(f for f in os.listdir(tmp_path / "run" / "checkpoints") if f.endswith(".index")) == ["ckpt-200.index", "ckpt-400.index"]
    # the validation camera is the config's own ([0.47, -0.65, 0.6] x 5); its second parameter set is the one the teacher's images were made with
    vds = D.Dataset(dict(val["data_loader_config"]), {"module": "network.pixel_sampler.Full"}, {"module": "network.ray_sampler.Proxy"}, dict(val["proxy_config"]),
                    n_epochs=1, device=dev())
    view = list(vds)[1]
    want = render_image(Renderer(model=teacher, n_samples=48, perturb=False), vds, view)[0]
    err = lambda img: float((img - want).abs().mean())
    assert err(out["images"][400][1]) < 0.8 * err(out["images"][200][1]) or err(out["images"][400][1]) < 0.05, (err(out["images"][200][1]), err(out["images"][400][1]))
    more = Train(str(tmp_path / "run"), n_iters=420, logger_config={"i_print": 10, "i_img": 0, "i_checkpoint": 0}, **kw)
    assert more["step"] == 420 and more["trainer"].iterations == 420 and [s for s, _ in more["loss"]] == [410, 420]
    # CheckpointManager's two rules (logger.py:34): the newest `max_to_keep`, and beyond them one every `keep_every_n_hours` (0: every one)
    Train(str(tmp_path / "run"), n_iters=450, logger_config={"i_print": 0, "i_img": 0, "i_checkpoint": 10, "max_to_keep": 1, "keep_every_n_hours": 0}, **kw)
    assert sorted(f for f in os.listdir(tmp_path / "run" / "checkpoints") if f.endswith(".index")) == [f"ckpt-{k}.index" for k in (410, 420, 430, 440, 450)]   # (resumed from 400: the run to 420 wrote none)
    Train(str(tmp_path / "run"), n_iters=470, logger_config={"i_print": 0, "i_img": 0, "i_checkpoint": 10, "max_to_keep": 1}, **kw)
    # the resumed run rotates what the `checkpoint` file lists (ckpt-450, the last run's newest max_to_keep) and what it saves itself; the four
    # the last run kept for good are not in that list and stay (globbing the directory deleted them)
    assert sorted(f for f in os.listdir(tmp_path / "run" / "checkpoints") if f.endswith(".index")) == [f"ckpt-{k}.index" for k in (410, 420, 430, 440, 470)]


def test_two_ranks_run_the_loop_data_parallel(tmp_path):
    """`Train` under torch.distributed (one process per GPU; here two ranks sharing GPU 0 on gloo): rank 0's state is everybody's start,
    every rank draws its own batches (seed + rank), the gradients are averaged each step, only rank 0 logs and checkpoints
    (tests/_dp_loop_worker.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29657",
                          os.path.join(root, "tests", "_dp_loop_worker.py"), str(tmp_path / "run")], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "DP_LOOP_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_main_runs_a_training_config_file(tmp_path):
    """`python -m nerf_tex_amd.main <config>` on a TRAINING config written like the shipped ones (main.py:17-50): the seed set, the target
    folder made with a copy of the config as `config_train.py`, `network.train.Train` instantiated with the file's blocks."""
    from nerf_tex_amd import main as m
    tfr, _, _, _ = teacher_set(str(tmp_path), n_views=4)
    cfg, train, val = carpet_blocks(tfr)
    config = {"module": "network.train.Train", "target_path": str(tmp_path / "out"), "seed": 0, "override": True,
              "train_dataset_config": train, "val_dataset_config": val, "model_config": cfg["model_config"], "loss_config": cfg["loss_config"],
              "n_iters": 30, "lrate": cfg["lrate"], "lrate_decay": cfg["lrate_decay"], "renderer_config": dict(cfg["renderer_config"], n_samples=32),
              "logger_config": {"module": "network.logger.Logger", "i_print": 10, "i_img": 30, "i_checkpoint": 30}}
    path = str(tmp_path / "config_tiny_train.py")
    with open(path, "w") as f:
        f.write("config = " + repr(config) + "\n")
    out = m.main([path])
    assert out["step"] == 30 and [s for s, _ in out["loss"]] == [10, 20, 30] and sorted(out["images"]) == [30]
    assert os.path.exists(tmp_path / "out" / "config_train.py") and os.path.exists(tmp_path / "out" / "checkpoints" / "ckpt-30.index")
    assert os.path.exists(tmp_path / "out" / "media" / "validation" / "30" / "0.png")


def test_the_example_training_config_runs(tmp_path, monkeypatch):
    """README's quick start for training: `python tools/make_example_dataset.py` then `python -m nerf_tex_amd.main configs/example_carpet_train.py`
    (fewer views, a smaller image and 40 steps here): the config file as it is, but for `n_iters` and the Logger's cadences."""
    import importlib.util
    from nerf_tex_amd import main as m
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_example_dataset", os.path.join(root, "tools", "make_example_dataset.py"))
    maker = importlib.util.module_from_spec(spec); spec.loader.exec_module(maker)
    monkeypatch.chdir(tmp_path)
    files, _ = maker.make("datasets/example_carpet", n_views=5, size=64, n_samples=48)
    assert [os.path.basename(f) for f in files] == ["train.tfr"]
    raw = m.load_config(os.path.join(root, "configs", "example_carpet_train.py"))
    assert raw.train_dataset_config.data_loader_config.tfr_path == "datasets/example_carpet/tfr" and raw.module == "network.train.Train"
    raw.n_iters = 40; raw.logger_config.update(i_print=20, i_img=40, i_checkpoint=40)
    raw.val_dataset_config.data_loader_config.update(height=32, width=32)
    np.random.seed(raw.seed)
    from nerf_tex_amd import util
    out = util.instantiate(m.prepare(raw))
    assert out["step"] == 40 and len(out["loss"]) == 2 and np.isfinite([v for _, v in out["loss"]]).all()
    assert os.path.exists("logs/example_carpet_train/checkpoints/ckpt-40.index") and os.path.exists("logs/example_carpet_train/media/validation/40/0.png")


def test_independent_pixels_and_frustum_rays_over_a_folder(tmp_path):
    """The other samplers on the same path (pixel_sampler.Independent: iid pixels, misses included; ray_sampler.Frustum: near / far instead of
    a proxy) over `network.dataset.FileFolder`, view by view (no proxy hits to keep: the per-view path), with the background composited in."""
    from nerf_tex_amd import png, util
    rng = np.random.default_rng(2)
    os.makedirs(tmp_path / "nerf" / "train")
    imgs, frames = [], []
    from nerf_tex_amd.dataset import look_at
    for k in range(3):
        imgs.append(rng.integers(0, 256, (40, 56, 4), dtype=np.uint8))
        png.write_png(str(tmp_path / "nerf" / "train" / f"r_{k}.png"), imgs[-1])
        frames.append({"transform_matrix": look_at(np.asarray([np.cos(k), np.sin(k), 0.5]) * 4).tolist(), "driver_parameters": {"a": 0.5, "b": float(k)}})
    json.dump({"camera_angle_x": 0.7, "frames": frames}, open(tmp_path / "nerf" / "transforms_train.json", "w"))
    ds = util.instantiate({"module": "network.dataset.Dataset",
                           "data_loader_config": {"module": "network.dataset.FileFolder", "imgs_path": str(tmp_path / "nerf" / "train"),
                                                  "poses_path": str(tmp_path / "nerf" / "transforms_train.json"), "idxs": [0, 1, 2], "composite_bkgd": True, "bkgd_color": [0.2, 0.4, 1.0]},
                           "pixel_sampler_config": {"module": "network.pixel_sampler.Independent", "n_samples": 300},
                           "ray_sampler_config": {"module": "network.ray_sampler.Frustum", "near": 2.0, "far": 6.0},
                           "n_epochs": 1, "batchsize": 2, "device": dev(), "seed": 4})
    assert (ds.height, ds.width, ds.n_samples, ds.n_parameters, ds.composite_bkgd) == (40, 56, 300, 2, True) and not ds._can_fuse(dev())
    batches = list(ds)
    assert [b["color"].shape[0] for b in batches] == [2, 1] and batches[0]["t"].shape == (2, 300, 2)
    assert torch.equal(batches[0]["t"][..., 0], torch.full((2, 300), 2.0, device=dev())) and torch.equal(batches[0]["t"][..., 1], torch.full((2, 300), 6.0, device=dev()))
    k = 0
    for b in batches:
        for e in range(b["color"].shape[0]):
            c2w = torch.as_tensor(np.asarray(frames[k]["transform_matrix"], np.float32), device=dev())
            d_cam = b["rays_d"][e] @ c2w[:3, :3]
            d_cam = d_cam / -d_cam[:, 2:3]
            j = torch.round(d_cam[:, 0] * ds.focal + 56 / 2 - 0.5).long(); i = torch.round(-d_cam[:, 1] * ds.focal + 40 / 2 - 0.5).long()
            assert int(i.min()) >= 0 and int(i.max()) < 40 and int(j.min()) >= 0 and int(j.max()) < 56
            px = torch.as_tensor(imgs[k], device=dev())[i, j].float() * torch.tensor(1.0 / 255)
            want = px[:, :3] * px[:, 3:] + (1 - px[:, 3:]) * torch.tensor([0.2, 0.4, 1.0], device=dev())
            assert torch.equal(b["color"][e], want) and torch.equal(b["alpha"][e], px[:, 3]) and float(b["parameters"][e, 1]) == float(k)
            k += 1


@pytest.mark.parametrize("fam", ["carpet", "fur", "grass", "grass_filtered", "plush"])
def test_every_shipped_training_config_runs_from_its_blocks(tmp_path, fam):
    """All five shipped training configs (tests/golden/train_configs.json: the reference's config modules, evaluated): `Train(**config)`
    with every block as written -- the TFRecord dataset (its `tfr_path` pointed at a small file of random images carrying the family's number
    of parameters), the Proxy samplers in the family's box, batch 4 x 256 rays, the generated validation views (rendered at 48 x 48), the
    model, loss, schedule and renderer blocks incl. grass_filtered's blur_idx 0 and raw_noise_std 0.1 -- for six steps at 256 samples a ray."""
    from nerf_tex_amd import dataset as D, tfrecord
    from nerf_tex_amd.train import Train
    cfg = json.load(open(os.path.join(GOLDEN, "train_configs.json")))[fam]
    P = sum(cfg["model_config"]["n_parameters"])
    rng = np.random.default_rng(3)
    from nerf_tex_amd import png
    recs = []
    for k in range(5):
        pose = D.look_at(np.asarray([np.cos(k), np.sin(k), 0.7]) * 6)
        recs.append(tfrecord.make_example({"image": png.encode_png(rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)), "pose": tfrecord.serialize_tensor(pose), "angle": 0.6,
                                           "parameters": tfrecord.serialize_tensor(rng.uniform(0, 1, P).astype(np.float32))}))
    tfrecord.write_records(str(tmp_path / "train.tfr"), recs)
    train = json.loads(json.dumps(cfg["train_dataset_config"])); train["data_loader_config"]["tfr_path"] = str(tmp_path / "train.tfr"); train["seed"] = 0
    val = json.loads(json.dumps(cfg["val_dataset_config"])); val["data_loader_config"].update(height=48, width=48)
    np.random.seed(0)
    out = Train(str(tmp_path / "run"), train_dataset_config=train, val_dataset_config=val, model_config=cfg["model_config"], loss_config=cfg["loss_config"],
                n_iters=6, lrate=cfg["lrate"], lrate_decay=cfg["lrate_decay"], renderer_config=cfg["renderer_config"],
                logger_config=dict(cfg["logger_config"], i_print=2, i_img=6, i_checkpoint=6))
    tr = out["trainer"]
    assert out["step"] == 6 and tr.iterations == 6 and tr.n_samples == 256 and tr.max_rays == 1024 and np.isfinite([v for _, v in out["loss"]]).all() and len(out["loss"]) == 3
    assert (tr.blur_idx, tr.raw_noise_std) == ((0, 0.1) if fam == "grass_filtered" else (None, 0.0))
    assert len(out["images"][6]) == cfg["val_views_reference"]["n"] and all(tuple(im.shape) == (48, 48, 4) and bool(torch.isfinite(im).all()) for im in out["images"][6])
    assert os.path.exists(tmp_path / "run" / "checkpoints" / "ckpt-6.index")


@pytest.mark.parametrize("path", ["volume", "instanced"])
def test_two_ranks_render_one_image(tmp_path, path):
    """`Render` under torch.distributed (one process per GPU; here two ranks sharing GPU 0 on gloo): the ranks render bands of each view, rank 0
    gathers and writes; the image is the one-GPU image bit for bit, with the reference's default jitter on -- and through `InstanceRenderer` with
    random patch choices, whose draws are keyed by a ray's place among the image's proxy hits (tests/_dp_render_worker.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29659",
                          os.path.join(root, "tests", "_dp_render_worker.py"), str(tmp_path / "run"), path], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "DP_RENDER_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_main_under_torch_distributed_run(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m nerf_tex_amd.main configs/example_carpet_render.py` (both ranks on GPU 0 over gloo,
    the development knob): the entry point joins the process group, `Render` shards each view, rank 0 writes the two images."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29661",
                          "-m", "nerf_tex_amd.main", os.path.join(root, "configs", "example_carpet_render.py")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NTX_MAIN_SHARE_GPU="1", PYTHONPATH=root))
    assert out.returncode == 0 and "rendered 2 image(s)" in out.stdout and "rendered 0 image(s)" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    assert sorted(os.listdir(tmp_path / "logs" / "example_carpet" / "media" / "test")) == ["0.npy", "0.png", "1.npy", "1.png"]
    assert np.load(tmp_path / "logs" / "example_carpet" / "media" / "test" / "1.npy").shape == (1, 128, 128, 4)


def test_the_library_loads_before_torch_in_a_fresh_process():
    """PyTorch-ROCm bundles its own libamdhip64 and this library names /opt/rocm's, one SONAME: mapped first, the library's copy left torch with
    "No HIP GPUs are available" (a fresh `python -m nerf_tex_amd.main <training config>` makes the trainer before anything touched torch).
    `_lib` imports torch before it loads the library; a process that starts with the package must still have its GPU in torch."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from nerf_tex_amd import _lib\n"
            "from nerf_tex_amd.layer import FourierFeatures\nfrom nerf_tex_amd.model import ParamNerf\nfrom nerf_tex_amd.train import Trainer\n"
            "tr = Trainer(ParamNerf(FourierFeatures(10), FourierFeatures(4), FourierFeatures(4), [1, 6])['model'], max_rays=64, n_samples=16)\n"
            "import torch\nprint('ORDER_OK', torch.cuda.device_count(), float(torch.ones(3, device='cuda').sum()))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ORDER_OK 1 3.0" in out.stdout or "ORDER_OK" in out.stdout and " 3.0" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])

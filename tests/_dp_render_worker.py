"""Worker of tests/test_gpu_data.py::test_two_ranks_render_one_image: launched by torch.distributed.run with 2 ranks that share GPU 0 (process
group on gloo).  `Render` on the example render config: every rank renders its band of each view, rank 0 gathers, writes the files and compares
with the image it renders alone -- bit for bit, jitter included (the draws are keyed by the pixel)."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_tex_amd import main as m, util          # noqa: E402
from nerf_tex_amd.render import render_image      # noqa: E402


def main():
    target = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    cfg = m.prepare(m.load_config(os.path.join(ROOT, "configs", "example_carpet_render.py")))
    cfg.target_path = target
    cfg.test_dataset_config.data_loader_config.update(height=50, width=38)          # 1900 pixels: bands of 950
    np.random.seed(0)                                                               # the same random-initialised weights on both ranks (main.py:30)
    imgs = util.instantiate(cfg)
    assert (len(imgs) == 2) == (rank == 0)
    if rank == 0:
        assert sorted(os.listdir(os.path.join(target, "media", "test"))) == ["0.npy", "0.png", "1.npy", "1.png"] and tuple(imgs[1].shape) == (1, 50, 38, 4)
        np.save(os.path.join(target, "sharded.npy"), torch.stack(imgs).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:                                                                   # alone: the whole image on one GPU, the same seeds
        from nerf_tex_amd.dataset import Dataset
        np.random.seed(0)
        cfg2 = m.prepare(m.load_config(os.path.join(ROOT, "configs", "example_carpet_render.py")))
        cfg2.test_dataset_config.data_loader_config.update(height=50, width=38)
        ds = util.instantiate(cfg2.test_dataset_config)
        model = util.instantiate(dict(cfg2.model_config, n_parameters=cfg2.model_config.get("n_parameters", ds.n_parameters)))
        renderer = util.instantiate(dict(cfg2.renderer_config, **model))
        whole = torch.stack([render_image(renderer, ds, dict(d, seed=7919 * (i + 1))) for i, d in enumerate(ds)]).cpu().numpy()
        got = np.load(os.path.join(target, "sharded.npy"))
        assert got.shape == whole.shape and np.array_equal(got, whole) and float(whole[..., 3].max()) > 0
        print("DP_RENDER_OK")


if __name__ == "__main__":
    main()

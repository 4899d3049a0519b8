"""Worker of tests/test_gpu_data.py::test_two_ranks_run_the_loop_data_parallel: launched by torch.distributed.run with 2 ranks that share GPU 0
(process group on gloo).  Each rank runs `Train` on the carpet config's blocks over its own draws of an image dataset; the ranks start from
rank 0's weights (theirs differ), average their gradients every step, end with the same weights bit for bit, and only rank 0 writes files."""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_tex_amd import dataset as D                     # noqa: E402
from nerf_tex_amd.train import Train                      # noqa: E402


def main():
    target = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "train_configs.json")))["carpet"]
    rng = np.random.default_rng(5)                           # the same images on both ranks
    H = W = 64
    views = [{"pose": D.look_at(np.asarray([np.cos(a), np.sin(a), 0.6]) * 5), "parameters": [1, 1, 1, .1, 0, -.707, .707],
              "rgba": rng.integers(0, 256, (H, W, 4), dtype=np.uint8)} for a in np.linspace(0, 5, 6)]
    train = json.loads(json.dumps(cfg["train_dataset_config"]))
    train["data_loader_config"] = {"module": "nerf_tex_amd.dataset.FromViews", "views": views, "height": H, "width": W, "angle": 0.63}
    train.update(batchsize=2, seed=11)
    train["pixel_sampler_config"]["n_samples"] = 128
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.layer import FourierFeatures
    table = ParamNerf(FourierFeatures(10), FourierFeatures(4), FourierFeatures(4), [1, 6])["model"].layer_table()
    mine = synthetic.synthetic_weights(table, seed=100 + rank)                  # the ranks come with different weights
    out = Train(target, train_dataset_config=train, model_config=cfg["model_config"], loss_config=cfg["loss_config"], lrate=cfg["lrate"],
                lrate_decay=cfg["lrate_decay"], renderer_config=dict(cfg["renderer_config"], n_samples=32), n_iters=12,
                logger_config=dict(i_print=4, i_img=0, i_checkpoint=6), weights=mine)
    w = torch.from_numpy(out["trainer"].weights())
    both = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    assert all(torch.equal(both[0], b) for b in both)                           # in step bit for bit after 12 averaged steps
    zero = synthetic.synthetic_weights(table, seed=100)
    assert not np.array_equal(w.numpy(), zero) and out["step"] == 12 and out["trainer"].iterations == 12
    assert (len(out["loss"]) == 3) == (rank == 0) and (len(out["checkpoints"]) == 2) == (rank == 0)
    dist.barrier()
    if rank == 0:
        kept = sorted(f for f in os.listdir(os.path.join(target, "checkpoints")) if f.endswith(".index"))
        assert kept == ["ckpt-12.index", "ckpt-6.index"], kept
        print("DP_LOOP_OK")


if __name__ == "__main__":
    main()

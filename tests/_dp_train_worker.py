"""Worker of tests/test_gpu_train.py::test_two_ranks_train_data_parallel: launched by torch.distributed.run with 2 ranks that share GPU 0
(process group on gloo).  Each rank takes the gradient of ITS half of a batch, the ranks average (Trainer.sync_gradients), and the result is
the gradient of the whole batch, which every rank also computes alone; then both take the same Adam step."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import make_model, batch, make_loss, layer_slices, rel_linf   # noqa: E402
from nerf_tex_amd.train import Trainer                                                  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 128, 48
    ro, rd, t, cone, params, color, alpha = batch(21, n, S, 7, "carpet")
    okw, loss = make_loss("nerf_mse")                       # a mean over rays and channels: the mean of the halves' losses is the whole's
    z = None
    whole = Trainer(model, max_rays=n, n_samples=S, perturb=False)
    whole.gradients_step(ro, rd, t, params, cone, color, alpha, loss)
    g_whole = whole.gradients()
    lo, hi = rank * n // world, (rank + 1) * n // world
    mine = Trainer(model, max_rays=hi - lo, n_samples=S, perturb=False, lrate=5e-4)
    mine.gradients_step(ro[lo:hi], rd[lo:hi], t[lo:hi], params[lo:hi], cone[lo:hi], color[lo:hi], alpha[lo:hi], loss)
    g_half = mine.gradients()
    mine.sync_gradients(group=None)
    g = mine.gradients()
    assert np.abs(g_whole).max() > 1e-6 and rel_linf(g_half, g_whole) > 1e-3             # the halves differ from the whole
    worst = max(rel_linf(g[sl], g_whole[sl]) for _, sl in layer_slices(spec))
    assert worst <= 2e-5, worst                                                          # float32 sums in another order
    mine.apply_gradients()
    w = torch.from_numpy(mine.weights())
    both = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    assert all(torch.equal(both[0], b) for b in both)                                    # the ranks stay in step bit for bit
    assert not np.array_equal(mine.weights(), np.asarray(model.get_blob(), np.float32).reshape(-1))
    # unequal shards: the mean of per-rank means would weigh rank 0's rays more -- refused on every rank before anything is averaged
    k = 40 if rank == 0 else 24
    odd = Trainer(model, max_rays=k, n_samples=S, perturb=False)
    odd.gradients_step(ro[:k], rd[:k], t[:k], params[:k], cone[:k], color[:k], alpha[:k], loss)
    before = odd.gradients()
    try:
        odd.sync_gradients(group=None)
        raise AssertionError("unequal shards were averaged")
    except ValueError as e:
        assert "unequal shards" in str(e) and np.array_equal(odd.gradients(), before)
    dist.barrier()
    if rank == 0:
        print("DP_TRAIN_OK", worst)


if __name__ == "__main__":
    main()

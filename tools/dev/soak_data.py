"""Randomised soak of nerf_tex_amd.dataset.Dataset on the GPU: random image sizes, numbers of views, boxes, batch sizes, shuffle buffers, epochs,
background compositing -- every batch made in one piece (the fused path) against the same batch made view by view, and its colours / alphas
against numpy on the pixels the rays themselves name.
    python tools/dev/soak_data.py [--cases 200] [--seed 0]"""
import argparse, json, os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=200); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    from nerf_tex_amd import dataset as D, util
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(a.seed)
    failed, batches, t0 = 0, 0, time.time()
    for case in range(a.cases):
        H, W = int(rng.integers(24, 97)), int(rng.integers(24, 97))
        nv = int(rng.integers(2, 10))
        half = float(rng.uniform(0.6, 1.4))
        views = []
        for k in range(nv):
            az, z = rng.uniform(0, 2 * np.pi), rng.uniform(0.1, 0.9)
            views.append({"pose": D.look_at(np.asarray([np.cos(az) * np.sqrt(1 - z * z), np.sin(az) * np.sqrt(1 - z * z), z]) * rng.uniform(4, 7)),
                          "parameters": rng.uniform(0, 1, 3).astype(np.float32), "rgba": rng.integers(0, 256, (H, W, 4), dtype=np.uint8)})
        cb = bool(rng.integers(2))
        block = {"module": "network.dataset.Dataset",
                 "data_loader_config": {"module": "nerf_tex_amd.dataset.FromViews", "views": views, "angle": float(rng.uniform(0.3, 0.7)), "composite_bkgd": cb, "bkgd_color": [0.1, 0.6, 0.9]},
                 "pixel_sampler_config": {"module": "network.pixel_sampler.Proxy", "n_samples": int(rng.integers(1, 40)), "downsample_factor": int(rng.choice([1, 2, 4, 8]))},
                 "ray_sampler_config": {"module": "network.ray_sampler.Proxy"}, "proxy_config": {"module": "network.proxy.AABB", "b_0": [-half] * 3, "b_1": [half] * 3},
                 "batchsize": int(rng.integers(1, 6)), "shuffle_buffer_size": int(rng.integers(1, 12)), "n_epochs": int(rng.integers(1, 4)), "seed": int(rng.integers(1 << 30)), "device": dev}
        try:
            fresh = lambda **kw: dict(block, pixel_sampler_config=dict(block["pixel_sampler_config"]), ray_sampler_config=dict(block["ray_sampler_config"]),
                                      data_loader_config=dict(block["data_loader_config"]), **kw)       # (Dataset writes height / width / focal into the sampler blocks)
            one = list(util.instantiate(fresh()))
            two = list(util.instantiate(fresh(fused_batches=False)))
        except ValueError as e:                                   # fewer hit pixels than n_samples in some view: both paths refuse it
            if "pixel sampler found" in str(e):
                continue
            raise
        ok = len(one) == len(two) == -(-nv * block["n_epochs"] // block["batchsize"]) and all(torch.equal(x[k], y[k]) for x, y in zip(one, two) for k in x)
        focal = W / np.tan(block["data_loader_config"]["angle"] / 2) / 2
        for b in one:
            for e in range(b["color"].shape[0]):
                o = b["rays_o"][e, 0].cpu().numpy()
                k = int(np.argmin([np.abs(v["pose"][:3, 3] - o).max() for v in views]))
                c2w = torch.as_tensor(views[k]["pose"], device=dev)
                d = b["rays_d"][e] @ c2w[:3, :3]; d = d / -d[:, 2:3]
                j = torch.round(d[:, 0] * focal + W / 2 - 0.5).long(); i = torch.round(-d[:, 1] * focal + H / 2 - 0.5).long()
                px = torch.as_tensor(views[k]["rgba"], device=dev)[i.clamp(0, H - 1), j.clamp(0, W - 1)].float() * torch.tensor(1.0 / 255)
                want = px[:, :3] * px[:, 3:]
                if cb:
                    want = want + (1 - px[:, 3:]) * torch.tensor([0.1, 0.6, 0.9], device=dev)
                ok = ok and torch.equal(b["color"][e], want) and torch.equal(b["alpha"][e], px[:, 3]) and torch.equal(b["parameters"][e].cpu(), torch.as_tensor(views[k]["parameters"]))
            batches += 1
        if not ok:
            failed += 1
            print("FAILED case", case, {k: v for k, v in block.items() if k not in ("data_loader_config", "device")})
    print(json.dumps({"cases": a.cases, "seed": a.seed, "batches": batches, "failed": failed, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()

"""The whole training path on something it can learn, at the shipped config's batch: a teacher network's renders of 24 cameras (128 x 128,
uint8 RGBA PNGs as the Logger writes them) -> NeRF folder -> TFRecord shards (tfrecord.convert_folder) -> `Train(**config)` with the carpet
training config's blocks as written (TFRecord dataset, Proxy pixel / ray samplers, 4 x 256 rays x 256 samples, AlphaLoss(smape, mse), Adam +
decay, GenerateData validation views) for --steps steps; every --every steps the validation view's PSNR against the teacher's render of the
same camera, the loss, the loop's rate.  Also: device memory at the start and the end (nothing grows).
    python tools/dev/soak_loop.py [--steps 4000] [--every 500]"""
import argparse, json, os, sys, tempfile, time
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--every", type=int, default=500)
    ap.add_argument("--size", type=int, default=128)
    args = ap.parse_args()
    from nerf_tex_amd import dataset as D
    from nerf_tex_amd.render import render_image
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Train
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "train_configs.json")))["carpet"]
    dev = torch.device("cuda", 0)
    H = W = args.size
    root = tempfile.mkdtemp()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_example_dataset
    _, teacher = make_example_dataset.make(root, 24, H)                                   # PNGs -> NeRF folder -> TFRecord shards
    renderer = Renderer(model=teacher, n_samples=256, perturb=False)
    train = json.loads(json.dumps(cfg["train_dataset_config"])); train["data_loader_config"]["tfr_path"] = os.path.join(root, "tfr")
    val = json.loads(json.dumps(cfg["val_dataset_config"])); val["data_loader_config"].update(height=H, width=W)
    vds = D.Dataset(dict(val["data_loader_config"]), {"module": "network.pixel_sampler.Full"}, {"module": "network.ray_sampler.Proxy"}, dict(val["proxy_config"]), n_epochs=1, device=dev)
    want = render_image(renderer, vds, list(vds)[1])[0]                                  # the second validation parameter set is the teacher's
    psnr = lambda img: float(-10 * torch.log10(((img - want) ** 2).mean()))
    kw = dict(train_dataset_config=train, val_dataset_config=val, model_config=cfg["model_config"], loss_config=cfg["loss_config"], lrate=cfg["lrate"],
              lrate_decay=cfg["lrate_decay"], renderer_config=cfg["renderer_config"])
    torch.cuda.synchronize(); mem0 = None; done = 0; t0 = time.perf_counter()
    print(f"{'step':>6} {'loss':>9} {'PSNR of the validation view (dB)':>34} {'loop ms/step':>13}")
    while done < args.steps:
        done += args.every
        t1 = time.perf_counter()
        out = Train(os.path.join(root, "run"), n_iters=done, logger_config={"i_print": args.every, "i_img": args.every, "i_checkpoint": args.every, "max_to_keep": 2}, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t1
        if mem0 is None:
            mem0 = torch.cuda.memory_allocated(dev)
        print(f"{done:6d} {out['loss'][-1][1]:9.5f} {psnr(out['images'][done][1]):34.2f} {1e3 * dt / args.every:13.2f}", flush=True)
    kept = sorted(f for f in os.listdir(os.path.join(root, "run", "checkpoints")) if f.endswith(".index"))
    print(json.dumps({"steps": done, "wall_s": time.perf_counter() - t0, "checkpoints_kept": kept, "torch_bytes_after_first_leg": mem0, "torch_bytes_at_end": torch.cuda.memory_allocated(dev),
                      "note": "each leg is a NEW Train(...) call resuming from the last checkpoint: its time includes making the trainer, restoring, the validation render and the checkpoint"}))


if __name__ == "__main__":
    main()

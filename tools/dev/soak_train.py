#!/usr/bin/env python3
"""GPU box: the training step against the float64 restatement on RANDOM cases -- family, batch shape (ragged sample counts, fewer / more row
groups than workgroups), loss, perturb, raw_noise_std, blur, background compositing, rays that miss the proxy -- each held to the gates of
tests/test_gpu_train.py (loss 1e-5, predictions and every layer's gradient 1e-4 rel-Linf with the oracle following the float32 ReLU branches).
    python tools/dev/soak_train.py [--cases 150] [--seed 0]"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import make_model, step_errors                                           # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=150); ap.add_argument("--seed", type=int, default=0); ap.add_argument("--only", type=int, nargs="*", default=None)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
FAMS = [("carpet", (1, 6)), ("grass", (1, 4)), ("fur", (1, 4)), ("grass_filtered", (2, 3))]
worst, t0, fails, waived = {"loss": 0.0, "pred": 0.0, "grad": 0.0}, time.time(), [], 0
models = {}
for case in range(a.cases):
    fam, npar = FAMS[rng.integers(len(FAMS))]
    if npar not in models: models[npar] = make_model(npar, dense_media=True)
    model, spec, wts = models[npar]
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), rng.integers(400, 1500)]))
    S = int(rng.choice([rng.integers(2, 12), rng.integers(12, 70), rng.integers(70, 130)]))
    if n * S > 60000: n = max(1, 60000 // S)
    loss_name = ["alpha_smape", "alpha_mse_soft", "nerf_mse"][rng.integers(3)]
    perturb, bkgd = bool(rng.integers(2)), bool(rng.integers(2))
    noise_std = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    blur = None if rng.integers(2) else int(rng.integers(sum(npar)))
    seed = int(rng.integers(1 << 30))
    batch_seed = int(rng.integers(1 << 20))
    miss = rng.uniform(size=n) < float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    cap = n + int(rng.integers(0, 50))
    if a.only is not None and case not in a.only: continue
    knobs = dict(perturb=perturb, bkgd=bkgd, noise_std=noise_std, blur=blur, seed=seed, batch_seed=batch_seed, miss=miss, cap=cap)
    e = step_errors(model, spec, wts, fam, n, S, loss_name, **knobs)
    M = n * S
    tiny = 5.0 if M < 100 else 1.0
    e_loss, e_pred, e_grad = e["e_loss"], e["e_pred"], e["e_grad"]
    ok = e_loss <= (1e-5 if M >= 1000 else 1e-4) * tiny and e_pred <= 1e-4 * tiny and e_grad <= 1e-4 * tiny and e["finite"]
    if not ok and e["finite"]:
        # a case beyond the gate: is it float32 itself?  The same restatement, same branches, under float32 torch autograd is the floor of this
        # comparison (few coarse steps: 1 - exp(-sigma dist) at dist ~ 1 carries sigma's rounding far); the step has to stay within 4 floors
        f = step_errors(model, spec, wts, fam, n, S, loss_name, floors=True, **knobs)
        floor_pred, floors = f["floor_pred"], f["floors"]
        ok = e_pred <= max(1e-4 * tiny, 4 * floor_pred) and all(v <= max(1e-4 * tiny, 4 * floors[nm]) for nm, v in e["layers"].items()) \
            and e_loss <= max(1e-4, 4 * floor_pred)
        if ok:
            waived += 1
            print("beyond the gate, within 4 float32 floors:", json.dumps(dict(case=case, n=n, S=S, e_pred=e_pred, floor_pred=floor_pred, e_grad=e_grad, floor_grad=max(floors.values()))), flush=True)
    desc = dict(case=case, fam=fam, n=n, S=S, loss=loss_name, perturb=perturb, bkgd=bkgd, noise=noise_std, blur=blur, missing=int(miss.sum()), e_loss=e_loss, e_pred=e_pred, e_grad=e_grad)
    if not ok:
        fails.append(desc); print("FAIL", json.dumps(desc), flush=True)
        for nm, v in e["layers"].items():
            if v > 1e-4: print("    ", nm, "rel-Linf %.3e" % v, flush=True)
    if M >= 100:
        worst = {"loss": max(worst["loss"], e_loss), "pred": max(worst["pred"], e_pred), "grad": max(worst["grad"], e_grad)}
print(json.dumps({"cases": a.cases, "seed": a.seed, "failed": len(fails), "waived_within_4_float32_floors": waived, "worst_rel_loss": worst["loss"],
                  "worst_rel_linf_predictions": worst["pred"], "worst_rel_linf_layer_gradient": worst["grad"], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if fails else 0)

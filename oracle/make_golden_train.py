"""Generate tests/golden/train_grads.npz: parameter gradients of one training step of the IMPORTED
REFERENCE (Renderer_TensoIR_train with is_train=True + the loss of train_tensoIR.py:262-311 + backward)
on the seeded small scene of tests/golden/small_scene.npz.

Run in the build container (needs the read-only reference checkout):
    python oracle/make_golden_train.py
Pins oracle.train_step_grads (tests/test_oracle_train.py), which in turn checks the HIP backward
kernels on the GPU box.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import tensoir_oracle as O  # noqa: E402
from oracle.make_golden import build_reference_model, npy  # noqa: E402
from tests.helpers import golden_checkpoint  # noqa: E402

SEED = 20211202
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref = ref_loader.load()
    g0 = np.load(os.path.join(OUT, "small_scene.npz"))
    ckpt = golden_checkpoint(g0)
    ckpt["kwargs"]["light_rotation"] = [int(r) for r in ckpt["kwargs"]["light_rotation"]]
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    vol = torch.from_numpy(np.array(g0["scene/alpha_volume"]))
    model = build_reference_model(ref, ckpt, envh, envw)
    model.alphaMask = ref.tensorf.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
    model.train()
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    rays = torch.from_numpy(np.array(g0["rays/rays"]))
    light_idx = torch.from_numpy(np.array(g0["rays/light_idx"]))
    B = rays.shape[0]
    rgb_gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(SEED + 11))
    S = 64
    g = {"train/rgb_gt": npy(rgb_gt), "train/n_samples": np.array([S], np.int64),
         "train/second": np.array([args.second_nSample, args.second_near, args.second_far], np.float64),
         "train/weights": np.array([O.TRAIN_WEIGHTS[k] for k in sorted(O.TRAIN_WEIGHTS)], np.float64)}

    for tag, relight, method in (("A", False, "fixed_envirmap"), ("B", True, "fixed_envirmap"),
                                 ("C", True, "stratified_sampling")):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(SEED + 12)
        ret = ref.renderer.Renderer_TensoIR_train(
            rays, None, light_idx, model, N_samples=S, white_bg=True, is_train=True, is_relight=relight,
            sample_method=method, chunk_size=777, device="cpu", args=args)
        loss = O.training_loss(ret, rgb_gt, relight)
        loss.backward()
        g[f"train_{tag}/loss"] = npy(loss).reshape(1)
        for k, v in ret.items():
            g[f"train_{tag}/out/{k}"] = npy(v)
        for name, p in model.named_parameters():
            g[f"train_{tag}/grad/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
        # RNG replay (CPU generator): rand [B,1] ray jitter (:717), then -- relight only -- randn [A,3]
        # (:937), then for 'stratified_sampling' two rand [envh,envw] (:520-521)
        torch.manual_seed(SEED + 12)
        g[f"train_{tag}/ray_jitter"] = npy(torch.rand(B, 1))

    path = os.path.join(OUT, "train_grads.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

"""Generate tests/golden/normals_kinds.npz: the IMPORTED REFERENCE's forward maps and one training step's
parameter gradients for normals_kind = 'purely_predicted' (the class default,
models/tensorBase_rotated_lights.py:357) and 'purely_derived' on the seeded small scene of
tests/golden/small_scene.npz.  In both kinds normals_diff / normals_orientation_loss stay zero: only the
'derived_plus_predicted' branch fills them (:946-960).

Run in the build container (needs the read-only reference checkout):
    python oracle/make_golden_kinds.py
TEST INFRASTRUCTURE ONLY.
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
NAMES = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map",
         "acc_map", "normals_diff_map", "normals_orientation_loss_map", "acc_mask",
         "albedo_smoothness_loss", "roughness_smoothness_loss"]


def main():
    ref = ref_loader.load()
    g0 = np.load(os.path.join(OUT, "small_scene.npz"))
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    vol = torch.from_numpy(np.array(g0["scene/alpha_volume"]))
    rays = torch.from_numpy(np.array(g0["rays/rays"]))
    light_idx = torch.from_numpy(np.array(g0["rays/light_idx"]))
    B = rays.shape[0]
    rgb_gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(SEED + 11))
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    S = 64
    g = {"rgb_gt": npy(rgb_gt), "n_samples": np.array([S], np.int64)}
    normal_gt = torch.nn.functional.normalize(torch.randn(B, 3, generator=torch.Generator().manual_seed(SEED + 13)), dim=-1)
    g["normal_gt"] = npy(normal_gt)
    for kind in ("purely_predicted", "purely_derived", "gt_normals", "residue_prediction"):
        ckpt = golden_checkpoint(g0)
        ckpt["kwargs"]["light_rotation"] = [int(r) for r in ckpt["kwargs"]["light_rotation"]]
        ckpt["kwargs"]["normals_kind"] = kind
        if kind in ("purely_derived", "gt_normals"):      # no renderModule_normal in those configurations (:422-428)
            ckpt["state_dict"] = {k: v for k, v in ckpt["state_dict"].items() if not k.startswith("renderModule_normal")}
        if kind == "residue_prediction":                  # MLPNormal_normal_and_PExyz: layer 1 is 153 wide (:240); seeded weights
            gen = torch.Generator().manual_seed(SEED + 17)
            w0 = ckpt["state_dict"]["renderModule_normal.mlp.0.weight"]
            wide = torch.empty(w0.shape[0], w0.shape[1] + 3).uniform_(-1, 1, generator=gen) / float(w0.shape[1] + 3) ** 0.5
            ckpt["state_dict"]["renderModule_normal.mlp.0.weight"] = wide
            g[f"{kind}/w0_normal_decoder"] = npy(wide)
        model = build_reference_model(ref, ckpt, envh, envw)
        model.alphaMask = ref.tensorf.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
        # eval forward (the randn_like draw of :937 only feeds the smoothness losses)
        model.eval()
        torch.manual_seed(SEED + 3)
        out = model(rays, light_idx, white_bg=True, is_train=False, is_relight=True, N_samples=-1)
        for n, v in zip(NAMES, out):
            g[f"{kind}/fwd/{n}"] = npy(v)
        # the boundary call in eval mode (gt_normals: the ground-truth normals replace the zero map before shading, renderer.py:82-83)
        ngt = normal_gt if kind == "gt_normals" else None
        torch.manual_seed(SEED + 3)
        with torch.no_grad():
            ret = ref.renderer.Renderer_TensoIR_train(
                rays, ngt, light_idx, model, N_samples=-1, white_bg=True, is_train=False, is_relight=True,
                sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
        for k, v in ret.items():
            g[f"{kind}/eval_render/{k}"] = npy(v)
        # one training step through the boundary call
        model.train()
        model.zero_grad(set_to_none=True)
        torch.manual_seed(SEED + 12)
        ret = ref.renderer.Renderer_TensoIR_train(
            rays, ngt, light_idx, model, N_samples=S, white_bg=True, is_train=True, is_relight=True,
            sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
        loss = O.training_loss(ret, rgb_gt, True)
        loss.backward()
        g[f"{kind}/train/loss"] = npy(loss).reshape(1)
        for k, v in ret.items():
            g[f"{kind}/train/out/{k}"] = npy(v)
        for name, p in model.named_parameters():
            g[f"{kind}/train/grad/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
        torch.manual_seed(SEED + 12)
        g[f"{kind}/train/ray_jitter"] = npy(torch.rand(B, 1))
    path = os.path.join(OUT, "normals_kinds.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

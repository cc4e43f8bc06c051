"""Generate tests/golden/general_lights.npz and tests/golden/mask_maintenance.npz by running the IMPORTED REFERENCE:

* general multi-light variant (models/tensoRF_general_multi_lights.py + tensorBase_general_multi_lights.py:463-479,
  :566-582): one SG set per light -- environment radiance ``get_light_rgbs`` on seeded directions, and one training
  step through ``Renderer_TensoIR_train`` (forward maps, loss, gradients of every parameter incl. the three
  ``lgtSGs_list`` entries) on the seeded small scene of tests/golden/small_scene.npz;
* occupancy-mask maintenance (models/tensorBase_rotated_lights.py:737-811): ``getDenseAlpha`` (with and without an
  existing mask), ``updateAlphaMask`` (new volume + returned aabb), ``filtering_rays`` (both modes).

Run in the build container (needs the read-only reference checkout):
    python oracle/make_golden_general.py
Pins oracle.light_rgbs (lgtSGs_list branch), oracle.dense_alpha / update_alpha_mask / filtering_rays
(tests/test_oracle_general.py) and, on the GPU box, the HIP path.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import contextlib
import importlib
import io
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
LIGHTS = ["sunset", "snow", "courtyard"]


def filter_rays_input():
    """Seeded rays around the camera of the small scene: hits, misses and zero direction components."""
    gen = torch.Generator().manual_seed(51)
    o = torch.randn(3000, 3, generator=gen) * 0.5 + torch.tensor([0.0, 0.0, 4.0])
    d = torch.nn.functional.normalize(torch.randn(3000, 3, generator=gen) * torch.tensor([0.6, 0.6, 0.3]) - torch.tensor([0, 0, 1.0]), dim=-1)
    rays = torch.cat([o, d], -1)
    rays[:5, 3] = 0.0
    return rays


def general(ref, g0):
    gen_mod = importlib.import_module("models.tensoRF_general_multi_lights")
    ckpt = golden_checkpoint(g0)
    kw = dict(ckpt["kwargs"])
    for k in ("light_num", "light_rotation"):
        kw.pop(k, None)
    aabb, grid = kw.pop("aabb"), kw.pop("gridSize")
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    with contextlib.redirect_stdout(io.StringIO()):
        model = gen_mod.TensorVMSplit(aabb, grid, "cpu", envmap_h=envh, envmap_w=envw, light_name_list=LIGHTS, **kw)
    sd = {k: v for k, v in ckpt["state_dict"].items() if k != "lgtSGs"}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing       # lgtSGs_list is a plain list, not state
    vol = torch.from_numpy(np.array(g0["scene/alpha_volume"]))
    model.alphaMask = gen_mod.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
    gen = torch.Generator().manual_seed(41)
    base = ckpt["state_dict"]["lgtSGs"]
    g = {}
    for i, sg in enumerate(model.lgtSGs_list):
        with torch.no_grad():
            sg.copy_(base + 0.3 * torch.randn(base.shape, generator=gen))
        g[f"sg/{i}"] = npy(sg)
    dirs = torch.nn.functional.normalize(torch.randn(50, 3, generator=gen), dim=-1)
    g["env/dirs"] = npy(dirs)
    with torch.no_grad():
        g["env/light_rgbs"] = npy(model.get_light_rgbs(dirs, device="cpu"))
    rays = torch.from_numpy(np.array(g0["rays/rays"]))
    light_idx = torch.from_numpy(np.array(g0["rays/light_idx"]))
    B, S = rays.shape[0], 64
    rgb_gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(SEED + 11))
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    g["train/rgb_gt"], g["train/n_samples"] = npy(rgb_gt), np.array([S], np.int64)
    # eval maps (fixed sample positions), then one training step
    model.eval()
    torch.manual_seed(SEED + 3)
    with torch.no_grad():
        ret = ref.renderer.Renderer_TensoIR_train(rays, None, light_idx, model, N_samples=-1, white_bg=True, is_train=False,
                                                  is_relight=True, sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
    for k, v in ret.items():
        g[f"eval/out/{k}"] = npy(v)
    model.train()
    model.zero_grad(set_to_none=True)
    for sg in model.lgtSGs_list:
        sg.grad = None
    torch.manual_seed(SEED + 12)
    ret = ref.renderer.Renderer_TensoIR_train(rays, None, light_idx, model, N_samples=S, white_bg=True, is_train=True,
                                              is_relight=True, sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
    loss = O.training_loss(ret, rgb_gt, True)
    loss.backward()
    g["train/loss"] = npy(loss).reshape(1)
    for k, v in ret.items():
        g[f"train/out/{k}"] = npy(v)
    for name, p in model.named_parameters():
        g[f"train/grad/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
    for i, sg in enumerate(model.lgtSGs_list):
        g[f"train/grad/lgtSGs_list.{i}"] = npy(sg.grad)
    torch.manual_seed(SEED + 12)
    g["train/ray_jitter"] = npy(torch.rand(B, 1))
    path = os.path.join(OUT, "general_lights.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def mask(ref, g0):
    ckpt = golden_checkpoint(g0)
    ckpt["kwargs"]["light_rotation"] = [int(r) for r in ckpt["kwargs"]["light_rotation"]]
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    model = build_reference_model(ref, ckpt, envh, envw)
    vol = torch.from_numpy(np.array(g0["scene/alpha_volume"]))
    grid = (22, 26, 30)
    g = {"grid": np.array(grid, np.int64)}
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        a, d = model.getDenseAlpha(grid)                      # no mask yet: every lattice point evaluated
        g["nomask/alpha"], g["dense_xyz"] = npy(a), npy(d)
        model.alphaMask = ref.tensorf.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
        a, _ = model.getDenseAlpha(grid)                      # culled by the existing mask
        g["masked/alpha"] = npy(a)
        g["update/aabb"] = npy(model.updateAlphaMask(grid))   # replaces the mask
        g["update/volume"] = npy(model.alphaMask.alpha_volume[0, 0])
        g["update/mask_aabb"] = npy(model.alphaMask.aabb)
        rays = filter_rays_input()
        g["filter/rays"] = npy(rays)
        kept, m = model.filtering_rays(rays, N_samples=80, bbox_only=False)
        g["filter/mask_alpha"], g["filter/kept_alpha"] = npy(m), npy(kept)
        kept, m = model.filtering_rays(rays, bbox_only=True)
        g["filter/mask_bbox"] = npy(m)
        # a second update on top of the new mask, finer lattice (train_tensoIR.py:385-399 runs it twice)
        g["update2/aabb"] = npy(model.updateAlphaMask((33, 29, 31)))
        g["update2/volume"] = npy(model.alphaMask.alpha_volume[0, 0])
    path = os.path.join(OUT, "mask_maintenance.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def pixel(ref, g0):
    """light_kind == 'pixel' (models/tensorBase_rotated_lights.py:459-460, :585-605): the learnable envmap_h x envmap_w image behind
    softplus(beta=5), looked up per rotated direction with grid_sample(align_corners=False) -> tests/golden/pixel_light.npz."""
    ckpt = golden_checkpoint(g0)
    ckpt["kwargs"]["light_rotation"] = [int(r) for r in ckpt["kwargs"]["light_rotation"]]
    ckpt["kwargs"]["light_kind"] = "pixel"
    ckpt["state_dict"] = {k: v for k, v in ckpt["state_dict"].items() if k != "lgtSGs"}
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    gen = torch.Generator().manual_seed(SEED + 21)
    raw = torch.rand(envh * envw, 3, generator=gen) * 3.0 - 0.5          # uniform(0, 3) as the reference initialises, some negative
    ckpt["state_dict"]["_light_rgbs"] = raw
    model = build_reference_model(ref, ckpt, envh, envw)
    vol = torch.from_numpy(np.array(g0["scene/alpha_volume"]))
    model.alphaMask = ref.tensorf.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
    assert model.light_kind == "pixel" and torch.equal(model._light_rgbs.data, raw)
    g = {"light_rgbs_raw": npy(raw)}
    dirs = torch.nn.functional.normalize(torch.randn(60, 3, generator=gen), dim=-1)
    # next to the poles and on both sides of the +-pi seam.  (EXACTLY at a pole the lookup column is atan2(+-0, +-0) of the rotated
    # direction: it depends on the signed zeros the rotation matmul happens to produce -- not a property to pin.)
    dirs[:4] = torch.nn.functional.normalize(torch.tensor([[1e-3, 2e-3, 1.0], [-2e-3, 1e-3, -1.0], [-1.0, 1e-4, 0.0],
                                                           [-1.0, -1e-4, 0.0]]), dim=-1)
    g["env/dirs"] = npy(dirs)
    with torch.no_grad():
        g["env/light_rgbs"] = npy(model.get_light_rgbs(dirs, device="cpu"))
        g["env/light_rgbs_fixed"] = npy(model.get_light_rgbs(model.fixed_viewdirs, device="cpu"))
    rays = torch.from_numpy(np.array(g0["rays/rays"]))
    light_idx = torch.from_numpy(np.array(g0["rays/light_idx"]))
    B, S = rays.shape[0], 64
    rgb_gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(SEED + 11))
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    g["train/rgb_gt"], g["train/n_samples"] = npy(rgb_gt), np.array([S], np.int64)
    model.eval()
    torch.manual_seed(SEED + 3)
    with torch.no_grad():
        ret = ref.renderer.Renderer_TensoIR_train(rays, None, light_idx, model, N_samples=-1, white_bg=True, is_train=False,
                                                  is_relight=True, sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
    for k, v in ret.items():
        g[f"eval/out/{k}"] = npy(v)
    model.train()
    model.zero_grad(set_to_none=True)
    torch.manual_seed(SEED + 12)
    ret = ref.renderer.Renderer_TensoIR_train(rays, None, light_idx, model, N_samples=S, white_bg=True, is_train=True,
                                              is_relight=True, sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
    loss = O.training_loss(ret, rgb_gt, True)
    loss.backward()
    g["train/loss"] = npy(loss).reshape(1)
    for k, v in ret.items():
        g[f"train/out/{k}"] = npy(v)
    for name, p in model.named_parameters():
        g[f"train/grad/{name}"] = npy(torch.zeros_like(p) if p.grad is None else p.grad)
    torch.manual_seed(SEED + 12)
    g["train/ray_jitter"] = npy(torch.rand(B, 1))
    # light_kind == 'gt' (:592-593): the data set's probe looked up as it is -- same lookup, no activation, nothing to train
    probe = torch.rand(envh * envw, 3, generator=gen) * 4.0
    gt_ck = golden_checkpoint(g0)
    gt_ck["kwargs"]["light_rotation"] = [int(r) for r in gt_ck["kwargs"]["light_rotation"]]
    gt_ck["kwargs"]["light_kind"] = "gt"
    gt_ck["kwargs"]["dataset"] = types.SimpleNamespace(lights_probes=probe)
    gt_ck["state_dict"] = {k: v for k, v in gt_ck["state_dict"].items() if k != "lgtSGs"}
    gt_model = build_reference_model(ref, gt_ck, envh, envw)
    gt_model.alphaMask = ref.tensorf.AlphaGridMask("cpu", torch.from_numpy(np.array(g0["scene/alpha_aabb"])), vol)
    g["gt/probe"] = npy(probe)
    gt_model.eval()
    with torch.no_grad():
        g["gt/light_rgbs"] = npy(gt_model.get_light_rgbs(dirs, device="cpu"))
        torch.manual_seed(SEED + 3)
        ret = ref.renderer.Renderer_TensoIR_train(rays, None, light_idx, gt_model, N_samples=-1, white_bg=True, is_train=False,
                                                  is_relight=True, sample_method="fixed_envirmap", chunk_size=777, device="cpu", args=args)
    for k, v in ret.items():
        g[f"gt/eval/out/{k}"] = npy(v)
    path = os.path.join(OUT, "pixel_light.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def main():
    ref = ref_loader.load()
    g0 = np.load(os.path.join(OUT, "small_scene.npz"))
    torch.manual_seed(SEED)
    general(ref, g0)
    mask(ref, g0)
    pixel(ref, g0)


if __name__ == "__main__":
    main()

"""The call sequence of ``reconstruction()`` (train_tensoIR.py:110-461) driven through the PRODUCT API only, as a reusable
function: tests/test_gpu_train_loop.py (the script-level drop-in evidence), tests/test_gpu_precision_policy.py and
tools/precision_sweep.py (a TRAINED checkpoint for the indirect-light precision policy) all run it.

  analytic dataset -> TensorVMSplit(aabb, reso, device, **script kwargs) (:170-192) -> get_optparam_groups + Adam (:196,
  :206) -> filtering_rays(bbox_only=True) on the host ray tensor (:228) -> iterations (:237-317): host batch indexing,
  Renderer_TensoIR_train(is_train=True, is_relight=flag, sample_method='stratified_sampling'), image loss + L1 / TV
  regularisers + relighting losses, backward, Adam step, lr decay -> updateAlphaMask + shrink + relighting on (:385-398)
  -> second mask update + ray re-filtering (:401-406) -> upsample_volume_grid + fresh optimizer (:409-422)."""
import types

import numpy as np
import torch


def n_to_reso(n_voxels, bbox):
    size = bbox[1] - bbox[0]
    voxel = (size.prod() / n_voxels).pow(1 / 3)
    return (size / voxel).long().tolist()


def tv(x):
    h, w = x.shape[2], x.shape[3]
    ch = x.shape[1] * (h - 1) * w
    cw = x.shape[1] * h * max(w - 1, 1)
    return 2 * ((x[:, :, 1:, :] - x[:, :, :-1, :]).pow(2).sum() / ch + (x[:, :, :, 1:] - x[:, :, :, :-1]).pow(2).sum() / cw) / x.shape[0]


def reconstruct(variant="single_light", n_iters=150, batch=1024, upsamp=(100, 130), mask_updates=(60, 110),
                dataset="synthetic:views=4,res=32", grid0=32, grid1=64, device="cuda:0", seed=20211202, model_kw=None, on_iter=None):
    """-> namespace(model, ds, args, rays_f, rgbs_f, lidx_f, losses, grids, n_samples).
    variant = which of the reference's three training scripts is mirrored: train_tensoIR.py (one light),
    train_tensoIR_rotated_multi_lights.py (one SG set seen under three rotations) or
    train_tensoIR_general_multi_lights.py (one SG set per light, models/tensoRF_general_multi_lights.py)."""
    from tensoir_amd import Renderer_TensoIR_train
    from tensoir_amd.optim import Adam       # what tensoir_amd.run binds torch.optim.Adam to (train_tensoIR.py:197)
    from tensoir_amd.synth_dataset import SyntheticDataset
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device(device)
    if variant == "general_multi_lights":
        from tensoir_amd.general_multi_lights import TensorVMSplit
        light_kw = dict(light_rotation=None, light_name_list=["sunset", "snow", "courtyard"])
        ds = SyntheticDataset(dataset, "none", split="train", light_name_list=light_kw["light_name_list"])
    else:
        from tensoir_amd import TensorVMSplit
        rot = ["000"] if variant == "single_light" else ["000", "120", "240"]
        light_kw = dict(light_rotation=rot)
        ds = SyntheticDataset(dataset, "none", split="train", light_rotation=rot)
    args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
    upsamp, mask_updates = list(upsamp), list(mask_updates)
    # tests/data/synthetic_train.txt (the reference itself trains the default schedule to 26.8 dB on its CPU path: density
    # has emerged well before the first mask update)
    voxel_list = torch.round(torch.exp(torch.linspace(np.log(grid0 ** 3), np.log(grid1 ** 3), len(upsamp) + 1))).long().tolist()[1:]
    aabb = ds.scene_bbox.to(dev)
    reso = n_to_reso(grid0 ** 3, aabb)
    n_samples = min(10 ** 6, int(np.linalg.norm(reso) / 0.5))
    m = TensorVMSplit(aabb, reso, dev, density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                      near_far=ds.near_far, shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=-10,
                      distance_scale=25, pos_pe=2, view_pe=2, fea_pe=2, featureC=128, step_ratio=0.5,
                      fea2denseAct="softplus", normals_kind="derived_plus_predicted", light_kind="sg", dataset=ds,
                      numLgtSGs=128, **light_kw, **(model_kw or {}))
    lr_factor = 0.1 ** (1 / n_iters)
    opt = Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    all_rays, all_rgbs, all_lidx = ds.all_rays, ds.all_rgbs, ds.all_light_idx
    rays_f, keep = m.filtering_rays(all_rays, bbox_only=True)
    rgbs_f, lidx_f = all_rgbs[keep], all_lidx[keep]
    assert 0 < rays_f.shape[0] <= all_rays.shape[0]
    relight, l1_w, tv_d, tv_a = False, 8e-5, 0.05, 0.005
    losses, grids = [], [tuple(m.gridSize.tolist())]
    for it in range(n_iters):
        idx = torch.from_numpy(np.random.permutation(rays_f.shape[0])[:batch])
        rays_b, rgb_b, lidx_b = rays_f[idx], rgbs_f[idx].to(dev), lidx_f[idx].to(dev)     # host rays, as the script passes
        ret = Renderer_TensoIR_train(rays=rays_b, normal_gt=None, light_idx=lidx_b, tensoIR=m, N_samples=n_samples,
                                     white_bg=ds.white_bg, ndc_ray=0, device=dev, sample_method="stratified_sampling",
                                     chunk_size=160000, is_train=True, is_relight=relight, args=args)
        assert set(ret) >= {"rgb_map", "rgb_with_brdf_map", "normals_diff_map", "albedo_smoothness_loss"}
        loss_rgb = torch.mean((ret["rgb_map"] - rgb_b) ** 2)
        total = loss_rgb + l1_w * m.density_L1()
        if tv_d > 0:
            tv_d *= lr_factor
            tv_a *= lr_factor
            total = total + m.TV_loss_density(tv) * tv_d + m.TV_loss_app(tv) * tv_a
        if relight:
            total = total + 0.2 * torch.mean((ret["rgb_with_brdf_map"] - rgb_b) ** 2)
            total = total + 5e-4 * ret["normals_diff_map"].mean() + 1e-3 * ret["normals_orientation_loss_map"].mean()
            total = total + 1e-3 * ret["roughness_smoothness_loss"] + 1e-3 * ret["albedo_smoothness_loss"]
        opt.zero_grad()
        total.backward()
        opt.step()
        assert torch.isfinite(total), it
        losses.append(float(loss_rgb.detach()))
        if on_iter is not None:
            on_iter(it, m, ret)
        for g in opt.param_groups:
            g["lr"] *= lr_factor
        if it in mask_updates:
            new_aabb = m.updateAlphaMask(tuple(reso))
            if it == mask_updates[0]:
                m.shrink(new_aabb)
                l1_w, relight, tv_d, tv_a = 4e-5, True, 0, 0
            else:
                rays_f, keep = m.filtering_rays(all_rays, bbox_only=True)
                rgbs_f, lidx_f = all_rgbs[keep], all_lidx[keep]
        if it in upsamp:
            reso = n_to_reso(voxel_list.pop(0), m.aabb)
            n_samples = min(10 ** 6, int(np.linalg.norm(reso) / 0.5))
            m.upsample_volume_grid(reso)
            opt = Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
            grids.append(tuple(m.gridSize.tolist()))
    return types.SimpleNamespace(model=m, ds=ds, args=args, rays_f=rays_f, rgbs_f=rgbs_f, lidx_f=lidx_f, losses=losses, grids=grids,
                                 n_samples=n_samples, device=dev)

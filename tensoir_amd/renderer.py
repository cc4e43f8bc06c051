"""Host-side mirror of the hot-path part of ``renderer.py`` (reference lines 57-127)."""
from __future__ import annotations

import torch

from . import ops, relight


def Renderer_TensoIR_train(rays=None, normal_gt=None, light_idx=None, tensoIR=None, N_samples=-1, ndc_ray=False,
                           white_bg=True, is_train=False, is_relight=True, sample_method="fixed_envirmap",
                           chunk_size=15000, device="cuda", args=None):
    """renderer.py:57-127: primary pass + physically-based re-render of the rays with acc > 0.5.
    Same signature and the same 12-key dict."""
    rays = ops.to_device(rays, device)
    light_idx = ops.to_device(light_idx, device, torch.int32)
    # Record capacities (primary w > thres samples, secondary records) come from the previous call; the device-side
    # counts are read once, after every launch of the pass has been queued; an overflow (rare) re-runs the pass with
    # exact counts.  Same for the training forward (the autograd graph of a discarded attempt is simply dropped).
    for attempt in range(2):
        (rgb_map, depth_map, normal_map, albedo_map, roughness_map, fresnel_map, acc_map, normals_diff_map,
         normals_orientation_loss_map, acc_mask, albedo_smoothness_loss, roughness_smoothness_loss), maps = \
            tensoIR(rays, light_idx, is_train=is_train, white_bg=white_bg, is_relight=is_relight, ndc_ray=ndc_ray,
                    N_samples=N_samples, _return_maps=True, _defer_check=attempt == 0, _want_mask=False)
        if tensoIR.normals_kind == "gt_normals" and normal_gt is not None:       # renderer.py:82-83
            normal_map = ops.to_device(normal_gt, device).to(torch.float32)
            maps = torch.cat([maps[:, :4], normal_map, maps[:, 7:]], dim=1)      # the shading kernels read the map rows
        if is_relight:
            # all rays go through the shading kernels; rows with acc <= 0.5 (acc_mask, :1031) spawn no secondary
            # rays and get the white background (renderer.py:86-106) -- no boolean-mask compaction, no host sync
            rgb_with_brdf = relight.shade_from_maps(tensoIR, maps, rays, light_idx, sample_method, args,
                                                    acc_thres=0.5, _defer_check=attempt == 0)
        else:
            rgb_with_brdf = torch.ones_like(rgb_map)
        ok = tensoIR._finish_primary()     # record-capacity checks, after everything is queued
        if relight.finish_pending(tensoIR) and ok:
            break
    return {
        "rgb_map": rgb_map, "depth_map": depth_map, "normal_map": normal_map, "albedo_map": albedo_map,
        "acc_map": acc_map, "roughness_map": roughness_map, "fresnel_map": fresnel_map,
        "rgb_with_brdf_map": rgb_with_brdf, "normals_diff_map": normals_diff_map,
        "normals_orientation_loss_map": normals_orientation_loss_map,
        "albedo_smoothness_loss": albedo_smoothness_loss, "roughness_smoothness_loss": roughness_smoothness_loss,
    }

"""Host-side mirror of the hot-path part of ``renderer.py`` (reference lines 57-127)."""
from __future__ import annotations

import os

import torch

from . import ops, relight


# Inference calls of the boundary replay a captured HIP graph when the same call shape comes back (the reference's evaluation loops,
# renderer.py:225-249, call it chunk after chunk with batch_size_test rays): a small per-model cache of GraphedRenderer objects keyed by
# (rays, samples, flags, secondary-march arguments, device).  The graph itself notices parameter / grid / mask / policy changes
# (GraphedRenderer._stale) and re-captures.  A key is captured the SECOND time it is seen (a ragged last chunk that never comes
# back costs nothing); TENSOIR_BOUNDARY_GRAPHS=0 switches the cache off, TENSOIR_BOUNDARY_GRAPH_MIN_RAYS sets the smallest call
# worth a graph.
BOUNDARY_GRAPHS = os.environ.get("TENSOIR_BOUNDARY_GRAPHS", "1") != "0"
BOUNDARY_GRAPH_MIN_RAYS = int(os.environ.get("TENSOIR_BOUNDARY_GRAPH_MIN_RAYS", "1024"))
BOUNDARY_GRAPH_SLOTS = 4


def _boundary_graph(tensoIR, rays, N_samples, white_bg, is_relight, device, args):
    """The cached GraphedRenderer of this call shape, or None (first sighting of the key, or graphs not applicable)."""
    dev = torch.device(device)
    if dev.type != "cuda" or args is None:
        return None
    key = (int(rays.shape[0]), int(N_samples), bool(white_bg), bool(is_relight), int(args.second_nSample), float(args.second_near),
           float(args.second_far), str(dev))
    cache = tensoIR.__dict__.setdefault("_boundary_graphs", {})
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= BOUNDARY_GRAPH_SLOTS:                 # drop the least recently used key
            cache.pop(min(cache, key=lambda k: cache[k][1]))
        hit = cache[key] = [None, 0, 0]                        # [renderer, last use, sightings]
    tick = tensoIR.__dict__["_boundary_tick"] = tensoIR.__dict__.get("_boundary_tick", 0) + 1
    hit[1], hit[2] = tick, hit[2] + 1
    if hit[0] is None and hit[2] >= 2:
        from .graph import GraphedRenderer
        hit[0] = GraphedRenderer(tensoIR, key[0], N_samples=N_samples, white_bg=white_bg, is_relight=is_relight, args=args, device=dev)
    return hit[0]


def Renderer_TensoIR_train(rays=None, normal_gt=None, light_idx=None, tensoIR=None, N_samples=-1, ndc_ray=False,
                           white_bg=True, is_train=False, is_relight=True, sample_method="fixed_envirmap",
                           chunk_size=15000, device="cuda", args=None, _no_graph=False):
    """renderer.py:57-127: primary pass + physically-based re-render of the rays with acc > 0.5.
    Same signature and the same 12-key dict."""
    if (BOUNDARY_GRAPHS and not _no_graph and not is_train and not torch.is_grad_enabled() and sample_method == "fixed_envirmap"
            and not ndc_ray and rays is not None and rays.shape[0] >= BOUNDARY_GRAPH_MIN_RAYS and tensoIR.__dict__.get("_capture") is None
            and not (tensoIR.normals_kind == "gt_normals" and normal_gt is not None)
            and getattr(ops, "TIMING", None) is None and getattr(ops, "STATS", None) is None):     # (instrumented passes issue the launches themselves)
        gr = _boundary_graph(tensoIR, rays, N_samples, white_bg, is_relight, device, args)
        if gr is not None:
            try:
                return gr(rays, light_idx)                     # fresh output tensors (the graph's own buffers are reused by the next call)
            except ops._lib.TensoirHipError:                   # capture refused for this call shape: the eager launches are the same work
                tensoIR.__dict__["_boundary_graphs"].pop(next(k for k, v in tensoIR.__dict__["_boundary_graphs"].items() if v[0] is gr), None)
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

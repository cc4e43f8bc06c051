"""Host-side mirror of ``models/relight_utils.py`` (live functions only): same names, argument
meaning and return shapes; per-sample work runs in libtensoir_hip.so."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .field_model import safe_l2_normalize  # noqa: F401  (re-exported, as the reference module does)

MAP_STRIDE = ops.MAP_STRIDE


_CONST_CACHE = {}


def _z_table(n_sample, near, far, device):
    """sample_ray_equally's distances (models/relight_utils.py:716-717), computed by the same torch ops
    on the same device as the reference would; cached per (n, near, far, device)."""
    key = ("z", int(n_sample), float(near), float(far), str(device))
    z = _CONST_CACHE.get(key)
    if z is None:
        t = torch.linspace(0.0, 1.0, n_sample, device=device)
        z = (near * (1.0 - t) + far * t).contiguous()
        _CONST_CACHE[key] = z
    return z


def _on_device(tensoIR, name, src, device):
    """Device-resident copy of a small host-side table of the model (direction grid, solid angles ...)."""
    cache = tensoIR.__dict__.setdefault("_dev_tables", {})
    key = (name, str(device), src.data_ptr(), src._version)
    t = cache.get(name)
    if t is None or t[0] != key:
        t = (key, src.to(device, torch.float32).contiguous())
        cache[name] = t
    return t[1]


def _rec_capacity(n_rays):
    return int(min(max(1 << 20, 16 * n_rays), 1 << 28))


def _gather_then_decode(tensoIR, f, fh, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs, n_dev, full=False):
    """Appearance gather, then the radiance decoder, as two launches (features through HBM).  full: the primary-stage kernels
    (fp32 taps, split-bf16 x3 decoder) whatever the precision policy says."""
    if fh is not None and not full:   # indirect-light precision policy: fp16 shadow taps, fp16 matrix operands
        feat = ops.vm_app_h16(f, fh, rec_xyz, light_idx, rec_ray, light_div, n_dev)
    else:
        feat = ops.vm_app(f, rec_xyz, light_idx, rec_ray, True, False, None, light_div, n_dev)[0]
    # view direction of a record = that of its ray (ray id -> direction via aux_mod on the dense [point][direction] grid)
    return ops.mlp(tensoIR.renderModule.packed(), feat, dirs, rec_ray if dir_map is None else
                   dir_map[rec_ray.long().clamp_(0, dir_map.numel() - 1)].contiguous(), None if full else ops.secondary_mlp_impl(),
                   n_dirs if dir_map is None else 0, n_dev)


def _indirect_state(tensoIR):
    return tensoIR.__dict__.setdefault("_indirect_state", {"verdict": None, "key": None, "storage": None, "age": 0, "why": None,
                                                           "stats": None, "probes": 0, "fallbacks": 0})


def _indirect_key(tensoIR):
    """(parameter versions, parameter storage) of everything the indirect-light kernels read: appearance field + radiance decoder."""
    if tensoIR._field_key is None:          # (callers inside a pass have just refreshed it: the key walk over ~35 parameters
        tensoIR.packed_field()              #  costs ~70 us of host time, and the training loop is host-bound)
    tensoIR.renderModule.packed()
    fk = tensoIR._field_key[0]
    key = (fk, tensoIR.renderModule._key)
    storage = (tuple((a, c) for a, _, c in fk), tuple(a for a, _ in tensoIR.renderModule._key))
    return key, storage


def _indirect_mode(tensoIR, training=False):
    """Which kernels decode this pass's secondary-ray records: "full" (primary-stage kernels), "f16" (the precision policy's fast
    kernels), "hp" (the high-precision fused kernel, ops.indirect_fused_hp) or "probe" (auto policy, no valid verdict for the
    current parameters: run f16, self-check against full, if that fails self-check hp against full, decide).

    auto (ops.INDIRECT_GUARD): a verdict belongs to one parameter version.  Inference passes always use a verdict of exactly
    the current version (so the same parameters render the same image whatever was rendered before).  TRAINING passes
    (`training`: the forward of an optimizer step, where indirect light is a no_grad constant of the loss) carry it over to later
    versions of the SAME storage -- an optimizer step moves a parameter by at most the learning rate -- for
    ops.INDIRECT_PROBE["interval"] versions, then re-establish it; new storage (load, upsample, shrink) re-establishes it at
    once.  The range guard is evaluated for EVERY version (HalfRange, no extra synchronisation) by the caller."""
    if ops.secondary_app_impl() != "h16" and ops.secondary_mlp_impl() in (None, "hp"):
        return "hp" if ops.secondary_mlp_impl() == "hp" else "full"
    if not ops.INDIRECT_GUARD:
        return "f16"
    st = _indirect_state(tensoIR)
    key, storage = _indirect_key(tensoIR)
    # `key` is the version a probe MEASURED; `carried_key` the latest version a training pass carried that verdict over to.
    # Only a training pass may ride on a carried verdict: an inference pass at a version that was never probed probes.
    if st["verdict"] is not None and st["key"] == key and (training or not st.get("train_limit")):
        return st["verdict"]         # (an inference pass never rides on a verdict taken with the training limit)
    if training and st["verdict"] is not None and st["storage"] == storage:
        if st.get("carried_key") == key:
            return st["verdict"]     # (another pass at a version already counted)
        if st["age"] < ops.INDIRECT_PROBE["interval"]:
            st["age"] += 1
            st["carried_key"] = key
            return st["verdict"]
    return "probe"


def _set_verdict(tensoIR, verdict, why, stats=None, train_limit=False):
    st = _indirect_state(tensoIR)
    key, storage = _indirect_key(tensoIR)
    if verdict != "f16" and st["verdict"] != verdict:       # (a version that left the fast kernels: to hp, or all the way to full)
        st["fallbacks"] += 1
    st.update(verdict=verdict, key=key, carried_key=None, storage=storage, age=0, why=why, train_limit=bool(train_limit))
    if stats is not None:
        st["stats"] = stats


def _probe_indirect(tensoIR, f, rgb, n_valid, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs):
    """The self-check of the auto policy: an evenly strided subset of this pass's records decoded by the primary-stage kernels
    and compared with the f16 path's `rgb` rows.  One host synchronisation (only in passes that establish a verdict)."""
    lim = ops.INDIRECT_PROBE
    n_valid = min(int(n_valid), rgb.shape[0])
    if n_valid <= 0:
        return True, {"records": 0}
    step = max(1, n_valid // lim["records"])
    sel = torch.arange(0, n_valid, step, device=rgb.device)[:lim["records"]]
    ref = _gather_then_decode(tensoIR, f, None, rec_xyz[sel].contiguous(), light_idx, rec_ray[sel].contiguous(), light_div, dirs,
                              dir_map, n_dirs, None, full=True)
    d = (rgb[sel] - ref).double()
    v = torch.stack([d.mean(0).abs().max(), d.pow(2).mean().sqrt(), d.abs().max(), ref.double().pow(2).mean().sqrt()]).tolist()
    est = max(lim["w_bias"] * v[0] + lim["w_rms"] * v[1], lim["w_max"] * v[2])     # estimated max error on rgb_with_brdf_map (ops.INDIRECT_PROBE)
    stats = {"kind": "records", "records": int(sel.numel()), "of": n_valid, "bias": v[0], "rms": v[1], "max": v[2], "radiance_rms": v[3],
             "estimate": est}
    ok = est <= lim["limit"]                                   # (NaN fails)
    _indirect_state(tensoIR)["probes"] += 1
    return bool(ok), stats


def _secondary(tensoIR, origins, dirs, n_rays, z, org_map, dir_map, active, light_idx, light_div,
               want_indirect, want_nerfactor=False, n_dirs=0, keep_records=False, ids=None, defer=False, training=False,
               probe_map=None):
    """Shared driver of compute_transmittance / compute_radiance / render_with_BRDF:
    march (+ record the w > thres samples) -> appearance gather -> radiance decoder -> per-ray sum.

    No host synchronisation on the record count: the record buffers are sized from the previous call's
    count (x1.5), the gather / decoder kernels read the actual count from device memory (n_dev), and the
    count is checked once after everything has been queued; an overflow (rare) re-runs the stage."""
    f = tensoIR.packed_field()
    dev = origins.device
    if not want_indirect:
        vis, oma, _ = ops.march_secondary(f, origins, dirs, z, n_rays, org_map, dir_map, active,
                                          tensoIR.march_t_stop, False, 0, want_nerfactor, n_dirs)
        return vis, oma, None
    hints = tensoIR.__dict__.setdefault("_rec_cap_hints", {})      # record capacity learnt per problem size
    cap = hints.get(n_rays, 0)
    first = cap <= 0
    capture = tensoIR.__dict__.get("_capture")
    if capture is not None and first:
        raise ops._lib.TensoirHipError("graph capture needs a warmed-up secondary record-capacity hint")
    if first:
        cap = _rec_capacity(n_rays)
    # a record counter the primary march of this pass has already zeroed on the device (no fill launch); first attempt only
    armed = tensoIR.__dict__.pop("_rec_counter_armed", None)
    force_full = False
    while True:
        extra = {} if ids is None else dict(ray_ids=ids["pair_ids"], n_ids_dev=ids["n_active"], vis=ids["vis"],
                                            rec_cnt=ids["rec_cnt"])
        if armed is not None and armed.device == dev:
            extra["counter"] = armed
        armed = None
        vis, oma, rec = ops.march_secondary(f, origins, dirs, z, n_rays, org_map, dir_map, active,
                                            tensoIR.march_t_stop, True, cap, want_nerfactor, n_dirs, **extra)
        n_total, n_dev = rec["counter"][0:1], rec["counter"][1:2]      # all records / the written prefix consumers may read
        # later attempts: the count travels to the host while the gather / decoder launches are queued
        total_host = None if (first or capture is not None) else ops.AsyncCount(n_total)
        if first:                                      # no history yet: learn the count before sizing buffers
            total = int(n_total.item())
            if total > cap:
                cap = int(total * 1.25) + 1024
                continue
            n_rows = total
        else:
            n_rows = cap
        indirect = None
        mode = "full" if force_full else _indirect_mode(tensoIR, training)
        if mode == "probe" and capture is not None:
            raise ops._lib.TensoirHipError("graph capture needs an established indirect-light precision verdict (run the pass eagerly first)")
        rng = None
        if n_rows > 0:
            rec_ray, rec_w, rec_xyz = rec["ray"][:n_rows], rec["w"][:n_rows], rec["xyz"][:n_rows]
            # light index / view direction of a record = those of its ray (ray id -> point via idx_div,
            # ray id -> direction via aux_mod on the dense [point][direction] grid)
            fh = tensoIR.packed_field_half(_fresh=True) if (mode != "full" and ops.secondary_app_impl() == "h16") else None
            rng = tensoIR.half_range(_fresh=True) if (fh is not None and ops.INDIRECT_GUARD) else None
            if rng is not None and (mode == "probe" or rng.ready()) and not rng.ok():
                # range guard (tir_pack_half_checked's contract): an fp16 product could overflow -> the primary-stage kernels
                _set_verdict(tensoIR, "full", "range", {"bound": rng.bound, "maxima": rng.maxima})
                mode, fh, rng = "full", None, None
            if capture is not None and rng is not None and not rng.ready():
                raise ops._lib.TensoirHipError("graph capture needs a finished range check of the fp16 field shadow (run the pass eagerly first)")

            fusable = dir_map is None and n_dirs > 0 and dirs.shape[0] * 8 <= max(n_rows, 1) and int(f.app_dim) == 27 and int(f.n_acomp) == 48

            def decode(kind, fh=fh):
                """kind: "f16" (fp16 shadow + fp16 decoder), "hp" (fp32 taps, fp16 + fp8-residue weights) or "full"."""
                if kind == "f16" and fh is not None and fusable and ops.fused_indirect() and int(f.n_lights) <= 16:
                    # gather -> basis contraction -> radiance decoder in ONE launch, the feature rows never reach HBM
                    return ops.indirect_fused(f, fh, tensoIR.renderModule.packed(), rec_xyz, light_idx, rec_ray, light_div, dirs, n_dirs, n_dev)
                if kind == "hp" and fusable and ops.AUX_TABLE and ops.MLP_IMPL == "bf16x3" and int(f.n_lights) <= 8:
                    return ops.indirect_fused_hp(f, tensoIR.renderModule.packed(), rec_xyz, light_idx, rec_ray, light_div, dirs, n_dirs, n_dev)
                return _gather_then_decode(tensoIR, f, fh, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs, n_dev,
                                           full=kind != "f16")

            def packed(rgb):
                if keep_records:       # the caller's integration kernel sums the records itself (tir_shade_integrate_records)
                    return {"off": rec["off"], "cnt": rec["cnt"], "w": rec_w, "rgb": rgb}
                return ops.accumulate_records(rec["off"], rec["cnt"], rec_w, rgb, n_rays)

            rgb = decode("f16" if mode == "probe" else mode)
            if mode == "probe":        # auto policy, no verdict for these parameters yet: self-check on this pass's own records
                n_valid = min(total if first else total_host.get(), n_rows)
                verdict = "f16"
                if probe_map is not None:
                    # the caller renders rgb_with_brdf_map from BOTH decodes of ALL records of this pass: the quantity the
                    # tolerance is stated on, measured -- not estimated
                    rgb_full = decode("full")
                    # a TRAINING pass renders the map as a no_grad constant of the loss (models/relight_utils.py:344): there the
                    # policy accepts up to the contract's own tolerance; inference / export use the strict limit
                    map_limit = ops.INDIRECT_PROBE["train_map_limit" if training else "map_limit"]

                    def measure(cand):
                        d = (cand[:n_valid] - rgb_full[:n_valid]).double()
                        delta = probe_map(vis, packed(cand), packed(rgb_full))
                        v = (torch.stack([d.mean(0).abs().max(), d.pow(2).mean().sqrt(), d.abs().max()]).tolist() if n_valid else [0.0, 0.0, 0.0])
                        return delta, {"kind": "map", "map_max_abs": delta, "records": n_valid, "rays": n_rays, "bias": v[0], "rms": v[1], "max": v[2],
                                       "limit": map_limit}
                    delta, stats = measure(rgb)
                    _indirect_state(tensoIR)["probes"] += 1
                    if not delta <= map_limit:                               # (NaN fails)
                        verdict = "full"
                        if ops.INDIRECT_HP:                                  # first fallback: the high-precision fused kernel, checked the same way
                            rgb_hp = decode("hp")
                            delta_hp, stats_hp = measure(rgb_hp)
                            stats = {**stats_hp, "f16": stats}
                            if delta_hp <= map_limit:
                                verdict, rgb = "hp", rgb_hp
                        if verdict == "full":
                            rgb = rgb_full
                else:
                    ok, stats = _probe_indirect(tensoIR, f, rgb, n_valid, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs)
                    if not ok:
                        verdict = "full"
                        if ops.INDIRECT_HP and fusable and ops.AUX_TABLE and ops.MLP_IMPL == "bf16x3":
                            rgb_hp = decode("hp")
                            ok_hp, stats_hp = _probe_indirect(tensoIR, f, rgb_hp, n_valid, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs)
                            stats = {**stats_hp, "f16": stats}
                            if ok_hp:
                                verdict, rgb = "hp", rgb_hp
                        if verdict == "full":
                            rgb = decode("full")
                _set_verdict(tensoIR, verdict, "probe", stats, train_limit=training and probe_map is not None)
                rng = None             # (evaluated above)
            indirect = packed(rgb)
        else:
            indirect = torch.zeros((n_rays, 3), dtype=torch.float32, device=dev)

        def range_failed(rng=rng):
            """The range guard of a carried-over verdict, evaluated once the maxima of THIS version's shadow have arrived."""
            if rng is None or rng.ok():
                return False
            _set_verdict(tensoIR, "full", "range", {"bound": rng.bound, "maxima": rng.maxima})
            return True

        if first:
            if range_failed():
                force_full = True
                continue
            break
        if capture is not None:                        # HIP-graph capture: the graph owner reads the counter after replay
            if rng is not None and not rng.ok():
                raise ops._lib.TensoirHipError("graph capture: the fp16 field shadow fails the range guard")
            capture.append((n_total, cap, ("secondary", n_rays)))
            return vis, oma, indirect
        if defer:                                      # the caller checks after ITS remaining launches are queued too
            def check(total_host=total_host, cap=cap, range_failed=range_failed):
                total = total_host.get()
                if total > cap:
                    hints.pop(n_rays, None)            # the re-run learns the count first
                    return False
                hints[n_rays] = max(int(total * 1.5) + 4096, 1 << 14, int(0.97 * hints.get(n_rays, 0)))
                return not range_failed()              # (a failed range guard: the re-run decodes with the primary-stage kernels)
            tensoIR.__dict__.setdefault("_pending_checks", []).append(check)
            return vis, oma, indirect
        total = total_host.get()                       # waits for the march only, not for what was queued behind it
        if total <= cap:
            if range_failed():
                force_full = True
                continue
            break
        cap = int(total * 1.25) + 1024                 # overflow: some rays were dropped -> redo with room
    if len(hints) > 32:
        hints.clear()
    hints[n_rays] = max(int(total * 1.5) + 4096, 1 << 14, int(0.97 * hints.get(n_rays, 0)))
    return vis, oma, indirect


@torch.no_grad()
def compute_transmittance(tensoIR, surf_pts, light_in_dir, nSample=128, vis_near=0.1, vis_far=2, device="cuda"):
    """models/relight_utils.py:657-705 -> (nerv_vis [N], nerfactor_vis [N]).
    nSample <= 256 (the secondary-march kernels' limit; the reference's configs use 96): larger values raise TensoirHipError."""
    surf_pts = surf_pts.to(torch.float32).contiguous()
    light_in_dir = light_in_dir.to(torch.float32).contiguous()
    z = _z_table(nSample, vis_near, vis_far, surf_pts.device)
    vis, oma, _ = _secondary(tensoIR, surf_pts, light_in_dir, surf_pts.shape[0], z, None, None, None,
                             None, 0, False, True)
    return vis, oma


@torch.no_grad()
def compute_radiance(tensoIR, surf_pts, light_in_dir, light_idx, nSample=128, vis_near=0.05, vis_far=1.5,
                     device=None):
    """models/relight_utils.py:777-834 -> (nerv_vis [N], nerfactor_vis [N], indirect [N,3]).  nSample <= 256 (see compute_transmittance)."""
    surf_pts = surf_pts.to(torch.float32).contiguous()
    light_in_dir = light_in_dir.to(torch.float32).contiguous()
    li = light_idx.reshape(-1).to(surf_pts.device, torch.int32).contiguous()
    z = _z_table(nSample, vis_near, vis_far, surf_pts.device)
    return _secondary(tensoIR, surf_pts, light_in_dir, surf_pts.shape[0], z, None, None, None, li, 0,
                      True, True)


@torch.no_grad()
def compute_secondary_shading_effects(tensoIR, surface_pts, surf2light, light_idx, nSample=96, vis_near=0.05,
                                      vis_far=1.5, chunk_size=15000, device="cuda"):
    """models/relight_utils.py:344-399 -> (visibility [N,1], indirect [N,3]).  chunk_size only bounded
    the reference's intermediates; the fused march needs no chunking."""
    vis, _, ind = compute_radiance(tensoIR, surface_pts, surf2light, light_idx, nSample, vis_near, vis_far)
    return vis.reshape(-1, 1), ind.reshape(-1, 3)


def GGX_specular(normal, pts2c, pts2l, roughness, fresnel):
    """models/relight_utils.py:17-50 -> tir_ggx_specular."""
    return ops.ggx_specular(normal, pts2c, pts2l, roughness, fresnel)


brdf_specular = GGX_specular


def linear2srgb_torch(tensor_0to1):
    """models/relight_utils.py:489-515 (elementwise glue for callers outside the fused kernels)."""
    if isinstance(tensor_0to1, np.ndarray):
        x = np.clip(tensor_0to1, 0, 1)
        return np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x + 1e-6, 1 / 2.4) - (1.055 - 1))
    x = tensor_0to1.clamp(0, 1)
    return torch.where(x <= 0.0031308, x * 12.92, 1.055 * torch.pow(x + 1e-6, 1 / 2.4) - (1.055 - 1))


def _maps_from_parts(depth_map, normal_map, albedo_map, roughness_map, fresnel_map):
    M = depth_map.shape[0]
    maps = torch.zeros((M, MAP_STRIDE), dtype=torch.float32, device=depth_map.device)
    maps[:, 3] = depth_map.reshape(-1)
    maps[:, 4:7] = normal_map
    maps[:, 7:10] = albedo_map
    maps[:, 10] = roughness_map.reshape(M, -1)[:, 0]
    maps[:, 11:14] = fresnel_map
    return maps


def finish_pending(tensoIR):
    """Run the deferred record-capacity checks of the shading stage; False = something overflowed (re-run the pass)."""
    ok = True
    for check in tensoIR.__dict__.pop("_pending_checks", []):
        ok = check() and ok
    return ok


def shade_from_maps(tensoIR, maps, rays, light_idx, sample_method="fixed_envirmap", args=None,
                    use_linear2srgb=True, acc_thres=-1e30, return_aux=False, _defer_check=False):
    """The body of render_with_BRDF (models/relight_utils.py:417-480) on packed [M,20] map rows.
    Rows with acc <= acc_thres are background: no secondary rays, white output (renderer.py:86-106)."""
    dev = maps.device
    M = maps.shape[0]
    rays = ops.to_device(rays, dev, torch.float32).contiguous()
    li = ops.to_device(light_idx.reshape(-1), dev, torch.int32).contiguous()
    if sample_method == "fixed_envirmap":
        dirs = _on_device(tensoIR, "fixed_viewdirs", tensoIR.fixed_viewdirs, dev)
    else:
        dirs = ops.to_device(tensoIR.gen_light_incident_dirs(method=sample_method), dev, torch.float32).contiguous()
    D = dirs.shape[0]
    area = _on_device(tensoIR, "light_area_weight", tensoIR.light_area_weight, dev)
    z = _z_table(int(args.second_nSample), args.second_near, args.second_far, dev)
    if M == 0:
        out = torch.zeros((0, 3), dtype=torch.float32, device=dev)
        return (out, None) if return_aux else out
    train = torch.is_grad_enabled() and (maps.requires_grad or any(t.requires_grad for t in tensoIR.light_parameters()))
    fuse = not train and not return_aux
    with torch.no_grad():      # compute_secondary_shading_effects is @torch.no_grad (models/relight_utils.py:344)
        # only the pairs that pass the cosine / acc masks get a secondary ray: compacted id list (the reference's
        # boolean-mask indexing, :440-441); the pair counter is re-armed by the integration kernel at the end
        n_active = tensoIR.__dict__.get("_pair_counter")
        if n_active is None or n_active.device != dev:
            n_active = tensoIR.__dict__["_pair_counter"] = torch.zeros((1,), dtype=torch.int32, device=dev)
        surf, active, pair_ids, vis0, cnt0 = ops.shade_setup_compact(maps.detach(), rays, dirs, acc_thres, n_active)
        ids = {"pair_ids": pair_ids, "n_active": n_active, "vis": vis0, "rec_cnt": cnt0}
        equal_area = sample_method == "stratifed_sample_equal_areas"
        w_d = None if equal_area else area

        def probe_map(vis_p, ind_a, ind_b):
            """max |d rgb_with_brdf_map| over this pass's rays between two decodes of the secondary records (auto policy):
            the integration kernel run on both (no_grad; the pair counter is left alone).  One host synchronisation."""
            env_p = tensoIR.get_light_rgbs(dirs, device=dev).detach()
            outs = []
            for ind in (ind_a, ind_b):
                if isinstance(ind, dict):
                    outs.append(ops.shade_integrate_records(maps.detach(), rays, dirs, li, vis_p.view(M, D), ind["off"], ind["cnt"], ind["w"],
                                                            ind["rgb"], env_p, w_d, equal_area, use_linear2srgb, acc_thres, reset_counter=None))
                else:
                    outs.append(ops.shade_integrate(maps.detach(), rays, dirs, li, vis_p.view(M, D), ind.view(M, D, 3), env_p, w_d, equal_area,
                                                    use_linear2srgb, acc_thres))
            return float((outs[0] - outs[1]).abs().max())

        try:
            vis, _, ind = _secondary(tensoIR, surf, dirs, M * D, z, None, None, None, li, D, True, False, D,
                                     keep_records=fuse, ids=ids, defer=_defer_check, training=train, probe_map=probe_map)
        except BaseException:
            # the pair counter is re-armed by the integration kernel at the end of the pass; if the pass is abandoned
            # in between (capacity error under capture, OOM ...) it must not stay non-zero for the next call
            if tensoIR.__dict__.get("_capture") is None:
                n_active.zero_()
            else:
                tensoIR.__dict__.pop("_pair_counter", None)
            raise
    env = tensoIR.get_light_rgbs(dirs, device=dev)
    if not isinstance(ind, dict):
        ids["n_active"].zero_()            # the fused integration kernel (which re-arms the pair counter) is not on this route
    if train:
        from . import training
        rgb = training.ShadeFn.apply(maps, env, rays, dirs, li, vis.view(M, D), ind.view(M, D, 3), w_d, equal_area,
                                     use_linear2srgb, acc_thres)
    elif isinstance(ind, dict):
        rgb = ops.shade_integrate_records(maps, rays, dirs, li, vis.view(M, D), ind["off"], ind["cnt"], ind["w"],
                                          ind["rgb"], env, w_d, equal_area, use_linear2srgb, acc_thres,
                                          reset_counter=ids["n_active"])
    else:
        rgb = ops.shade_integrate(maps, rays, dirs, li, vis.view(M, D), ind.view(M, D, 3), env, w_d, equal_area,
                                  use_linear2srgb, acc_thres)
    if return_aux:
        return rgb, {"vis": vis.view(M, D), "indirect": ind.view(M, D, 3), "env": env, "surf": surf,
                     "active": active}
    return rgb


def render_with_BRDF(depth_map, normal_map, albedo_map, roughness_map, fresnel_map, rays, tensoIR, light_idx,
                     sample_method="fixed_envirmap", chunk_size=15000, device="cuda", use_linear2srgb=True,
                     args=None):
    """models/relight_utils.py:403-483 (same positional arguments; roughness_map is [M,3] or [M,1])."""
    maps = _maps_from_parts(depth_map.to(torch.float32), normal_map, albedo_map, roughness_map, fresnel_map)
    return shade_from_maps(tensoIR, maps, rays, light_idx, sample_method, args, use_linear2srgb)


class Environment_Light:
    """models/relight_utils.py:110-205 with the HDR maps handed in as arrays (the reference reads
    ``*.hdr`` files with OpenCV, which is I/O outside the hot path)."""

    def __init__(self, hdr_path=None, device="cuda", hdr_maps=None):
        self.hdr_rgbs, self.hdr_pdf_sample, self.hdr_pdf_return, self.hdr_dir = {}, {}, {}, {}
        self.hdr_row_cdf, self.hdr_col_cdf, self.hdr_cdf_guide, self._cell_records = {}, {}, {}, {}
        self._draws = 0
        maps = dict(hdr_maps or {})
        if torch.device(device).type == "cuda" and not torch.cuda.is_available():
            raise ops._lib.TensoirHipError("Environment_Light: no GPU is visible; tensoir_amd has no CPU path")
        if hdr_path is not None and str(hdr_path).startswith("synthetic"):       # no *.hdr files offline: seeded maps
            from . import synth
            from .synth_dataset import parse_spec
            spec = {"h": 32, "w": 64, **{k: v for k, v in parse_spec(hdr_path).items() if k in ("h", "w")}}
            maps.update(synth.make_hdr_maps(synth.HDR_NAMES, int(spec["h"]), int(spec["w"])))
        elif hdr_path is not None:
            for file in os.listdir(hdr_path):
                if file.endswith(".hdr"):
                    maps[file.split(".")[0]] = torch.from_numpy(read_hdr(os.path.join(hdr_path, file)))
        for name, rgbs in maps.items():
            rgbs = torch.as_tensor(rgbs, dtype=torch.float32).cpu()
            H, W, _ = rgbs.shape
            inten = torch.sum(rgbs, dim=2, keepdim=True)
            sin_theta = torch.sin(torch.linspace(0 + 0.5 / H, np.pi - 0.5 / H, H))
            pdf = inten * sin_theta.view(-1, 1, 1)
            pdf = pdf / torch.sum(pdf)
            pdf_ret = pdf * H * W / (2 * np.pi * np.pi * sin_theta.view(-1, 1, 1))
            lat, lng = np.pi / H, 2 * np.pi / W
            phi, theta = torch.meshgrid([torch.linspace(np.pi / 2 - 0.5 * lat, -np.pi / 2 + 0.5 * lat, H),
                                         torch.linspace(np.pi - 0.5 * lng, -np.pi + 0.5 * lng, W)], indexing="ij")
            dirs = torch.stack([torch.cos(theta) * torch.cos(phi), torch.sin(theta) * torch.cos(phi),
                                torch.sin(phi)], dim=-1).view(H, W, 3)
            self.hdr_rgbs[name] = rgbs.to(device)
            self.hdr_pdf_sample[name] = pdf.to(device)
            self.hdr_pdf_return[name] = pdf_ret.to(device)
            self.hdr_dir[name] = dirs.to(device)
            # inverse-CDF tables of the device sampler (tir_env_sample_setup): row marginal + row-conditional columns,
            # accumulated in fp64 so that the fp32 tables are the correctly rounded prefix sums
            p2 = pdf.view(H, W).double()
            rows = p2.sum(dim=1)
            row_cdf = torch.cumsum(rows, 0) / rows.sum()
            col_cdf = torch.cumsum(p2, 1) / rows.clamp(min=1e-300).unsqueeze(1)
            row_cdf[-1] = 1.0
            col_cdf[:, -1] = 1.0
            self.hdr_row_cdf[name] = row_cdf.float().to(device).contiguous()
            self.hdr_col_cdf[name] = col_cdf.float().to(device).contiguous()
            # guide tables of that search (ops.cdf_guide_tables): ~4 dependent loads per draw instead of log2(H) + log2(W)
            self.hdr_cdf_guide[name] = (ops.cdf_guide_tables(self.hdr_row_cdf[name], self.hdr_col_cdf[name])
                                        if os.environ.get("TENSOIR_CDF_GUIDE", "1") != "0" else None)
            # :144-146, the tables of sample_type="uniform" (solid-angle-uniform cells; like the reference they belong to the
            # size of the LAST map read)
            updf = torch.ones(H, W, 1) * sin_theta.view(-1, 1, 1) / (H * W)
            updf = updf / torch.sum(updf)
            self.envir_map_uniform_pdf = updf.to(device)
            self.envir_map_uniform_pdf_return = (updf * H * W / (2 * np.pi * np.pi * sin_theta.view(-1, 1, 1))).to(device)

    @torch.no_grad()
    def sample_light(self, light_name, bs, num_samples, sample_type="importance"):
        """:150-188.  The reference draws torch.multinomial over an expanded [bs, H*W] pdf; sampling bs*num
        indices from the 1-D pdf is the same distribution without materialising bs copies."""
        if sample_type == "importance":
            pdf, pdf_ret = self.hdr_pdf_sample[light_name].view(-1), self.hdr_pdf_return[light_name].view(-1)
        elif sample_type == "uniform":                   # :174-188 (no caller in the reference's scripts)
            pdf, pdf_ret = self.envir_map_uniform_pdf.view(-1), self.envir_map_uniform_pdf_return.view(-1)
            if pdf.numel() != self.hdr_dir[light_name].shape[0] * self.hdr_dir[light_name].shape[1]:
                raise ValueError(f"sample_light(uniform): map {light_name!r} has another size than the uniform tables (built for the last map read)")
        else:
            raise ValueError(f"sample_light: unknown sample_type {sample_type!r}")
        idx = torch.multinomial(pdf, bs * num_samples, replacement=True).view(bs, num_samples)
        d = self.hdr_dir[light_name].view(-1, 3)[idx]
        rgb = self.hdr_rgbs[light_name].view(-1, 3)[idx]
        p = pdf_ret[idx].unsqueeze(-1)
        return d, rgb, p

    @torch.no_grad()
    def sample_cells(self, light_name, normal, num_samples):
        """Device-side sample_light + cosine mask (tir_env_sample_setup): per surface point `num_samples` cells of the map
        drawn from hdr_pdf_sample by inverse-CDF search with Philox uniforms keyed by the framework's CUDA seed.  Returns
        (cell [M, Ns] int32, active [M, Ns] uint8): direction / radiance / pdf are hdr_dir / hdr_rgbs / hdr_pdf_return
        at the cell -- the [M, Ns, 3] tensors of the reference are never materialised."""
        self._draws += 1
        return ops.env_sample_setup(self.hdr_row_cdf[light_name], self.hdr_col_cdf[light_name],
                                    self.hdr_dir[light_name].view(-1, 3), normal, num_samples,
                                    torch.cuda.initial_seed(), self._draws)

    @torch.no_grad()
    def sample_cells_listed(self, light_name, normal, num_samples, bins=(1, 1), block_pairs=256, m_dev=None):
        """sample_cells with the same draws (same Philox counters) + the compacted list of the unmasked pairs, optionally
        direction-binned inside blocks of `block_pairs` pairs (tir_env_sample_setup_list).  Returns (cell [M, Ns],
        vis [M, Ns] with the masked pairs' zeros, pair_ids [M * Ns], n_active [1] on the device)."""
        self._draws += 1
        rec = self.cell_records(light_name)               # the direction comes from the records the integration reads later
        return ops.env_sample_setup_list(self.hdr_row_cdf[light_name], self.hdr_col_cdf[light_name],
                                           self.hdr_dir[light_name].view(-1, 3) if rec is None else rec, normal, num_samples,
                                           torch.cuda.initial_seed(), self._draws, bins, block_pairs,
                                           self.hdr_cdf_guide.get(light_name), m_dev)

    def cell_records(self, light_name):
        """[H*W, 8] records {direction, pdf_return, radiance, 0} of a map (ops.pack_env_cells), built on first use: the
        integration kernel reads one 32-byte record per sample instead of three tables.  TENSOIR_ENV_RECORDS=0: None."""
        if os.environ.get("TENSOIR_ENV_RECORDS", "1") == "0":
            return None
        rec = self._cell_records.get(light_name)
        if rec is None:
            rec = ops.pack_env_cells(self.hdr_dir[light_name], self.hdr_rgbs[light_name], self.hdr_pdf_return[light_name])
            self._cell_records[light_name] = rec
        return rec

    def get_light(self, light_name, incident_dir):
        """:191-205 (background lookup, bilinear, align_corners=True) -> tir_env_lookup."""
        return ops.env_lookup(self.hdr_rgbs[light_name], incident_dir.reshape(-1, 3))


def read_hdr(path):
    import cv2
    with open(path, "rb") as h:
        buf = np.frombuffer(h.read(), np.uint8)
    return cv2.cvtColor(cv2.imdecode(buf, cv2.IMREAD_UNCHANGED), cv2.COLOR_BGR2RGB)


@torch.no_grad()
def relight_with_envmap(tensoIR, surface_xyz, normal, albedo, roughness, fresnel, rays_d, light_dir,
                        light_rgb, light_pdf, nSample=96, vis_near=0.05, vis_far=1.5):
    """Loop body of scripts/relight_importance.py:119-170 for one environment map, given the samples
    drawn by Environment_Light.sample_light: cosine mask -> visibility march -> BRDF*L*cos/pdf mean -> sRGB."""
    dev = surface_xyz.device
    M, Ns = light_dir.shape[:2]
    light_dir = light_dir.to(torch.float32).contiguous()
    normal = normal.to(torch.float32).contiguous()
    cosine = torch.einsum("ijk,ik->ij", light_dir, normal)
    active = (cosine > 1e-6).to(torch.uint8).contiguous()
    key = ("orgmap", M, Ns, str(dev))
    org_map = _CONST_CACHE.get(key)
    if org_map is None:
        pair = torch.arange(M * Ns, dtype=torch.int32, device=dev)
        org_map = torch.div(pair, Ns, rounding_mode="floor").to(torch.int32)
        if len(_CONST_CACHE) > 64:
            _CONST_CACHE.clear()
        _CONST_CACHE[key] = org_map
    z = _z_table(nSample, vis_near, vis_far, dev)
    vis, _, _ = ops.march_secondary(tensoIR.packed_field(), surface_xyz.to(torch.float32).contiguous(),
                                    light_dir.view(-1, 3), z, M * Ns, org_map, None, active.view(-1),
                                    tensoIR.march_t_stop, False, 0, False)
    return ops.relight_importance(normal, albedo, roughness, fresnel, rays_d, light_dir, light_rgb,
                                  light_pdf, vis.view(M, Ns))


@torch.no_grad()
def relight_importance_sampled(tensoIR, env, light_name, surface_xyz, normal, albedo, roughness, fresnel, rays_d,
                               num_samples=512, nSample=96, vis_near=0.05, vis_far=1.5, m_dev=None):
    """The loop body of scripts/relight_importance.py:119-170 for one environment map, entirely on the device:
    importance sampling + cosine mask (tir_env_sample_setup) -> visibility march of the unmasked (point, cell) pairs with
    the map's direction table as `dirs` and the cell index as `dir_map` -> BRDF x radiance x cosine / pdf mean -> sRGB
    (tir_relight_importance_cells).  Per sample 5 bytes of bookkeeping instead of the reference's 28 + masks.
    m_dev (int32 device scalar): the arrays have capacity M rows of which only the first m_dev are surface points
    (ops.surface_compact); rows beyond are neither sampled nor marched, their output rows are left unwritten."""
    dev = surface_xyz.device
    normal = normal.to(torch.float32).contiguous()
    M = normal.shape[0]
    if m_dev is not None and (not torch.is_tensor(m_dev) or m_dev.numel() != 1 or m_dev.dtype != torch.int32 or m_dev.device != dev):
        raise ValueError("m_dev: expected one int32 element on the points' device")
    if M == 0:
        # A call consumes one draw counter whatever M is (since round 5: the device-compacted chunk call, relight_chunk, cannot know
        # that a chunk is all background, and both routes must draw the same sequence).  Images rendered by rounds <= 4 with the
        # host-masked loop are therefore not reproduced bit for bit on views that contain all-background chunks (HISTORY, round 5).
        env._draws += 1
        return torch.zeros((0, 3), dtype=torch.float32, device=dev)
    order, bins, block_pairs = ops.c5_pair_order()
    if m_dev is not None and (order == "mask" or env.cell_records(light_name) is None):
        raise ops._lib.TensoirHipError("a device-side surface-point count needs the pair-list sampler and the packed cell records "
                                       "(TENSOIR_C5_PAIRS != mask, TENSOIR_ENV_RECORDS != 0)")
    if order == "mask":
        cell, active = env.sample_cells(light_name, normal, num_samples)
        listed = {}
    else:
        # :127-131 query visibility for the unmasked pairs only: so does the march, from a compacted list
        cell, vis0, pair_ids, n_active = env.sample_cells_listed(light_name, normal, num_samples, bins, block_pairs, m_dev)
        active = None
        listed = dict(ray_ids=pair_ids, n_ids_dev=n_active, vis=vis0.view(-1))
    key = ("orgmap", M, num_samples, str(dev))
    org_map = _CONST_CACHE.get(key)
    if org_map is None:
        pair = torch.arange(M * num_samples, dtype=torch.int32, device=dev)
        org_map = torch.div(pair, num_samples, rounding_mode="floor").to(torch.int32)
        if len(_CONST_CACHE) > 64:
            _CONST_CACHE.clear()
        _CONST_CACHE[key] = org_map
    z = _z_table(nSample, vis_near, vis_far, dev)
    env_dir = env.hdr_dir[light_name].view(-1, 3)
    vis, _, _ = ops.march_secondary(tensoIR.packed_field(), surface_xyz.to(torch.float32).contiguous(), env_dir, z,
                                    M * num_samples, org_map, cell.view(-1), None if active is None else active.view(-1),
                                    tensoIR.march_t_stop, False, 0, False, **listed)
    return ops.relight_importance_cells(normal, albedo, roughness, fresnel, rays_d, cell, env_dir,
                                        env.hdr_rgbs[light_name].view(-1, 3), env.hdr_pdf_return[light_name].view(-1),
                                        vis.view(M, num_samples), env_cell=env.cell_records(light_name), m_dev=m_dev)


@torch.no_grad()
def relight_chunk(tensoIR, env, light_names, rays, light_idx, num_samples=512, nSample=96, vis_near=0.05, vis_far=1.5,
                  N_samples=-1, out=None):
    """One chunk of scripts/relight_importance.py:93-185 for ALL environment maps without a host round trip: primary pass ->
    the acc > 0.5 rows compacted ON THE DEVICE (the script's boolean-mask indexing, :99-113, is a synchronisation per chunk
    plus ~12 indexing launches) -> per map: importance sampling, visibility march and BRDF integration bounded by the
    device-side point count -> relit colour where a ray hit, background lookup elsewhere (:166-171), written side by side into
    ONE [B, 3 * n_maps] buffer.  Same Philox counters per (point, sample) as the host-compacted sequence: identical colours.
    Returns (out [B, 3 n_maps], primary-pass tuple, compacted surface dict).
    What the caller still applies, as the script does AFTER its loop body (scripts/relight_importance.py:166-176): hit rows hold the
    relit colour already clamped and sRGB-encoded by the integration kernel; BACKGROUND rows hold the raw environment lookup
    (Environment_Light.get_light) -- the script's clamp, tone mapping of the background and its `acc > 0.9` blend of the two
    (`:172-176`) are the caller's (prim[6] is acc_map)."""
    dev = rays.device
    rays = rays.to(torch.float32).contiguous()
    B = rays.shape[0]
    names = list(light_names)
    prim, maps = tensoIR(rays, light_idx, N_samples=N_samples, _return_maps=True)
    c = ops.surface_compact(maps, rays, 0.5)
    if out is None:
        out = torch.empty((B, 3 * max(len(names), 1)), dtype=torch.float32, device=dev)
    for i, name in enumerate(names):
        rgb = relight_importance_sampled(tensoIR, env, name, c["surf"], c["normal"], c["albedo"], c["rough"], c["fresnel"], c["rays_d"],
                                         num_samples, nSample, vis_near, vis_far, m_dev=c["n_hit"])
        ops.env_compose(env.hdr_rgbs[name], rays, c["slot"], rgb, out, 3 * i)
    return out, prim, c

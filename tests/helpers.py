"""Shared test helpers: golden fixture -> checkpoint / oracle Scene, error metrics."""
import numpy as np
import torch

from oracle import tensoir_oracle as O


def golden_checkpoint(g):
    """Rebuild the reference-format checkpoint stored in tests/golden/small_scene.npz."""
    sd = {k[3:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd/")}
    rot = [int(r) for r in g["scene/light_rotation"]]
    kwargs = {
        "aabb": torch.from_numpy(np.array(g["scene/aabb"])), "gridSize": [int(x) for x in g["scene/grid"]],
        "density_n_comp": [16, 16, 16], "appearance_n_comp": [48, 48, 48], "app_dim": 27,
        "density_shift": -10, "alphaMask_thres": 0.001, "distance_scale": 25,
        "rayMarch_weight_thres": 0.0001, "fea2denseAct": "softplus", "near_far": [2.0, 6.0],
        "step_ratio": 0.5, "shadingMode": "MLP_Fea", "pos_pe": 2, "view_pe": 2, "fea_pe": 2,
        "featureC": 128, "normals_kind": "derived_plus_predicted", "light_num": len(rot),
        "light_kind": "sg", "numLgtSGs": 128, "light_rotation": rot,
    }
    vol = torch.from_numpy(np.array(g["scene/alpha_volume"]))
    ckpt = {"kwargs": kwargs, "state_dict": sd,
            "alphaMask.shape": tuple(vol.shape),
            "alphaMask.mask": np.packbits(vol.bool().numpy().reshape(-1)),
            "alphaMask.aabb": torch.from_numpy(np.array(g["scene/alpha_aabb"]))}
    return ckpt


def scene_from_checkpoint(ckpt, envmap_h, envmap_w):
    vol = aabb = None
    if "alphaMask.aabb" in ckpt:
        n = int(np.prod(ckpt["alphaMask.shape"]))
        vol = torch.from_numpy(np.unpackbits(ckpt["alphaMask.mask"])[:n].reshape(ckpt["alphaMask.shape"])).float()
        aabb = ckpt["alphaMask.aabb"]
    return O.scene_from_state_dict(ckpt["state_dict"], ckpt["kwargs"], vol, aabb, envmap_h, envmap_w)


def golden_scene(g):
    h, w = [int(x) for x in g["scene/envmap_hw"]]
    return scene_from_checkpoint(golden_checkpoint(g), h, w)


def T(g, key):
    return torch.from_numpy(np.array(g[key]))


def max_err(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max()) if a.numel() else 0.0


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor): the parity metric (north_star: 1e-4 relative)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    if a.numel() == 0:
        return 0.0
    return float(((a - b).abs() / b.abs().clamp(min=floor)).max())


def scene_from_model(ckpt, model, envmap_h, envmap_w):
    """Oracle Scene of a synthetic checkpoint whose occupancy mask was built on the device
    (model.updateAlphaMask): same parameters, same 0/1 volume, same mask aabb."""
    ck = dict(ckpt)
    if model.alphaMask is not None:
        vol = model.alphaMask.alpha_volume[0, 0].bool().cpu()
        ck["alphaMask.shape"] = tuple(vol.shape)
        ck["alphaMask.mask"] = np.packbits(vol.numpy().reshape(-1))
        ck["alphaMask.aabb"] = model.alphaMask.aabb.cpu()
    return scene_from_checkpoint(ck, envmap_h, envmap_w)


def parity_metrics(a, b, min_norm=1e-2):
    """The three error figures reported for a rendered map (hip a vs reference b):
    max_abs, max |d| / max(|ref|, 1) (the test metric; maps live in [0,1] / unit normals / depth ~4) and the
    true per-pixel relative error ||d|| / ||ref|| over pixels with ||ref|| > min_norm (vector maps: L2 over channels)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    if a.numel() == 0:
        return {"max_abs": 0.0, "max_rel_floor1": 0.0, "max_rel_pixel": 0.0}
    d = (a - b).abs()
    a2, b2, d2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1), (a - b).reshape(a.shape[0], -1)
    nb = b2.norm(dim=-1)
    sel = nb > min_norm
    relpix = float((d2.norm(dim=-1)[sel] / nb[sel]).max()) if bool(sel.any()) else 0.0
    return {"max_abs": float(d.max()), "max_rel_floor1": float((d / b.abs().clamp(min=1.0)).max()),
            "max_rel_pixel": relpix}


def ggx_flip_rays(ref_normal_map, rays, eps=1e-5):
    """Rays on which the reference's physically-based colour is DISCONTINUOUS in its inputs: GGX_specular flips the normal by
    sign(N.V) (models/relight_utils.py:30-31: ``N = N * NoV.sign()``), so a ray whose composited normal is perpendicular to the
    view direction within fp32 noise (|N.V| < eps; the maps themselves agree to ~1e-5) takes either branch -- the specular
    term, hence rgb_with_brdf_map, jumps by percents (measured: one ray of the 4096-ray bench batch, oracle N.V = +1.0e-6, split-bf16
    decoders -8e-8, exact fp32 decoders +1.0e-6: profiles/r05_outlier_diag.json).  Such rays are reported, not compared."""
    n = torch.as_tensor(ref_normal_map).detach().double().cpu()
    d = torch.as_tensor(rays).detach().double().cpu()[:, 3:6]
    n = n / n.norm(dim=-1, keepdim=True).clamp(min=1e-6)
    v = -d / d.norm(dim=-1, keepdim=True).clamp(min=1e-6)
    return (n * v).sum(-1).abs() < eps


def depth_discontinuity_rays(O, sc, ref, rays, light_idx, candidates, n_sample, tol=1e-5, limit=16):
    """Rays on which the reference's physically-based colour is DISCONTINUOUS in the ray's own depth: the secondary rays start at
    rays_o + depth * rays_d (models/relight_utils.py:433), and a secondary sample that sits on an occupancy-cell or bounding-box
    boundary within the last bits is culled or not (:683-695, :803-815) -- the visibility of that light direction, hence
    rgb_with_brdf_map, jumps.  The criterion involves the ORACLE only: its own render_with_brdf on its own maps, with its own depth
    moved by -2 ... +2 ulps, varies by more than `tol` (measured on the headline batch: two rays of 4096 jump by 2.1e-4 and 1.3e-4 for
    ONE ulp of depth, their neighbours by 1e-7 per ulp; profiles/r06_depth_discontinuity_rays.txt).  Only `candidates` (ray indices into
    the oracle's rows, at most `limit`) are examined.  -> {ray: the oracle's own spread}."""
    import numpy as np
    out = {}
    rays = torch.as_tensor(rays).detach().cpu().float()
    light_idx = torch.as_tensor(light_idx).detach().cpu()
    for i in [int(c) for c in candidates][:limit]:
        g = lambda k: ref[k][i:i + 1]
        d0 = g("depth_map").reshape(1).float().numpy()
        vals = []
        for u in (-2, -1, 0, 1, 2):
            d = d0.copy()
            for _ in range(abs(u)):
                d = np.nextafter(d, np.float32(np.inf if u > 0 else -np.inf))
            with torch.no_grad():
                o = O.render_with_brdf(sc, torch.from_numpy(d), g("normal_map"), g("albedo_map"), g("roughness_map").reshape(1, -1).expand(1, 3),
                                       g("fresnel_map"), rays[i:i + 1], light_idx[i:i + 1], n_sample=n_sample)
            vals.append(o.reshape(-1))
        v = torch.stack(vals)
        spread = float((v.max(0).values - v.min(0).values).max())
        if spread > tol:
            out[i] = spread
    return out

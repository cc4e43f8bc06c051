"""Synthetic TensoIR scenes + camera rays (no dataset is available offline).

Produces a checkpoint in the reference's own format (``{'kwargs', 'state_dict',
'alphaMask.*'}``, models/tensorBase_rotated_lights.py:675-683) so the same
object feeds ``tensoir_amd.TensorVMSplit.load`` and -- in tests -- the CPU oracle.
Recipe: SURVEY.md section 8(d) (Gaussian-blob density so that rays hit a surface).
"""
from __future__ import annotations

import math

import numpy as np
import torch

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


def _linear_init(gen, out_f, in_f, zero_bias=False):
    # torch.nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = torch.zeros(out_f) if zero_bias else (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def _fibonacci_sphere(n):
    # models/tensorBase_rotated_lights.py:49-67
    pts = []
    phi = np.pi * (3.0 - np.sqrt(5.0))
    for i in range(n):
        z = 1 - (i / float(n - 1)) * 2
        r = np.sqrt(1 - z * z)
        t = phi * i
        pts.append([np.cos(t) * r, np.sin(t) * r, z])
    return np.array(pts, dtype=np.float32)


def make_checkpoint(grid=(128, 128, 128), seed=20211202, light_rotation=("000",),
                    aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)), density_n_comp=(16, 16, 16),
                    app_n_comp=(48, 48, 48), app_dim=27, feature_c=128, pe=2, num_sgs=128,
                    step_ratio=0.5, blob_sigma=0.35, blob_gain=20.0):
    """Random-init VM field with a separable Gaussian blob in density component 0."""
    gen = torch.Generator().manual_seed(seed)
    grid = [int(g) for g in grid]
    sd = {}
    for name, comps in (("density", density_n_comp), ("app", app_n_comp)):
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            sd[f"{name}_plane.{i}"] = 0.1 * torch.randn(1, comps[i], grid[m1], grid[m0], generator=gen)
            sd[f"{name}_line.{i}"] = 0.1 * torch.randn(1, comps[i], grid[VEC_MODE[i]], 1, generator=gen)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        g0 = torch.exp(-torch.linspace(-1, 1, grid[m0]) ** 2 / (2 * blob_sigma ** 2))
        g1 = torch.exp(-torch.linspace(-1, 1, grid[m1]) ** 2 / (2 * blob_sigma ** 2))
        gv = torch.exp(-torch.linspace(-1, 1, grid[VEC_MODE[i]]) ** 2 / (2 * blob_sigma ** 2))
        sd[f"density_plane.{i}"][0, 0] = blob_gain * g1[:, None] * g0[None, :]
        sd[f"density_line.{i}"][0, 0, :, 0] = gv
    n_app = sum(app_n_comp)
    L = len(light_rotation)
    sd["basis_mat.weight"], _ = _linear_init(gen, app_dim, n_app)
    sd["light_line.weight"] = torch.randn(L, n_app, generator=gen)
    in_c = 2 * pe * 3 + 2 * pe * app_dim + 3 + app_dim
    for mod, outc in (("renderModule", 3), ("renderModule_normal", 3), ("renderModule_brdf", 4)):
        sd[f"{mod}.mlp.0.weight"], sd[f"{mod}.mlp.0.bias"] = _linear_init(gen, feature_c, in_c)
        sd[f"{mod}.mlp.2.weight"], sd[f"{mod}.mlp.2.bias"] = _linear_init(gen, feature_c, feature_c)
        sd[f"{mod}.mlp.4.weight"], sd[f"{mod}.mlp.4.bias"] = _linear_init(gen, outc, feature_c, True)
    # spherical Gaussians: models/tensorBase_rotated_lights.py:461-476
    sg = torch.randn(num_sgs, 7, generator=gen)
    sg[:, -2:] = sg[:, -3:-2].expand(-1, 2)
    sg[:, 3:4] = 10.0 + torch.abs(sg[:, 3:4] * 20.0)
    lam, mu = torch.abs(sg[:, 3:4]), torch.abs(sg[:, 4:])
    energy = mu * 2.0 * np.pi / lam * (1.0 - torch.exp(-2.0 * lam))
    sg[:, 4:] = torch.abs(sg[:, 4:]) / torch.sum(energy, dim=0, keepdim=True) * 2.0 * np.pi * 0.8
    lobes = torch.from_numpy(_fibonacci_sphere(num_sgs // 2))
    sg[:num_sgs // 2, :3] = lobes
    sg[num_sgs // 2:, :3] = lobes
    sd["lgtSGs"] = sg
    kwargs = {
        "aabb": torch.tensor(aabb, dtype=torch.float32), "gridSize": grid,
        "density_n_comp": list(density_n_comp), "appearance_n_comp": list(app_n_comp),
        "app_dim": app_dim, "density_shift": -10, "alphaMask_thres": 0.001,
        "distance_scale": 25, "rayMarch_weight_thres": 0.0001, "fea2denseAct": "softplus",
        "near_far": [2.0, 6.0], "step_ratio": step_ratio, "shadingMode": "MLP_Fea",
        "pos_pe": pe, "view_pe": pe, "fea_pe": pe, "featureC": feature_c,
        "normals_kind": "derived_plus_predicted", "light_num": L, "light_kind": "sg",
        "numLgtSGs": num_sgs, "light_rotation": [int(r) for r in light_rotation],
    }
    return {"kwargs": kwargs, "state_dict": sd}


def make_rays(h, w, cam_z=4.0, fov=0.6911, narrow=0.45, device="cpu"):
    """Pin-hole camera at (0,0,cam_z) looking at -z; directions L2-normalised
    (as dataLoader/tensoIR_rotation_setting.py:105-106).  Returns [h*w, 6] fp32."""
    focal = 0.5 * w / math.tan(0.5 * fov * narrow)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                          torch.arange(w, dtype=torch.float32), indexing="ij")
    d = torch.stack([(i - w / 2 + 0.5) / focal, -(j - h / 2 + 0.5) / focal,
                     -torch.ones_like(i)], dim=-1).reshape(-1, 3)
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    o = torch.tensor([0.0, 0.0, cam_z]).expand_as(d)
    return torch.cat([o, d], dim=-1).contiguous().to(device)


HDR_NAMES = ("bridge", "city", "fireplace", "forest", "night")       # the maps scripts/relight_importance.py:361 asks for


def make_hdr_maps(names=HDR_NAMES, H=1024, W=2048, seed=71):
    """Seeded HDR environment maps (exp(N(0, 1.5^2)) low-pass filtered + one 100x sun disc each) -- there are no
    *.hdr files offline.  ``Environment_Light("synthetic:h=32,w=64")`` builds them for HDR_NAMES."""
    gen = torch.Generator().manual_seed(seed)
    maps = {}
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for i, name in enumerate(names):
        hdr = torch.exp(torch.randn(max(H // 8, 2), max(W // 8, 2), 3, generator=gen) * 1.5)
        hdr = torch.nn.functional.interpolate(hdr.permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                              align_corners=False)[0].permute(1, 2, 0).contiguous()
        cy, cx, rad = (200 + 100 * i) * H // 1024, 300 * (i + 1) * W // 2048, max(20 * H // 1024, 1)
        hdr[((yy - cy) ** 2 + (xx - cx) ** 2) < rad ** 2] *= 100.0
        maps[name] = hdr
    return maps

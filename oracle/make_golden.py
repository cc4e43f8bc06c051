"""Generate tests/golden/*.npz by running the IMPORTED REFERENCE on seeded inputs.

Run in the build container (needs the read-only reference checkout):
    python oracle/make_golden.py
The fixtures pin oracle/tensoir_oracle.py (tests/test_oracle_golden.py) and, on
the GPU box, the HIP path (tests/test_gpu_*.py).  TEST INFRASTRUCTURE ONLY.
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
from tensoir_amd import synth  # noqa: E402

SEED = 20211202
OUT = os.path.join(ROOT, "tests", "golden")


def build_reference_model(ref, ckpt, envmap_h, envmap_w, alpha_grid=None):
    kw = dict(ckpt["kwargs"])
    kw.pop("light_num", None)
    aabb = kw.pop("aabb")
    grid = kw.pop("gridSize")
    kw["light_rotation"] = [f"{r:03d}" for r in kw["light_rotation"]]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.TensorVMSplit(aabb, grid, "cpu", envmap_h=envmap_h, envmap_w=envmap_w, **kw)
        model.load_state_dict(ckpt["state_dict"])
        if alpha_grid is not None:
            model.updateAlphaMask(tuple(alpha_grid))
    model.eval()
    return model


def npy(x):
    if x is None:
        return np.zeros(0, np.float32)
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def main():
    ref = ref_loader.load()
    RU = ref.RU
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(SEED)
    np.random.seed(SEED)

    # small anisotropic scene: non-cubic grid and aabb catch axis mix-ups
    grid = (20, 24, 28)
    aabb = ((-1.5, -1.4, -1.3), (1.5, 1.4, 1.6))
    ckpt = synth.make_checkpoint(grid=grid, seed=SEED, light_rotation=("000", "120", "240"),
                                 aabb=aabb)
    envh, envw = 4, 8
    model = build_reference_model(ref, ckpt, envh, envw, alpha_grid=(16, 18, 20))
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)

    g = {}
    # ---- scene (checkpoint format) ----
    for k, v in ckpt["state_dict"].items():
        g["sd/" + k] = npy(v)
    g["scene/aabb"] = npy(ckpt["kwargs"]["aabb"])
    g["scene/grid"] = np.array(grid, np.int64)
    g["scene/light_rotation"] = np.array(ckpt["kwargs"]["light_rotation"], np.int64)
    g["scene/envmap_hw"] = np.array([envh, envw], np.int64)
    g["scene/alpha_volume"] = npy(model.alphaMask.alpha_volume[0, 0])
    g["scene/alpha_aabb"] = npy(model.alphaMask.aabb)
    g["scene/nSamples"] = np.array([model.nSamples], np.int64)
    g["scene/stepSize"] = npy(model.stepSize).reshape(1)

    gen = torch.Generator().manual_seed(SEED + 1)
    with torch.no_grad():
        # ---- K2 density / K4 appearance features ----
        xyz = torch.rand(600, 3, generator=gen) * 2 - 1
        xyz[:8] = torch.tensor([[-1, -1, -1], [1, 1, 1], [0, 0, 0], [1, -1, 0.5],
                                [-1, 1, -0.25], [0.999999, 0.3, -0.7], [0.1, -1, 1], [0.5, 0.5, 1]])
        lidx = torch.randint(0, 3, (600, 1), generator=gen).int()
        g["feat/xyz"] = npy(xyz)
        g["feat/light_idx"] = npy(lidx)
        g["feat/density"] = npy(model.compute_densityfeature(xyz))
        g["feat/sigma"] = npy(model.feature2density(model.compute_densityfeature(xyz)))
        g["feat/app"] = npy(model.compute_appfeature(xyz, lidx))
        r, i = model.compute_bothfeature(xyz, lidx)
        g["feat/both_rad"], g["feat/both_int"] = npy(r), npy(i)
        g["feat/intrin"] = npy(model.compute_intrinfeature(xyz))
        # ---- K5 decoders ----
        vd = torch.nn.functional.normalize(torch.randn(600, 3, generator=gen), dim=-1)
        g["mlp/viewdirs"] = npy(vd)
        g["mlp/rgb"] = npy(model.renderModule(xyz, vd, r))
        g["mlp/brdf"] = npy(model.renderModule_brdf(xyz, i))
        g["mlp/normal"] = npy(model.renderModule_normal(xyz, i))
        # ---- occupancy ----
        world = model.aabb[0] + (xyz * 0.5 + 0.5) * (model.aabb[1] - model.aabb[0])
        g["occ/xyz_world"] = npy(world)
        g["occ/alpha"] = npy(model.alphaMask.sample_alpha(world))

    # ---- K6 derived normals (autograd through the custom grid_sample) ----
    xg = (torch.rand(300, 3, generator=gen) * 1.6 - 0.8)
    g["normals/xyz"] = npy(xg)
    g["normals/derived"] = npy(model.compute_derived_normals(xg.clone()))

    with torch.no_grad():
        # ---- rays ----
        rays = synth.make_rays(6, 8)
        rays = torch.cat([rays, torch.tensor([[0.0, 0.0, 4.0, 0.9, 0.0, -0.43589]])])  # a miss
        rays[-1, 3:] = rays[-1, 3:] / rays[-1, 3:].norm()
        B = rays.shape[0]
        light_idx = (torch.arange(B) % 3).view(-1, 1).int()
        g["rays/rays"] = npy(rays)
        g["rays/light_idx"] = npy(light_idx)
        # K1 sample_ray: eval and train (the rand_like draw replayed as explicit jitter)
        pts, z, valid = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=-1)
        g["march/pts"], g["march/z"], g["march/valid"] = npy(pts), npy(z), npy(valid)
        torch.manual_seed(SEED + 2)
        pts_t, z_t, valid_t = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=True, N_samples=40)
        torch.manual_seed(SEED + 2)
        g["march/train_jitter"] = npy(torch.rand(B, 1))
        g["march/train_z"], g["march/train_valid"] = npy(z_t), npy(valid_t)
        # K3 raw2alpha
        sig = torch.rand(5, 33, generator=gen) * 3
        dist = torch.rand(5, 33, generator=gen) * 0.5
        a, w, bg = ref.base.raw2alpha(sig, dist)
        g["r2a/sigma"], g["r2a/dist"] = npy(sig), npy(dist)
        g["r2a/alpha"], g["r2a/weight"], g["r2a/bg"] = npy(a), npy(w), npy(bg)

    # ---- primary forward (K1-K6 + compositing) ----
    names = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map",
             "acc_map", "normals_diff_map", "normals_orientation_loss_map", "acc_mask",
             "albedo_smoothness_loss", "roughness_smoothness_loss"]
    torch.manual_seed(SEED + 3)
    out = model(rays, light_idx, white_bg=True, is_train=False, is_relight=True, N_samples=-1)
    for n, v in zip(names, out):
        g["fwd/" + n] = npy(v)
    torch.manual_seed(SEED + 3)
    out = model(rays, light_idx, white_bg=True, is_train=False, is_relight=False, N_samples=-1)
    g["fwd_norelight/rgb_map"], g["fwd_norelight/depth_map"], g["fwd_norelight/acc_map"] = \
        npy(out[0]), npy(out[1]), npy(out[6])
    torch.manual_seed(SEED + 3)
    out = model(rays, light_idx, white_bg=False, is_train=False, is_relight=True, N_samples=57)
    for n, v in zip(names, out):
        g["fwd_blackbg57/" + n] = npy(v)

    with torch.no_grad():
        # ---- environment light ----
        area, dirs = model.generate_envir_map_dir(envh, envw)
        g["env/area"], g["env/dirs"] = npy(area), npy(dirs)
        g["env/light_rgbs"] = npy(model.get_light_rgbs(dirs, device="cpu"))
        torch.manual_seed(SEED + 4)
        g["env/strat_dirs"] = npy(model.gen_light_incident_dirs(method="stratified_sampling"))
        torch.manual_seed(SEED + 4)
        g["env/strat_u_phi"] = npy(torch.rand(envh, envw))
        g["env/strat_u_theta"] = npy(torch.rand(envh, envw))
        # ---- GGX ----
        M, D = 40, 9
        n_ = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=-1)
        v_ = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=-1)
        l_ = torch.nn.functional.normalize(torch.randn(M, D, 3, generator=gen), dim=-1)
        rg = (torch.rand(M, 1, generator=gen) * 0.9 + 0.09).repeat(1, 3)
        fr = torch.full((M, 3), 0.04)
        g["ggx/normal"], g["ggx/v"], g["ggx/l"], g["ggx/rough"], g["ggx/fresnel"] = \
            npy(n_), npy(v_), npy(l_), npy(rg), npy(fr)
        g["ggx/spec"] = npy(RU.GGX_specular(n_, v_, l_, rg, fr))
        # ---- sRGB ----
        lin = torch.cat([torch.linspace(0, 1, 101), torch.tensor([0.0031308, 0.00313, 0.0032, 1e-7])]).view(-1, 3)
        g["srgb/in"], g["srgb/out"] = npy(lin), npy(RU.linear2srgb_torch(lin))
        # ---- secondary rays (K7) ----
        P = 150
        sp = (torch.rand(P, 3, generator=gen) * 2 - 1) * torch.tensor([1.0, 1.0, 1.0])
        sdir = torch.nn.functional.normalize(torch.randn(P, 3, generator=gen), dim=-1)
        sl = torch.randint(0, 3, (P, 1), generator=gen).int()
        g["sec/pts"], g["sec/dirs"], g["sec/light_idx"] = npy(sp), npy(sdir), npy(sl)
        tv, tn = RU.compute_transmittance(model, sp, sdir, nSample=96, vis_near=0.05, vis_far=1.5)
        g["sec/trans_vis"], g["sec/trans_nerfactor"] = npy(tv), npy(tn)
        rv, rn, ri = RU.compute_radiance(model, sp, sdir, sl, nSample=96, vis_near=0.05, vis_far=1.5)
        g["sec/rad_vis"], g["sec/rad_nerfactor"], g["sec/rad_indirect"] = npy(rv), npy(rn), npy(ri)

    # ---- full boundary: Renderer_TensoIR_train (renderer.py:57-127) ----
    for tag, method in (("fixed", "fixed_envirmap"), ("strat", "stratified_sampling")):
        torch.manual_seed(SEED + 5)
        ret = ref.renderer.Renderer_TensoIR_train(
            rays, None, light_idx, model, N_samples=-1, white_bg=True, is_train=False,
            is_relight=True, sample_method=method, chunk_size=777, device="cpu", args=args)
        for k, v in ret.items():
            g[f"render_{tag}/" + k] = npy(v)
    # RNG replay for 'strat': forward draws randn_like [A,3] (:937) first, then the two
    # rand_like [envh,envw] of gen_light_incident_dirs (:520)
    g["render/second"] = np.array([args.second_nSample, args.second_near, args.second_far], np.float64)

    with torch.no_grad():
        # ---- HDR importance relight loop body (scripts/relight_importance.py:115-171) ----
        Hh, Wh = 16, 32
        hdr = torch.exp(torch.randn(Hh, Wh, 3, generator=gen) * 1.5)
        hdr[3:5, 10:13] += 100.0
        el = RU.Environment_Light.__new__(RU.Environment_Light)
        inten = torch.sum(hdr, dim=2, keepdim=True)
        sin_t = torch.sin(torch.linspace(0.5 / Hh, np.pi - 0.5 / Hh, Hh))
        pdf = inten * sin_t.view(-1, 1, 1)
        pdf = pdf / torch.sum(pdf)
        el.hdr_rgbs = {"syn": hdr}
        el.hdr_pdf_sample = {"syn": pdf}
        el.hdr_pdf_return = {"syn": pdf * Hh * Wh / (2 * np.pi * np.pi * sin_t.view(-1, 1, 1))}
        lat, lng = np.pi / Hh, 2 * np.pi / Wh
        phi, theta = torch.meshgrid([torch.linspace(np.pi / 2 - 0.5 * lat, -np.pi / 2 + 0.5 * lat, Hh),
                                     torch.linspace(np.pi - 0.5 * lng, -np.pi + 0.5 * lng, Wh)], indexing="ij")
        el.hdr_dir = {"syn": torch.stack([torch.cos(theta) * torch.cos(phi), torch.sin(theta) * torch.cos(phi),
                                          torch.sin(phi)], dim=-1).view(Hh, Wh, 3)}
        g["hdr/map"] = npy(hdr)
        g["hdr/pdf_sample"] = npy(el.hdr_pdf_sample["syn"])
        g["hdr/pdf_return"] = npy(el.hdr_pdf_return["syn"])
        g["hdr/dirs"] = npy(el.hdr_dir["syn"])
        torch.manual_seed(SEED + 6)
        out = model(rays, light_idx, white_bg=True, is_train=False, is_relight=True, N_samples=-1)
        rgb_c, depth_c, normal_c, albedo_c, rough_c, fres_c, acc_c = out[:7]
        mask = acc_c > 0.5
        surf = (rays[:, :3] + depth_c.unsqueeze(-1) * rays[:, 3:])[mask]
        Ms, Ns = int(mask.sum()), 64
        torch.manual_seed(SEED + 7)
        ldir, lrgb, lpdf = el.sample_light("syn", Ms, Ns)
        surf2c = RU.safe_l2_normalize(-rays[:, 3:][mask], dim=-1)
        cosine = torch.einsum("ijk,ik->ij", ldir, normal_c[mask])
        cmask = cosine > 1e-6
        vis = torch.zeros(*cmask.shape, 1)
        v, _ = RU.compute_transmittance(model, surf[:, None, :].expand(Ms, Ns, 3)[cmask], ldir[cmask],
                                        nSample=96, vis_near=0.05, vis_far=1.5)
        vis[cmask] = v.unsqueeze(-1)
        spec = RU.brdf_specular(normal_c[mask], surf2c, ldir, rough_c[mask], fres_c[mask])
        brdf = albedo_c[mask].unsqueeze(1).expand(-1, Ns, -1) / np.pi + spec
        contrib = brdf * (vis * lrgb) * cosine[:, :, None] / lpdf
        rel = torch.clamp(torch.mean(contrib, dim=1), 0.0, 1.0)
        rel = RU.linear2srgb_torch(rel)
        g["hdr/surf"], g["hdr/normal"], g["hdr/albedo"] = npy(surf), npy(normal_c[mask]), npy(albedo_c[mask])
        g["hdr/rough"], g["hdr/fresnel"], g["hdr/rays_d"] = npy(rough_c[mask]), npy(fres_c[mask]), npy(rays[:, 3:][mask])
        g["hdr/light_dir"], g["hdr/light_rgb"], g["hdr/light_pdf"] = npy(ldir), npy(lrgb), npy(lpdf)
        g["hdr/relit"] = npy(rel)
        g["hdr/bg"] = npy(el.get_light("syn", rays[:, 3:]))

    path = os.path.join(OUT, "small_scene.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

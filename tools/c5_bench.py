"""BASELINE configs[4] timing on one GPU: relighting an 800x800 view with 2048x1024 HDR environment maps, 512 importance
samples per surface point (scripts/relight_importance.py:99-171): per 4096-ray chunk one primary pass, then per
environment map: device-side importance sampling + cosine mask, visibility march of the unmasked (point, cell) pairs,
BRDF x radiance integration, background lookup.  Reference loop: 5 maps per chunk.  (GPU box.)"""
import contextlib, io, json, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import relight, synth

grid = int(os.environ.get("GRID", 400))          # ficus: N_voxel_final = 400^3 (configs/single_light/ficus.txt)
n_maps = int(os.environ.get("MAPS", 5))
ck = synth.make_checkpoint(grid=(grid,) * 3, seed=20211202)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=8, envmap_w=16)
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    m.updateAlphaMask((128, 128, 128))
H, W = 1024, 2048
gen = torch.Generator().manual_seed(71)
maps = {}
for i in range(n_maps):
    hdr = torch.exp(torch.randn(H // 8, W // 8, 3, generator=gen) * 1.5)
    hdr = torch.nn.functional.interpolate(hdr.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).contiguous()
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    hdr[((yy - 200 - 100 * i) ** 2 + (xx - 300 * (i + 1)) ** 2) < 20 ** 2] *= 100.0
    maps[f"env{i}"] = hdr
env = relight.Environment_Light(hdr_maps=maps, device="cuda")
rays = synth.make_rays(800, 800, narrow=float(os.environ.get("NARROW", 1.0))).cuda()
lidx = torch.zeros(rays.shape[0], 1, dtype=torch.int32, device="cuda")
Ns = 512

@torch.no_grad()
def image():
    n_hit = 0
    for c in torch.split(torch.arange(rays.shape[0], device="cuda"), 4096):
        r, l = rays[c], lidx[c]
        out = m(r, l, N_samples=-1)                              # geometry + material maps (the is_relight=True forward)
        depth, normal, albedo, rough, fres, acc = out[1], out[2], out[3], out[4], out[5], out[6]
        mask = acc > 0.5
        surf = (r[:, :3] + depth.unsqueeze(-1) * r[:, 3:])[mask]
        nrm, alb, rgh, fr, rd = normal[mask], albedo[mask], rough[mask], fres[mask], r[:, 3:][mask]
        n_hit += int(surf.shape[0])
        for name in maps:
            rgb = relight.relight_importance_sampled(m, env, name, surf, nrm, alb, rgh, fr, rd, num_samples=Ns)
            bg = env.get_light(name, r[:, 3:])
            img = torch.where(mask[:, None], torch.zeros_like(bg).index_put_((mask.nonzero()[:, 0],), rgb), bg)
    return n_hit

image(); torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 2
for _ in range(reps):
    n_hit = image()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"config": f"C5: 800x800 view, {grid}^3 field, {n_maps} HDR maps 2048x1024, {Ns} importance samples per surface point, "
                            f"96 visibility samples per (point, sample) pair, 1 GPU",
                  "s_per_view": round(dt, 4), "camera_rays_per_s": round(rays.shape[0] / dt, 1),
                  "surface_points": n_hit, "secondary_rays_per_s": round(n_hit * Ns * n_maps / dt, 1),
                  "relit_images_per_s": round(n_maps / dt, 3)}))

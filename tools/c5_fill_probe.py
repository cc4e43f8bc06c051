"""Slot fill of the gather passes in the visibility march of the C5 workload (importance-sampled, incoherent rays).
Needs the EXP_COUNT_ITERS build: TENSOIR_HIP_LIB=gpurun_scratch/lib_iters.so python tools/c5_fill_probe.py"""
import contextlib, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, relight, synth
grid = int(os.environ.get("GRID", 400))
ck = synth.make_checkpoint(grid=(grid,) * 3, seed=20211202)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=8, envmap_w=16)
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    m.updateAlphaMask((128, 128, 128))
H, W = 1024, 2048
gen = torch.Generator().manual_seed(71)
hdr = torch.exp(torch.randn(H // 8, W // 8, 3, generator=gen) * 1.5)
hdr = torch.nn.functional.interpolate(hdr.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).contiguous()
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
hdr[((yy - 200) ** 2 + (xx - 300) ** 2) < 20 ** 2] *= 100.0
env = relight.Environment_Light(hdr_maps={"env0": hdr}, device="cuda")
rays = synth.make_rays(800, 800, narrow=1.0).cuda()
c = torch.arange(400 * 800, 400 * 800 + 4096, device="cuda")
r = rays[c]; l = torch.zeros(4096, 1, dtype=torch.int32, device="cuda")
with torch.no_grad():
    out = m(r, l, N_samples=-1)
    depth, normal, albedo, rough, fres, acc = out[1], out[2], out[3], out[4], out[5], out[6]
    mask = acc > 0.5
    surf = (r[:, :3] + depth.unsqueeze(-1) * r[:, 3:])[mask]
    args = (surf, normal[mask], albedo[mask], rough[mask], fres[mask], r[:, 3:][mask])
    relight.relight_importance_sampled(m, env, "env0", *args, num_samples=512)
    torch.cuda.synchronize()
    ops.STATS = {}
    t0 = time.perf_counter()
    relight.relight_importance_sampled(m, env, "env0", *args, num_samples=512)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for k, v in ops.STATS.items():
        x = int(v.item()); valid, it = x & 0xffffffff, x >> 32
        print(k, "rays", surf.shape[0] * 512, "valid", valid, "passes", it, "fill", valid / 16 / max(1, it), "ms", 1e3 * dt)

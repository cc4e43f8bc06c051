"""BASELINE configs[3] on one GPU: a full 800x800 image (640 000 rays, 157 chunks of 4096, light index = pixel mod 3)
through dist.render_sharded; per-image time and rays/s (GPU box)."""
import json, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import Renderer_TensoIR_train, synth
from tensoir_amd import dist as tdist
import contextlib, io
ck = synth.make_checkpoint(grid=(300, 300, 300), seed=20211202, light_rotation=("000", "120", "240"))
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=8, envmap_w=16)
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    m.updateAlphaMask((128, 128, 128))
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
out = {}
for name, narrow in (("object fills the frame", 0.45), ("full field of view (corners are background)", 1.0)):
    rays = synth.make_rays(800, 800, narrow=narrow).cuda()
    lidx = (torch.arange(rays.shape[0], device="cuda") % 3).to(torch.int32).view(-1, 1)
    fn = lambda r, l: Renderer_TensoIR_train(r, None, l, m, N_samples=-1, white_bg=True, is_train=False, is_relight=True,
                                             sample_method="fixed_envirmap", device="cuda", args=args)
    if os.environ.get("IMAGE_BENCH_EAGER") != "1":      # default: one captured graph per chunk shape, checks deferred per image
        fn = tdist.GraphedChunkRenderer(m, 4096, args)
    with torch.no_grad():
        img = tdist.render_sharded(fn, rays, lidx, rank=0, world=1, chunk=4096)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            img = tdist.render_sharded(fn, rays, lidx, rank=0, world=1, chunk=4096)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    out[name] = {"launch": "eager" if os.environ.get("IMAGE_BENCH_EAGER") == "1" else "hip-graph replay per chunk", "s_per_image": round(dt, 4), "rays_per_s": round(rays.shape[0] / dt, 1), "n_samples": m.nSamples,
                 "hit_fraction": round(float((img["acc_map"] > 0.5).float().mean()), 3)}
print(json.dumps({"config": "C4: 800x800, 300^3 field, 3 light rotations, 128 dirs x 96 secondary samples, 1 GPU", "results": out}))

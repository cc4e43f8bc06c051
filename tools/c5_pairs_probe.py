"""C5 (tools/c5_bench.py's workload) under the pair-order modes of the visibility march: masked list, compacted list, direction-binned
lists of several bin grids / block sizes (TENSOIR_C5_PAIRS / _BINS / _BLOCK_PAIRS).  One process, same scene; per mode seconds per
relit view and whether the image equals the masked march's bit for bit.  Usage (GPU box): python tools/c5_pairs_probe.py [out.json]"""
import contextlib, io, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import relight, synth

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "c5_pairs_probe.json")
grid = int(os.environ.get("GRID", 400))
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
rays = synth.make_rays(800, 800).cuda()
lidx = torch.zeros(rays.shape[0], 1, dtype=torch.int32, device="cuda")
Ns = 512


@torch.no_grad()
def image(keep):
    env._draws = 0
    imgs = []
    for c in torch.split(torch.arange(rays.shape[0], device="cuda"), 4096):
        r, l = rays[c], lidx[c]
        out = m(r, l, N_samples=-1)
        depth, normal, albedo, rough, fres, acc = out[1], out[2], out[3], out[4], out[5], out[6]
        mask = acc > 0.5
        surf = (r[:, :3] + depth.unsqueeze(-1) * r[:, 3:])[mask]
        nrm, alb, rgh, fr, rd = normal[mask], albedo[mask], rough[mask], fres[mask], r[:, 3:][mask]
        for name in maps:
            rgb = relight.relight_importance_sampled(m, env, name, surf, nrm, alb, rgh, fr, rd, num_samples=Ns)
            if keep:
                imgs.append(rgb)
    return torch.cat(imgs) if keep else None


from tensoir_amd import ops  # noqa: E402

EVENTS = {}


def timed(name):
    fn = getattr(ops, name)

    def wrap(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        EVENTS.setdefault(name, []).append((e0, e1))
        return out
    setattr(ops, name, wrap)


for fn_name in ("env_sample_setup", "env_sample_setup_list", "march_secondary", "relight_importance_cells"):
    timed(fn_name)

MODES = [("mask", None, None), ("compact", None, 512), ("binned", "15x17", 512), ("binned", "8x8", 512), ("binned", "8x8", 4096),
         ("mask", None, None)]
res, base = [], None
for mode, bins, block in MODES:
    os.environ["TENSOIR_C5_PAIRS"] = mode
    for k, v in (("TENSOIR_C5_BINS", bins), ("TENSOIR_C5_BLOCK_PAIRS", block)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    torch.manual_seed(3)
    img = image(True)
    torch.cuda.synchronize()
    if base is None:
        base = img
    t = []
    EVENTS.clear()
    for _ in range(2):
        t0 = time.perf_counter()
        image(False)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    per = {k: round(sum(a.elapsed_time(b) for a, b in v) / 2 / 1e3, 4) for k, v in EVENTS.items()}      # s per view
    res.append({"pairs": mode, "bins": bins, "block_pairs": block, "s_per_view": round(min(t), 4), "s_per_view_all": [round(x, 4) for x in t],
                "gpu_s_per_view": per, "bit_identical_to_masked_march": bool(torch.equal(img, base))})
    print(json.dumps(res[-1]), flush=True)
with open(out_path, "w") as fh:
    json.dump({"config": f"C5: 800x800 view, {grid}^3 field, {n_maps} HDR maps 2048x1024, {Ns} importance samples, 96 visibility samples", "modes": res}, fh, indent=1)

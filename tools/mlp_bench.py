"""Decoder kernel micro-benchmark + accuracy vs an fp64 reference (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
from oracle import tensoir_oracle as O
from tests.helpers import scene_from_checkpoint

ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
# make the decoders "trained-like": larger weights so activations are O(1..10)
with torch.no_grad():
    for mod in (m.renderModule, m.renderModule_brdf, m.renderModule_normal):
        for p in mod.parameters():
            p.mul_(3.0)
sc = O.scene_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, ck["kwargs"])
n = int(os.environ.get("N", 2_000_000))
g = torch.Generator().manual_seed(0)
feat = torch.zeros(n, 32)                       # [A,32] rows as the appearance kernel writes them (16-B aligned: VEC path)
feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
feat = feat.cuda()
aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
sc64 = sc.to(torch.float64)
k = 4096
f64 = feat[:k, :27].cpu().double()
ref = {"rgb": O.render_rgb(sc64, aux[:k].cpu().double(), f64),
       "brdf": O.render_brdf(sc64, aux[:k].cpu().double(), f64),
       "normal": O.render_normal(sc64, aux[:k].cpu().double(), f64)}
mods = {"rgb": m.renderModule, "brdf": m.renderModule_brdf, "normal": m.renderModule_normal}
with torch.no_grad():
    for impl in os.environ.get("IMPLS", "mfma,bf16x3,bf16").split(","):
        for name, mod in mods.items():
            pk = mod.packed()
            out = ops.mlp(pk, feat, aux, None, impl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = ops.mlp(pk, feat, aux, None, impl)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = n * 2 * (150 * 128 + 128 * 128 + 128 * pk.out_dim)
            err = float((out[:k].cpu().double() - ref[name]).abs().max())
            print(f"{impl:7s} {name:6s} n={n} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  max_abs_err_vs_fp64={err:.3e}", flush=True)

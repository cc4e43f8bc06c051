"""Decoder accuracy probe (GPU box): max abs error of each decoder vs fp64 on 4096 rows, per impl."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
from oracle import tensoir_oracle as O
ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
sc64 = O.scene_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, ck["kwargs"]).to(torch.float64)
g = torch.Generator().manual_seed(0)
n = 4096 + 77
feat = torch.zeros(n, 32); feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
f64, a64 = feat[:, :27].double(), aux.double()
ref = {"rgb": O.render_rgb(sc64, a64, f64), "brdf": O.render_brdf(sc64, a64, f64), "normal": O.render_normal(sc64, a64, f64)}
mods = {"rgb": m.renderModule, "brdf": m.renderModule_brdf, "normal": m.renderModule_normal}
with torch.no_grad():
    for impl in os.environ.get("IMPLS", "mfma,bf16x3").split(","):
        for name, mod in mods.items():
            out = ops.mlp(mod.packed(), feat.cuda(), aux.cuda(), None, impl).cpu().double()
            e = (out - ref[name]).abs()
            print(f"{impl:7s} {name:6s} max_abs_err={float(e.max()):.3e} mean={float(e.mean()):.3e} worst_row={int(e.max(-1).values.argmax())}")

"""Decoder kernel timing + accuracy vs fp64 (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
from oracle import tensoir_oracle as O
ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
sc = O.scene_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, ck["kwargs"]).to(torch.float64)
g = torch.Generator().manual_seed(0)
for n in (400_000, 2_000_000):
    feat = torch.zeros(n, 32); feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
    aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    ref = O.render_rgb(sc, aux[:4096].double(), feat[:4096, :27].double())
    feat, aux = feat.cuda(), aux.cuda()
    pk = m.renderModule.packed()
    for impl in ("bf16x3", "mfma", "bf16x3"):
        with torch.no_grad():
            out = ops.mlp(pk, feat, aux, None, impl); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): out = ops.mlp(pk, feat, aux, None, impl)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = n * 2 * (150 * 128 + 128 * 128 + 128 * 3)
        print(f"n={n} {impl:7s}: {ms:7.4f} ms  {fl/ms/1e9:7.1f} TFLOP/s  max|err vs fp64|={float((out[:4096].cpu().double()-ref).abs().max()):.2e}", flush=True)

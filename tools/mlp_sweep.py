"""Sweep the decoder kernel's wave-stagger delay (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
g = torch.Generator().manual_seed(0)
for n in (400_000, 2_000_000):
    base = None
    feat = torch.zeros(n, 32); feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
    feat = feat.cuda()
    aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    pk = m.renderModule.packed()
    base = None
    for var, st in [(v, st) for v in (1, 3, 1, 3) for st in (0,)]:
        os.environ["TIR_MLP_STAGGER"] = str(st)
        os.environ["TIR_MLP_VARIANT"] = str(var)
        with torch.no_grad():
            out = ops.mlp(pk, feat, aux, None, "bf16x3"); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): out = ops.mlp(pk, feat, aux, None, "bf16x3")
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if base is None: base = out.clone()
        same = bool(torch.equal(out, base))
        fl = n * 2 * (150 * 128 + 128 * 128 + 128 * 3)
        print(f"n={n} variant={var} stagger={st:2d}: {ms:7.4f} ms  {fl/ms/1e9:7.1f} TFLOP/s  identical={same}", flush=True)

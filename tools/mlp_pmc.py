"""Single decoder launch set for PMC collection (GPU box): python tools/mlp_pmc.py [impl] [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
impl = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
g = torch.Generator().manual_seed(0)
feat = torch.zeros(n, 32); feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
feat = feat.cuda()
aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
pk = m.renderModule.packed()
with torch.no_grad():
    for _ in range(3):
        out = ops.mlp(pk, feat, aux, None, impl)
torch.cuda.synchronize()

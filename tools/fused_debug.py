import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
ck = synth.make_checkpoint(grid=(64,) * 3, seed=1)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=8, envmap_w=16)
pm = m.renderModule.packed(); fld, fh = m.packed_field(), m.packed_field_half()
gen = torch.Generator().manual_seed(3)
D = 16
dirs = torch.nn.functional.normalize(torch.randn(D, 3, generator=gen), dim=-1).cuda()
for npts in (1, 31, 32, 33, 255, 256, 257, 383, 384, 385, 5003, 100000):
    pts = (torch.rand(npts, 3, generator=gen) * 1.6 - 0.8).cuda()
    lpt = torch.zeros(64, dtype=torch.int32).cuda()
    pair = torch.randint(0, 64 * D, (npts,), generator=gen).int().cuda()
    feat = ops.vm_app_h16(fld, fh, pts, lpt, pair, D)
    two = ops.mlp(pm, feat, dirs, pair, "f16", D) if D * 8 <= npts else ops.mlp(pm, feat, dirs, pair, "mfma", D)
    one = ops.indirect_fused(fld, fh, pm, pts, lpt, pair, D, dirs, D)
    d = (one - two).abs().max(dim=1).values
    bad = (d > 1e-4).nonzero().view(-1)
    print(npts, "max diff", float(d.max()), "rows > 1e-4:", int(bad.numel()), bad[:8].tolist(), (bad % 32)[:8].tolist())

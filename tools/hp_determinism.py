"""Is tir_indirect_fused_hp_fwd bit-reproducible?  (GPU box.)  Random records on the golden scene: the launch twice and once with a
device-side count 7 below n -- every common row must be identical; the worst records against the exact route are listed with their
position in the tile (a packed-fp32 form of the product chains failed this in lanes 48-63 of a wave: DESIGN 8).
Usage: python tools/hp_determinism.py"""
import sys, os, types, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoir_amd
from tensoir_amd import ops
from tests.helpers import golden_checkpoint
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "small_scene.npz"))
eh, ew = [int(x) for x in g["scene/envmap_hw"]]
m = tensoir_amd.model_from_checkpoint(golden_checkpoint(g), "cuda", envmap_h=eh, envmap_w=ew)
gen = torch.Generator().manual_seed(12)
D, npt = 16, 40
dirs = torch.nn.functional.normalize(torch.randn(D, 3, generator=gen), dim=-1).cuda()
lpt = torch.randint(0, m.light_num, (npt,), generator=gen).int().cuda()
fld, pm = m.packed_field(), m.renderModule.packed()
for npts in (37, 255, 5003, 70001):
    pts = (torch.rand(npts, 3, generator=gen) * 1.9 - 0.95).cuda()
    pair = torch.randint(0, npt * D, (npts,), generator=gen).int().cuda()
    exact = ops.mlp(pm, ops.vm_app(fld, pts, lpt, pair, True, False, None, D)[0], dirs, pair, "mfma", D)
    hp = ops.indirect_fused_hp(fld, pm, pts, lpt, pair, D, dirs, D)
    hp2 = ops.indirect_fused_hp(fld, pm, pts, lpt, pair, D, dirs, D)
    d = (hp - exact).abs().amax(1)
    print(npts, "max", float(d.max()), "repeat equal", bool(torch.equal(hp, hp2)), "n>1e-4:", int((d > 1e-4).sum()))
    idx = torch.topk(d, min(5, npts)).indices
    for i in idx.tolist():
        print("   rec", i, "in tile pos", i % 256, "wave", (i % 256) // 32, "lane-rec", i % 32, "err", float(d[i]), "pt", pts[i].tolist())
    if npts > 10:
        n_dev = torch.tensor([npts - 7], dtype=torch.int32, device="cuda")
        part = ops.indirect_fused_hp(fld, pm, pts, lpt, pair, D, dirs, D, n_dev)
        ne = (part[:npts - 7] != hp[:npts - 7]).any(1)
        print("   n_dev run differs on", int(ne.sum()), "records", ne.nonzero().view(-1)[:10].tolist())

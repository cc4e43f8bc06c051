"""Determinism stress of the graph-replayed step (GPU box): N replays, every map of every replay must equal the first
replay's bit for bit (the smoothness losses excepted: their jitter noise advances per pass by design).  Catches a stale
cross-workgroup read in the last-workgroup scan / mean or a mis-armed counter, which would show up as a rare mismatch."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd.graph import GraphedRenderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
gr = GraphedRenderer(model, 4096, N_samples=512, args=args, device="cuda")
gr.rays.copy_(rays); gr.lidx.copy_(lidx)
first = {k: v.clone() for k, v in gr(clone_outputs=False).items() if torch.is_tensor(v)}
keys = [k for k in first if not k.endswith("smoothness_loss")]
bad = 0
smooth = set()
for i in range(n):
    out = gr(clone_outputs=False, defer_check=True)
    for k in keys:
        if not torch.equal(out[k], first[k]):
            bad += 1
            print("MISMATCH replay", i, k, float((out[k] - first[k]).abs().max()), flush=True)
    smooth.add(float(out["albedo_smoothness_loss"]))
    if i % 64 == 63:
        assert gr.validate()
print(f"{n} replays, {bad} mismatching maps, {len(smooth)} distinct smoothness values (fresh jitter per pass)")
sys.exit(1 if bad else 0)

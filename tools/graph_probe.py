"""Where does a graph-replayed step spend its time? (GPU box)"""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import ops
from tensoir_amd.graph import GraphedRenderer
a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
ops.MLP_IMPL = os.environ.get("IMPL", "bf16x3")
gr = GraphedRenderer(model, 4096, N_samples=512, args=args, device="cuda")
gr.rays.copy_(rays); gr.lidx.copy_(lidx)
gr(clone_outputs=False)
def run(n, raw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if raw: gr.graph.replay()
        else: gr(clone_outputs=False, defer_check=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for raw in (True, False, True, False):
    for n in (20, 200):
        h, t = run(n, raw)
        print(f"raw_replay={raw} n={n}: host {h:.3f} ms/step, total {t:.3f} ms/step", flush=True)
print("captures", gr.captures, "valid", gr.validate())

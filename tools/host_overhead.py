"""How much of a step is host-side (GPU box): wall per step vs CPU issue time vs summed kernel time."""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tensoir_amd import Renderer_TensoIR_train, ops

a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096, samples=512, second_samples=96)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda", 0), 0)
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)

def step():
    with torch.no_grad():
        return Renderer_TensoIR_train(rays, None, lidx, model, N_samples=512, white_bg=True, is_train=False,
                                      is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device="cuda", args=args)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/30:.3f} ms/step, wall {1e3*(t2-t0)/30:.3f} ms/step")
# python-only cost: stub out the C calls
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)

import os, sys, types, torch
ROOT='/root/repo'; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import bench
from tensoir_amd import ops, Renderer_TensoIR_train
a = types.SimpleNamespace(rays=4096, samples=512, grid=300, env_h=8, env_w=16, second_samples=96)
dev = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, dev, 0)
model.march_t_stop = 1e-6
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
kw = dict(N_samples=512, white_bg=True, is_train=False, is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=dev, args=args)
with torch.no_grad():
    for _ in range(2): Renderer_TensoIR_train(rays, None, lidx, model, **kw)
    ops.STATS = {}
    Renderer_TensoIR_train(rays, None, lidx, model, **kw)
    torch.cuda.synchronize()
    for k, v in ops.STATS.items():
        x = int(v.item()); print(k, "valid", x & 0xffffffff, "iters", x >> 32, "fill", (x & 0xffffffff) / 16 / max(1, x >> 32))

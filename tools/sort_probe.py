"""Probe: how much faster is the appearance gather of the secondary records when the records are visited in spatial
(coarse-cell) order?  Captures the arguments of the largest tir_vm_app_fwd call of one step and times it as is, and
on inputs permuted by a G^3 cell key (G = 8, 16, 32).  Run with TIR_XCD=0 and TIR_XCD=1 (XCD-chunked dispatch)."""
import os, sys, types, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import Renderer_TensoIR_train, ops

a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096, samples=512, second_samples=96)
dev = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, dev, 0)
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
calls = []
orig = ops.vm_app

def spy(field, xyz, *rest, **kw):
    calls.append((field, xyz, rest, kw))
    return orig(field, xyz, *rest, **kw)

ops.vm_app = spy
with torch.no_grad():
    for _ in range(2):
        calls.clear()
        Renderer_TensoIR_train(rays, None, lidx, model, N_samples=512, white_bg=True, is_train=False, is_relight=True,
                               sample_method="fixed_envirmap", chunk_size=160000, device=dev, args=args)
ops.vm_app = orig
field, xyz, rest, kw = max(calls, key=lambda c: c[1].shape[0])
rest = list(rest)
n_dev = rest[6] if len(rest) > 6 else kw.get("n_dev")
n = int(n_dev.item()) if n_dev is not None else xyz.shape[0]
xyz = xyz[:n].contiguous()
light_idx, idx_map = rest[0], rest[1]
idx_map = idx_map[:n].contiguous()
print("records", n, "xcd", os.environ.get("TIR_XCD", "0"))

def run(x, m):
    args2 = [light_idx, m] + rest[2:6] + [None]
    for _ in range(2):
        orig(field, x, *args2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        orig(field, x, *args2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5

print("as is          %.3f ms" % run(xyz, idx_map))
for G in (8, 16, 32):
    q = ((xyz * 0.5 + 0.5).clamp(0, 0.9999) * G).long()
    key = (q[:, 2] * G + q[:, 1]) * G + q[:, 0]
    perm = torch.argsort(key)
    print("sorted G=%-2d    %.3f ms" % (G, run(xyz[perm].contiguous(), idx_map[perm].contiguous())))

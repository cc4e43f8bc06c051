"""Appearance-gather kernel micro-benchmark: exact fp32 MFMA vs split-bf16 MFMA vs VALU (GPU box)."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import ops
a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
f = model.packed_field()
g = torch.Generator().manual_seed(0)
for n in (400_000, 2_200_000):
    xyz = ((torch.rand(n, 3, generator=g) * 2 - 1) * 0.6).cuda()
    # records of a real pass are ordered along rays: sort by a space-filling-ish key so neighbours share texels
    li = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref = None
    for impl in ("mfma", "bf16x3", "mfma", "bf16x3"):
        for want in ((True, False), (True, True)):
            with torch.no_grad():
                r, i = ops.vm_app(f, xyz, li, None, want[0], want[1], impl); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): r, i = ops.vm_app(f, xyz, li, None, want[0], want[1], impl)
                e1.record(); torch.cuda.synchronize()
            if ref is None: ref = r.clone()
            err = float((r - ref).abs().max()); mag = float(ref.abs().max())
            print(f"n={n} impl={impl:7s} rad+int={want[1]}: {e0.elapsed_time(e1)/5:7.4f} ms  max|diff vs exact|={err:.2e} (max|feat|={mag:.2f})", flush=True)

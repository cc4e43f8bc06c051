"""Occupancy statistics of the secondary march on the bench scene (GPU box), read from an instrumented library variant:
  tools/build_variant.sh tir_march steps -DEXP_COUNT_STEPS   -> wave-steps executed / of which no sample passed the cull
  tools/build_variant.sh tir_march iters -DEXP_COUNT_ITERS   -> valid samples / 16-sample gather passes
Usage: TENSOIR_HIP_LIB=gpurun_scratch/lib_steps.so python tools/march_stats.py steps"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import ops, relight
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
with torch.no_grad():
    out, maps = model(rays, lidx, N_samples=512, _return_maps=True)
    mask = out[9]
    maps, r = maps[mask], rays[mask]
    dirs = model.gen_light_incident_dirs(method="fixed_envirmap").cuda().contiguous()
    surf, active = ops.shade_setup(maps, r, dirs)
    M, D = maps.shape[0], dirs.shape[0]
    pair = torch.arange(M * D, dtype=torch.int32, device="cuda")
    org_map = torch.div(pair, D, rounding_mode="floor").to(torch.int32)
    dir_map = (pair - org_map * D).to(torch.int32)
    z = relight._z_table(96, 0.05, 1.5, "cuda")
    f = model.packed_field()
    ops.STATS = {}
    vis, oma, rec = ops.march_secondary(f, surf, dirs, z, M * D, org_map, dir_map, active.view(-1), 1e-6, True, 8_000_000, False)
    torch.cuda.synchronize()
    st = {k: int(v.item()) for k, v in ops.STATS.items()}
    print(kind, "pairs", M * D, "active", int(active.sum()), "records", int(rec["counter"][0]), flush=True)
    for k, v in st.items():
        print(f"  {k}: low32 = {v & 0xffffffff}  high32 = {v >> 32}", flush=True)

"""Time the secondary march kernel alone on the bench scene (GPU box)."""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import ops, relight
a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
with torch.no_grad():
    out, maps = model(rays, lidx, N_samples=512, _return_maps=True)
    mask = out[9]
    maps, r, li = maps[mask], rays[mask], lidx[mask].view(-1)
    dirs = model.gen_light_incident_dirs(method="fixed_envirmap").cuda().contiguous()
    surf, active = ops.shade_setup(maps, r, dirs)
    M, D = maps.shape[0], dirs.shape[0]
    pair = torch.arange(M * D, dtype=torch.int32, device="cuda")
    org_map = torch.div(pair, D, rounding_mode="floor").to(torch.int32)
    dir_map = (pair - org_map * D).to(torch.int32)
    z = relight._z_table(96, 0.05, 1.5, "cuda")
    f = model.packed_field()
    print("pairs", M * D, "active", int(active.sum()))
    import ctypes
    from tensoir_amd import _lib
    L = _lib.lib()
    exps = [0]
    if hasattr(L, "tir_set_experiment"):
        exps = [0, 1, 2, 3, 16]
    for ex in exps:
      if hasattr(L, "tir_set_experiment"):
        L.tir_set_experiment(ctypes.c_int(ex)); print("== experiment flags", ex)
      for want_rec in (False, True):
        for t_stop in (1e-6,):
            for it in range(3):
                ops.STATS = {} if it == 0 else None
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                vis, oma, rec = ops.march_secondary(f, surf, dirs, z, M * D, org_map, dir_map, active.view(-1), t_stop, want_rec, 8_000_000 if want_rec else 0, False)
                e1.record(); torch.cuda.synchronize()
            g = 0
            print(f"records={want_rec} t_stop={t_stop:g}: {e0.elapsed_time(e1):.3f} ms gathered={g} "
                  f"nrec={int(rec['counter']) if rec else 0} vis_sum={float(vis.sum()):.3f}", flush=True)

"""Time the three decode routes of the secondary-ray records on the bench scene (GPU box): the fp16 fused kernel, the high-precision
fused kernel and the unfused full-precision pair, each alone on the stream (HIP events, median of 20), plus their deviation from the
exact route (fp32 gather + exact fp32 decoder).  Usage: python tools/indirect_bench.py [grid=300]"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd import ops, relight
a = types.SimpleNamespace(grid=int(sys.argv[1]) if len(sys.argv) > 1 else 300, env_h=8, env_w=16, rays=4096)
ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
with torch.no_grad():
    out, maps = model(rays, lidx, N_samples=512, _return_maps=True)
    mask = out[9]
    maps, r, li = maps[mask], rays[mask], lidx[mask].view(-1).int().contiguous()
    dirs = model.gen_light_incident_dirs(method="fixed_envirmap").cuda().contiguous()
    surf, active = ops.shade_setup(maps, r, dirs)
    M, D = maps.shape[0], dirs.shape[0]
    z = relight._z_table(96, 0.05, 1.5, "cuda")
    f = model.packed_field()
    vis, oma, rec = ops.march_secondary(f, surf, dirs, z, M * D, None, None, active.view(-1), 1e-6, True, 8_000_000, False, D)
    n = int(rec["counter"][0])
    xyz, ray = rec["xyz"][:n].contiguous(), rec["ray"][:n].contiguous()
    fh, pm = model.packed_field_half(), model.renderModule.packed()
    print("records", n, "points", M, "dirs", D, flush=True)

    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2], o
    routes = {
        "f16 fused": lambda: ops.indirect_fused(f, fh, pm, xyz, li, ray, D, dirs, D),
        "hp fused": lambda: ops.indirect_fused_hp(f, pm, xyz, li, ray, D, dirs, D),
        "full (2 launches)": lambda: ops.mlp(pm, ops.vm_app(f, xyz, li, ray, True, False, None, D)[0], dirs, ray, None, D),
    }
    exact = ops.mlp(pm, ops.vm_app(f, xyz, li, ray, True, False, None, D)[0], dirs, ray, "mfma", D)
    for name, fn in routes.items():
        ms, o = t(fn)
        d = (o - exact).double()
        print(f"{name:20s} {ms:8.3f} ms   vs exact: max {float(d.abs().max()):.2e} rms {float(d.pow(2).mean().sqrt()):.2e} mean {float(d.mean(0).abs().max()):.2e}", flush=True)

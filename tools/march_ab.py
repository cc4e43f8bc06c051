"""A/B timing of the secondary march on the bench scene (GPU box): every library given is loaded in its own process (TENSOIR_HIP_LIB),
the processes are run interleaved `rounds` times, and each reports the median HIP-event time of the march with records (the batch
workload's launch) and without (visibility only, the C5 launch) plus a checksum of its outputs (bit-identity across variants).
Usage: python tools/march_ab.py [--rounds 3] name=path.so [name=path.so ...]      (no arguments: the shipped library)
       python tools/march_ab.py --child                                             (one measurement, used by the parent)"""
import os, subprocess, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def child():
    import torch
    import bench
    from tensoir_amd import ops, relight
    a = types.SimpleNamespace(grid=300, env_h=8, env_w=16, rays=4096)
    ckpt, model, rays, lidx = bench.build_scene(a, torch.device("cuda"), 0)
    with torch.no_grad():
        out, maps = model(rays, lidx, N_samples=512, _return_maps=True)
        mask = out[9]
        maps, r = maps[mask], rays[mask]
        dirs = model.gen_light_incident_dirs(method="fixed_envirmap").cuda().contiguous()
        M, D = maps.shape[0], dirs.shape[0]
        n_active = torch.zeros((1,), dtype=torch.int32, device="cuda")
        surf, active, pair_ids, vis0, cnt0 = ops.shade_setup_compact(maps, r, dirs, 0.5, n_active)      # the product's pair list
        z = relight._z_table(96, 0.05, 1.5, "cuda")
        f = model.packed_field()

        def t(fn, reps=30):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2], o
        vis0b = vis0.clone()
        kw = dict(n_dirs=D, ray_ids=pair_ids, n_ids_dev=n_active)
        ms_rec, (vis, oma, rec) = t(lambda: ops.march_secondary(f, surf, dirs, z, M * D, None, None, None, 1e-6, True, 8_000_000, False, vis=vis0, rec_cnt=cnt0, **kw))
        ms_vis, (vis2, oma2, _) = t(lambda: ops.march_secondary(f, surf, dirs, z, M * D, None, None, None, 1e-6, False, 0, False, vis=vis0b, **kw))
        n = int(rec["counter"][0])
        ck = (float(vis.double().sum()), n, float(rec["w"][:n].double().sum()), float(rec["xyz"][:n].double().sum()),
              float(vis2.double().sum()))
        print(f"RESULT with_records_ms {ms_rec:.4f} visibility_only_ms {ms_vis:.4f} checksum {ck}", flush=True)


def main():
    if "--child" in sys.argv:
        return child()
    args = sys.argv[1:]
    rounds = 3
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    libs = [x.split("=", 1) for x in args] or [["shipped", os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so")]]
    res = {n: [] for n, _ in libs}
    for rnd in range(rounds):
        for name, path in libs:
            env = dict(os.environ, TENSOIR_HIP_LIB=os.path.abspath(path))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT")), None)
            if line is None:
                print(name, "FAILED", p.stderr[-800:], flush=True)
                continue
            res[name].append(line)
            print(f"round {rnd} {name:12s} {line}", flush=True)
    for name, lines in res.items():
        rec = sorted(float(l.split()[2]) for l in lines); vis = sorted(float(l.split()[4]) for l in lines)
        if rec:
            print(f"{name:12s} with records: median {rec[len(rec) // 2]:.4f} ms   visibility only: median {vis[len(vis) // 2]:.4f} ms   (n = {len(rec)})")


if __name__ == "__main__":
    main()

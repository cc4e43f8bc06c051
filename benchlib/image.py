"""bench.py --workload image (BASELINE configs[3]: one 800 x 800 image sharded over the ranks), the sharp-surface variant of the
batch step, and the simulated-rank scaling estimate."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

from benchlib.common import *  # noqa: F401,F403
from benchlib.common import ROOT


def sharp_scene_line(a, device, args):
    """The same step on a scene whose density rises 3-4x faster across the surface (blob sigma 0.2, gain 2000): the
    number of appearance samples per ray drops from ~56 to what a trained scene keeps (10-30), which moves the kernel mix
    from the decoders toward the gathers.  Informative only -- never the headline value."""
    from tensoir_amd import ops
    from tensoir_amd.graph import GraphedRenderer
    _ck, model, rays, lidx = build_scene(a, device, 0, blob_sigma=0.2, blob_gain=2000.0)
    B = rays.shape[0]
    gr = GraphedRenderer(model, B, N_samples=a.samples, args=args, device=device)
    gr.rays.copy_(rays)
    gr.lidx.copy_(lidx)
    ret = gr(clone_outputs=False)
    parity = None
    if not a.no_cpu_baseline:        # the graph-replay maps of this scene against the oracle on every 32nd ray (default policy)
        from oracle import tensoir_oracle as O          # checker only
        from tests.helpers import scene_from_model
        sc = scene_from_model(_ck, model, a.env_h, a.env_w)
        stride = max(1, B // 128)
        with torch.no_grad():
            ref = O.renderer_train(sc, rays.cpu()[::stride], lidx.cpu()[::stride], n_samples=a.samples, second_n_sample=a.second_samples)
        got = {k: v.clone() for k, v in ret.items() if torch.is_tensor(v)}
        parity = map_parity(got, ref, MAP_KEYS, slice(0, None, stride), rays.cpu()[::stride], disc=(O, sc, lidx.cpu()[::stride], a.second_samples))
        parity["rays_compared"] = int(ref["rgb_map"].shape[0])
        parity["indirect_precision"] = model.indirect_precision()
    for _ in range(5):
        gr(clone_outputs=False, defer_check=True)
    torch.cuda.synchronize()
    n = max(10, a.steps)
    t0 = time.perf_counter()
    for _ in range(n):
        gr(clone_outputs=False, defer_check=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = gr.validate()
    # per-kernel attribution: one eager pass bracketed by events
    from tensoir_amd import Renderer_TensoIR_train
    ops.TIMING = []
    with torch.no_grad():
        for _ in range(3):
            Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False,
                                   is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args, _no_graph=True)
    torch.cuda.synchronize()
    agg = {}
    for name, e0, e1 in ops.TIMING:
        name = ALIAS.get(name, name)
        agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1) / 3
    ops.TIMING = None
    totals = [int(c[0].item()) for c in gr.checks] if gr.checks else []
    return {"value": round(B * n / el, 1), "unit": "rays/s", "ms_per_step": round(1e3 * el / n, 4), "capacity_checks_ok": bool(ok),
            "scene": "blob sigma 0.2, gain 2000 (headline: 0.35 / 20)",
            "surface_points": int((ret["acc_map"] > 0.5).sum()),
            "app_samples_per_ray": round(totals[0] / B, 1) if totals else None,
            "secondary_records": totals[1] if len(totals) > 1 else None, "parity": parity,
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]}}


def bench_image(a, embed=False):
    """BASELINE configs[3]: an 800x800 image (640 000 rays in chunks of 4096, light index = pixel mod 3) rendered
    data-parallel -- every rank its shard of the rays (row tiles or interleaved tiles), ONE all-gather of the 96-B per-ray
    records per image (renderer.py:225-249 is the reference's sequential chunk loop).  A step = one image; strong scaling."""
    import contextlib
    import io
    import torch.distributed as dist
    import tensoir_amd
    from tensoir_amd import _lib, synth
    from tensoir_amd import dist as tdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: tensoir_amd has no CPU path")
    local = local_device(a)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if a.tile < 0:      # auto: interleaved chunk-sized tiles as soon as the image is shared (profiles/r05_predicted_scaling.json:
        a.tile = a.rays if world > 1 else 0          # 8 ranks -> 7.5-7.8x predicted; two chunks 7.5-7.6x, four 7.3x, row tiles 5.6x)
    assert _lib.lib().tir_device_check() == 0
    use_dist = world > 1 or a.force_dist
    ck = synth.make_checkpoint(grid=(a.grid,) * 3, seed=20211202, light_rotation=("000", "120", "240"))
    model = tensoir_amd.model_from_checkpoint(ck, device, envmap_h=a.env_h, envmap_w=a.env_w)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        model.updateAlphaMask((128, 128, 128))
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    side = a.image_side
    rays = synth.make_rays(side, side, narrow=1.0).to(device)
    n = rays.shape[0]
    lidx = (torch.arange(n, device=device) % 3).to(torch.int32).view(-1, 1)
    fn = tdist.GraphedChunkRenderer(model, a.rays, args, device=device, lanes=max(1, a.in_flight))
    with torch.no_grad():          # capture + capacity learning on this rank's own shard, BEFORE any RCCL thread exists:
        mine = tdist.shard_rows(n, rank, world, a.tile).to(device)       # repeat until the captured capacities of every lane
        for _ in range(4):                                                # hold for the heaviest chunk of the shard
            tdist._render_chunks(fn, rays, lidx, mine, a.rays)
            if fn.validate():
                break
    if use_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)               # --force-dist in a bare single process
        dist.init_process_group(a.backend, **({"device_id": device} if a.backend == "nccl" else {}))
    gw = world if use_dist else 1

    def one():
        with torch.no_grad():
            return tdist.render_sharded_timed(fn, rays, lidx, rank=rank, world=gw, chunk=a.rays, tile=a.tile)
    for _ in range(3 + a.warmup):      # clock settle + warm-up: a fixed count (every image ends in a collective)
        one()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loc, exch = [], []
    for _ in range(a.steps):
        img, tl, te = one()
        loc.append(tl)
        exch.append(te)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [sum(loc) / len(loc)]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        pr = torch.zeros((gw,), dtype=torch.float64, device=device)
        pr[rank] = per_rank[0]
        dist.all_reduce(pr)
        per_rank = pr.tolist()
    sim = None
    if a.simulate_ranks >= 2 and world == 1:
        def render_shard(mine):
            mine = mine.to(device)
            with torch.no_grad():
                for _ in range(4):
                    parts = tdist._render_chunks(fn, rays, lidx, mine, a.rays)
                    if fn.validate():
                        break
                return torch.cat(parts, dim=0) if parts else None
        sim = simulate_ranks(render_shard, n, a.rays, elapsed / a.steps, tdist.RECORD * 4, a.simulate_ranks,
                             local_exchange_s=sum(exch) / len(exch))
    roofline = parity = cpu = kernels = None
    if rank == 0 and not a.no_cpu_baseline:
        # dominant kernel, in-run parity and CPU baseline on ONE chunk of the image (the middle one: rays cross the object)
        from oracle import tensoir_oracle as O          # checker / CPU baseline only
        from tests.helpers import scene_from_model
        from tensoir_amd import Renderer_TensoIR_train
        c0 = (n // a.rays // 2) * a.rays
        rc, lc = rays[c0:c0 + a.rays].contiguous(), lidx[c0:c0 + a.rays].contiguous()

        def run():
            with torch.no_grad():
                return Renderer_TensoIR_train(rc, None, lc, model, N_samples=-1, white_bg=True, is_train=False, is_relight=True,
                                              sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args, _no_graph=True)
        ret_c = run()
        Mc, Dn = int((ret_c["acc_map"] > 0.5).sum()), a.env_h * a.env_w
        rows, gpu_ms, ev_over = attribute_kernels(run, 2, a.rays * 40 + a.rays * model.nSamples * 4, Mc * Dn * 40, device)
        roofline = dominant_roofline(rows)
        kernels = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:6]]
        sc = scene_from_model(ck, model, a.env_h, a.env_w)
        stride = max(1, a.rays // 128)
        r_cpu, l_cpu = rc.cpu()[::stride], lc.cpu()[::stride]
        ref, med, ts = timed_cpu(lambda: O.renderer_train(sc, r_cpu, l_cpu, n_samples=-1, second_n_sample=a.second_samples), 1, 3)
        parity = map_parity(ret_c, ref, MAP_KEYS, slice(0, None, stride), r_cpu, disc=(O, sc, l_cpu, a.second_samples))
        parity["rays_compared"] = int(r_cpu.shape[0])
        cpu = {"value": round(r_cpu.shape[0] / med, 2), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"every {stride}th ray of the image's middle chunk ({r_cpu.shape[0]} rays x {model.nSamples} samples, "
                         f"{Dn} dirs x {a.second_samples}), 1 warm-up + {len(ts)} timed calls, median; host nproc={os.cpu_count()}"}
    if rank == 0:
        hit = float((img["acc_map"] > 0.5).float().mean())
        line = {
            "metric": "full-image primary+secondary rays/sec, one 800x800 image sharded over the GPUs", "value": round(n * a.steps / elapsed, 1),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 io; decoders split-bf16 x3 (hi/lo operands, 3 MFMA products), fp32 accumulate", "data": "synthetic",
            "config": {"workload": f"C4: {side}x{side} image = {n} rays in chunks of {a.rays}, VM grid {a.grid}^3, 3 light rotations "
                                   f"(light index = pixel mod 3), N_samples=-1 ({model.nSamples} per ray), secondary {a.env_h * a.env_w} dirs x "
                                   f"{a.second_samples}; full field of view ({hit:.2f} of the rays hit the object)",
                       "sharding": ("contiguous row tiles" if a.tile <= 0 else f"interleaved tiles of {a.tile} rays") +
                                   f", one all_gather_into_tensor of {tdist.RECORD * 4} B/ray records per image",
                       "launch": "hip-graph replay per chunk, one capacity check per image", "in_flight": max(1, a.in_flight)},
            "world_size": gw, "device_count": torch.cuda.device_count(), "backend": a.backend if use_dist else None,
            "per_rank_render_ms": [round(1e3 * x, 3) for x in per_rank],
            "load_imbalance": round(max(per_rank) / max(min(per_rank), 1e-9), 3),
            "exchange_ms": round(1e3 * sum(exch) / len(exch), 3),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels_middle_chunk": kernels,
            "roofline_note": "dominant kernel of the image's middle chunk, one eager pass bracketed by events (calibrated)",
        }
        if sim is not None:
            line["simulated_ranks"] = sim
        if embed:
            return line
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["ok"]:
            raise SystemExit(f"[bench] PARITY FAILURE vs the oracle (image workload): {parity}")
    if use_dist:
        dist.destroy_process_group()


def simulate_ranks(render_shard, n_rays, chunk, t1_s, record_bytes, max_world, passes=3, tiles=None, local_exchange_s=0.0):
    """The multi-GPU row on ONE GPU: for W = 2, 4, ... <= max_world and every sharding (contiguous row
    tiles; interleaved tiles of 1, 2, 4 chunks) render each rank's shard of the W-rank job ALONE on this GPU -- what rank r would
    do on its own device, the field being replicated -- and time it (device drained around each shard; best of `passes` after
    two untimed passes that let the captured capacities settle).  Predicted time of the W-rank job = max_r t_r + exchange, where
    exchange = the measured local reassembly of the gathered records (`local_exchange_s`, the world = 1 figure) + the wire time
    of ONE all_gather_into_tensor over xGMI modelled at 60 % of the 153 GB/s per-link peak, every rank receiving (W - 1) shards
    over W - 1 links in parallel (MI355X_MICROARCH.md: fully connected, 7 links per GPU).  Predicted speed-up = t(1) / that.
    RCCL itself has still not run with N > 1: this bounds the load-balance part of the scaling curve, not the collective."""
    from tensoir_amd import dist as tdist
    if tiles is None:
        tiles = [0, chunk, 2 * chunk, 4 * chunk]
    worlds = [w for w in (2, 4, 8, 16) if w <= max_world]
    out = {"method": simulate_ranks.__doc__.split("\n")[0].strip(), "t1_ms": round(1e3 * t1_s, 3), "record_bytes_per_ray": record_bytes,
           "link_GBps_assumed": round(0.6 * 153.0, 1), "local_reassembly_ms": round(1e3 * local_exchange_s, 3), "configs": []}
    for w in worlds:
        for tile in tiles:
            per = []
            for r in range(w):
                mine = tdist.shard_rows(n_rays, r, w, tile)
                drain = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
                render_shard(mine)                       # untimed, twice: capacities / caches / re-captures of this shard's chunking
                render_shard(mine)
                best = None
                for _ in range(passes):
                    drain()
                    t0 = time.perf_counter()
                    render_shard(mine)
                    drain()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                per.append(best)
            wire = (w - 1) / w * n_rays * record_bytes / (0.6 * 153e9 * (w - 1))
            t_w = max(per) + local_exchange_s + wire
            out["configs"].append({"world": w, "sharding": "row tiles" if tile <= 0 else f"interleaved tiles of {tile // chunk} chunk(s)",
                                   "tile": tile, "per_shard_ms": [round(1e3 * x, 3) for x in per], "max_ms": round(1e3 * max(per), 3),
                                   "mean_ms": round(1e3 * sum(per) / w, 3), "imbalance_max_over_mean": round(max(per) / (sum(per) / w), 3),
                                   "sum_over_t1": round(sum(per) / t1_s, 3), "exchange_model_ms": round(1e3 * (local_exchange_s + wire), 3),
                                   "predicted_ms": round(1e3 * t_w, 3), "predicted_speedup": round(t1_s / t_w, 2)})
    best = {}
    for c in out["configs"]:
        if c["world"] not in best or c["predicted_speedup"] > best[c["world"]]["predicted_speedup"]:
            best[c["world"]] = c
    out["best_per_world"] = {str(w): {"sharding": c["sharding"], "predicted_speedup": c["predicted_speedup"], "imbalance_max_over_mean": c["imbalance_max_over_mean"]}
                             for w, c in best.items()}
    return out

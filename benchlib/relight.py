"""bench.py --workload relight (BASELINE configs[4]: HDR relighting, 400^3 field, 5 maps x 512 importance samples)."""
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


from benchlib.image import simulate_ranks


def synthetic_hdr_maps(n_maps, H=1024, W=2048):
    """Seeded 2048x1024 HDR environment maps (tensoir_amd.synth.make_hdr_maps)."""
    from tensoir_amd import synth
    return synth.make_hdr_maps([f"env{i}" for i in range(n_maps)], H, W)


def bench_relight(a, embed=False):
    """BASELINE configs[4] (ficus relighting_test): one 800x800 view of the 400^3 field relit under `--maps` 2048x1024 HDR
    environment maps with 512 importance samples per surface point -- the loop body of scripts/relight_importance.py:93-185.
    Per 4096-ray chunk one primary pass, then per map: importance sampling + cosine mask on the device, visibility march of
    the unmasked (point, cell) pairs (96 samples), BRDF x radiance x cosine / pdf, sRGB, background lookup.  Chunks are
    sharded over the ranks (interleaved tiles), ONE all-gather of the relit colours per view.  A step = one view."""
    import contextlib
    import io
    import torch.distributed as dist
    import tensoir_amd
    from tensoir_amd import _lib, ops, relight, synth
    from tensoir_amd import dist as tdist
    pair_order = ops.c5_pair_order()
    world, rank = (int(os.environ.get(k, "0" if k != "WORLD_SIZE" else "1")) for k in ("WORLD_SIZE", "RANK"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: tensoir_amd has no CPU path")
    local = local_device(a)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    assert _lib.lib().tir_device_check() == 0
    grid = a.grid if a.grid != 300 else 400                 # ficus: N_voxel_final = 400^3 (configs/relighting_test/ficus.txt)
    ck = synth.make_checkpoint(grid=(grid,) * 3, seed=20211202)
    model = tensoir_amd.model_from_checkpoint(ck, device, envmap_h=a.env_h, envmap_w=a.env_w)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        model.updateAlphaMask((128, 128, 128))
    maps = synthetic_hdr_maps(a.maps)
    env = relight.Environment_Light(hdr_maps=maps, device=device)
    side, Ns = a.image_side, 512
    rays = synth.make_rays(side, side, narrow=1.0).to(device)
    n = rays.shape[0]
    lidx = torch.zeros(n, 1, dtype=torch.int32, device=device)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        dist.init_process_group(a.backend, **({"device_id": device} if a.backend == "nccl" else {}))
    gw = world if use_dist else 1
    tile = a.rays if (a.tile < 0 and gw > 1) else max(a.tile, 0)
    mine = tdist.shard_rows(n, rank, gw, tile).to(device)

    @torch.no_grad()
    def chunk_pass_host(c, names, counts=None):
        """One chunk the way the reference script drives it (scripts/relight_importance.py:99-113, :166-171): boolean-mask
        indexing of the hit rows on the host side of the call (a synchronisation + ~12 indexing launches per chunk), per
        environment map the relit colours, get_light + index_put_ for the background.  --c5-host-masking times this."""
        r, l = rays[c], lidx[c]
        out = model(r, l, N_samples=-1)
        depth, normal, albedo, rough, fres, acc = out[1:7]
        mask = acc > 0.5
        surf = (r[:, :3] + depth.unsqueeze(-1) * r[:, 3:])[mask]
        nrm, alb, rgh, fr, rd = normal[mask], albedo[mask], rough[mask], fres[mask], r[:, 3:][mask]
        if counts is not None:
            counts[0] += int(surf.shape[0])
        cols = []
        rows_hit = mask.nonzero()[:, 0]
        for name in names:
            rgb = relight.relight_importance_sampled(model, env, name, surf, nrm, alb, rgh, fr, rd, num_samples=Ns)
            img = env.get_light(name, r[:, 3:]).index_put_((rows_hit,), rgb)      # (scripts/relight_importance.py:166-171
            cols.append(img)                                                          #  tone-maps the background too: host side)
        return (torch.cat(cols, dim=1) if cols else None), (surf, nrm, alb, rgh, fr, rd, r[mask], l[mask])

    @torch.no_grad()
    def chunk_pass(c, names, counts=None):
        """One chunk through the product's chunk call (relight.relight_chunk): primary maps, device-side compaction of the hit
        rows, per environment map the relit colours, background composed in -- no host round trip between the launches.
        counts[1] collects the device-side hit counters (summed once per view)."""
        if a.c5_host_masking:
            return chunk_pass_host(c, names, counts)
        r, l = rays[c], lidx[c]
        out, _prim, cc = relight.relight_chunk(model, env, names, r, l, num_samples=Ns)
        if counts is not None:
            counts[1].append(cc["n_hit"])
        return (out if names else None), cc

    def view(counts=None):
        if counts is not None:
            counts[1] = []
        parts = [chunk_pass(c, list(maps), counts)[0] for c in torch.split(mine, a.rays) if c.numel()]
        if counts is not None and counts[1]:
            counts[0] += int(torch.cat(counts[1]).sum().item())       # one read-back per view, after every chunk is queued
        local_rec = torch.cat(parts, dim=0) if parts else torch.zeros((0, 3 * len(maps)), device=device)
        return tdist.gather_records(local_rec, n, rank, gw, tile)

    for _ in range(1 + a.warmup):
        view()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = [0, []]
    for _ in range(a.steps):
        img = view(counts)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed / a.steps]
    if use_dist:
        pr = torch.zeros((gw,), dtype=torch.float64, device=device)
        pr[rank] = elapsed / a.steps
        dist.all_reduce(pr)
        per_rank, elapsed = pr.tolist(), float(pr.max().item()) * a.steps
        cnt = torch.tensor(counts[:1], dtype=torch.float64, device=device)
        dist.all_reduce(cnt)
        counts = [int(cnt.item()), []]
    sim = None
    if a.simulate_ranks >= 2 and world == 1:
        def render_shard(mine_r):
            mine_r = mine_r.to(device)
            parts = [chunk_pass(c, list(maps))[0] for c in torch.split(mine_r, a.rays) if c.numel()]
            return torch.cat(parts, dim=0) if parts else None
        sim = simulate_ranks(render_shard, n, a.rays, elapsed / a.steps, 12 * len(maps), a.simulate_ranks, passes=2, tiles=[0, a.rays])
    roofline = parity = cpu = kernels = None
    if rank == 0 and not a.no_cpu_baseline:
        from oracle import tensoir_oracle as O          # checker / CPU baseline only
        from tests.helpers import parity_metrics, scene_from_model
        c0 = (n // a.rays // 2) * a.rays
        c = torch.arange(c0, c0 + a.rays, device=device)
        _, (surf, nrm, alb, rgh, fr, rd, r_hit, l_hit) = chunk_pass_host(c, [])
        M = int(surf.shape[0])
        rows, gpu_ms, ev_over = attribute_kernels(lambda: chunk_pass(c, list(maps)), 1,
                                                  a.rays * 40 + a.rays * model.nSamples * 4, 0, device)
        for r in rows:                                  # the visibility march: launches per step = maps; io = pair bookkeeping
            if r["kernel"] == "tir_march_secondary_fwd":
                r["note"] = f"{len(maps)} launches (one per environment map), {M} surface points x {Ns} importance samples each"
        roofline = dominant_roofline(rows)
        kernels = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:6]]
        # parity: the fused device path vs the oracle's loop body, fed the cells the device drew, map 0
        sc = scene_from_model(ck, model, a.env_h, a.env_w)
        name = next(iter(maps))
        with torch.no_grad():
            draws = env._draws
            cell, _active = env.sample_cells(name, nrm.contiguous(), Ns)
            env._draws = draws                           # the same draw again inside relight_importance_sampled
            got = relight.relight_importance_sampled(model, env, name, surf, nrm, alb, rgh, fr, rd, num_samples=Ns)
            ldir = env.hdr_dir[name].view(-1, 3)[cell.long()]
            lrgb = env.hdr_rgbs[name].view(-1, 3)[cell.long()]
            lpdf = env.hdr_pdf_return[name].view(-1)[cell.long()].unsqueeze(-1)
        sel = slice(0, M, max(1, M // 96))
        cc = lambda t: t[sel].cpu()
        pts = int(cc(surf).shape[0])
        ref, med, ts = timed_cpu(lambda: O.relight_importance(sc, cc(surf), cc(nrm), cc(alb), cc(rgh), cc(fr), cc(rd), cc(ldir), cc(lrgb),
                                                              cc(lpdf), n_sample=96, near=0.05, far=1.5), 1, 3)
        m = parity_metrics(cc(got), ref)
        parity = {"ok": m["max_rel_floor1"] < 1e-4, "tolerance": 1e-4, "relit_rgb": {k: float(f"{v:.3e}") for k, v in m.items()},
                  "surface_points_compared": pts, "note": "every k-th surface point of the view's middle chunk, environment map 0, the "
                  "device-drawn cells fed to the oracle's restatement of scripts/relight_importance.py:119-170"}
        vis_rays = pts * Ns
        # the same unit as `value` (camera rays/s of a whole view): the oracle's primary pass on the rays of those surface points
        # + its relight loop body once per environment map; the view's background rays (1 - hit fraction of the rays) are
        # counted as free for the CPU (their primary pass is a bounding-box miss) -- which can only flatter the CPU figure
        _, med_p, _ = timed_cpu(lambda: O.forward_primary(sc, cc(r_hit), cc(l_hit).to(torch.int32), -1, True, True, None, None, "aten"), 0, 1)
        hit_frac = counts[0] / max(1, a.steps) / n
        cpu = {"value": round((pts / max(hit_frac, 1e-9)) / (med_p + len(maps) * med), 2), "unit": "rays/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"{pts} surface points of the middle chunk: primary pass of their camera rays ({med_p:.2f} s) + the relight loop "
               f"body ({Ns} samples x 96 visibility steps, {med:.2f} s per map, 1 warm-up + {len(ts)} timed calls, median) x {len(maps)} maps; scaled "
               f"to camera rays by the view's hit fraction {hit_frac:.3f} (background rays free); host nproc={os.cpu_count()}",
               "pairs_per_s": round(vis_rays / med, 1), "gpu_pairs_per_s": round(counts[0] * Ns * len(maps) / elapsed, 1)}
    if rank == 0:
        line = {
            "metric": "relit camera rays/sec: one 800x800 view under 2048x1024 HDR maps, 512 importance samples per surface point",
            "value": round(n * a.steps / elapsed, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 io; primary-pass decoders split-bf16 x3, fp32 accumulate; relight integration f32", "data": "synthetic",
            "config": {"workload": f"C5: {side}x{side} view = {n} rays in chunks of {a.rays}, VM grid {grid}^3, {len(maps)} HDR maps "
                                   f"2048x1024, {Ns} importance samples per surface point, 96 visibility samples per pair",
                       "sharding": ("contiguous row tiles" if tile <= 0 else f"interleaved tiles of {tile} rays") +
                                   f", one all_gather_into_tensor of {12 * len(maps)} B/ray relit colours per view",
                       "launch": "eager per chunk (primary pass + per-map relight kernels); " + ("host-side masking per chunk as the reference script "
                                 "does (--c5-host-masking)" if a.c5_host_masking else "relight.relight_chunk: hit rows compacted on the device, no host round "
                                 "trip per chunk (one hit-count read-back per view)"),
                       "visibility_pairs": "{} (bins {}x{}, blocks of {} pairs): only the pairs that pass the cosine mask are marched "
                                           "(scripts/relight_importance.py:127-131); TENSOIR_C5_PAIRS".format(
                                               pair_order[0], pair_order[1][0], pair_order[1][1], pair_order[2])},
            "surface_points_per_view": counts[0] // max(1, a.steps),
            "visibility_pairs_per_s": round(counts[0] * Ns * len(maps) / elapsed, 1),
            "relit_images_per_s": round(len(maps) * a.steps / elapsed, 3),
            "world_size": gw, "device_count": torch.cuda.device_count(), "backend": a.backend if use_dist else None,
            "per_rank_ms_per_step": [round(1e3 * x, 3) for x in per_rank],
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels_middle_chunk": kernels,
        }
        if sim is not None:
            line["simulated_ranks"] = sim
        if embed:
            return line
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["ok"]:
            raise SystemExit(f"[bench] PARITY FAILURE vs the oracle (relight workload): {parity}")
    if use_dist:
        dist.destroy_process_group()

"""Shared by the bench workloads: hardware constants, the synthetic scene, per-kernel attribution (HIP events on the launch stream),
roofline objects, PMC files, CPU-baseline helpers, launch plumbing.  (Split out of bench.py in round 6; `import bench` still exposes
every name.)"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate over the 8 XCDs = 256 CUs x 64 B/clk: also the
                               # rate at which the vector L1s can be filled, the bound of the ray-coherent gathers
GATHER_BENCH_TAPS = 157e9      # tools/gather_bench.hip (profiles/r01_v5_gather_bench.txt): 192-B taps/s when
                               # consecutive samples share cells -- the measured ceiling of this access pattern
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0     # v_mfma_f32_32x32x16_bf16 / _f16 dense peak
# VALU issue ceiling: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 4 cycles at the 2.4 GHz peak shader clock
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 4.0
# the density gather's arithmetic floor: 3 planes x 16 channels x 7 FMAs per valid sample (4 bilinear + 2 line + 1 product-sum,
# DESIGN 4.1) = 336 lane operations = 5.25 wave64 instructions per sample (84 per 16-sample gather pass)
VALU_FLOOR_PER_DENSITY_SAMPLE = 3 * 16 * 7 / 64.0
# algorithmic bytes per unit of work ("gather-bytes model")
B_DENSITY_SAMPLE = 1184        # occupancy 8x4 + planes 3x4x16x4 + lines 3x2x16x4
B_APP_GATHER = 3456            # planes 3x4x48x4 + lines 3x2x48x4
B_APP_GATHER_H16 = 1728        # the same 18 taps from the fp16 shadow (indirect-light precision policy)
SETTLE_STEPS = 300             # untimed clock-settle steps before the --warmup steps (stated in the JSON as `settle_steps`)


def pose_batches(rays, n, rank=0):
    """`n` distinct ray batches from the base camera batch: the camera orbits the object (azimuth about y, elevation about x;
    the pose sequence of every rank starts at another angle), same pin-hole, same distance.  The synthetic blob is isotropic
    (SURVEY 8d), so every pose still sees the whole object; what changes from batch to batch is where the rays walk through
    the field's planes / lines, the occupancy box and the secondary rays' directions relative to the fixed light grid."""
    import math
    out = []
    for k in range(n):
        az = 2.0 * math.pi * (k + 0.37 * rank) / max(n, 1)
        el = 0.35 * math.sin(1.7 * (k + rank))
        if k == 0 and rank == 0:
            out.append(rays.clone())                   # batch 0 of rank 0 = the canonical SURVEY 8d camera
            continue
        ca, sa, ce, se = math.cos(az), math.sin(az), math.cos(el), math.sin(el)
        Ry = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])
        Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, ce, -se], [0.0, se, ce]])
        R = (Ry @ Rx).to(rays.dtype)
        d = rays[:, 3:] @ R.T
        out.append(torch.cat([rays[:, :3] @ R.T, d / d.norm(dim=-1, keepdim=True)], dim=-1).contiguous())
    return out


def build_scene(a, device, rank, **blob):
    import tensoir_amd
    from tensoir_amd import synth
    ckpt = synth.make_checkpoint(grid=(a.grid,) * 3, seed=20211202, **blob)
    model = tensoir_amd.model_from_checkpoint(ckpt, device, envmap_h=a.env_h, envmap_w=a.env_w)
    with torch.no_grad():
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            model.updateAlphaMask((128, 128, 128))
    side = int(round(a.rays ** 0.5))
    rays = synth.make_rays(side, a.rays // side)
    rays = rays.to(device).contiguous()
    lidx = torch.zeros(rays.shape[0], 1, dtype=torch.int32, device=device)
    return ckpt, model, rays, lidx


# entry points that launch the same kernel as another one (same roofline model, same PMC key)
ALIAS = {"tir_march_secondary_ids_fwd": "tir_march_secondary_fwd", "tir_shade_integrate_records": "tir_shade_integrate",
         # one launch for the primary stage's four decoders: the same device code (mlp_bf16_body) as the single-decoder launch
         "tir_mlp_fwd_multi_bf16x3": "tir_mlp_fwd_bf16x3",
         # the aux-table variants of both (view-direction columns folded into a per-direction accumulator start, 9 k-blocks):
         # same decoder, same useful FLOPs per row
         "tir_mlp_fwd_auxtab_bf16x3": "tir_mlp_fwd_bf16x3", "tir_mlp_fwd_multi_auxtab_bf16x3": "tir_mlp_fwd_bf16x3",
         # the primary stage's two appearance gathers in one launch: app_mfma_body twice, the grid split between them
         "tir_vm_app_primary_fwd": "tir_vm_app_fwd", "tir_vm_app_jitter_fwd": "tir_vm_app_fwd"}
# rocprofv3 kernel names behind each row (the trace under profiles/ lists these)
ROCPROF_KERNELS = {
    "tir_mlp_fwd_bf16x3": ["k_mlp_bf16_multi<3> (the four primary-stage decoders in one launch)",
                           "k_mlp_bf16_auxt<true, false> (one decoder with the aux table; the secondary-ray records when the indirect "
                           "precision policy is `full`)"],
    "tir_mlp_fwd_auxtab_f16": ["k_mlp_f16_auxt<true> (radiance decoder of the secondary-ray records, single-product fp16)"],
    "tir_vm_app_fwd_h16": ["k_vm_app_h16 (radiance features of the secondary-ray records from the fp16 shadow planes)"],
    "tir_indirect_fused_fwd": ["k_indirect_fused (secondary-ray records: fp16-shadow gather + basis contraction + fp16 radiance decoder in one pass)"],
    "tir_vm_app_fwd": ["k_vm_app_primary<12> (primary stage: records + jittered records)", "k_vm_app_mfma<12, ...> (fp32 gather)"],
    "tir_march_secondary_fwd": ["k_march_secondary_lds<4, 3, 512, true> (false: the visibility-only launches of a C5 view)"],
    "tir_march_primary_fwd": ["k_march_primary"],
}


def event_bracket_overhead_ms(device, n=300):
    """What a (record, one-workgroup launch, record) bracket of ops._call reads when the kernel between the events is
    (next to) empty: the events' own cost on the stream.  Subtracted from every bracketed launch below, so that the sum of
    the per-kernel durations does not exceed the one-stream step and agrees with a rocprofv3 kernel trace (round 2: the
    raw brackets summed to 8 % more than the step).  The probe kernel's own ~2 us ride along, i.e. the corrected durations
    are low by at most that."""
    from tensoir_amd import ops
    probe = torch.zeros(64, dtype=torch.int32, device=device)
    for _ in range(20):
        ops.exclusive_scan(probe)
    old, ops.TIMING = ops.TIMING, []
    for _ in range(n):
        ops.exclusive_scan(probe)
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for _, e0, e1 in ops.TIMING)
    ops.TIMING = old
    return ts[len(ts) // 2]


def kernel_table(timing, stats, steps, shapes, overhead_ms=0.0):
    """Aggregate (name, e0, e1) event pairs into per-kernel totals and roofline figures.  Every roofline row carries an
    INDEPENDENT ceiling (a hardware rate from MI355X_MICROARCH.md times the algorithm's own work per unit), so frac <= 1."""
    agg = {}
    for name, e0, e1 in timing:
        name = ALIAS.get(name, name)
        ms = max(e0.elapsed_time(e1) - overhead_ms, 1e-4)
        k = agg.setdefault(name, {"ms": 0.0, "launches": 0})
        k["ms"] += ms
        k["launches"] += 1
    rows = []
    for name, k in agg.items():
        avg_ms = k["ms"] / k["launches"]
        sec = avg_ms * 1e-3
        row = {"kernel": name, "launches_per_step": k["launches"] / steps, "avg_ms": avg_ms,
               "ms_per_step": k["ms"] / steps}
        units = shapes.get(name)
        if name in ("tir_march_primary_fwd", "tir_march_secondary_fwd") and stats and name in stats:
            gathered = int(stats[name].item()) / (k["launches"] / steps)      # counters come from ONE step
            extra = units["io_bytes"] if units else 0
            by = gathered * B_DENSITY_SAMPLE + extra
            # The 70 MB field is cache resident and the march is bound by VALU issue (PMC: SQ_ACTIVE_INST_VALU), so the ceiling
            # is the VALU issue rate over the FMAs the algorithm needs per valid sample -- not a memory level.
            peak = VALU_PEAK_WAVE_INSTR / VALU_FLOOR_PER_DENSITY_SAMPLE / 1e9
            row.update(bound="valu", units=gathered, unit="valid density samples/launch", gather_bytes=by,
                       achieved=gathered / sec / 1e9, peak=round(peak, 2), runit="G valid density samples/s",
                       gather_GBps=by / sec / 1e9)
        elif name in ("tir_vm_app_fwd", "tir_vm_app_fwd_h16", "tir_indirect_fused_fwd") and units and units["n"] > 0:
            per = B_APP_GATHER if name == "tir_vm_app_fwd" else B_APP_GATHER_H16
            by = units["n"] / k["launches"] * (per + units["out_bytes"])
            row.update(bound="l2", units=units["n"] / k["launches"], unit="appearance gathers/launch", gather_bytes=by,
                       achieved=by / sec / 1e9, peak=L2_PEAK_GBS, runit="GB/s", taps_per_s=units["n"] / k["launches"] * 18 / sec)
            if name == "tir_indirect_fused_fwd":       # the fused kernel also carries the decoder's matrix work: second reading
                row["decoder_TFLOPs"] = units["flops"] / k["launches"] / sec / 1e12
                row["decoder_frac_of_dense_fp16_peak"] = row["decoder_TFLOPs"] / BF16_MFMA_PEAK_TF
        elif name.startswith("tir_mlp_fwd") and units and units["n"] > 0:
            fl = units["flops"] / k["launches"]
            # split-bf16 issues 3 bf16 MFMAs per fp32-equivalent product: price it against the dense bf16 peak / 3; the
            # single-product fp16 decoder against the dense peak itself; the exact decoder against the fp32 MFMA peak
            peak = F32_MFMA_PEAK_TF if name == "tir_mlp_fwd" else (BF16_MFMA_PEAK_TF if name.endswith("_f16") else BF16_MFMA_PEAK_TF / 3.0)
            row.update(bound="mfma", units=units["n"] / k["launches"], unit="decoder rows/launch",
                       achieved=fl / sec / 1e12, peak=round(peak, 1), runit="TFLOP/s")
        if "achieved" in row:
            row["frac"] = row["achieved"] / row["peak"]
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def attribute_kernels(run_eager, psteps, io_primary, io_secondary, device, stat_steps=1):
    """Per-kernel attribution of the inference path: `run_eager()` issues one eager pass (every C call bracketed by events on
    the launch stream, ops.TIMING); the ops wrappers are instrumented to count the rows each gather / decoder launch really
    processed (device-side counts), one extra pass reads the counters of gathered density samples.  Returns
    (rows, gpu_ms_per_step, event_overhead_ms)."""
    from tensoir_amd import ops
    # ---- per-kernel attribution: pass 1 brackets every C call with events on the launch stream (no counters),
    #      pass 2 (one step) reads the device-side counters of gathered density samples ------------------
    ops.TIMING, ops.STATS = [], None
    # rows actually processed: min(buffer rows, device-side count) -- the counts are read back after the pass; keyed by the
    # entry point that really runs (the indirect-light policy sends the secondary records to the h16 gather / f16 decoder)
    pending = []
    DEC = lambda o: 2 * (150 * 128 + 128 * 128 + 128 * o)          # useful FLOPs of one decoder row
    orig = {k: getattr(ops, k) for k in ("vm_app", "vm_app_h16", "mlp", "vm_app_primary", "vm_app_jitter", "mlp_multi", "indirect_fused")}

    def fused_wrap(field, fh, m, xyz, light_idx, rec_map, idx_div, dirs, n_dirs, n_dev=None):
        pending.append(("tir_indirect_fused_fwd", xyz.shape[0], n_dev, 4 * m.out_dim, DEC(m.out_dim)))
        return orig["indirect_fused"](field, fh, m, xyz, light_idx, rec_map, idx_div, dirs, n_dirs, n_dev)

    def app_wrap(field, xyz, *args, **kw):
        n_dev = kw.get("n_dev", args[6] if len(args) > 6 else None)
        want_rad = kw.get("want_rad", args[2] if len(args) > 2 else True)
        want_int = kw.get("want_int", args[3] if len(args) > 3 else False)
        pending.append(("tir_vm_app_fwd", xyz.shape[0], n_dev, 27 * 4 * (int(want_rad) + int(want_int)), 0))
        return orig["vm_app"](field, xyz, *args, **kw)

    def h16_wrap(field, fh, xyz, *args, **kw):
        n_dev = kw.get("n_dev", args[3] if len(args) > 3 else None)
        pending.append(("tir_vm_app_fwd_h16", xyz.shape[0], n_dev, 27 * 4, 0))
        return orig["vm_app_h16"](field, fh, xyz, *args, **kw)

    def mlp_wrap(m, feat, aux, aux_map=None, impl=None, aux_mod=0, n_dev=None):
        impl_eff = impl or ops.MLP_IMPL
        tabled = ops.AUX_TABLE and (aux_map is not None or aux_mod > 0) and aux.shape[0] * 8 <= max(feat.shape[0], 1)
        key = "tir_mlp_fwd_auxtab_f16" if (impl_eff == "f16" and tabled) else ("tir_mlp_fwd" if impl_eff == "mfma" else "tir_mlp_fwd_bf16x3")
        pending.append((key, feat.shape[0], n_dev, 0, DEC(m.out_dim)))
        return orig["mlp"](m, feat, aux, aux_map, impl, aux_mod, n_dev)

    def prim_wrap(field, xyz, *args, **kw):
        n_dev = kw.get("n_dev", args[4] if len(args) > 4 else None)
        pending.append(("tir_vm_app_fwd", xyz.shape[0], n_dev, 27 * 4 * 2, 0))              # records: radiance + intrinsic features
        pending.append(("tir_vm_app_fwd", xyz.shape[0], n_dev, 27 * 4 + 12, 0))             # jittered records: intrinsic features + the points
        return orig["vm_app_primary"](field, xyz, *args, **kw)

    def jit_wrap(field, xyz, *args, **kw):
        n_dev = kw.get("n_dev", args[4] if len(args) > 4 else None)
        pending.append(("tir_vm_app_fwd", xyz.shape[0], n_dev, 27 * 4 + 12, 0))
        return orig["vm_app_jitter"](field, xyz, *args, **kw)

    def multi_wrap(jobs, n_dev=None):
        for m, _, _, _ in jobs:
            pending.append(("tir_mlp_fwd_bf16x3" if ops.MLP_IMPL != "mfma" else "tir_mlp_fwd", jobs[0][1].shape[0], n_dev, 0, DEC(m.out_dim)))
        return orig["mlp_multi"](jobs, n_dev)

    ops.vm_app, ops.vm_app_h16, ops.mlp = app_wrap, h16_wrap, mlp_wrap
    ops.vm_app_primary, ops.vm_app_jitter, ops.mlp_multi = prim_wrap, jit_wrap, multi_wrap
    ops.indirect_fused = fused_wrap
    try:
        for _ in range(psteps):
            run_eager()
        torch.cuda.synchronize()
    finally:
        for k, v in orig.items():
            setattr(ops, k, v)
    shapes = {"tir_march_primary_fwd": {"io_bytes": io_primary}, "tir_march_secondary_fwd": {"io_bytes": io_secondary}}
    for key, rows_, n_dev, out_bytes, flops in pending:
        n = rows_ if n_dev is None else min(rows_, int(n_dev.item()))
        e = shapes.setdefault(key, {"n": 0, "out_total": 0.0, "flops": 0.0})
        e["n"] += n
        e["out_total"] += n * out_bytes
        e["flops"] += n * flops
    for e in shapes.values():
        if "n" in e:
            e["out_bytes"] = e["out_total"] / max(1, e["n"])
    timing = ops.TIMING
    ops.TIMING, ops.STATS = None, {}
    for _ in range(stat_steps):                       # counters accumulate over the passes; kernel_table wants them per step
        run_eager()
    torch.cuda.synchronize()
    stats = {k: v.clone() // stat_steps for k, v in ops.STATS.items()}
    ops.STATS = None
    ev_over = event_bracket_overhead_ms(device)
    rows = kernel_table(timing, stats, psteps, shapes, ev_over)
    return rows, sum(r["ms_per_step"] for r in rows), ev_over


def library_info():
    """Which library the numbers of this run come from: the source hash csrc/build.sh stamps next to the .so it links."""
    p = os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so.srchash")
    try:
        h = open(p).read().strip()
    except OSError:
        h = None
    return {"so": "tensoir_amd/libtensoir_hip.so", "source_hash": h,
            "source_hash_of": "sha256 over every csrc/*.hip + tir_common.hpp + include/tensoir_hip.h + compiler flags (csrc/build.sh)"}


def load_pmc():
    """(bytes per launch, issue fractions, meta) from the separate rocprofv3 --pmc passes kept under profiles/ (tools/
    tools/round_evidence.sh + tools/summarize_prof.py).  The files carry the source hash of the library they were collected with; when it
    is not the library being timed now, every PMC-derived field of this run is marked `"stale": true`."""
    out = []
    for name in ("pmc_traffic.json", "pmc_issue.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            out.append(json.load(open(path)))
        except Exception:
            out.append({})
    lib = library_info()["source_hash"]
    hashes = {d.get("_library_source_hash") for d in out if d}
    meta = {"library_source_hash": sorted(h for h in hashes if h), "current_library": lib,
            "stale": not (len(hashes) == 1 and lib is not None and lib in hashes),
            "source": "profiles/pmc_traffic.json, profiles/pmc_issue.json (separate rocprofv3 --pmc passes of this command)"}
    return out[0], out[1], meta


def roofline_object(r, pmc_traffic, pmc_issue, pmc_meta=None):
    """One roofline object: achieved / peak with an INDEPENDENT peak (frac <= 1 by construction).
    bound 'mfma': useful decoder FLOPs vs the matrix-core ceiling of the operand scheme.
    bound 'l2'  : the appearance gathers read a cache-resident field; the bounding resource is the fill rate of the vector
                  L1s = the aggregate L2 bandwidth (256 CUs x 64 B/clk): gather bytes (SURVEY 8d model) / launch time vs 34.5 TB/s.
    bound 'valu': the density march; PMC says VALU issue is the binding pipe, so the ceiling is the VALU issue rate over the
                  FMAs the algorithm needs per valid sample; the PMC instruction count shows how much of the issued work is that.
    The SURVEY 8d gather-bytes-over-HBM-peak figure is kept as the labelled `sec8d_hbm_model` (a ratio that exceeds 1 for a
    cache-resident field -- NOT a roofline fraction), next to the counter-measured HBM traffic."""
    stale = bool(pmc_meta and pmc_meta.get("stale"))
    t = pmc_traffic.get(r["kernel"])
    o = {"kernel": r["kernel"], "bound": r["bound"], "achieved": round(r["achieved"], 3),
         "peak": r["peak"], "unit": r["runit"], "frac": round(r["frac"], 4),
         "traffic": t, "avg_launch_ms": round(r["avg_ms"], 4),
         "units_per_launch": round(r["units"], 1), "unit_of_work": r["unit"]}
    if t is not None:
        o["traffic_stale"] = stale
    if r["kernel"] in ROCPROF_KERNELS:
        o["rocprof_kernels"] = ROCPROF_KERNELS[r["kernel"]]
    iss = pmc_issue.get(r["kernel"]) or {}
    if r["bound"] in ("l2", "valu"):
        gb = r.get("gather_GBps", r["achieved"]) if r["bound"] == "valu" else r["achieved"]
        o["sec8d_hbm_model"] = {"gather_bytes_per_launch": round(r["gather_bytes"], 1), "gather_GBps": round(gb, 2),
                                "hbm_peak_GBps": HBM_PEAK_GBS, "gather_GBps_over_hbm_peak": round(gb / HBM_PEAK_GBS, 3),
                                "note": "SURVEY 8d gather-bytes model; the field is cache resident, so this ratio is "
                                        "not bounded by 1 and is not a roofline fraction"}
        if t:
            o["hbm_traffic"] = {"bytes_per_launch": t, "GBps": round(t / (r["avg_ms"] * 1e-3) / 1e9, 2),
                                "frac_of_hbm_peak": round(t / (r["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "stale": stale,
                                "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/pmc_traffic.json"}
    if r["bound"] == "l2":
        o["peak_source"] = "MI355X_MICROARCH.md L2 aggregate 34.5 TB/s = 256 CUs x 64 B/clk of vector-L1 fill"
        if "taps_per_s" in r:
            o["taps_per_s"] = round(r["taps_per_s"], 1)
    elif r["bound"] == "valu":
        o["peak_source"] = (f"VALU issue ceiling {VALU_PEAK_WAVE_INSTR / 1e9:.1f} G wave64 instructions/s (256 CUs x 4 SIMDs x 2.4 GHz / 4) over the "
                            f"algorithm's {VALU_FLOOR_PER_DENSITY_SAMPLE:.2f} FMA instructions per valid sample (3 planes x 16 channels x 7 / 64 lanes)")
        # the same ceiling if every FMA of the floor were a packed v_pk_fma_f32 (two channels per instruction: the kernel's
        # interpolation IS packed; the product-sum and everything per sample is not) -- the stricter yardstick
        o["packed_fma_floor"] = {"peak": round(o["peak"] * 2.0, 2), "frac": round(o["frac"] / 2.0, 4), "unit": o.get("unit"),
                                 "note": f"{VALU_FLOOR_PER_DENSITY_SAMPLE / 2:.3f} packed instructions per valid sample (42 per 16-sample pass)"}
        o["l2_model"] = {"bound": "l2", "achieved": round(r["gather_GBps"], 2), "peak": L2_PEAK_GBS, "unit": "GB/s",
                         "frac": round(r["gather_GBps"] / L2_PEAK_GBS, 4), "note": "gather bytes through the vector L1s (density lines staged in LDS "
                         "are counted although they never reach the L1)"}
        if iss.get("valu_instructions_per_launch"):
            # what the kernel really issues, from the SQ counter pass: VALU instructions per 16-sample gather pass against the
            # 84 the FMAs need; the rest is index / weight / occupancy / compositing arithmetic -- the headroom of this kernel
            per_pass = iss["valu_instructions_per_launch"] / max(r["units"] / 16.0, 1.0)
            o["valu"] = {"valu_per_pass": round(per_pass, 1), "fma_floor_per_pass": 16 * VALU_FLOOR_PER_DENSITY_SAMPLE,
                         "useful_valu_frac": round(16 * VALU_FLOOR_PER_DENSITY_SAMPLE / per_pass, 4),
                         "valu_issue_frac": iss.get("valu_issue_frac"), "wait_frac_of_wave_cycles": iss.get("wait_frac_of_wave_cycles"),
                         "stale": stale, "source": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES, profiles/pmc_issue.json; "
                         "pass = 16 valid samples (4 lanes each); units_per_launch of THIS run"}
    else:
        o["frac_of_dense_bf16_peak"] = round(r["achieved"] / BF16_MFMA_PEAK_TF, 4)
        if iss:
            o["pmc"] = {"mfma_busy_frac": iss.get("mfma_busy_frac"), "valu_issue_frac": iss.get("valu_issue_frac"), "stale": stale}
        if r["kernel"] == "tir_mlp_fwd_bf16x3":
            o["aggregation"] = "all launches of this row run the same device code (mlp_bf16_body); avg_launch_ms / units_per_launch are means over them"
        if r["kernel"].endswith("bf16x3"):
            o["power_limited"] = {
                "note": "back-to-back launches of this kernel on random data run at the board power cap: the shader clock "
                        "settles below the 2.4 GHz the peak assumes; all-zero data (same instructions) runs at 2.39 GHz and "
                        "15-26 % faster", "board_power_W": "1330-1400", "sustained_sclk_GHz": "1.93-2.07",
                "frac_at_sustained_clock": round(r["frac"] * 2.4 / 2.0, 4),
                "bf16_matrix_rate_TF": round(3.0 * r["achieved"], 1),
                "source": "tools/mlp_power.py -> profiles/r02_mlp_power.txt (rocm-smi polled during the launches)"}
        o["peak_source"] = ("dense bf16 MFMA 2.5 PF / 3 products of the split-bf16 scheme" if r["kernel"].endswith("bf16x3")
                            else "dense fp16 MFMA 2.5 PF (single product)" if r["kernel"].endswith("_f16") else "dense f32 MFMA 157.3 TF")
    if iss.get("valu_issue_frac") is not None and iss.get("mfma_busy_frac") is not None:
        # On a gfx950 SIMD fp32 VALU work and the matrix pipe do not run at the same time: a matrix-only wave and an fp32-VALU-only wave
        # sharing a SIMD take the SUM of their times; packed-fp16 and integer instructions overlap ~60 % (tools/mfma_valu_overlap.hip ->
        # profiles/r06_mfma_valu_overlap.txt).  The fraction of the
        # launch in which a SIMD executes one or the other is therefore the sum of the two counters -- the occupancy of the binding
        # resource of a kernel that is neither memory- nor single-pipe-bound.
        o["simd_issue"] = {"valu_issue_frac": iss["valu_issue_frac"], "mfma_busy_frac": iss["mfma_busy_frac"],
                           "sum": round(iss["valu_issue_frac"] + iss["mfma_busy_frac"], 4), "stale": stale,
                           "note": "fp32 VALU and matrix instructions are mutually exclusive on a SIMD, packed-fp16 / integer VALU overlap ~60 % (measured: "
                                   "profiles/r06_mfma_valu_overlap.txt); sum ~ share of the launch in which the SIMDs execute either"}
    return o


def dominant_roofline(rows):
    pmc_traffic, pmc_issue, pmc_meta = load_pmc()
    dom = next((r for r in rows if "achieved" in r), None)
    return roofline_object(dom, pmc_traffic, pmc_issue, pmc_meta) if dom else None


def trained_300_verdict():
    """What the auto policy decided on a checkpoint TRAINED to 300^3 through the product API (tools/precision_300.py, run on a
    GPU box; the JSON is committed evidence, stamped with the library hash it was measured with)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_precision_trained_300.json")))
        dec = d["policy"]["decision"]
        probe = dec["probe"]
        # (a verdict other than f16 carries the rejected f16 measurement under "f16" and the accepted tier's own at the top)
        f16 = probe.get("f16", probe).get("map_max_abs")
        return {"mode": dec["mode"], "why": dec["why"], "map_max_abs_f16_vs_full": f16,
                "map_max_abs_decided_vs_full": probe.get("map_max_abs"), "limit": probe.get("limit"),
                "iterations": d.get("iterations"), "grids": d.get("grids"), "library_source_hash": d.get("library_source_hash"),
                "stale": d.get("library_source_hash") != library_info().get("source_hash"),
                "worst_map_vs_oracle": max(v["max_rel_floor1"] for v in d["oracle"].values() if isinstance(v, dict) and "max_rel_floor1" in v),
                "source": "profiles/r06_precision_trained_300.json"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def port_vs_reference(port_value):
    """What is known about the oracle's speed relative to the imported reference's CPU path (which cannot run on the GPU box).
    profiles/port_over_reference.json is written by oracle/calibrate_port.py in the build container (both implementations, same
    inputs, same threads); profiles/r03_ref_on_gpu.json holds the one staged run of the reference on the GPU box's host cores."""
    out = {"note": "no calibration file"}
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "port_over_reference.json")))
        out = {"port_over_reference_time": cal["port_over_reference"],
               "reference_equivalent_rays_per_s": round(port_value * cal["port_over_reference"], 2),
               "measured": f"oracle/calibrate_port.py in the build container ({cal.get('threads')} threads): reference {cal.get('reference_rays_per_s')} rays/s, "
                           f"port {cal.get('port_rays_per_s')} rays/s on {cal.get('sample', '?').split(',')[0]}",
               "caveat": "the ratio depends on thread count and batch size (the CPU path is dominated by per-op overheads and memory "
                         "traffic, not FLOPs); the staged reference run on THIS kind of box (profiles/r03_ref_on_gpu.json, 128 threads, "
                         "full batch) measured 361-373 rays/s"}
    except Exception:
        pass
    return out


def timed_cpu(fn, warm, calls):
    """Median wall time of `fn()` on the host cores (warm-ups first, time.perf_counter, no_grad)."""
    out, ts = None, []
    with torch.no_grad():
        for i in range(warm + calls):
            t0 = time.perf_counter()
            out = fn()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
    ts.sort()
    return out, ts[len(ts) // 2], ts


def map_parity(got, ref, keys, sel=None, rays=None, disc=None):
    """max |hip - oracle| / max(|oracle|, 1) over the named maps (the test metric), per map and worst.  rays (the oracle's rows):
    rgb_with_brdf_map is compared on the rays where the reference's GGX normal flip is not within fp32 noise of its
    discontinuity (tests/helpers.py ggx_flip_rays).  disc = (oracle module, scene, light_idx rows, secondary samples): rays more
    than 1e-5 off whose ORACLE colour itself jumps by more than 1e-5 for +-2 ulps of the oracle's own depth are listed, not compared
    (tests/helpers.py depth_discontinuity_rays; at most max(1, rays / 500) of them)."""
    from tests.helpers import depth_discontinuity_rays, ggx_flip_rays, parity_metrics
    per, worst = {}, 0.0
    keep = ~ggx_flip_rays(ref["normal_map"], rays) if rays is not None and "normal_map" in ref else None
    jumps = {}
    if disc is not None and keep is not None and "rgb_with_brdf_map" in keys:
        g = got["rgb_with_brdf_map"].detach().cpu()
        g = g[sel] if sel is not None else g
        d_all = (g - ref["rgb_with_brdf_map"]).abs().max(dim=-1).values
        cand = torch.nonzero(keep & (d_all > 1e-5)).reshape(-1).tolist()
        if cand:
            O_, sc_, l_rows, n_s = disc
            jumps = depth_discontinuity_rays(O_, sc_, ref, rays, l_rows, cand, n_s)
            jumps = dict(list(jumps.items())[: max(1, int(keep.numel()) // 500)])
            for i_ in jumps:
                keep[i_] = False
    for k in keys:
        g = got[k].detach().cpu()
        g = g[sel] if sel is not None else g
        r = ref[k]
        if keep is not None and k == "rgb_with_brdf_map":
            g, r = g[keep], r[keep]
        m = parity_metrics(g, r)
        per[k] = {kk: float(f"{vv:.3e}") for kk, vv in m.items()}
        worst = max(worst, m["max_rel_floor1"])
    worst_px = max(v["max_rel_pixel"] for v in per.values()) if per else 0.0
    return {"ok": worst < 1e-4 and worst_px < 1e-4, "tolerance": 1e-4, "max_rel_floor1": float(f"{worst:.3e}"),
            "max_rel": float(f"{worst_px:.3e}"), "per_map": per,
            "depth_discontinuity_rays": {str(k_): float(f"{v_:.3e}") for k_, v_ in jumps.items()},
            "metric": "both asserted: max |hip - oracle| / max(|oracle|, 1) per map and the true per-pixel relative error "
                      "||d|| / ||ref|| over pixels with ||ref|| > 1e-2"}


MAP_KEYS = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map", "acc_map",
            "rgb_with_brdf_map", "normals_diff_map", "normals_orientation_loss_map"]


def local_device(a):
    """This rank's GPU index: LOCAL_RANK, or LOCAL_RANK modulo the visible devices with --allow-shared-gpu (plumbing runs of
    several ranks on one device)."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.allow_shared_gpu and torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()
    return local


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher.  Re-executes this command
    under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, the
    reference's own env rendezvous: train_tensoIR.py:22-27 reads RANK / WORLD_SIZE / MASTER_* the same way) and exits with
    the launcher's return code; rank 0's JSON line is the child's stdout, passed through."""
    import socket
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() < a.gpus and not a.allow_shared_gpu:
        raise SystemExit(f"[bench] --gpus {a.gpus} but only {torch.cuda.device_count()} visible GPU(s) "
                         "(--allow-shared-gpu: plumbing runs of several ranks on one device, not a measurement)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    env["TENSOIR_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), *sys.argv[1:]]
    print(f"[bench] --gpus {a.gpus} without a launcher environment: starting {a.gpus} ranks: {' '.join(cmd[1:9])} bench.py ...",
          file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def side_summary(line, wall_s):
    """The fields of a side workload's full line that the headline run carries in its `workloads` block."""
    rf, par, cpu = line.get("roofline") or {}, line.get("parity") or {}, line.get("cpu_baseline") or {}
    worst = {k: par[k] for k in ("max_rel_floor1", "max_rel", "loss_abs_diff", "grad_max_rel", "field_grad_rel_l2", "field_grad_outlier_share",
                                 "field_grad_max_rel", "record_mask_mismatches") if k in par}
    if "relit_rgb" in par:
        worst.update(par["relit_rgb"])
    if "maps_max_abs" in par:
        worst["maps_max_abs"] = max(par["maps_max_abs"].values())
    return {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
            "warmup": line["warmup"], "scaling": line["scaling"], "workload": line["config"]["workload"],
            "parity": {"ok": par.get("ok"), **worst},
            "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")} if rf else None,
            "cpu_baseline": {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cpu else None,
            "wall_s_incl_setup_and_cpu_checks": round(wall_s, 1)}


def check_launch(a):
    """--gpus N must be the number of ranks actually running, each with a GPU of its own: a scaling
    line must not be printable from fewer processes or devices than it claims.  A bare `python bench.py --gpus N` (no
    WORLD_SIZE / RANK in the environment) starts its N ranks itself (self_launch); a launcher environment whose WORLD_SIZE
    differs from --gpus is refused."""
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU, e.g. python -m "
                         f"torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} "
                         f"(or unset WORLD_SIZE / RANK and bench.py starts its ranks itself)")
    if torch.cuda.is_available() and torch.cuda.device_count() < world and not a.allow_shared_gpu:
        raise SystemExit(f"[bench] {world} ranks but only {torch.cuda.device_count()} visible GPU(s) "
                         "(--allow-shared-gpu: plumbing tests of several ranks on one device, not a measurement)")
    if world > 1:
        print(f"[bench] rank {os.environ.get('RANK', '0')}/{world} up (local rank {os.environ.get('LOCAL_RANK', '0')}, "
              f"backend {a.backend})", file=sys.stderr, flush=True)

#!/usr/bin/env python
"""Headline benchmark: primary+secondary rays/s at 4096 rays x 512 samples (BASELINE.json), one
MI355X per rank.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of 4096 synthetic rays per rank:
``Renderer_TensoIR_train`` = primary march + appearance gather + 3 decoders + analytic normals +
compositing, then for every ray with acc > 0.5 the secondary march over 8x16 = 128 light directions x
96 samples (visibility + indirect radiance) and the GGX x SG-environment integration.  Rays and the
field are resident in HBM before the timed region.  With N > 1 every rank renders its own 4096 rays
(weak scaling; rays are independent) and the per-ray output records are all-gathered over RCCL once
per step.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the workloads live in benchlib/ (round 6 split); every name stays reachable as bench.<name> for tests and tools
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.image import bench_image, sharp_scene_line, simulate_ranks  # noqa: E402,F401
from benchlib.relight import bench_relight, synthetic_hdr_maps  # noqa: E402,F401
from benchlib.train import (ATOMIC_SEGMENTS_PER_S, FIELD_TENSORS, TRAIN_W, _to_fp64, atomic_segments_after_combining, bench_train,  # noqa: E402,F401
                            grad_deviation, single_ray_bisect, train_loss, train_parity_and_cpu)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--env-h", type=int, default=8)
    ap.add_argument("--env-w", type=int, default=16)
    ap.add_argument("--second-samples", type=int, default=96)
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays in the CPU-baseline sample (default: the full batch)")
    ap.add_argument("--cpu-calls", type=int, default=2, help="timed CPU-baseline calls (after 1 warm-up)")
    ap.add_argument("--boundary-calls", type=int, default=50, help="timed eager boundary calls (after 10 warm-ups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct ray batches (camera poses) rotated through the lanes inside the timed region; 1 = the same "
                         "rays every step (round-2 behaviour)")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="a second timed region of max(this, --steps) steps, reported next to the K-step figure")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--decoder", type=str, default="bf16x3", choices=["mfma", "bf16x3"],
                    help="decoder matrix-core mode: bf16x3 = split-bf16 (parity grade), mfma = exact fp32")
    ap.add_argument("--no-exact-pass", action="store_true", help="skip the extra timed pass with the exact fp32 decoders")
    ap.add_argument("--no-full-pass", action="store_true", help="skip the extra timed pass with the indirect-light policy forced to `full`")
    ap.add_argument("--breakdown", type=str, default="", help="write the per-kernel table to this file")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying the HIP graph")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="independent batches in flight: that many captured graphs of the step, replayed round-robin on as many "
                         "HIP streams (a step's kernels have tails in which CUs idle; another batch fills them).  1 = one stream")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--no-sharp-scene", action="store_true",
                    help="skip the secondary (informative) line on a sharp-surface scene (10-30 appearance samples per ray, as a "
                         "trained scene keeps; the headline blob keeps ~56)")
    ap.add_argument("--maps", type=int, default=5, help="relight workload: environment maps per view (the reference loop uses 5)")
    ap.add_argument("--workload", default="batch", choices=["batch", "image", "relight", "train"],
                    help="batch = the headline 4096-ray step (weak scaling); image = one 800x800 image of BASELINE configs[3] "
                         "(3 light rotations, 1036 samples per ray) sharded over the ranks, one all-gather per image (strong scaling)")
    ap.add_argument("--tile", type=int, default=-1,
                    help="image workload: 0 = contiguous row tiles, >0 = interleaved tiles of that many rays, -1 (default) = "
                         "interleaved tiles of one chunk when there is more than one rank (background rows finish early: with "
                         "row tiles the ranks that hold the object set the pace, SURVEY 8e), row tiles on one rank")
    ap.add_argument("--image-side", type=int, default=800)
    ap.add_argument("--allow-shared-gpu", action="store_true", help="let several ranks share one GPU (gloo plumbing tests only)")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="batch workload, one GPU: skip the reduced-repetition image / relight / train lines of the `workloads` block")
    ap.add_argument("--c5-host-masking", action="store_true",
                    help="relight workload: drive every chunk the way the reference script does (boolean-mask indexing, nonzero, "
                         "index_put_: one host synchronisation per chunk) instead of the product's sync-free chunk call")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="image / relight workload on ONE GPU: render each rank's shard of a W-rank job alone (W = 2, 4, ... up to this), "
                         "row tiles and interleaved tiles, and report per-shard times, max / mean and the predicted speed-up")
    ap.add_argument("--force-dist", action="store_true",
                    help="single process: create a 1-rank RCCL group anyway and run the multi-rank code path (all-gather per step)")
    return ap.parse_args()


def main():
    a = parse()
    check_launch(a)
    if a.workload == "image":
        return bench_image(a)
    if a.workload == "relight":
        return bench_relight(a)
    if a.workload == "train":
        return bench_train(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: tensoir_amd has no CPU path")
    local = local_device(a)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    n_gpus = world
    use_dist = world > 1 or a.force_dist        # one process per GPU over RCCL (--force-dist: the same path on 1 rank)

    from tensoir_amd import Renderer_TensoIR_train, _lib, ops
    from tensoir_amd import dist as tdist
    assert _lib.lib().tir_device_check() == 0
    ckpt, model, rays, lidx = build_scene(a, device, rank)
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    B = rays.shape[0]
    # the batches the timed region rotates through (resident in HBM; a step copies its batch into the lane's graph inputs)
    batches = [b.to(device) for b in pose_batches(rays.cpu(), max(1, a.batches), rank)]
    rays = batches[0]

    from tensoir_amd.graph import GraphedRenderer
    graphed = {}

    def make_graph(impl):
        """One captured step per batch in flight (each with its own input / output buffers and device-side pass state)."""
        ops.MLP_IMPL = impl
        grs = []
        for _ in range(max(1, a.in_flight) if impl == a.decoder else 1):
            gr = GraphedRenderer(model, B, N_samples=a.samples, args=args, device=device)
            gr.rays.copy_(rays)                      # the batch is resident in HBM: it sits in the graph's input buffers
            gr.lidx.copy_(lidx)
            gr(clone_outputs=False)                  # capture + one checked replay
            for b in batches[1:] + batches[:1]:      # every pose once, checked: a pose that needs more record room than the
                gr(rays=b, clone_outputs=False)      # captured capacity re-captures with room for it (converges to the heaviest)
            grs.append(gr)
        graphed[impl] = grs

    # Every rank captures its step graph(s) BEFORE the process group exists: no RCCL thread is alive yet that could
    # issue a call into the runtime while the stream is capturing.  Replays and the per-step all-gather then simply
    # follow each other on the stream.
    if not a.no_graph:
        try:
            for impl in dict.fromkeys([a.decoder] + (["mfma"] if a.decoder != "mfma" and not a.no_exact_pass else [])):
                make_graph(impl)
        except Exception as e:                       # capture refused on this box: the eager path is the same work
            print(f"[bench] HIP-graph capture unavailable ({type(e).__name__}: {e}); using eager launches",
                  file=sys.stderr, flush=True)
            graphed.clear()
            a.no_graph = True
        ops.MLP_IMPL = a.decoder
    if use_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)               # --force-dist in a bare single process
        dist.init_process_group(a.backend, **({"device_id": device} if a.backend == "nccl" else {}))
    lanes = max(1, a.in_flight)
    gathered = [torch.empty((world * B, tdist.RECORD), dtype=torch.float32, device=device) for _ in range(lanes)] if use_dist else None
    streams = [torch.cuda.Stream(device=device) for _ in range(lanes)]
    state = {"i": 0, "lanes": lanes, "b": 0, "last_b": 0, "e": 0}

    def fork():
        """The lanes' streams start behind everything queued on the current stream."""
        cur = torch.cuda.current_stream()
        for st in streams[:state["lanes"]]:
            st.wait_stream(cur)

    def join():
        cur = torch.cuda.current_stream()
        for st in streams[:state["lanes"]]:
            cur.wait_stream(st)

    def step(eager=False):
        """One pass over one batch.  Graph replays go round-robin over the lanes (batch i on stream i mod lanes, through that
        lane's own captured graph); eager passes and the one-lane mode run on the current stream.  Timed steps walk through
        the pose batches (step i renders batch i mod --batches); the eager attribution passes render batch 0."""
        if eager:                                       # the attribution passes walk through the poses as the timed steps do
            bi = state["e"] % len(batches)
            state["e"] += 1
        else:
            bi = state["b"] % len(batches)
            state["b"] += 1
            state["last_b"] = bi
        if eager or a.no_graph or state["lanes"] == 1:
            return step_on(0, eager, bi)
        lane = state["i"] % state["lanes"]
        state["i"] += 1
        with torch.cuda.stream(streams[lane]):
            return step_on(lane, eager, bi)

    def step_on(lane, eager, bi=0):
        rays = batches[bi]
        with torch.no_grad():
            if a.no_graph or eager:       # (_no_graph: the per-kernel attribution needs the launches themselves, not the boundary's cached graph)
                ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True,
                                             is_train=False, is_relight=True, sample_method="fixed_envirmap",
                                             chunk_size=160000, device=device, args=args, _no_graph=True)
            else:           # the same launches, replayed as one HIP graph per decoder mode (tensoir_amd/graph.py)
                try:
                    if ops.MLP_IMPL not in graphed:
                        if use_dist:
                            raise RuntimeError("no graph was captured for this decoder mode before the process group was created")
                        make_graph(ops.MLP_IMPL)
                    grs = graphed[ops.MLP_IMPL]
                    gr = grs[lane % len(grs)]
                    # outputs stay in the graph's buffers (valid until the next step); the record-capacity check of all
                    # queued replays is made once, inside the timed region, by validate() below
                    # the batch moves into the graph's input buffers on this lane's stream (98 KB, device to device)
                    ret = gr(rays=rays, clone_outputs=False, defer_check=not getattr(a, "no_defer", False))
                except Exception as e:      # capture refused on this box: the eager path is the same work
                    print(f"[bench] HIP-graph replay unavailable ({type(e).__name__}: {e}); using eager launches",
                          file=sys.stderr, flush=True)
                    a.no_graph = True
                    return step_on(lane, eager, bi)
            if use_dist:   # the one exchange step: all-gather of the rendered per-ray records
                dist.all_gather_into_tensor(gathered[lane], tdist.pack_records(ret))
        return ret

    def settle(n_steps=SETTLE_STEPS):
        """Untimed: bring the GPU out of its idle power state (a fresh box reports 'low-power state'; the first ~50 ms of
        work run at ramping clocks: 2.5 ms per step instead of 1.85 measured right after process start).  A FIXED number
        of steps (~0.5 s), not a time budget: with several ranks every step ends in a collective, so all ranks must run the
        same number of them."""
        fork()
        for i in range(n_steps):
            step()
            if i % 10 == 9:
                torch.cuda.synchronize()
        join()

    def timed(n_warm, n_steps):
        fork()
        for _ in range(n_warm):
            step()
        join()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = None
        fork()
        for _ in range(n_steps):
            r = step()
        join()
        torch.cuda.synchronize()
        valid = all(g.validate() for grs in graphed.values() for g in grs)     # sticky overflow flag of every replay queued above
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:                                               # re-timing is a collective decision (taken off the clock)
            ok = torch.tensor([1 if valid else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            valid = bool(ok.item())
        if not valid:                                              # a capacity overflowed: time again with per-step checks
            print("[bench] a deferred record-capacity check failed; re-timing with per-step checks", file=sys.stderr, flush=True)
            if use_dist:
                a.no_graph = True          # a re-capture would run next to live RCCL threads: finish eagerly instead
            else:
                a.no_defer = True
            return timed(n_warm, n_steps)
        if use_dist:
            pr = torch.zeros((dist.get_world_size(),), dtype=torch.float64, device=device)
            pr[dist.get_rank()] = el
            dist.all_reduce(pr)                                   # every rank's own clock (the value uses the MAX)
            state["per_rank_s"] = (pr / n_steps).tolist()
            el = float(pr.max().item())
        else:
            state["per_rank_s"] = [el / n_steps]
        return el, r

    ops.MLP_IMPL = a.decoder
    settle()
    state["b"] = 0
    elapsed, _ = timed(a.warmup, a.steps)
    per_rank_ms = [round(1e3 * x, 4) for x in state["per_rank_s"]]
    n_sus = max(a.sustained_steps, a.steps)
    state["b"] = 0
    el_sus, _ = timed(0, n_sus)                          # a longer region over the same rotation, reported next to the K steps
    sustained = {"steps": n_sus, "value": round(n_gpus * B * n_sus / el_sus, 1), "ms_per_step": round(1e3 * el_sus / n_sus, 4),
                 "batches_rotated": len(batches)}
    single = None
    if state["lanes"] > 1 and not a.no_graph:            # the same steps one at a time on one stream, for reference
        state["lanes"] = 1
        n1 = max(len(batches), a.steps // 2, n_sus // 4)
        state["b"] = 0
        el1, _ = timed(1, n1)
        state["lanes"] = lanes
        single = {"in_flight": 1, "steps": n1, "value": round(n_gpus * B * n1 / el1, 1), "ms_per_step": round(1e3 * el1 / n1, 4)}
        if graphed.get(a.decoder) and len(graphed[a.decoder]) > 1:      # every lane's graph computes the same maps, bit for bit
            outs = [g(rays=batches[-1], clone_outputs=True) for g in graphed[a.decoder]]
            for o in outs[1:]:
                for k, v in outs[0].items():
                    if torch.is_tensor(v) and "smoothness" not in k and not torch.equal(v, o[k]):
                        raise SystemExit(f"[bench] lanes disagree on {k}")
    # the canonical batch once more through the timed route (graph replay unless --no-graph): surface points, parity rows
    with torch.no_grad():
        if graphed.get(a.decoder):
            ret = graphed[a.decoder][0](rays=batches[0], clone_outputs=True)
        else:
            ret = Renderer_TensoIR_train(batches[0], None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False,
                                         is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args)
    torch.cuda.synchronize()
    exact = None
    if a.decoder != "mfma" and not a.no_exact_pass:      # same workload with the exact-fp32 decoders, for reference
        ops.MLP_IMPL = "mfma"
        state["lanes"] = 1                               # one captured graph for this reference pass
        n2 = max(len(batches), a.steps // 2)
        state["b"] = 0
        el2, _ = timed(1, n2)
        state["lanes"] = lanes
        exact = {"decoder": "mfma (exact fp32)", "steps": n2, "value": round(n_gpus * rays.shape[0] * n2 / el2, 1),
                 "ms_per_step": round(1e3 * el2 / n2, 4)}
        ops.MLP_IMPL = a.decoder

    # ---- the same steps with the indirect-light policy forced to `full` (TENSOIR_INDIRECT_PRECISION=full): what a checkpoint
    #      pays whose self-check rejects the fp16 kernels -- a 300^3 TRAINED checkpoint does (profiles/r06_precision_trained_300.json),
    #      the freshly initialised field of this bench does not.  Same graphs, re-captured under the forced policy; single process
    #      only (a re-capture must not run next to live RCCL threads).
    #      `hp` = the auto policy's first fallback (round 6: one launch, fp32 taps, fp16 + fp8-residue decoder weights), `full` its last.
    forced_lines = {}
    if not use_dist and not a.no_graph and not a.no_full_pass and graphed.get(a.decoder) and ops.secondary_mlp_impl() is not None:
        saved = (ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL)
        try:
            rets = {}
            for pol, flags, route in (("full", (False, None, None), ops.full_indirect_route()), ("hp", (False, "hp", None), ops.hp_indirect_route())):
                ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = flags
                for gr in graphed[a.decoder]:
                    for b in batches:                        # re-capture + record capacities of every pose
                        gr(rays=b, clone_outputs=False)
                state["b"] = 0
                elf, _ = timed(a.warmup, a.steps)
                with torch.no_grad():
                    rets[pol] = graphed[a.decoder][0](rays=batches[0], clone_outputs=True)["rgb_with_brdf_map"]
                forced_lines[pol] = {"policy": pol, "steps": a.steps, "value": round(n_gpus * B * a.steps / elf, 1),
                                     "ms_per_step": round(1e3 * elf / a.steps, 4), "in_flight": lanes, "kernels": route,
                                     "rgb_with_brdf_max_abs_vs_default_policy": float(f"{float((rets[pol] - ret['rgb_with_brdf_map']).abs().max()):.3e}")}
            forced_lines["hp"]["rgb_with_brdf_max_abs_vs_full"] = float(f"{float((rets['hp'] - rets['full']).abs().max()):.3e}")
        finally:
            ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = saved
            for gr in graphed[a.decoder]:                    # back under the default policy for everything that follows
                for b in batches:
                    gr(rays=b, clone_outputs=False)
        torch.cuda.synchronize()

    M = int((ret["acc_map"] > 0.5).sum())
    D = a.env_h * a.env_w
    # every pose once per round: the per-kernel durations are averages over the same rotation the timed region and a rocprofv3
    # trace of this command see
    n_rot = len(batches) * max(1, min(a.profile_steps, a.steps) // len(batches))
    state["e"] = 0
    rows, gpu_ms, ev_over = attribute_kernels(lambda: step(eager=True), n_rot, B * (24 + 4 + 12) + B * a.samples * 4,
                                              M * D * (24 + 16), device, stat_steps=len(batches))

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    pmc_traffic, pmc_issue, pmc_meta = load_pmc()
    roof = lambda r: roofline_object(r, pmc_traffic, pmc_issue, pmc_meta)
    dom = next((r for r in rows if "achieved" in r), None)
    roofline = roof(dom) if dom else None
    # the fused VM-sample (density gather + march) kernel the north star names, whatever its rank in the table
    pick = lambda *names: next((r for nm in names for r in rows if r["kernel"] == nm and "achieved" in r), None)
    vm = pick("tir_march_secondary_fwd")
    roofline_vm = roof(vm) if vm else None
    vapp = pick("tir_indirect_fused_fwd", "tir_vm_app_fwd_h16", "tir_vm_app_fwd")   # the secondary-record gather (fused with its decoder by default)
    roofline_app = roof(vapp) if vapp else None
    vdec = pick("tir_mlp_fwd_auxtab_f16", "tir_mlp_fwd_bf16x3", "tir_mlp_fwd")      # (with the fused kernel: the primary-stage decoders)
    roofline_dec = roof(vdec) if vdec else None

    # ---- the reference's boundary call, eagerly, host rays in (renderer.py:74-75 does the H2D per call) ----------
    boundary = None
    if world == 1:
        ops.MLP_IMPL = a.decoder
        r_hosts, l_host = [b.cpu().pin_memory() for b in batches], lidx.cpu().pin_memory()
        stream = torch.cuda.current_stream()
        ts = []
        with torch.no_grad():
            for i in range(10 + a.boundary_calls):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(stream)
                Renderer_TensoIR_train(r_hosts[i % len(r_hosts)], None, l_host, model, N_samples=a.samples, white_bg=True, is_train=False,
                                       is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device,
                                       args=args)
                e1.record(stream)
                e1.synchronize()
                if i >= 10:
                    ts.append(e0.elapsed_time(e1))
        med = sorted(ts)[len(ts) // 2]
        boundary = {"rays_per_s": round(B / (med * 1e-3), 1), "ms": round(med, 4), "min_ms": round(min(ts), 4),
                    "max_ms": round(max(ts), 4),
                    "protocol": f"BASELINE.md 2.1: Renderer_TensoIR_train(host rays) as the unmodified scripts call it, incl. H2D of rays, hipEvent pair "
                                f"per call, 10 warm-ups, median of {len(ts)}; call i renders pose i mod {len(r_hosts)}; the boundary replays its cached "
                                f"HIP graph of this call shape (tensoir_amd/renderer.py, round 6; TENSOIR_BOUNDARY_GRAPHS=0: eager launches)",
                    "boundary_graphs": bool(__import__("tensoir_amd.renderer", fromlist=["x"]).BOUNDARY_GRAPHS)}

    # ---- CPU baseline: the oracle (same algorithm, ATen CPU ops) on a bounded sample; its outputs double as a
    #      full-size parity check of the HIP maps (rays are independent; sharding is bit-exact) --------------
    cpu = parity = cal_ctx = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle import tensoir_oracle as O          # checker / CPU baseline only
        from tests.helpers import parity_metrics, scene_from_model
        sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
        stride = max(1, B // a.cpu_rays)
        r_cpu, l_cpu = rays.cpu()[::stride][: a.cpu_rays], lidx.cpu()[::stride][: a.cpu_rays]
        # BASELINE.md 2.1 wants the reference's own CPU path timed.  The reference (Python) cannot travel to the GPU box in any
        # form, so what is timed here is the oracle (kind "port"); `vs_reference` relates it to the imported reference itself,
        # measured in the build container where both exist (oracle/calibrate_port.py -> profiles/port_over_reference.json).
        times, ref = [], None
        with torch.no_grad():
            for i in range(1 + a.cpu_calls):
                t1 = time.perf_counter()
                ref = O.renderer_train(sc, r_cpu, l_cpu, n_samples=a.samples, second_n_sample=a.second_samples)
                if i >= 1:
                    times.append(time.perf_counter() - t1)
        if True:      # (one CPU baseline kind: the port)
            med = sorted(times)[len(times) // 2]
            cpu = {"value": round(r_cpu.shape[0] / med, 2), "unit": "rays/s", "cores": torch.get_num_threads(),
                   "kind": "port", "reference_checkout": False,
                   "note": "the reference checkout (/root/reference) does not exist on the GPU box; the timed code is the oracle, "
                           "a functional restatement on the same ATen CPU ops (F.grid_sample, cumprod, F.linear), pinned to "
                           "the imported reference by tests/golden/.  `vs_reference` relates it to the imported reference itself",
                   "sample": (f"the full batch ({r_cpu.shape[0]} rays" if stride == 1 else f"every {stride}th ray of the batch ({r_cpu.shape[0]} rays") +
                             f" x {a.samples} samples, {D} dirs x {a.second_samples}), 1 warm-up + {len(times)} timed calls, median "
                             f"(min {min(times):.2f} s, max {max(times):.2f} s); host nproc={os.cpu_count()}"}
            cpu["vs_reference"] = port_vs_reference(cpu["value"])
            cal_ctx = (sc, O)           # the port at the calibration's thread count is timed at the very END of the run (see there)
        # parity of the timed HIP path (graph replay outputs `ret`) against those oracle rows
        maps = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map", "acc_map",
                "rgb_with_brdf_map", "normals_diff_map", "normals_orientation_loss_map"]
        worst = {"max_abs": 0.0, "max_rel_floor1": 0.0, "max_rel_pixel": 0.0}
        per_map = {}
        # rays on which the reference itself is discontinuous (GGX_specular flips the normal by sign(N.V), models/relight_utils.py:
        # 30-31: a composited normal perpendicular to the view direction within fp32 noise takes either branch and the specular
        # term jumps by percents -- tests/helpers.py ggx_flip_rays, profiles/r05_outlier_diag.json): counted and listed, not compared
        from tests.helpers import ggx_flip_rays
        flip = ggx_flip_rays(ref["normal_map"], r_cpu)
        keep = ~flip
        for k in maps:
            kk_ = keep if k == "rgb_with_brdf_map" else slice(None)
            m = parity_metrics(ret[k].detach().cpu()[::stride][: a.cpu_rays][kk_], ref[k][kk_])
            per_map[k] = {kk: float(f"{vv:.3e}") for kk, vv in m.items()}
            for kk in worst:
                worst[kk] = max(worst[kk], m[kk])
        # how many rays carry the worst figure: a decision flip (one secondary sample on the other side of an occupancy-cell
        # boundary because the surface point differs in its last bits) shows up as ONE ray far above the rest
        d_all = (ret["rgb_with_brdf_map"].detach().cpu()[::stride][: a.cpu_rays] - ref["rgb_with_brdf_map"]).abs().max(dim=-1).values
        # ... and is examined on the ORACLE alone: a ray whose reference colour moves by more than 1e-5 when the oracle's own depth
        # moves by one or two ulps is a ray on which the reference is discontinuous (tests/helpers.py depth_discontinuity_rays) --
        # listed with the oracle's own jump, not compared.  Both exemption classes together are capped at max(2, rays / 500).
        from tests.helpers import depth_discontinuity_rays
        cand = torch.nonzero(keep & (d_all > 1e-5)).reshape(-1).tolist()
        jumps = depth_discontinuity_rays(O, sc, ref, r_cpu, l_cpu, cand, a.second_samples) if cand else {}
        for i_ in jumps:
            keep[i_] = False
        if jumps:
            m = parity_metrics(ret["rgb_with_brdf_map"].detach().cpu()[::stride][: a.cpu_rays][keep], ref["rgb_with_brdf_map"][keep])
            per_map["rgb_with_brdf_map"] = {kk: float(f"{vv:.3e}") for kk, vv in m.items()}
            worst = {kk: max(max(v_[kk] for v_ in per_map.values()), 0.0) for kk in worst}
        d_b = d_all[keep]
        top2 = torch.topk(d_b, min(2, d_b.numel())).values.tolist()
        parity = {"ok": worst["max_rel_floor1"] < 1e-4 and worst["max_rel_pixel"] < 1e-4 and int(flip.sum()) + len(jumps) <= max(2, flip.numel() // 500),
                  "tolerance": 1e-4,
                  "ggx_normal_flip_rays": {"count": int(flip.sum()), "criterion": "|N.V| < 1e-5 for the oracle's composited normal",
                                           "rgb_with_brdf_abs_diff_there": [float(f"{v:.3e}") for v in d_all[flip].tolist()[:8]]},
                  "depth_discontinuity_rays": {"count": len(jumps), "criterion": "the ORACLE's rgb_with_brdf on its own maps varies by > 1e-5 when its own depth moves by -2 ... +2 ulps "
                                               "(a secondary sample on an occupancy / box boundary, models/relight_utils.py:433, :683-695); candidates: rays > 1e-5 off",
                                               "rays": {str(k_): {"oracle_own_jump": float(f"{v_:.3e}"), "hip_abs_diff": float(f"{float(d_all[k_]):.3e}")} for k_, v_ in jumps.items()}},
                  "rgb_with_brdf_rays_over_1e-5": int((d_b > 1e-5).sum()), "rgb_with_brdf_second_worst_abs": float(f"{top2[-1]:.3e}"),
                  "metric": "BOTH asserted < 1e-4: max |hip - oracle| / max(|oracle|, 1) per map (maps live in [0,1], unit normals, "
                            "depth ~4) and max_rel = the true per-pixel relative error ||d|| / ||ref|| over pixels with ||ref|| > 1e-2 "
                            "(north_star: 1e-4 relative on rendered RGB / normals)",
                  "max_abs": float(f"{worst['max_abs']:.3e}"), "max_rel_floor1": float(f"{worst['max_rel_floor1']:.3e}"),
                  "max_rel": float(f"{worst['max_rel_pixel']:.3e}"), "rays_compared": int(r_cpu.shape[0]),
                  "maps": maps, "per_map": per_map,
                  "excluded": "albedo/roughness smoothness losses (depend on the device-side jitter draw)"}

    # ---- informative second scene: a sharp surface (what a trained checkpoint looks like) -------------------------
    sharp = None
    if world == 1 and not a.no_sharp_scene:
        try:
            sharp = sharp_scene_line(a, device, args)
        except Exception as e:                       # never let the informative line break the headline
            sharp = {"error": f"{type(e).__name__}: {e}"}

    # the reference marches every ray to its last sample; the product stops a ray once its transmittance is below march_t_stop
    # (1e-6: bounded, tested deviation).  The same rotation with march_t_stop = 0, one stream, for the record (single process
    # only: the changed constant re-captures the graph, which must not happen next to live RCCL threads; LAST measurement of the
    # run: it enlarges the record-capacity hints, which would slow the per-kernel passes above by a few per cent).
    exact_march = None
    if not use_dist and not a.no_graph and not a.no_exact_pass and float(model.march_t_stop) != 0.0:
        old_stop = float(model.march_t_stop)
        model.march_t_stop = 0.0
        try:
            state["lanes"] = 1
            for b in batches:                                # re-capture + capacities of every pose (rays march further now)
                graphed[a.decoder][0](rays=b, clone_outputs=False)
            n3 = max(len(batches), a.steps // 2)
            state["b"] = 0
            el3, _ = timed(1, n3)
            exact_march = {"march_t_stop": 0.0, "in_flight": 1, "steps": n3, "value": round(n_gpus * B * n3 / el3, 1),
                           "ms_per_step": round(1e3 * el3 / n3, 4)}
        finally:
            model.march_t_stop = old_stop
            state["lanes"] = lanes

    # ---- the other three workloads of BASELINE.json in the same run, at reduced repetition (one image, one relit view, 60
    #      training steps): value, parity against the oracle, dominant-kernel roofline and CPU baseline of each -- the full lines
    #      come from `--workload image|relight|train` (profiles/r04_*_bench.json)
    side = None
    if world == 1 and not a.no_side_workloads:
        side = {}
        for wl, fn, kw in (("image", bench_image, dict(steps=1, warmup=0)), ("relight", bench_relight, dict(steps=1, warmup=0)),
                           ("train", bench_train, dict(steps=60, warmup=5))):
            b = argparse.Namespace(**dict(vars(a), workload=wl, **kw))
            t_side = time.perf_counter()
            try:
                torch.cuda.empty_cache()
                side[wl] = side_summary(fn(b, embed=True), time.perf_counter() - t_side)
            except (Exception, SystemExit) as e:                         # never let a side line break the headline
                side[wl] = {"error": f"{type(e).__name__}: {e}"}

    # ... and the port once more at the thread count that ratio was calibrated with: the
    # reference-equivalent figure is then port x ratio at EQUAL threads, with no cross-thread-count extrapolation.  LAST
    # measurement of the process: resizing PyTorch's OpenMP team and back leaves every later tiny CPU op of this process with a
    # team start-up (measured: the embedded training workload went from 4.3 to 77 ms per step behind it).
    if cpu is not None and cpu.get("kind") == "port" and cal_ctx is not None:
        sc, O = cal_ctx
        try:
            cal = json.load(open(os.path.join(ROOT, "profiles", "port_over_reference.json")))
            cal_thr, n_thr = int(cal["threads"]), torch.get_num_threads()
            sub = max(1, B // 512)
            r_s, l_s = rays.cpu()[::sub], lidx.cpu()[::sub]
            torch.set_num_threads(cal_thr)
            try:
                _, med_c, ts_c = timed_cpu(lambda: O.renderer_train(sc, r_s, l_s, n_samples=a.samples, second_n_sample=a.second_samples), 1, 2)
            finally:
                torch.set_num_threads(n_thr)
            cpu["vs_reference"]["at_calibration_threads"] = {
                "threads": cal_thr, "port_rays_per_s": round(r_s.shape[0] / med_c, 2),
                "reference_equivalent_rays_per_s": round(r_s.shape[0] / med_c * cal["port_over_reference"], 2),
                "sample": f"every {sub}th ray of the batch ({r_s.shape[0]} rays), 1 warm-up + {len(ts_c)} timed calls, median"}
        except Exception as e:                   # the calibration file is optional
            cpu["vs_reference"]["at_calibration_threads"] = {"error": f"{type(e).__name__}: {e}"}
        ops.MLP_IMPL = a.decoder

    value = n_gpus * B * a.steps / elapsed
    rccl = None
    if use_dist and a.backend == "nccl":
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = "unknown"
    out = {
        "metric": "primary+secondary rays/sec at 4096 rays x 512 samples",
        "value": round(value, 1), "unit": "rays/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32 io and field; decoders split-bf16 x3 (hi/lo operands, 3 MFMA products), fp32 accumulate" +
                  ("; indirect light (secondary-ray records): fp16 shadow taps + fp16 single-product decoder, fp32 accumulate"
                   if ops.secondary_mlp_impl() == "f16" else "")) if a.decoder == "bf16x3" else "f32 (exact fp32 MFMA decoders)",
        "data": "synthetic",
        "config": {"workload": f"C2+C3: Renderer_TensoIR_train, {B} rays x {a.samples} samples per GPU, VM grid "
                               f"{a.grid}^3 (16/48 comps), occupancy 128^3, 3 decoders 150-128-128, secondary "
                               f"{D} dirs x {a.second_samples} samples on {M} surface points, SG env light",
                   "value_is": f"whole-job rays/s over exactly --steps = {a.steps} steps (after {SETTLE_STEPS} untimed clock-settle steps "
                               f"and --warmup = {a.warmup}): HIP-graph replay, "
                               f"{1 if a.no_graph else lanes} batch(es) in flight per GPU, rays resident in HBM, the timed region "
                               f"rotates through {len(batches)} distinct camera poses (batch i = pose i mod {len(batches)}); "
                               "`sustained` is the same over >= 200 steps, `protocol_2_1` the BASELINE.md 2.1 figure "
                               "(eager boundary call incl. H2D, hipEvent, median of 50)",
                   "rays_per_gpu": B, "samples": a.samples, "grid": a.grid, "light_dirs": D,
                   "second_samples": a.second_samples, "surface_points": M, "ray_batches": len(batches),
                   "sharding": f"dp{n_gpus} over rays, all-gather of {tdist.RECORD * 4} B/ray records",
                   "launch": "eager" if a.no_graph else "hip-graph replay (one graph per step)",
                   "in_flight": 1 if a.no_graph else lanes,
                   "in_flight_note": "independent batches in flight per GPU: batch i replays lane (i mod in_flight)'s captured graph "
                                     "on that lane's HIP stream; every batch is a full step, lanes checked bit-identical; "
                                     "per-kernel rooflines are measured one kernel at a time (eager pass on one stream)",
                   "march_t_stop": float(model.march_t_stop)},
        "sustained": sustained,
        "single_stream": single,
        "protocol_2_1": boundary,
        "world_size": (dist.get_world_size() if use_dist else 1), "device_count": torch.cuda.device_count(),
        "backend": (a.backend if use_dist else None), "rccl_version": rccl, "per_rank_ms_per_step": per_rank_ms,
        "decoder": {"mode": a.decoder, "note": "bf16x3 = x=hi+lo bf16 split, 3 MFMA products, fp32 accumulate; parity-tested at 1e-4"
                    if a.decoder == "bf16x3" else "exact fp32 MFMA"},
        "exact_fp32_decoders": exact,
        "march_to_the_end": exact_march,
        "roofline": roofline,
        "roofline_vm_sample": roofline_vm,
        "roofline_app_gather": roofline_app,
        "roofline_decoder": roofline_dec,
        "pmc": pmc_meta,
        "library": library_info(),
        "settle_steps": SETTLE_STEPS,
        "precision_policy": {"indirect": model.indirect_precision(), "full": forced_lines.get("full"), "hp": forced_lines.get("hp"),
                             "trained_300": trained_300_verdict(),
                             "secondary_gather": ops.secondary_app_impl() or "fp32", "secondary_decoder": ops.secondary_mlp_impl() or a.decoder,
                             "fused_gather_decoder": bool(ops.fused_indirect()), "limits": dict(ops.INDIRECT_PROBE),
                             "note": "auto (default): radiance of the secondary-ray records (indirect light) from fp16 shadow planes + single-product "
                                     "fp16 decoder, fp32 accumulation, ONLY while this field / decoder version passes the range guard and the "
                                     "self-check probe against the full-precision kernels (`indirect.probe`; otherwise `mode` = full); every launch "
                                     "whose output is composited directly stays fp32 / split-bf16 x3 (DESIGN 4.1, profiles/r05_precision_*.json); "
                                     "TENSOIR_INDIRECT_PRECISION=full|f16 forces either"},
        "boundary_call": boundary,
        "cpu_baseline": cpu,
        "parity": parity,
        "sharp_surface_scene": sharp,
        "workloads": side,
        "gpu_kernel_ms_per_step": round(gpu_ms, 4),
        "event_bracket_overhead_ms": round(ev_over, 5),
        "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:8]],
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    if a.breakdown:
        with open(a.breakdown, "w") as fh:
            json.dump({"rows": rows, "elapsed_s": elapsed, "steps": a.steps}, fh, indent=1)
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"[bench] PARITY FAILURE vs the oracle at the headline size: {parity}")


if __name__ == "__main__":
    main()

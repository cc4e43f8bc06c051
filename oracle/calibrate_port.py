"""How fast is the oracle (the CPU *port* bench.py times as `cpu_baseline`) relative to the reference's own CPU path?

TEST INFRASTRUCTURE ONLY (see oracle/tensoir_oracle.py).  BASELINE.md 2.1 asks for the reference's CPU path timed on the
bench box's host cores; the reference checkout cannot travel to the GPU box, so bench.py times the oracle there
(`cpu_baseline.kind = "port"`).  This script measures, where the checkout exists (the build container), both
implementations on the SAME inputs -- the C2+C3 bench scene (300^3 field, reference-built 128^3 mask, 128 light
directions x 96 secondary samples), a strided subsample of the 4096-ray batch -- and writes

    profiles/port_over_reference.json = {"port_over_reference": t_port / t_reference, ...}

so that a bench line can state a reference-equivalent figure: cpu_baseline.value x port_over_reference.  It also checks
that the two agree on every map (the oracle is pinned to the reference by tests/golden/; this is the full-size
confirmation).  Usage: python oracle/calibrate_port.py [--rays 256] [--calls 2]"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle.ref_on_gpu import MAPS, metrics, reference_model, with_mask  # noqa: E402


def measure(n_rays=256, calls=2, grid=300, samples=512, second=96, envh=8, envw=16):
    from oracle import tensoir_oracle as O
    from tensoir_amd import synth
    from tests.helpers import scene_from_checkpoint
    ref = ref_loader.load()
    ckpt = synth.make_checkpoint(grid=(grid,) * 3, seed=20211202)
    rays_all = synth.make_rays(64, 64)
    stride = max(1, rays_all.shape[0] // n_rays)
    rays = rays_all[::stride][:n_rays].contiguous()
    lidx = torch.zeros(rays.shape[0], 1, dtype=torch.int32)
    args = types.SimpleNamespace(second_nSample=second, second_near=0.05, second_far=1.5)
    rmodel = reference_model(ref, ckpt, "cpu", envh, envw)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        rmodel.updateAlphaMask((128, 128, 128))
    ck = with_mask(ckpt, rmodel)
    sc = scene_from_checkpoint(ck, envh, envw)

    def run_ref():
        return ref.renderer.Renderer_TensoIR_train(rays, None, lidx, rmodel, N_samples=samples, white_bg=True, is_train=False,
                                                   is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device="cpu",
                                                   args=args)

    def run_port():
        return O.renderer_train(sc, rays, lidx, n_samples=samples, second_n_sample=second, chunk_size=160000)
    t = {"reference": [], "port": []}
    out = {}
    with torch.no_grad():
        out["reference"], out["port"] = run_ref(), run_port()          # warm-ups (allocator, thread pool, page cache)
        for _ in range(max(1, calls)):                                  # interleaved A/B: clock and cache state shared
            for name, fn in (("reference", run_ref), ("port", run_port)):
                t0 = time.perf_counter()
                out[name] = fn()
                t[name].append(time.perf_counter() - t0)
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    worst = max(metrics(out["port"][k], out["reference"][k])["max_rel_floor1"] for k in MAPS)
    return {
        "port_over_reference": round(med["port"] / med["reference"], 4),
        "reference_rays_per_s": round(rays.shape[0] / med["reference"], 2), "port_rays_per_s": round(rays.shape[0] / med["port"], 2),
        "reference_s": [round(x, 3) for x in t["reference"]], "port_s": [round(x, 3) for x in t["port"]],
        "port_vs_reference_max_rel_floor1": worst,
        "sample": (f"the full 4096-ray bench batch ({rays.shape[0]} rays" if stride == 1 else
                   f"every {stride}th ray of the 4096-ray bench batch ({rays.shape[0]} rays") + f" x {samples} samples, {envh * envw} dirs x {second}), "
                  f"VM grid {grid}^3, reference-built 128^3 mask; one warm-up each, then interleaved A/B timed calls, median",
        "threads": torch.get_num_threads(), "host_nproc": os.cpu_count(), "torch": torch.__version__,
        "reference": "renderer.py:57-127 Renderer_TensoIR_train on models/tensoRF_rotated_lights.py (imported from the checkout, CPU)",
        "port": "oracle/tensoir_oracle.py renderer_train",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "port_over_reference.json"))
    a = ap.parse_args()
    if not ref_loader.available():
        raise SystemExit(f"no reference checkout at {ref_loader.REF_ROOT}")
    rep = measure(a.rays, a.calls, grid=a.grid)
    rep["gpu_box_measurement_r03"] = ("profiles/r03_ref_on_gpu.json + r03_v9_bench.json (staged checkout on the MI355X box, 128 threads): reference "
                                      "361-373 rays/s on the full batch, port 222.65 rays/s on a quarter batch")
    with open(a.out, "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()

"""Evidence for the indirect-light precision policy: the TRAINED checkpoint and the adversarial scaling
sweep of tests/precision_cases.py with every number recorded (nothing asserted -- the tests do that):
  gpurun_out/r05_precision_trained.json, gpurun_out/r05_precision_sweep.json   (copied to profiles/ by hand)
Usage (GPU box): python tools/precision_sweep.py [--grid 128] [--iters 450] [--skip-sweep] [--tag long]
(--tag X writes r05_precision_trained_X.json: e.g. the same schedule trained three times as long, sweep skipped)"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def slim(res):
    return {k: v for k, v in res.items() if not torch.is_tensor(v)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--iters", type=int, default=450)
    ap.add_argument("--oracle-rays", type=int, default=128)
    ap.add_argument("--skip-sweep", action="store_true")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    from tests import precision_cases as P
    from tensoir_amd import _lib, ops
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    stamp = {"library_source_hash": open(os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so.srchash")).read().strip(),
             "limits": dict(ops.INDIRECT_PROBE), "device": torch.cuda.get_device_name(0)}
    t0 = time.time()
    r = P.trained(a.iters)
    during = r.model.indirect_precision()
    res, rep = P.trained_case(r)
    trained = {**stamp, "iterations": a.iters, "grids": r.grids, "psnr_last10": float(-10 * torch.log10(torch.tensor(r.losses[-10:]).mean())),
               "policy_during_training": during, "policy": slim(res), "oracle": rep, "seconds": round(time.time() - t0, 1)}
    # magnitudes of what the fp16 kernels read on this checkpoint
    m = r.model
    trained["magnitudes"] = {"app_plane_absmax": [float(p.abs().max()) for p in m.app_plane], "app_line_absmax": [float(p.abs().max()) for p in m.app_line],
                             "light_line_absmax": float(m.light_line.weight.abs().max()), "basis_absmax": float(m.basis_mat.weight.abs().max()),
                             "radiance_decoder_weight_rms": [float(m.renderModule.mlp[i].weight.pow(2).mean().sqrt()) for i in (0, 2, 4)]}
    trained["radiance_decoder_weight_absmax"] = [float(m.renderModule.mlp[i].weight.abs().max()) for i in (0, 2, 4)]
    with open(os.path.join(out, f"r05_precision_trained{'_' + a.tag if a.tag else ''}.json"), "w") as fh:
        json.dump(trained, fh, indent=1, default=str)
    print("trained:", json.dumps({k: trained[k] for k in ("psnr_last10", "policy_during_training", "policy")}, default=str)[:1500], flush=True)
    print("trained oracle:", json.dumps({k: v for k, v in rep.items()}, default=str)[:1500], flush=True)
    if a.skip_sweep:
        return
    sweep = {**stamp, "grid": a.grid, "cases": {}}
    for cfg in P.SWEEP:
        t0 = time.time()
        res, rep = P.sweep_case(cfg, a.grid, a.oracle_rays if cfg.get("oracle", True) else 0)
        sweep["cases"][cfg["name"]] = {"scales": {k: v for k, v in cfg.items() if k != "name"}, "policy": slim(res), "oracle": rep,
                                       "seconds": round(time.time() - t0, 1)}
        worst = None if rep is None else max(v["max_rel_floor1"] for v in rep.values() if isinstance(v, dict))
        print(cfg["name"], "| f16 vs full", res["f16_vs_full"], "| auto:", res["decision"]["mode"], res["decision"]["why"], res["decision"]["probe"],
              "| auto vs full", res["auto_vs_full_max_abs"], "| worst map vs oracle", worst, flush=True)
        with open(os.path.join(out, "r05_precision_sweep.json"), "w") as fh:
            json.dump(sweep, fh, indent=1, default=str)


if __name__ == "__main__":
    main()

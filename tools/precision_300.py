"""The indirect-light precision policy on a checkpoint TRAINED TO 300^3 through the product API: the
analytic dataset, the script's schedule compressed (two mask updates, shrink, four up-samplings 128^3 -> 300^3, relighting
losses from the first mask update on), >= 2000 iterations.  Records what `auto` decides on that checkpoint, every map against
the oracle under the decided mode, and WHERE the f16 kernels' deviation comes from: the gather / decoder precision matrix
(fp16 shadow taps x fp32 taps) x (fp16 single-product decoder x split-bf16 x3 decoder), each against the full kernels.
Reference stage: models/relight_utils.py:777-834; schedule: train_tensoIR.py:237-461.
Usage (GPU box): python tools/precision_300.py [--iters 2400] [--rays 4096] [--oracle-rays 256] [--out gpurun_out/r06_precision_trained_300.json]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def schedule(n_iters):
    from tests.precision_cases import schedule_300
    return schedule_300(n_iters)


def train(n_iters, batch, views, res):
    from tests.precision_cases import trained_300
    return trained_300(n_iters, batch, views, res)


def stats(a, b):
    d = (a - b).double()
    return {"max_abs": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "mean_signed": float(d.mean()),
            "rays_over_1e-5": int((d.abs().amax(-1) > 1e-5).sum()), "rays_over_2.5e-5": int((d.abs().amax(-1) > 2.5e-5).sum()),
            "finite": bool(torch.isfinite(a).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2400)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--res", type=int, default=160)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--oracle-rays", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_precision_trained_300.json"))
    a = ap.parse_args()
    from tests import precision_cases as P
    from tests.helpers import scene_from_model
    from tensoir_amd import ops
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    rep = {"library_source_hash": open(os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so.srchash")).read().strip(),
           "limits": dict(ops.INDIRECT_PROBE), "device": torch.cuda.get_device_name(0), "iterations": a.iters, "batch": a.batch,
           "dataset": f"synthetic:views={a.views},res={a.res}", "schedule": schedule(a.iters)}
    t0 = time.time()
    r = train(a.iters, a.batch, a.views, a.res)
    torch.cuda.synchronize()
    m = r.model
    rep["train_seconds"] = round(time.time() - t0, 1)
    rep["grids"] = r.grids
    rep["psnr_last10"] = float(-10 * torch.log10(torch.tensor(r.losses[-10:]).mean()))
    rep["policy_during_training"] = m.indirect_precision()
    rep["magnitudes"] = {"app_plane_absmax": [float(p.abs().max()) for p in m.app_plane], "app_line_absmax": [float(p.abs().max()) for p in m.app_line],
                         "app_plane_rms": [float(p.pow(2).mean().sqrt()) for p in m.app_plane],
                         "light_line_absmax": float(m.light_line.weight.abs().max()), "basis_absmax": float(m.basis_mat.weight.abs().max()),
                         "radiance_decoder_weight_rms": [float(m.renderModule.mlp[i].weight.pow(2).mean().sqrt()) for i in (0, 2, 4)],
                         "radiance_decoder_weight_absmax": [float(m.renderModule.mlp[i].weight.abs().max()) for i in (0, 2, 4)]}
    print("trained:", json.dumps({k: rep[k] for k in ("train_seconds", "grids", "psnr_last10", "policy_during_training")}, default=str), flush=True)

    n = min(a.rays, r.rays_f.shape[0])
    rays = r.rays_f[:n].cuda()
    lidx = r.lidx_f[:n].cuda().to(torch.int32).reshape(-1, 1)
    S = int(m.nSamples)
    noise = torch.randn(n, S, 3, generator=torch.Generator().manual_seed(5))
    out, res = P.three_policies(m, rays, lidx, noise, S)
    rep["n_samples"] = S
    rep["policy"] = {k: v for k, v in res.items() if not torch.is_tensor(v)}
    rep["policy"]["f16_vs_full"].update(stats(res["f16"], res["full"]))
    # feature magnitudes of the secondary records are what the fp16 products see: the radiance features of the surface points
    with torch.no_grad():
        pts = (torch.rand(20000, 3, device="cuda") - 0.5) * 1.6            # normalised coordinates
        feat = m.compute_appfeature(pts, torch.zeros(20000, dtype=torch.int32, device="cuda"))
        rep["magnitudes"]["radiance_feature_absmax"] = float(feat.abs().max())
        rep["magnitudes"]["radiance_feature_rms"] = float(feat.pow(2).mean().sqrt())
    # the precision matrix: which half of the f16 route carries the deviation (unfused launches; the fused kernel = h16 + f16)
    matrix = {}
    for name, mlp, app in (("h16 taps + f16 decoder (the fused kernel's arithmetic)", "f16", "h16"),
                           ("h16 taps + bf16x3 decoder", "bf16x3", "h16"),
                           ("fp32 taps + f16 decoder", "f16", None)):
        try:
            with P.policy(False, mlp, app):
                _, got = P.render(m, rays, lidx, noise, S)
            matrix[name] = stats(got, res["full"])
        except Exception as e:
            matrix[name] = {"error": f"{type(e).__name__}: {e}"}
        print(name, matrix[name], flush=True)
    rep["precision_matrix_vs_full"] = matrix
    to_cpu = lambda v: v.detach().cpu() if torch.is_tensor(v) else v
    ckpt = {"kwargs": {k: to_cpu(v) for k, v in m.get_kwargs().items()}, "state_dict": {k: to_cpu(v) for k, v in m.state_dict().items()}}
    sc = scene_from_model(ckpt, m, 8, 16)
    step = max(1, n // a.oracle_rays)
    t0 = time.time()
    rep["oracle"] = P.oracle_compare(sc, out, res["auto"], rays, lidx, noise, S, slice(0, n, step), fp64_floor=True)
    rep["oracle_seconds"] = round(time.time() - t0, 1)
    with open(a.out, "w") as fh:
        json.dump(rep, fh, indent=1, default=str)
    print("policy:", json.dumps(rep["policy"], default=str)[:1200], flush=True)
    print("oracle:", json.dumps({k: v for k, v in rep["oracle"].items() if k != "oracle_fp32_vs_fp64"}, default=str)[:2500], flush=True)


if __name__ == "__main__":
    main()

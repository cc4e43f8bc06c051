"""Which ray of the headline batch carries the 2e-2 rgb_with_brdf_map difference, and why?  Renders the bench batch with the
split-bf16 and the exact fp32 decoders, picks the rays whose rgb_with_brdf_map differs by > 1e-4 between the two (the oracle agrees
with the exact-decoder render), and dumps their maps, N.V, per-direction visibility / cosine and the oracle's values for them.
Usage (GPU box): python tools/r05_outlier_diag.py -> gpurun_out/r05_outlier_diag.json"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from oracle import tensoir_oracle as O          # checker only
    from tests.helpers import scene_from_model
    from tensoir_amd import ops, relight
    sys.argv = ["bench.py"]
    a = bench.parse()
    device = torch.device("cuda", 0)
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
    noise = torch.randn(rays.shape[0], a.samples, 3, generator=torch.Generator().manual_seed(7))
    res = {}
    for impl in ("bf16x3", "mfma"):
        ops.MLP_IMPL = impl
        with torch.no_grad():
            out, maps = model(rays, lidx, N_samples=a.samples, _brdf_jitter_dense=noise, _return_maps=True)
            brdf, aux = relight.shade_from_maps(model, maps, rays, lidx, "fixed_envirmap", args, acc_thres=0.5, return_aux=True)
        res[impl] = {"maps": maps.cpu(), "brdf": brdf.cpu(), "vis": aux["vis"].cpu(), "ind": aux["indirect"].cpu(), "surf": aux["surf"].cpu()}
    ops.MLP_IMPL = "bf16x3"
    d = (res["bf16x3"]["brdf"] - res["mfma"]["brdf"]).abs().max(-1).values
    bad = torch.nonzero(d > 1e-4).view(-1).tolist()[:8]
    print("rays differing > 1e-4 between decoder modes:", bad, d[bad].tolist())
    sel = torch.tensor(bad + [0, 1000], dtype=torch.long)
    sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
    with torch.no_grad():
        ref = O.renderer_train(sc, rays.cpu()[sel], lidx.cpu()[sel], n_samples=a.samples, brdf_jitter=noise[sel], second_n_sample=a.second_samples)
    rep = {"rays": sel.tolist(), "oracle_brdf": ref["rgb_with_brdf_map"].tolist(), "oracle_normal": ref["normal_map"].tolist(),
           "oracle_depth": ref["depth_map"].tolist()}
    for impl in ("bf16x3", "mfma"):
        m = res[impl]["maps"][sel]
        n = m[:, 4:7]
        v = -rays.cpu()[sel][:, 3:]
        v = v / v.norm(dim=-1, keepdim=True)
        nn = n / n.norm(dim=-1, keepdim=True).clamp(min=1e-6)
        rep[impl] = {"brdf": res[impl]["brdf"][sel].tolist(), "normal": n.tolist(), "NoV": (nn * v).sum(-1).tolist(), "depth": m[:, 3].tolist(),
                     "albedo": m[:, 7:10].tolist(), "rough": m[:, 10].tolist(), "acc": m[:, 14].tolist(),
                     "brdf_err_vs_oracle": (res[impl]["brdf"][sel] - ref["rgb_with_brdf_map"]).abs().max(-1).values.tolist()}
    vb, vm = res["bf16x3"]["vis"][sel], res["mfma"]["vis"][sel]
    rep["vis_max_diff_between_modes"] = (vb - vm).abs().max(-1).values.tolist()
    rep["indirect_max_diff_between_modes"] = (res["bf16x3"]["ind"][sel] - res["mfma"]["ind"][sel]).abs().reshape(len(sel), -1).max(-1).values.tolist()
    rep["surf_max_diff_between_modes"] = (res["bf16x3"]["surf"][sel] - res["mfma"]["surf"][sel]).abs().max(-1).values.tolist()
    nv_o = (ref["normal_map"] / ref["normal_map"].norm(dim=-1, keepdim=True).clamp(min=1e-6) * (-rays.cpu()[sel][:, 3:] / rays.cpu()[sel][:, 3:].norm(dim=-1, keepdim=True))).sum(-1)
    rep["oracle_NoV"] = nv_o.tolist()
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "r05_outlier_diag.json"), "w"), indent=1)
    print(json.dumps({k: rep[k] for k in ("rays", "oracle_NoV", "vis_max_diff_between_modes", "indirect_max_diff_between_modes", "surf_max_diff_between_modes")}))
    for impl in ("bf16x3", "mfma"):
        print(impl, "NoV", rep[impl]["NoV"], "err", rep[impl]["brdf_err_vs_oracle"])


if __name__ == "__main__":
    main()

"""Which rays carry the largest rgb_with_brdf_map error at the headline size, and under which precision policy?
(full 4096-ray batch against the oracle; both policies).  Usage (GPU box): python tools/parity_outliers.py"""
import json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    from oracle import tensoir_oracle as O
    from tests.helpers import scene_from_model
    from tensoir_amd import Renderer_TensoIR_train, ops
    sys.argv = ["bench.py"]
    a = bench.parse()
    device = torch.device("cuda", 0)
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
    sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
    with torch.no_grad():
        ref = O.renderer_train(sc, rays.cpu(), lidx.cpu(), n_samples=a.samples, second_n_sample=a.second_samples)
    kw = dict(N_samples=a.samples, white_bg=True, is_train=False, is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args)
    out = {}
    for name, (m, g) in {"full": (None, None), "f16dec": ("f16", None), "policy": ("f16", "h16")}.items():
        ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = m, g
        for stop in (1e-6, 0.0):
            model.march_t_stop = stop
            with torch.no_grad():
                ret = Renderer_TensoIR_train(rays, None, lidx, model, **kw)
            got = ret["rgb_with_brdf_map"].cpu()
            d = (got - ref["rgb_with_brdf_map"]).abs().max(dim=-1).values
            top = torch.topk(d, 6)
            out[f"{name}/t_stop={stop}"] = {"max_abs": float(d.max()), "n_gt_1e-5": int((d > 1e-5).sum()), "n_gt_3e-6": int((d > 3e-6).sum()),
                                            "top_rays": top.indices.tolist(), "top_err": [float(f"{x:.3e}") for x in top.values],
                                            "top_ref_rgb": [[round(float(v), 5) for v in ref["rgb_with_brdf_map"][i]] for i in top.indices[:3]],
                                            "top_acc": [round(float(ref["acc_map"][i]), 6) for i in top.indices[:3]]}
            print(name, stop, json.dumps(out[f"{name}/t_stop={stop}"]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_outliers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

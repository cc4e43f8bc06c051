"""Precision policy of the secondary-record radiance decoder: the bench step with that ONE launch in
split-bf16 x3 (default), single-product fp16 (tir_mlp_fwd_auxtab_f16) and single-product bf16 (tir_mlp_fwd_bf16), on the
headline blob and on the sharp-surface scene.  Per mode: graph-replay step time, the launch's own duration (events), the
decoder-output error against the x3 kernel on the same records, rgb_with_brdf_map against the x3 render (all 4096 rays) and
against the oracle (every 8th ray).  Usage (GPU box): python tools/prec_probe.py [out.json]"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prec_probe.json")
    from oracle import tensoir_oracle as O          # checker only
    from tests.helpers import parity_metrics, scene_from_model
    from tensoir_amd import Renderer_TensoIR_train, ops
    from tensoir_amd.graph import GraphedRenderer
    sys.argv = ["bench.py"]
    a = bench.parse()
    device = torch.device("cuda", 0)
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    res = {}
    for scene, blob in (("blob", {}), ("sharp", dict(blob_sigma=0.2, blob_gain=2000.0))):
        ckpt, model, rays, lidx = bench.build_scene(a, device, 0, **blob)
        model.march_t_stop = 1e-6
        B = rays.shape[0]
        sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
        stride = 8
        with torch.no_grad():
            ref = O.renderer_train(sc, rays.cpu()[::stride], lidx.cpu()[::stride], n_samples=a.samples, second_n_sample=a.second_samples)
        base = None
        per = {}
        for mode, app in ((None, None), ("f16", None), ("f16", "h16"), ("bf16", None)):
            ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = mode, app
            with torch.no_grad():
                for _ in range(3):
                    ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False,
                                                 is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args)
                ops.TIMING = []
                for _ in range(5):
                    ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False,
                                                 is_relight=True, sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args)
                torch.cuda.synchronize()
                agg = {}
                for name, e0, e1 in ops.TIMING:
                    agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1) / 5
                ops.TIMING = None
            gr = GraphedRenderer(model, B, N_samples=a.samples, args=args, device=device)
            gr.rays.copy_(rays)
            gr.lidx.copy_(lidx)
            gr(clone_outputs=False)
            for _ in range(100):
                gr(clone_outputs=False, defer_check=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                gr(clone_outputs=False, defer_check=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 200 * 1e3
            assert gr.validate()
            got = gr(clone_outputs=True)
            rgbb = got["rgb_with_brdf_map"].detach().cpu()
            if base is None:
                base = rgbb
            m_or = parity_metrics(rgbb[::stride], ref["rgb_with_brdf_map"])
            d = (rgbb - base).abs()
            per[f"decoder {mode or 'bf16x3'} / gather {app or 'fp32'}"] = {
                "step_ms_graph_1lane": round(ms, 4),
                "secondary_decoder_ms": {k: round(v, 4) for k, v in agg.items() if "mlp_fwd" in k and "multi" not in k},
                "secondary_gather_ms": {k: round(v, 4) for k, v in agg.items() if k in ("tir_vm_app_fwd", "tir_vm_app_fwd_h16")},
                "rgb_with_brdf_vs_x3_max_abs": float(d.max()), "rgb_with_brdf_vs_x3_rms": float(d.pow(2).mean().sqrt()),
                "rgb_with_brdf_vs_oracle": {k: float(f"{v:.3e}") for k, v in m_or.items()},
                "rgb_map_vs_oracle": {k: float(f"{v:.3e}") for k, v in parity_metrics(got["rgb_map"].cpu()[::stride], ref["rgb_map"]).items()},
            }
            del gr
        ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = "f16", "h16"
        # decoder outputs on one set of rows: f16 / bf16 vs the x3 kernel and vs fp64
        res[scene] = per
        print(scene, json.dumps(per, indent=1), flush=True)
        del model
    # the decoder launch alone on random rows: error of every mode against fp64
    from tensoir_amd import synth
    import tensoir_amd
    ck = synth.make_checkpoint(grid=(64,) * 3, seed=1)
    model = tensoir_amd.model_from_checkpoint(ck, device, envmap_h=8, envmap_w=16)
    pm = model.renderModule.packed()
    n = 1 << 20
    gen = torch.Generator().manual_seed(0)
    feat = torch.zeros(n, ops.FEAT_STRIDE)
    feat[:, :27] = torch.randn(n, 27, generator=gen) * 0.7
    dirs = torch.nn.functional.normalize(torch.randn(128, 3, generator=gen), dim=-1)
    amap = torch.randint(0, 128, (n,), generator=gen, dtype=torch.int32)
    f, dd, am = feat.to(device), dirs.to(device), amap.to(device)
    # the gather alone: fp16-shadow features against the fp32 kernel's on the same points
    fld, fh = model.packed_field(), model.packed_field_half()
    pts = (torch.rand(n, 3, generator=gen) * 1.6 - 0.8).to(device)
    lix = torch.zeros(n, dtype=torch.int32, device=device)
    r32 = ops.vm_app(fld, pts, lix, None, True, False)[0]
    r16 = ops.vm_app_h16(fld, fh, pts, lix)
    torch.cuda.synchronize()
    dg = (r16 - r32)[:, :27].double()
    res["gather_1M_random_points"] = {"max_abs": float(dg.abs().max()), "rms": float(dg.pow(2).mean().sqrt()), "mean_signed": float(dg.mean()),
                                      "feature_rms": float(r32[:, :27].double().pow(2).mean().sqrt()), "pad_columns_zero": bool((r16[:, 27:] == 0).all())}
    print(json.dumps(res["gather_1M_random_points"], indent=1))
    outs = {}
    for mode in ("mfma", "bf16x3", "f16", "bf16"):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            o = ops.mlp(pm, f, dd, am, impl=mode)
        ev0.record()
        for _ in range(10):
            o = ops.mlp(pm, f, dd, am, impl=mode)
        ev1.record()
        torch.cuda.synchronize()
        outs[mode] = (o.cpu().double(), ev0.elapsed_time(ev1) / 10)
    w = {k: v.detach().cpu().double() for k, v in model.renderModule.state_dict().items()}
    x = O.mlp_input(feat[:, :27], dirs[amap.long()], 2, 2).double()
    h = torch.relu(x @ w["mlp.0.weight"].T + w["mlp.0.bias"])
    h = torch.relu(h @ w["mlp.2.weight"].T + w["mlp.2.bias"])
    ref64 = torch.sigmoid(h @ w["mlp.4.weight"].T + w["mlp.4.bias"])
    res["decoder_1M_random_rows"] = {k: {"max_abs_vs_fp64": float((v[0] - ref64).abs().max()), "rms_vs_fp64": float((v[0] - ref64).pow(2).mean().sqrt()),
                                         "mean_signed": float((v[0] - ref64).mean()), "ms": round(v[1], 4)} for k, v in outs.items()}
    print(json.dumps(res["decoder_1M_random_rows"], indent=1))
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()

"""Verbose HIP-vs-oracle comparison on the GPU box (diagnostics; the asserting version is tests/test_gpu_*.py).
Writes gpurun_out/gpu_check.txt."""
import os, sys, time, types, traceback
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tensoir_oracle as O
from tests.helpers import golden_checkpoint, scene_from_checkpoint, T
import tensoir_amd
from tensoir_amd import ops, synth, relight

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "gpu_check.txt"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


def err(name, a, b, floor=1.0):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double().cpu()
    if a.shape != b.shape:
        log(f"  {name}: SHAPE MISMATCH {tuple(a.shape)} vs {tuple(b.shape)}"); return
    if a.numel() == 0:
        log(f"  {name}: empty"); return
    d = (a - b).abs()
    rel = d / b.abs().clamp(min=floor)
    log(f"  {name}: max_abs={d.max():.3e} max_rel(floor {floor})={rel.max():.3e} mean_abs={d.mean():.3e} "
        f"n>1e-4={(rel > 1e-4).sum().item()}/{a.numel()} nan={torch.isnan(a).sum().item()}")


def section(name, fn):
    log(f"== {name}")
    t0 = time.time()
    try:
        with torch.no_grad():
            fn()
    except Exception:
        log("  EXCEPTION\n" + traceback.format_exc())
    log(f"  ({time.time() - t0:.2f}s)")


def main():
    dev = "cuda"
    log("device:", torch.cuda.get_device_name(0), "lib check:", tensoir_amd._lib.lib().tir_device_check())
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_scene.npz"))
    ckpt = golden_checkpoint(g)
    eh, ew = [int(x) for x in g["scene/envmap_hw"]]
    sc = scene_from_checkpoint(ckpt, eh, ew)
    model = tensoir_amd.model_from_checkpoint(ckpt, dev, envmap_h=eh, envmap_w=ew)
    model.march_t_stop = 0.0
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)

    def s_density():
        xyz = T(g, "feat/xyz").to(dev)
        err("density vs golden", model.compute_densityfeature(xyz), g["feat/density"])
        _, sig = ops.vm_density(model.packed_field(), xyz, False, True)
        err("sigma vs golden", sig, g["feat/sigma"], 1e-3)
    section("K2 density", s_density)

    def s_app():
        xyz, li = T(g, "feat/xyz").to(dev), T(g, "feat/light_idx").to(dev)
        err("app", model.compute_appfeature(xyz, li), g["feat/app"])
        r, i = model.compute_bothfeature(xyz, li)
        err("both_rad", r, g["feat/both_rad"]); err("both_int", i, g["feat/both_int"])
        err("intrin", model.compute_intrinfeature(xyz), g["feat/intrin"])
    section("K4 app features", s_app)

    def s_app_impl():
        xyz, li = T(g, "feat/xyz").to(dev), T(g, "feat/light_idx").to(dev).view(-1).int()
        with torch.no_grad():
            for impl in ("valu", "mfma"):
                r, i = ops.vm_app(model.packed_field(), xyz, li, None, True, True, impl)
                err(f"both_rad[{impl}]", r[:, :27], g["feat/both_rad"]); err(f"both_int[{impl}]", i[:, :27], g["feat/both_int"])
                log("   pad max", float(r[:, 27:].abs().max()), float(i[:, 27:].abs().max()))
                r1 = ops.vm_app(model.packed_field(), xyz, li, None, True, False, impl)[0]
                i1 = ops.vm_app(model.packed_field(), xyz, None, None, False, True, impl)[1]
                err(f"rad-only[{impl}]", r1[:, :27], g["feat/both_rad"]); err(f"int-only[{impl}]", i1[:, :27], g["feat/both_int"])
    section("K4 impls", s_app_impl)

    def s_mlp():
        xyz, vd = T(g, "feat/xyz").to(dev), T(g, "mlp/viewdirs").to(dev)
        r, i = T(g, "feat/both_rad").to(dev), T(g, "feat/both_int").to(dev)
        for impl in ("valu", "mfma", "bf16x3", "bf16"):
            err(f"rgb[{impl}]", ops.mlp(model.renderModule.packed(), r, vd, None, impl), g["mlp/rgb"])
            err(f"brdf[{impl}]", ops.mlp(model.renderModule_brdf.packed(), i, xyz, None, impl), g["mlp/brdf"])
            err(f"normal[{impl}]", ops.mlp(model.renderModule_normal.packed(), i, xyz, None, impl), g["mlp/normal"])
    section("K5 decoders", s_mlp)

    def s_occ():
        w = T(g, "occ/xyz_world").to(dev)
        hit = model.alphaMask.sample_alpha(w)
        ref = torch.from_numpy(g["occ/alpha"]) > 0
        log("  occupancy mismatches:", int(((hit.cpu() > 0) != ref).sum()), "/", ref.numel())
    section("occupancy", s_occ)

    def s_normals():
        err("derived normals", model.compute_derived_normals(T(g, "normals/xyz").to(dev)), g["normals/derived"])
    section("K6 normals", s_normals)

    rays, lidx = T(g, "rays/rays").to(dev), T(g, "rays/light_idx").to(dev)
    names = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map", "acc_map",
             "normals_diff_map", "normals_orientation_loss_map", "acc_mask", "albedo_smoothness_loss",
             "roughness_smoothness_loss"]

    def s_march():
        f = model.packed_field()
        S = int(g["scene/nSamples"][0])
        w, acc, depth, tend, cnt = ops.march_primary(f, rays, None, S, 0.0)
        _, aux = O.forward_primary(sc, rays.cpu(), lidx.cpu(), return_aux=True)
        err("weight", w, aux.weight, 1e-3)
        err("acc", acc, aux.weight.sum(-1)); err("bg_T", tend, aux.bg.view(-1))
        log("  app_count mismatches:", int((cnt.cpu() != aux.app_mask.sum(-1)).sum()), "total A", int(cnt.sum()),
            "oracle", int(aux.app_mask.sum()))
        w2, acc2, _, _, cnt2 = ops.march_primary(f, rays, None, S, 1e-6)
        err("weight(t_stop=1e-6)", w2, aux.weight, 1e-3)
    section("K1-K3 primary march", s_march)

    def s_forward():
        B, S = rays.shape[0], int(g["scene/nSamples"][0])
        noise = torch.randn(B, S, 3, generator=torch.Generator().manual_seed(7))
        with torch.no_grad():
            out = model(rays, lidx, _brdf_jitter_dense=noise)
        ref = O.forward_primary(sc, rays.cpu(), lidx.cpu(), brdf_jitter=noise)
        for n, a, b in zip(names, out, ref):
            if n == "acc_mask":
                log("  acc_mask mismatches:", int((a.cpu() != b).sum()))
            else:
                err(n + " vs oracle", a, b, 1e-9 if n.endswith("loss") else 1.0)
        for n, a in zip(names, out):
            if n not in ("acc_mask", "albedo_smoothness_loss", "roughness_smoothness_loss"):
                err(n + " vs golden", a, g["fwd/" + n])
        with torch.no_grad():
            o2 = model(rays, lidx, is_relight=False)
        err("norelight rgb vs golden", o2[0], g["fwd_norelight/rgb_map"])
        err("norelight depth vs golden", o2[1], g["fwd_norelight/depth_map"])
    section("primary forward", s_forward)

    def s_secondary():
        p, d, l = T(g, "sec/pts").to(dev), T(g, "sec/dirs").to(dev), T(g, "sec/light_idx").to(dev)
        v, nf = relight.compute_transmittance(model, p, d, nSample=96, vis_near=0.05, vis_far=1.5)
        err("trans vis", v, g["sec/trans_vis"]); err("trans 1-acc", nf, g["sec/trans_nerfactor"])
        v, nf, ind = relight.compute_radiance(model, p, d, l, nSample=96, vis_near=0.05, vis_far=1.5)
        err("rad vis", v, g["sec/rad_vis"]); err("rad indirect", ind, g["sec/rad_indirect"])
    section("K7 secondary", s_secondary)

    def s_env():
        err("light_rgbs", model.get_light_rgbs(T(g, "env/dirs").to(dev), device=dev), g["env/light_rgbs"])
        err("ggx", relight.GGX_specular(T(g, "ggx/normal").to(dev), T(g, "ggx/v").to(dev), T(g, "ggx/l").to(dev),
                                        T(g, "ggx/rough").to(dev), T(g, "ggx/fresnel").to(dev)), g["ggx/spec"], 1e-3)
    section("env + GGX", s_env)

    def s_render():
        B, S = rays.shape[0], int(g["scene/nSamples"][0])
        from tensoir_amd import Renderer_TensoIR_train
        with torch.no_grad():
            ret = Renderer_TensoIR_train(rays, None, lidx, model, args=args, device=dev)
        for k, v in ret.items():
            err(k + " vs golden", v, g["render_fixed/" + k], 1e-9 if k.endswith("loss") else 1.0)
    section("Renderer_TensoIR_train", s_render)

    def s_hdr():
        out = relight.relight_with_envmap(model, T(g, "hdr/surf").to(dev), T(g, "hdr/normal").to(dev),
                                          T(g, "hdr/albedo").to(dev), T(g, "hdr/rough").to(dev),
                                          T(g, "hdr/fresnel").to(dev), T(g, "hdr/rays_d").to(dev),
                                          T(g, "hdr/light_dir").to(dev), T(g, "hdr/light_rgb").to(dev),
                                          T(g, "hdr/light_pdf").to(dev))
        err("hdr relit", out, g["hdr/relit"])
    section("K9 hdr relight", s_hdr)

    # ---- larger scene: HIP vs oracle, with timings ----
    def s_big():
        R = int(os.environ.get("TIR_CHECK_R", "128"))
        ck = synth.make_checkpoint(grid=(R, R, R), seed=1)
        sc2 = scene_from_checkpoint(ck, 8, 16)
        O.update_alpha_mask(sc2, (64, 64, 64))
        vol = sc2.alpha_volume
        ck["alphaMask.shape"] = tuple(vol.shape)
        ck["alphaMask.mask"] = np.packbits(vol.bool().numpy().reshape(-1))
        ck["alphaMask.aabb"] = sc2.alpha_aabb
        m2 = tensoir_amd.model_from_checkpoint(ck, dev, envmap_h=8, envmap_w=16)
        r2 = synth.make_rays(24, 24).to(dev)
        l2 = torch.zeros(r2.shape[0], 1, dtype=torch.int32, device=dev)
        S = 256
        noise = torch.randn(r2.shape[0], S, 3, generator=torch.Generator().manual_seed(3))
        a2 = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
        from tensoir_amd import Renderer_TensoIR_train
        import tensoir_amd.field_model as FM
        orig = FM.TensorVMSplit.forward
        def fwd(self, *a, **k):
            k.setdefault("_brdf_jitter_dense", noise)
            return orig(self, *a, **k)
        FM.TensorVMSplit.forward = fwd
        try:
            with torch.no_grad():
                ret = Renderer_TensoIR_train(r2, None, l2, m2, N_samples=S, args=a2, device=dev)
                torch.cuda.synchronize(); t0 = time.time()
                ret = Renderer_TensoIR_train(r2, None, l2, m2, N_samples=S, args=a2, device=dev)
                torch.cuda.synchronize(); log(f"  hip time {time.time() - t0:.4f}s for {r2.shape[0]} rays")
        finally:
            FM.TensorVMSplit.forward = orig
        t0 = time.time()
        ref = O.renderer_train(sc2, r2.cpu(), l2.cpu(), n_samples=S, brdf_jitter=noise)
        log(f"  oracle time {time.time() - t0:.2f}s")
        for k, v in ret.items():
            err(k, v, ref[k], 1e-9 if k.endswith("loss") else 1.0)
    section("big scene renderer vs oracle", s_big)
    section("big scene renderer vs oracle (2nd call, capacity-hint path)", s_big)


if __name__ == "__main__":
    main()

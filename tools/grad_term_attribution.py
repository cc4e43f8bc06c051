"""Which LOSS TERM carries the field-gradient deviation of the training step?  (the VM plane / line gradients are
2e-4 ... 2e-3 relative L2 from the fp32 oracle on ordinary states; profiles/r06_sigma_noise_sensitivity.txt ruled out HIP's sigma.)

Trains the bench's train workload for a while, freezes the state, and evaluates the 128-ray parity step with ONE term of the loss at a
time (train_tensoIR.py:262-311: rgb, rgb_with_brdf, normals_diff, normals_orientation, roughness / albedo smoothness; the others at
weight 0) on HIP, on the fp32 oracle and on the fp64 oracle: per term and parameter family the relative L2 distance HIP - fp32
oracle, HIP - fp64 oracle and fp32 oracle - fp64 oracle (the reference arithmetic's own noise on that term).
Then, for the rgb term, the march backward itself: d loss / d density-feature per SAMPLE as the HIP kernel returns it
(tir_march_primary_bwd's g_feature) against (a) the same formula evaluated in fp64 on the host from the kernel's own inputs (sigma, weights'
cotangents) -- the kernel's arithmetic error -- and (b) the oracle's autograd value for the same sample, split by where the sample sits on its ray.
Usage (GPU box): python tools/grad_term_attribution.py [train_steps=150]"""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    from oracle import tensoir_oracle as O          # checker only
    from tests.helpers import scene_from_model
    from tensoir_amd import Renderer_TensoIR_train, optim
    sys.argv = ["bench.py"]; a = bench.parse()
    device = torch.device("cuda", 0)
    ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
    model.march_t_stop = 1e-6
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    batches = [b.to(device) for b in bench.pose_batches(rays.cpu(), 8, 0)]
    with torch.no_grad():
        gts = [(0.8 * Renderer_TensoIR_train(b, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False, is_relight=True,
                                             sample_method="fixed_envirmap", device=device, args=args)["rgb_map"] + 0.1).contiguous() for b in batches]
    opt = optim.Adam(model.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99))
    S = a.samples
    for it in range(steps):
        ret = Renderer_TensoIR_train(batches[it % 8], None, lidx, model, N_samples=S, white_bg=True, is_train=True, is_relight=True,
                                     sample_method="stratified_sampling", device=device, args=args)
        loss = bench.train_loss(ret, gts[it % 8], True)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
    stride = 32
    r_all, l_all = batches[0][::stride].contiguous(), lidx[::stride].contiguous()
    Bs = r_all.shape[0]
    gen = torch.Generator().manual_seed(21)
    jitter, noise = torch.rand(Bs, 1, generator=gen), torch.randn(Bs, S, 3, generator=gen)
    g_all = torch.rand(Bs, 3, generator=gen)
    sc0 = scene_from_model(ckpt, model, a.env_h, a.env_w)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sc = O.scene_from_state_dict(sd, dict(ckpt["kwargs"]), sc0.alpha_volume, sc0.alpha_aabb, a.env_h, a.env_w)
    sc64 = bench._to_fp64(sc)
    orig_rand, orig_fwd, orig_loss = torch.rand, type(model).forward, O.training_loss
    TERMS = {"rgb": 1.0, **bench.TRAIN_W}

    def loss_of(ret, gt, w):
        return (w["rgb"] * torch.mean((ret["rgb_map"] - gt) ** 2) + w["rgb_brdf"] * torch.mean((ret["rgb_with_brdf_map"] - gt) ** 2)
                + w["normals_diff"] * ret["normals_diff_map"].mean() + w["normals_orientation"] * ret["normals_orientation_loss_map"].mean()
                + w["roughness_smoothness"] * ret["roughness_smoothness_loss"] + w["albedo_smoothness"] * ret["albedo_smoothness_loss"])

    def hip(w):
        model.zero_grad(set_to_none=True)

        def fake_rand(*aa, **k):
            if tuple(aa) == (Bs, 1) or (len(aa) == 1 and tuple(aa[0]) == (Bs, 1)):
                return jitter.clone()
            return orig_rand(*aa, **k)

        def fwd(self, rr, ll, **k):
            return orig_fwd(self, rr, ll, _brdf_jitter_dense=noise, **k)
        torch.rand, type(model).forward = fake_rand, fwd
        try:
            ret = Renderer_TensoIR_train(r_all, None, l_all, model, N_samples=S, white_bg=True, is_train=True, is_relight=True,
                                         sample_method="fixed_envirmap", device=device, args=args)
        finally:
            torch.rand, type(model).forward = orig_rand, orig_fwd
        loss_of(ret, g_all.to(device), w).backward()
        g = {nm: p.grad.detach().cpu().double().clone() for nm, p in model.named_parameters() if p.grad is not None}
        model.zero_grad(set_to_none=True)
        return g

    def oracle(scene, w, dt):
        O.training_loss = lambda ret, gt, rel, weights=None: loss_of(ret, gt, w)
        try:
            _, gr, _ = O.train_step_grads(scene, r_all.cpu().to(dt), l_all.cpu(), g_all.to(dt), is_relight=True, n_samples=S, ray_jitter=jitter.to(dt),
                                          brdf_jitter=noise.to(dt), second_n_sample=a.second_samples, weights=dict(bench.TRAIN_W))
        finally:
            O.training_loss = orig_loss
        return {k: v.double() for k, v in gr.items()}

    FAM = {"density": ("density_plane", "density_line"), "appearance": ("app_plane", "app_line"), "decoders+basis+light": None}

    def fam_of(nm):
        head = nm.split(".")[0]
        for f, heads in FAM.items():
            if heads and head in heads:
                return f
        return "decoders+basis+light"

    def dist(ga, gb):
        num, den = {f: 0.0 for f in FAM}, {f: 0.0 for f in FAM}
        for nm, ref in gb.items():
            if nm not in ga:
                continue
            f = fam_of(nm)
            num[f] += float((ga[nm] - ref).pow(2).sum()); den[f] += float(ref.pow(2).sum())
        return {f: ((num[f] / den[f]) ** 0.5 if den[f] > 0 else float("nan")) for f in FAM}
    cap_out = {}

    def per_sample():
        """d loss / d density feature per sample, rgb term only: HIP kernel / fp64 re-evaluation of its formula on its inputs / oracle autograd."""
        from tensoir_amd import ops
        w = {k: (v if k == "rgb" else 0.0) for k, v in TERMS.items()}
        cap = {}
        orig_bwd = ops.march_primary_bwd

        def spy_bwd(field, grad, rays_, jit_, sigma, weight, g_weight, g_acc, g_depth, want_g_feature=False):
            gf = orig_bwd(field, grad, rays_, jit_, sigma, weight, g_weight, g_acc, g_depth, want_g_feature=True)
            cap["hip"] = tuple(t.detach().cpu().double() for t in (sigma, weight, g_weight, g_acc, g_depth, gf))
            cap["delta"] = float(field.step_size) * float(field.distance_scale)
            return gf if want_g_feature else None

        def spy_r2a(sigma, dist):
            if sigma.requires_grad and "or_sigma" not in cap:
                cap["or_sigma"], cap["or_dist"] = sigma.detach().double(), dist.detach().double()      # dist = (z[k+1] - z[k]) * distance_scale in fp32
                sigma.register_hook(lambda g: cap.__setitem__("or_gsigma", g.detach().double()))
            return orig_r2a(sigma, dist)
        orig_r2a = O.raw2alpha
        ops.march_primary_bwd, O.raw2alpha = spy_bwd, spy_r2a
        try:
            hip(w); oracle(sc, w, torch.float32)
        finally:
            ops.march_primary_bwd, O.raw2alpha = orig_bwd, orig_r2a
        sig, wgt, gw, ga, gd, gf = cap["hip"]
        B_, S_ = sig.shape
        delta = cap["or_dist"]                                # the reference's own fp32 sample spacings (the kernel forms the same ones)
        x = sig * delta
        alpha = 1.0 - torch.exp(-x)
        v = 1.0 - alpha + 1e-10
        T = torch.cumprod(torch.cat([torch.ones(B_, 1, dtype=torch.float64), v], 1), 1)[:, :-1]
        wk = alpha * T
        gk = gw + ga.view(-1, 1)                              # (the depth cotangent is zero for this term)
        a_ = gk * wk
        sfx = torch.flip(torch.cumsum(torch.flip(a_, [1]), 1), [1]) - a_
        df64 = (gk * T - sfx / v) * delta * torch.exp(-x) * (-torch.expm1(-sig))
        print(f"  max |g_depth| {float(gd.abs().max()):.1e}; forward weights: max |w_hip - w_fp64(sigma_hip)| {float((wgt - wk).abs().max()):.2e}", flush=True)
        o_sig, o_gs = cap["or_sigma"], cap["or_gsigma"]
        df_or = o_gs * (-torch.expm1(-o_sig))
        print(f"  sigma: max |hip - oracle| / max {float((sig - o_sig).abs().max() / o_sig.abs().max()):.2e}, rel L2 {float((sig - o_sig).norm() / o_sig.norm()):.2e}", flush=True)
        # which input carries it: the same fp64 formula on the ORACLE's sigma with HIP's cotangents
        x2 = o_sig * delta
        al2 = 1.0 - torch.exp(-x2); v2 = 1.0 - al2 + 1e-10
        T2 = torch.cumprod(torch.cat([torch.ones(B_, 1, dtype=torch.float64), v2], 1), 1)[:, :-1]
        a2 = gk * (al2 * T2)
        sfx2 = torch.flip(torch.cumsum(torch.flip(a2, [1]), 1), [1]) - a2
        df64_os = (gk * T2 - sfx2 / v2) * delta * torch.exp(-x2) * (-torch.expm1(-o_sig))
        live = (sig > 0) & (o_sig > 0) & (T > 1e-4)
        rel = ((sig - o_sig).abs() / o_sig.abs().clamp_min(1e-30))[live]
        print(f"  sigma where both marched and T > 1e-4 ({int(live.sum())} samples): |hip - oracle| / oracle  max {float(rel.max()):.2e}  rms {float(rel.pow(2).mean().sqrt()):.2e}  median {float(rel.median()):.2e}", flush=True)
        relf = torch.where(live, (sig - o_sig).abs() / o_sig.abs().clamp_min(1e-30), torch.zeros_like(sig))
        top = torch.topk(relf.flatten(), 6).indices
        for ti in top.tolist():
            r_, k_ = ti // S_, ti % S_
            print(f"    worst: ray {r_} sample {k_}  sigma hip {float(sig[r_, k_]):.6e} oracle {float(o_sig[r_, k_]):.6e}  T {float(T[r_, k_]):.3e}  neighbours (oracle) {[float(f'{float(v):.3e}') for v in o_sig[r_, max(0, k_ - 2):k_ + 3]]}", flush=True)
        for lo_, hi_ in ((0, 1e-3), (1e-3, 1e-1), (1e-1, 10), (10, 1e9)):
            mm = live & (o_sig >= lo_) & (o_sig < hi_)
            if int(mm.sum()):
                rr = relf[mm]
                print(f"    sigma in [{lo_:g}, {hi_:g}): {int(mm.sum())} samples, rel diff rms {float(rr.pow(2).mean().sqrt()):.2e} max {float(rr.max()):.2e}", flush=True)
        print(f"  fp64(formula) on the ORACLE's sigma + HIP's cotangents - oracle autograd: {float((df64_os - df_or).norm()) / float(df64.norm()):.2e}"
              f"   (on HIP's sigma: {float((df64 - df_or).norm()) / float(df64.norm()):.2e})  -> the part of the distance that sigma carries", flush=True)
        cap_out["sig"], cap_out["T"], cap_out["o_sig"] = sig, T, o_sig
        regions = {"in front (T > 0.99)": T > 0.99, "surface (0.01 < T <= 0.99)": (T <= 0.99) & (T > 0.01), "behind (T <= 0.01)": T <= 0.01, "all samples": torch.ones_like(T, dtype=torch.bool)}
        tot = float(df64.norm())
        for name, m in regions.items():
            n_ = int(m.sum())
            print(f"  {name:28s} {n_:7d} samples  |df| {float(df64[m].norm()):.3e}   kernel - fp64(formula on its inputs) {float((gf - df64)[m].norm()) / tot:.2e}"
                  f"   kernel - oracle autograd {float((gf - df_or)[m].norm()) / tot:.2e}   fp64(formula) - oracle {float((df64 - df_or)[m].norm()) / tot:.2e}   (all relative to the total |df|)", flush=True)
    print(f"state after {steps} training steps, {Bs} rays; relative L2 over each parameter family (gradient of ONE loss term at its config weight)", flush=True)
    for term in list(TERMS) + ["all"]:
        w = {k: (v if (term == "all" or k == term) else 0.0) for k, v in TERMS.items()}
        t0 = time.time()
        gh = hip(w); g32 = oracle(sc, w, torch.float32); g64 = oracle(sc64, w, torch.float64)
        a_, b_, c_ = dist(gh, g32), dist(gh, g64), dist(g32, g64)
        norm = {f: sum(float(v.pow(2).sum()) for nm, v in g64.items() if fam_of(nm) == f) ** 0.5 for f in FAM}
        print(f"term {term:22s} ({time.time() - t0:.0f} s)", flush=True)
        for f in FAM:
            print(f"    {f:22s} |grad| {norm[f]:.3e}   hip-fp32 oracle {a_[f]:.2e}   hip-fp64 oracle {b_[f]:.2e}   fp32-fp64 oracle {c_[f]:.2e}", flush=True)
    print("per-sample d loss / d density feature of the rgb term (march backward):", flush=True)
    per_sample()
    # the density FEATURE itself on identical normalised coordinates: HIP's gather against the oracle's (fp32 and fp64)
    with torch.no_grad():
        pts, z, valid = O.sample_ray(sc, r_all[:, :3].cpu(), r_all[:, 3:6].cpu(), S, jitter)
        xyz = O.normalize_coord(sc, pts)[valid]
        f_or = O.density_feature(sc, xyz, "aten").double()
        f_64 = O.density_feature(sc64, xyz.double(), "aten")
        f_hip = model.compute_densityfeature(xyz.to(device)).cpu().double()
        # HIP's MARCH sigma against sigma from HIP's own gather at the ORACLE's sample coordinates: do the march's coordinates differ?
        sg = torch.zeros_like(cap_out["sig"])
        sg[valid] = torch.nn.functional.softplus(f_hip + float(sc.density_shift))
        m2 = (cap_out["sig"] > 0) & (sg > 0) & (cap_out["T"] > 1e-4)
        r2 = ((cap_out["sig"] - sg).abs() / sg.clamp_min(1e-30))[m2]
        r3 = ((cap_out["o_sig"] - sg).abs() / sg.clamp_min(1e-30))[m2]
        print(f"march sigma (HIP) vs softplus(HIP gather at the oracle's coordinates), {int(m2.sum())} samples: rel rms {float(r2.pow(2).mean().sqrt()):.2e} max {float(r2.max()):.2e};"
              f"   oracle's march sigma vs the same: rms {float(r3.pow(2).mean().sqrt()):.2e} max {float(r3.max()):.2e}", flush=True)
        near = (f_64 + float(sc.density_shift)) > -12.0          # samples whose sigma is not negligible
        for nm, m_ in (("all in-box samples", torch.ones_like(near)), ("feature + shift > -12", near)):
            d_h, d_o = (f_hip - f_64)[m_], (f_or - f_64)[m_]
            print(f"density feature, {nm} ({int(m_.sum())}): |f| rms {float(f_64[m_].pow(2).mean().sqrt()):.2f}   hip - fp64: rms {float(d_h.pow(2).mean().sqrt()):.2e} max {float(d_h.abs().max()):.2e}"
                  f"   fp32 oracle - fp64: rms {float(d_o.pow(2).mean().sqrt()):.2e} max {float(d_o.abs().max()):.2e}   hip - fp32 oracle: rms {float((f_hip - f_or)[m_].pow(2).mean().sqrt()):.2e} max {float((f_hip - f_or)[m_].abs().max()):.2e}   (absolute; d sigma / sigma <= this for sigma << 1)", flush=True)


if __name__ == "__main__":
    main()

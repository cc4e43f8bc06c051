"""Is the in-run training parity figure (bench.py train_parity_and_cpu) stable?  After N training steps, evaluate the HIP
gradients of the SAME subsampled step several times (fixed parameters, fixed jitter) and compare them with each other and
with the oracle.  Usage (GPU box): python tools/train_parity_repeat.py [steps]"""
import os, sys, types, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench


def worst_of(ga, gb):
    w = {}
    for n in ga:
        if n in gb and float(gb[n].abs().max()) > 0:
            w[n] = float((ga[n].double() - gb[n].double()).abs().max() / gb[n].double().abs().max())
    return [(k, float(f"{v:.2e}")) for k, v in sorted(w.items(), key=lambda kv: -kv[1])[:3]]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    from oracle import tensoir_oracle as O
    from tests.helpers import scene_from_model
    from tensoir_amd import Renderer_TensoIR_train, ops, optim
    sys.argv = ["bench.py"]; a = bench.parse()
    device = torch.device("cuda", 0)
    ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
    model.march_t_stop = 1e-6
    args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
    with torch.no_grad():
        gt = (0.8 * Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False, is_relight=True,
                                           sample_method="fixed_envirmap", device=device, args=args)["rgb_map"] + 0.1).contiguous()
    opt = optim.Adam(model.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99))
    batches = [b.to(device) for b in bench.pose_batches(rays.cpu(), 8, 0)]
    with torch.no_grad():
        gts = [(0.8 * Renderer_TensoIR_train(b, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False, is_relight=True,
                                             sample_method="fixed_envirmap", device=device, args=args)["rgb_map"] + 0.1).contiguous() for b in batches]
    for it in range(steps):
        rays, gt = batches[it % 8], gts[it % 8]
        ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=True, is_relight=True,
                                     sample_method="stratified_sampling", device=device, args=args)
        loss = bench.train_loss(ret, gt, True)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
    rays, gt = batches[0], gts[0]
    # ---- the check of bench.train_parity_and_cpu, HIP side repeated
    sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sc = O.scene_from_state_dict(sd, dict(ckpt["kwargs"]), sc.alpha_volume, sc.alpha_aabb, a.env_h, a.env_w)
    stride = 32
    r, l, g = rays[::stride].contiguous(), lidx[::stride].contiguous(), gt[::stride].contiguous()
    Bs, S = r.shape[0], a.samples
    gen = torch.Generator().manual_seed(21)
    jitter, noise = torch.rand(Bs, 1, generator=gen), torch.randn(Bs, S, 3, generator=gen)
    loss_ref, grads_ref, ret_ref = O.train_step_grads(sc, r.cpu(), l.cpu(), g.cpu(), is_relight=True, n_samples=S, ray_jitter=jitter, brdf_jitter=noise,
                                                       second_n_sample=a.second_samples, weights=dict(bench.TRAIN_W))
    # the same step with the oracle in fp64: which of the two fp32 implementations is closer to it?
    def to64(x):
        if torch.is_tensor(x):
            return x.double() if x.is_floating_point() else x
        if isinstance(x, (list, tuple)):
            return type(x)(to64(v) for v in x)
        if isinstance(x, dict):
            return {k: to64(v) for k, v in x.items()}
        if isinstance(x, types.SimpleNamespace):
            return type(x)(**{k: to64(v) for k, v in vars(x).items()})
        return x
    g_rand = torch.rand(Bs, 3, generator=torch.Generator().manual_seed(3))
    targets = {"fitted": g.cpu(), "random": g_rand}
    ref32, ref64 = {}, {}
    for tn, tg in targets.items():
        _, ref32[tn], _ = O.train_step_grads(sc, r.cpu(), l.cpu(), tg, is_relight=True, n_samples=S, ray_jitter=jitter, brdf_jitter=noise,
                                             second_n_sample=a.second_samples, weights=dict(bench.TRAIN_W))
        try:
            _, ref64[tn], _ = O.train_step_grads(to64(sc), r.cpu().double(), l.cpu(), tg.double(), is_relight=True, n_samples=S, ray_jitter=jitter.double(),
                                                 brdf_jitter=noise.double(), second_n_sample=a.second_samples, weights=dict(bench.TRAIN_W))
        except Exception as e:
            print("fp64 oracle failed:", type(e).__name__, e)
    # threshold decisions of the primary march: which (ray, sample) pairs are records (w > 1e-4) in the HIP march and in the oracle?
    with torch.no_grad():
        w_hip = ops.march_primary_train(model.packed_field(), r, jitter.to(device), S, 0.0)[0].cpu()
        _, aux = O.forward_primary(sc, r.cpu(), l.cpu(), n_samples=S, ray_jitter=jitter, brdf_jitter=noise, return_aux=True)
    thr = float(sc.weight_thres)
    mh, mo = w_hip > thr, aux.weight > thr
    mism = (mh != mo).nonzero()
    print(f"record masks: hip {int(mh.sum())} oracle {int(mo.sum())} mismatching (ray, sample) pairs {mism.shape[0]}:",
          [(int(i), int(k), float(f"{float(w_hip[i, k]):.6e}"), float(f"{float(aux.weight[i, k]):.6e}")) for i, k in mism[:6]],
          "max |w_hip - w_oracle|", float((w_hip - aux.weight).abs().max()), "samples within 1e-3 relative of the threshold:", int(((aux.weight / thr - 1).abs() < 1e-3).sum()))
    orig_rand, orig_fwd = torch.rand, type(model).forward

    def hip(target=None):
        model.zero_grad(set_to_none=True)
        g = gt[::stride].contiguous() if target is None else target.to(device)
        def fake_rand(*aa, **k):
            if tuple(aa) == (Bs, 1) or (len(aa) == 1 and tuple(aa[0]) == (Bs, 1)):
                return jitter.clone()
            return orig_rand(*aa, **k)
        def fwd(self, rr, ll, **k):
            return orig_fwd(self, rr, ll, _brdf_jitter_dense=noise, **k)
        torch.rand, type(model).forward = fake_rand, fwd
        try:
            ret = Renderer_TensoIR_train(r, None, l, model, N_samples=S, white_bg=True, is_train=True, is_relight=True,
                                         sample_method="fixed_envirmap", device=device, args=args)
        finally:
            torch.rand, type(model).forward = orig_rand, orig_fwd
        bench.train_loss(ret, g, True).backward()
        grads = {n: p.grad.detach().cpu().double().clone() for n, p in model.named_parameters() if p.grad is not None}
        maps = {k: ret[k].detach().cpu() for k in ("rgb_map", "acc_map", "depth_map", "rgb_with_brdf_map", "normal_map", "albedo_map")}
        return grads, maps
    g = gt[::stride].contiguous()
    runs = [hip() for _ in range(3)]
    for tn, tg in targets.items():
        gh = hip(tg)[0]
        if tn in ref64:
            top = worst_of(gh, ref64[tn])[0][0]
            d = (gh[top] - ref64[tn][top]).abs().reshape(-1)
            i = int(d.argmax())
            print(f"   [{tn}] worst tensor {top} {tuple(gh[top].shape)} at {i}: hip {float(gh[top].reshape(-1)[i]):.6e} fp64 {float(ref64[tn][top].reshape(-1)[i]):.6e} "
                  f"fp32-oracle {float(ref32[tn][top].reshape(-1)[i]):.6e} max|ref| {float(ref64[tn][top].abs().max()):.3e}; elements over 10% of worst: {int((d > 0.1 * d.max()).sum())} of {d.numel()}")
            print(f"[{tn} target] hip vs fp64 oracle:", worst_of(gh, ref64[tn]), "| fp32 oracle vs fp64 oracle:", worst_of({k: v.double() for k, v in ref32[tn].items()}, ref64[tn]),
                  "| hip vs fp32 oracle:", worst_of(gh, ref32[tn]))
    def worst(ga, gb):
        w = {}
        for n in ga:
            if n in gb and float(gb[n].abs().max()) > 0:
                w[n] = float((ga[n] - gb[n].double()).abs().max() / gb[n].double().abs().max())
        top = sorted(w.items(), key=lambda kv: -kv[1])[:3]
        return [(k, float(f"{v:.3e}")) for k, v in top]
    print("hip run0 vs run1:", worst(runs[0][0], runs[1][0]))
    print("hip run0 vs run2:", worst(runs[0][0], runs[2][0]))
    print("hip run0 vs oracle:", worst(runs[0][0], grads_ref))
    print("maps run0 vs run1:", {k: float((runs[0][1][k] - runs[1][1][k]).abs().max()) for k in runs[0][1]})
    print("maps run0 vs oracle:", {k: float(f"{float((runs[0][1][k] - ret_ref[k]).abs().max()):.3e}") for k in runs[0][1]})
    # where is the worst deviation of the worst tensor?
    name = worst(runs[0][0], grads_ref)[0][0]
    d = (runs[0][0][name] - grads_ref[name].double()).abs()
    idx = int(d.reshape(-1).argmax())
    print("worst tensor", name, tuple(d.shape), "flat index", idx, "hip", float(runs[0][0][name].reshape(-1)[idx]), "ref", float(grads_ref[name].reshape(-1)[idx]),
          "max|ref|", float(grads_ref[name].abs().max()), "n elements over 10% of worst:", int((d > 0.1 * d.max()).sum()))
    # per-ray map deviations: how many rays deviate
    for k in ("albedo_map", "normal_map", "rgb_map"):
        dm = (runs[0][1][k] - ret_ref[k]).abs().reshape(Bs, -1).max(dim=1).values
        print(k, "rays over 1e-5:", int((dm > 1e-5).sum()), "top", [float(f"{x:.2e}") for x in torch.topk(dm, 3).values])


if __name__ == "__main__":
    main()

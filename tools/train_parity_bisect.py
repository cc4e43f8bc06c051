"""Which RAY carries the deviation when the train workload's in-run gradient check misses its strict bound?
(one end state in ten is 2e-3 ... 1.6e-2 from the oracle on some dense tensor; HIP repeats to 5e-7 on a fixed
state and the fp32 oracle is within 5e-5 of fp64, so the deviation is HIP's and belongs to the state.)

Trains the bench's train workload, checks the 128-ray parity step every few steps (HIP vs the fp32 oracle, seeded random target),
and on a strict miss bisects the ray set: the loss is a mean over rays, so the gradient error is a sum of per-ray errors; the half
whose own check deviates more is kept until one ray is left.  Prints that ray's map rows on both sides, N.V, its record count, and
the check with that ray removed.  Then the ReLU masks of that ray's decoder rows are compared between the HIP
training forward (the hidden activations it saves for the backward) and the oracle (pre-activations of the same rows): which hidden
units are on one side of zero in HIP and on the other in the oracle, and how close to zero the oracle's pre-activation is there --
the same count over all 128 rays as the control.  With force=1 the last state is bisected even if it kept the strict bound.
Usage (GPU box): python tools/train_parity_bisect.py [seconds=150] [check_every=7] [force=0]"""
import os, sys, time, types, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    force = len(sys.argv) > 3 and sys.argv[3] not in ("0", "")
    t_start = time.time()
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
    stride = 32
    r_all, l_all = batches[0][::stride].contiguous(), lidx[::stride].contiguous()
    Bs = r_all.shape[0]
    gen = torch.Generator().manual_seed(21)
    jitter_all, noise_all = torch.rand(Bs, 1, generator=gen), torch.randn(Bs, S, 3, generator=gen)
    g_all = torch.rand(Bs, 3, generator=gen)
    W = dict(bench.TRAIN_W)
    FIELD = ("density_plane", "density_line", "app_plane", "app_line")
    orig_rand, orig_fwd = torch.rand, type(model).forward

    def train(n, it0):
        for it in range(it0, it0 + n):
            ret = Renderer_TensoIR_train(batches[it % 8], None, lidx, model, N_samples=S, white_bg=True, is_train=True, is_relight=True,
                                         sample_method="stratified_sampling", device=device, args=args)
            loss = bench.train_loss(ret, gts[it % 8], True)
            opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        return it0 + n

    def hip(idx):
        n = idx.numel()
        jit, noi = jitter_all[idx], noise_all[idx]
        model.zero_grad(set_to_none=True)

        def fake_rand(*aa, **k):
            if tuple(aa) == (n, 1) or (len(aa) == 1 and tuple(aa[0]) == (n, 1)):
                return jit.clone()
            return orig_rand(*aa, **k)

        def fwd(self, rr, ll, **k):
            return orig_fwd(self, rr, ll, _brdf_jitter_dense=noi, **k)
        torch.rand, type(model).forward = fake_rand, fwd
        try:
            ret = Renderer_TensoIR_train(r_all[idx.to(device)], None, l_all[idx.to(device)], model, N_samples=S, white_bg=True, is_train=True,
                                         is_relight=True, sample_method="fixed_envirmap", device=device, args=args)
        finally:
            torch.rand, type(model).forward = orig_rand, orig_fwd
        bench.train_loss(ret, g_all[idx].to(device), True).backward()
        grads = {nm: p.grad.detach().cpu().double().clone() for nm, p in model.named_parameters() if p.grad is not None}
        model.zero_grad(set_to_none=True)
        return grads, {k: v.detach().cpu() for k, v in ret.items() if torch.is_tensor(v)}

    def oracle(sc, idx):
        _, gr, ret = O.train_step_grads(sc, r_all[idx.to(device)].cpu(), l_all[idx.to(device)].cpu(), g_all[idx], is_relight=True, n_samples=S,
                                        ray_jitter=jitter_all[idx], brdf_jitter=noise_all[idx], second_n_sample=a.second_samples, weights=W)
        return gr, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ret.items()}

    def deviation(gh, gr):
        """(dense max-norm rel, field rel L2, name of the worst dense tensor, absolute L2 of the difference over all tensors)"""
        dense, l2, worst, tot = 0.0, 0.0, None, 0.0
        for nm, ref in gr.items():
            if nm not in gh or float(ref.abs().max()) == 0.0:
                continue
            d = (gh[nm] - ref.double()).abs()
            tot += float(d.pow(2).sum())
            if nm.split(".")[0] in FIELD:
                l2 = max(l2, float(d.norm() / ref.double().norm()))
            else:
                v = float(d.max() / ref.double().abs().max())
                if v > dense:
                    dense, worst = v, nm
        return dense, l2, worst, tot ** 0.5

    def scene_now():
        sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        return O.scene_from_state_dict(sd, dict(ckpt["kwargs"]), sc.alpha_volume, sc.alpha_aabb, a.env_h, a.env_w)


    def relu_masks(idx, label):
        """ReLU masks of the primary-stage decoder rows of rays `idx`: HIP's saved hidden activations against the oracle's pre-activations."""
        import torch.nn.functional as F
        from tensoir_amd import ops
        rec_h, rec_o = [], []
        orig_multi, orig_mlp3 = ops.mlp_multi, O.mlp3

        def spy_multi(jobs, n_dev=None, save_hidden=False):
            out = orig_multi(jobs, n_dev=n_dev, save_hidden=save_hidden)
            if save_hidden:
                rec_h.append([(h1.detach().cpu(), h2.detach().cpu(), int(o.shape[1])) for (o, h1, h2) in out])
            return out

        def spy_mlp3(w, x):
            z1 = F.linear(x, w["w0"], w["b0"]); z2 = F.linear(torch.relu(z1), w["w1"], w["b1"])
            if torch.is_grad_enabled():
                rec_o.append((z1.detach().reshape(-1, z1.shape[-1]), z2.detach().reshape(-1, z2.shape[-1]), int(w["w2"].shape[0])))
            return F.linear(torch.relu(z2), w["w2"], w["b2"])
        ops.mlp_multi, O.mlp3 = spy_multi, spy_mlp3
        try:
            hip(idx); oracle(sc, idx)
        finally:
            ops.mlp_multi, O.mlp3 = orig_multi, orig_mlp3
        if not rec_h or not rec_o:
            print(f"      [{label}] no decoder rows recorded (hip {len(rec_h)}, oracle {len(rec_o)})", flush=True)
            return
        names = ["rgb", "brdf", "brdf (jittered features)", "normal"]
        for j, (h1, h2, od) in enumerate(rec_h[0]):
            # the oracle call of this decoder: same output width, same row count, closest layer-1 activations
            best = None
            for (z1, z2, od_o) in rec_o:
                n = z1.shape[0]
                if od_o != od or n > h1.shape[0] or n == 0:
                    continue
                d = float((torch.relu(z1) - h1[:n]).abs().max())
                if best is None or d < best[0]:
                    best = (d, z1, z2, n)
            if best is None:
                print(f"      [{label}] {names[j] if j < 4 else j}: no oracle call matches", flush=True)
                continue
            d, z1, z2, n = best
            out = [f"{n} rows, max |relu(z1) - h1| {d:.2e}"]
            for lname, z, hh in (("layer 1", z1, h1[:n]), ("layer 2", z2, h2[:n])):
                flip = (z > 0) != (hh > 0)
                nf = int(flip.sum())
                if nf:
                    zz = z[flip].abs()
                    rows = sorted(set(torch.nonzero(flip)[:, 0].tolist()))
                    out.append(f"{lname}: {nf} of {z.numel()} units on the other side of zero (oracle |z| there: max {float(zz.max()):.2e}, median {float(zz.median()):.2e}; rows {rows[:8]}{'...' if len(rows) > 8 else ''})")
                else:
                    near = float(z.abs().min())
                    out.append(f"{lname}: masks identical (smallest oracle |z| {near:.2e})")
            print(f"      [{label}] {names[j] if j < 4 else j}: " + "; ".join(out), flush=True)

    it = train(100, 0)
    everyone = torch.arange(Bs)
    found = 0
    while time.time() - t_start < budget and found < 3:
        it = train(every, it)
        torch.cuda.synchronize()
        sc = scene_now()
        gh, ret_h = hip(everyone)
        gr, ret_o = oracle(sc, everyone)
        dense, l2, worst, _ = deviation(gh, gr)
        print(f"[state after {it} steps] dense {dense:.2e} ({worst}) field L2 {l2:.2e}", flush=True)
        last_try = force and time.time() - t_start >= budget - 60 and found == 0
        if dense < 2e-3 and l2 < 3e-3 and not last_try:
            continue
        if dense < 2e-3 and l2 < 3e-3:
            print("   (forced: this state keeps the strict bound; bisected anyway)", flush=True)
        found += 1
        cur = everyone
        while cur.numel() > 1 and time.time() - t_start < budget + 60:
            halves = (cur[:cur.numel() // 2], cur[cur.numel() // 2:])
            devs = []
            for h in halves:
                g1, _ = hip(h)
                g2, _ = oracle(sc, h)
                devs.append(deviation(g1, g2))
            k = 0 if devs[0][3] * halves[0].numel() >= devs[1][3] * halves[1].numel() else 1      # error sums weighted back by 1/n of the mean
            print(f"   bisect {cur.numel()} -> halves dense {devs[0][0]:.2e} / {devs[1][0]:.2e}, abs {devs[0][3] * halves[0].numel():.3e} / {devs[1][3] * halves[1].numel():.3e}"
                  f" -> keep {'first' if k == 0 else 'second'}", flush=True)
            cur = halves[k]
        ray = int(cur[0])
        rest = everyone[everyone != ray]
        g1, _ = hip(rest); g2, _ = oracle(sc, rest)
        d_rest = deviation(g1, g2)
        g1, r1 = hip(cur); g2, r2 = oracle(sc, cur)
        d_one = deviation(g1, g2)
        print(f"   -> ray {ray} of the subsample: alone dense {d_one[0]:.2e} ({d_one[2]}) field L2 {d_one[1]:.2e}; all rays but it: dense {d_rest[0]:.2e} field L2 {d_rest[1]:.2e}")
        rd = r_all[ray, 3:6].cpu()
        try:
            report_ray(ray, rd, ret_h, ret_o, Bs)
        except Exception as e:
            print("      (ray report failed:", type(e).__name__, e, ")", flush=True)
        try:
            relu_masks(cur, f"ray {ray}")
            relu_masks(everyone, "all 128 rays")
        except Exception as e:
            print("      (mask report failed:", type(e).__name__, e, ")", flush=True)
    print(f"done: {found} strict misses in {it} steps, {time.time() - t_start:.0f} s")


def report_ray(ray, rd, ret_h, ret_o, Bs):
    if True:
        for side, rr in (("hip", ret_h), ("oracle", ret_o)):
            nrm = rr["normal_map"][ray].reshape(-1)
            nv = float(-(nrm / nrm.norm().clamp(min=1e-12) * (rd / rd.norm())).sum())
            print(f"      {side:6s} acc {float(rr['acc_map'][ray]):.7f} depth {float(rr['depth_map'][ray]):.6f} N.V {nv:+.3e} |n| {float(nrm.norm()):.6f} rough {rr['roughness_map'][ray].reshape(-1)[:1].tolist()} "
                  f"albedo {[round(v, 6) for v in rr['albedo_map'][ray].reshape(-1).tolist()]} rgb {[round(v, 6) for v in rr['rgb_map'][ray].reshape(-1).tolist()]} "
                  f"rgb_brdf {[round(v, 7) for v in rr['rgb_with_brdf_map'][ray].reshape(-1).tolist()]} ndiff {float(rr['normals_diff_map'][ray].reshape(-1)[0]):.3e} "
                  f"norient {float(rr['normals_orientation_loss_map'][ray].reshape(-1)[0]):.3e}")
        dmap = {k: float((ret_h[k].reshape(Bs, -1)[ray] - ret_o[k].reshape(Bs, -1)[ray]).abs().max()) for k in
                ("rgb_map", "rgb_with_brdf_map", "normal_map", "albedo_map", "roughness_map", "normals_diff_map", "normals_orientation_loss_map", "acc_map")
                if k in ret_h and k in ret_o}
        print("      map rows, |hip - oracle| of that ray:", {k: float(f"{v:.2e}") for k, v in dmap.items()}, flush=True)


if __name__ == "__main__":
    main()

"""How smooth is the REFERENCE's training gradient as a function of the arithmetic?  (CPU only: the oracle against itself.)

The train workload's in-run gradient check misses its strict bound in about one end state in ten, always through ONE ray
(tools/train_parity_bisect.py), also with the exact fp32 decoders, while the maps agree to 1e-5.  This script shows the mechanism
without any HIP code: the oracle's gradients of one training step on a seeded scene, against the same step with every decoder
weight multiplied by (1 + eps N(0,1)) -- eps = 1e-5 stands for the split-bf16 decoders' distance from fp32, 1e-7 for one fp32
evaluation order against another.  A smooth function would answer with ~eps x condition number; the decoders are piecewise linear
(ReLU), so a pre-activation within eps of zero flips its mask on one side only and the gradient jumps by that unit's whole share.

Usage: python tools/grad_kink_sensitivity.py [n_scenes=24] [out.json]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import tensoir_oracle as O          # this tool IS about the oracle
from tensoir_amd import synth
from tests.helpers import scene_from_checkpoint

W = dict(rgb_brdf=0.2, normals_diff=0.0005, normals_orientation=0.001, albedo_smoothness=0.001, roughness_smoothness=0.001)
FIELD = ("density_plane", "density_line", "app_plane", "app_line")


def grads(ck, rays, lidx, gt, jit, noi, S):
    sc = scene_from_checkpoint(ck, 8, 16)
    return O.train_step_grads(sc, rays, lidx, gt, is_relight=True, n_samples=S, ray_jitter=jit, brdf_jitter=noi, second_n_sample=32, weights=W)[1]


def deviation(ga, gb):
    worst, name = 0.0, None
    for n in ga:
        den = float(ga[n].abs().max())
        if n.split(".")[0] in FIELD or den == 0.0:
            continue
        v = float((ga[n] - gb[n]).abs().max() / den)
        if v > worst:
            worst, name = v, n
    return worst, name


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    out = sys.argv[2] if len(sys.argv) > 2 else None
    S, B = 128, 48
    rays_all = synth.make_rays(64, 64)
    res = {1e-5: [], 1e-7: []}
    single = []
    t0 = time.time()
    for seed in range(n_scenes):
        ck = synth.make_checkpoint(grid=(48,) * 3, seed=100 + seed)
        gen = torch.Generator().manual_seed(seed)
        idx = torch.randperm(4096, generator=gen)[:B]
        rays, lidx = rays_all[idx], torch.zeros(B, 1, dtype=torch.int32)
        gt, jit, noi = torch.rand(B, 3, generator=gen), torch.rand(B, 1, generator=gen), torch.randn(B, S, 3, generator=gen)
        g0 = grads(ck, rays, lidx, gt, jit, noi, S)
        for eps in res:
            sd = dict(ck["state_dict"])
            for k in list(sd):
                if ".mlp." in k and k.endswith("weight"):
                    sd[k] = sd[k] * (1 + eps * torch.randn(sd[k].shape, generator=gen))
            ck2 = {**ck, "state_dict": sd}
            d = deviation(g0, grads(ck2, rays, lidx, gt, jit, noi, S))
            res[eps].append(d)
            if d[0] > 2e-3 and len(single) < 6:
                # is it ONE ray here too?  (the loss is a mean over rays: bench.single_ray_bisect)
                import bench
                dev_of = lambda ix: bench.grad_deviation(grads(ck2, rays[ix], lidx[ix], gt[ix], jit[ix], noi[ix], S),
                                                         grads(ck, rays[ix], lidx[ix], gt[ix], jit[ix], noi[ix], S))
                ray, alone, rest = bench.single_ray_bisect(B, dev_of)
                single.append({"scene": seed, "eps": eps, "all_rays": float(f"{d[0]:.2e}"), "ray": ray, "that_ray_alone": float(f"{alone['dense']:.2e}"),
                               "all_rays_but_it": float(f"{rest['dense']:.2e}")})
    report = {"what": "max over the decoder / basis / light gradient tensors of max|g(perturbed) - g| / max|g|, one training step of the ORACLE "
                      f"on {B} rays x {S} samples of a seeded 48^3 scene, decoder weights x (1 + eps N(0,1))", "scenes": n_scenes, "seconds": round(time.time() - t0, 1)}
    for eps, v in res.items():
        vals = sorted(x[0] for x in v)
        report[f"eps={eps:g}"] = {"median": float(f"{vals[len(vals) // 2]:.3e}"), "max": float(f"{vals[-1]:.3e}"),
                                  "share_over_2e-3": round(sum(x > 2e-3 for x in vals) / len(vals), 3),
                                  "median_over_eps": round(vals[len(vals) // 2] / eps, 1), "max_over_eps": round(vals[-1] / eps, 1),
                                  "sorted": [float(f"{x:.2e}") for x in vals], "worst_tensor": max(v, key=lambda t: t[0])[1]}
    report["single_ray_bisection_of_the_scenes_over_2e-3"] = single
    print(json.dumps(report, indent=1))
    if out:
        with open(out, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()

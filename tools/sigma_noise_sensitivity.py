"""Does the density's last-bit noise explain the deviation of the VM plane / line gradients?  (CPU only: the oracle against itself.)

Round 5's DESIGN attributed the 2e-4 ... 2e-3 relative-L2 deviation of the HIP field gradients to HIP's sigma being ~1e-6
(relative) from the fp32 reference's, amplified by the transmittance.  This multiplies the ORACLE's own sigma by (1 + eps N(0,1))
and compares the gradients of one training step with the unperturbed ones, on soft and sharp seeded scenes: eps = 1e-6 and 1e-7 give
the SAME deviation (1e-4 ... 2e-4 relative L2 on the sparse field tensors, 1e-6 ... 1e-5 on the dense ones) -- the floor is the fp32
summation noise of the sparse gradients themselves, not the size of the sigma error.  A libm-grade / fp64 sigma in the training march
would therefore not lower the figure (profiles/r06_sigma_noise_sensitivity.txt).
Usage: python tools/sigma_noise_sensitivity.py"""
import sys, torch, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tensoir_oracle as O
from tensoir_amd import synth
from tests.helpers import scene_from_checkpoint
import bench
W = dict(rgb_brdf=0.2, normals_diff=0.0005, normals_orientation=0.001, albedo_smoothness=0.001, roughness_smoothness=0.001)
S, B = 160, 48
rays_all = synth.make_rays(64, 64)
orig = O.feature2density
for name, kw in (("soft", {}), ("sharp", dict(blob_sigma=0.2, blob_gain=2000.0))):
    for seed in range(3):
        ck = synth.make_checkpoint(grid=(64,)*3, seed=100+seed, **kw)
        gen = torch.Generator().manual_seed(seed)
        idx = torch.randperm(4096, generator=gen)[:B]
        rays, lidx = rays_all[idx], torch.zeros(B, 1, dtype=torch.int32)
        gt, jit, noi = torch.rand(B,3,generator=gen), torch.rand(B,1,generator=gen), torch.randn(B,S,3,generator=gen)
        sc = scene_from_checkpoint(ck, 8, 16)
        def run():
            return O.train_step_grads(sc, rays, lidx, gt, is_relight=True, n_samples=S, ray_jitter=jit, brdf_jitter=noi, second_n_sample=32, weights=W)[1]
        g0 = run()
        for eps in (1e-6, 1e-7):
            g2 = torch.Generator().manual_seed(7)
            def f2d(sc_, f, eps=eps):
                s = orig(sc_, f)
                return s * (1 + eps * torch.randn(s.shape, generator=g2).to(s.dtype))
            O.feature2density = f2d
            try:
                g1 = run()
            finally:
                O.feature2density = orig
            d = bench.grad_deviation(g1, g0)
            print(name, seed, eps, {k: float(f"{v:.2e}") for k, v in d.items()}, flush=True)

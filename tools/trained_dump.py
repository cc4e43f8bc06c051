"""Train the analytic scene through the product API (tests/precision_cases.trained) and dump what a CPU-side analysis needs:
the checkpoint (reference layout), the probe rays / jitter noise and the HIP maps of the default policy ->
gpurun_out/r05_trained_dump.pt.  Usage (GPU box): python tools/trained_dump.py [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests import precision_cases as P
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 450
    r = P.trained(iters)
    m = r.model
    n = min(2048, r.rays_f.shape[0])
    rays = r.rays_f[:n].cuda()
    lidx = r.lidx_f[:n].cuda().to(torch.int32).reshape(-1, 1)
    S = int(m.nSamples)
    noise = torch.randn(n, S, 3, generator=torch.Generator().manual_seed(5))
    out, brdf = P.render(m, rays, lidx, noise, S)
    with P.policy(False, None, None):
        _, brdf_full = P.render(m, rays, lidx, noise, S)
    # the same with the exact fp32 decoders / gathers (ops.MLP_IMPL = mfma) and without early termination
    from tensoir_amd import ops
    old_impl, old_stop = ops.MLP_IMPL, m.march_t_stop
    ops.MLP_IMPL, m.march_t_stop = "mfma", 0.0
    try:
        out_exact, brdf_exact = P.render(m, rays, lidx, noise, S)
    finally:
        ops.MLP_IMPL, m.march_t_stop = old_impl, old_stop
    to_cpu = lambda v: v.detach().cpu() if torch.is_tensor(v) else v
    vol = m.alphaMask.alpha_volume[0, 0].bool().cpu()
    ckpt = {"kwargs": {k: to_cpu(v) for k, v in m.get_kwargs().items()}, "state_dict": {k: to_cpu(v).contiguous() for k, v in m.state_dict().items()},
            "alphaMask.shape": tuple(vol.shape), "alphaMask.mask": np.packbits(vol.numpy().reshape(-1)), "alphaMask.aabb": m.alphaMask.aabb.cpu()}
    torch.save({"ckpt": ckpt, "rays": rays.cpu(), "lidx": lidx.cpu(), "noise_seed": 5, "n_samples": S,
                "hip": {k: to_cpu(v) for k, v in zip(P.NAMES, out)}, "hip_brdf": brdf.cpu(), "hip_brdf_full": brdf_full.cpu(),
                "hip_exact": {k: to_cpu(v) for k, v in zip(P.NAMES, out_exact)}, "hip_brdf_exact": brdf_exact.cpu(),
                "losses": r.losses, "grids": r.grids, "policy": m.indirect_precision()},
               os.path.join(ROOT, "gpurun_out", "r05_trained_dump.pt"))
    print("dumped", n, "rays,", S, "samples, grid", r.grids[-1], "psnr", float(-10 * np.log10(np.mean(r.losses[-10:]))))


if __name__ == "__main__":
    main()

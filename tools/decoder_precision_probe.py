"""Which operand scheme does the radiance decoder of the secondary-ray records need on a TRAINED 300^3 checkpoint?  (GPU box.)

profiles/r06a_precision_trained_300_r5_kernels.json: on the trained checkpoint the auto policy rejects the fp16 kernels
(rgb_with_brdf_map 7.3e-5 from the full kernels; fp32 taps + fp16 decoder alone: 2.8e-5).  Before a fused full-precision kernel
is written this measures, with the FULL gather (fp32 taps) and the decoder EMULATED in torch (fp32 accumulate, operands rounded
as the matrix instructions would see them, aux columns + bias exact as in the aux-table kernels, layer 3 exact), the map-level
deviation of candidate operand schemes from the exact fp32 decoder -- through the real integration kernels, on the trained
model's own rays:
  f16x1   : W, x rounded to fp16 (the shipped fp16 decoder's arithmetic)
  f16x2   : x = hi + lo in fp16 (two products), W rounded to fp16            -- one LDS operand image
  f16x2w  : W = hi + lo in fp16 (two products), x rounded to fp16
  f16+fp8c: W = fp16 hi + fp8 lo (x 2^17), the lo product on v_mfma_f32_32x32x16_fp8_fp8 operands -- 1.5 LDS images
  f16x3l1 : layer 1 three products (x and W split), layer 2 as f16x2          -- 1.5 LDS images
  f16x3   : three products in both layers
  bf16x3  : the primary-stage scheme
Reference stage: models/relight_utils.py:818-832.  Usage: python tools/decoder_precision_probe.py [--iters 2400] [--rays 4096]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def split(x, dt):
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def prod(x, w, scheme):
    """x [N,K] @ w[U,K]^T with the operand rounding of `scheme`, fp32 accumulate."""
    if scheme == "exact":
        return x @ w.t()
    dt = torch.bfloat16 if scheme.startswith("bf16") else torch.float16
    xh, xl = split(x, dt)
    wh, wl = split(w, dt)
    if scheme in ("f16x1",):
        return xh @ wh.t()
    if scheme == "f16+fp8c":       # W = fp16 hi + fp8(e4m3) lo scaled by 2^17, the lo product on fp8 operands (x rounded to fp8 there)
        f8 = torch.float8_e4m3fn
        wl8 = ((w - wh) * 2.0 ** 17).clamp(-448, 448).to(f8).float()
        x8 = x.clamp(-448, 448).to(f8).float()
        return xh @ wh.t() + (x8 @ wl8.t()) * 2.0 ** -17
    if scheme == "f16x2":
        return xh @ wh.t() + xl @ wh.t()
    if scheme == "f16x2w":
        return xh @ wh.t() + xh @ wl.t()
    return xh @ wh.t() + xl @ wh.t() + xh @ wl.t()          # *x3


SCHEMES = {"exact": ("exact", "exact"), "f16x1": ("f16x1", "f16x1"), "f16x2": ("f16x2", "f16x2"), "f16x2w": ("f16x2w", "f16x2w"),
           "f16+fp8c": ("f16+fp8c", "f16+fp8c"), "f16x3l1": ("f16x3", "f16x2"), "f16x3": ("f16x3", "f16x3"), "bf16x3": ("bf16x3", "bf16x3")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2400)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_decoder_precision_probe.json"))
    a = ap.parse_args()
    from tests import precision_cases as P
    from tools.precision_300 import train, stats
    from tensoir_amd import ops, relight
    r = train(a.iters, 4096, 12, 160)
    m = r.model
    n = min(a.rays, r.rays_f.shape[0])
    rays = r.rays_f[:n].cuda()
    lidx = r.lidx_f[:n].cuda().to(torch.int32).reshape(-1, 1)
    S = int(m.nSamples)
    noise = torch.randn(n, S, 3, generator=torch.Generator().manual_seed(5))
    mlp = m.renderModule.mlp
    W0, b0, W1, b1, W2, b2 = (mlp[0].weight.detach(), mlp[0].bias.detach(), mlp[2].weight.detach(), mlp[2].bias.detach(),
                              mlp[4].weight.detach(), mlp[4].bias.detach())
    Fd, PE = 27, 2
    # input columns (models/tensorBase_rotated_lights.py:137-142, :12-17): [feat 27 | view 3 | sin PE(feat) 54 | cos PE(feat) 54 | sin PE(view) 6 | cos PE(view) 6]
    c_feat = list(range(0, Fd)) + list(range(Fd + 3, Fd + 3 + 4 * Fd))
    c_aux = list(range(Fd, Fd + 3)) + list(range(Fd + 3 + 4 * Fd, Fd + 3 + 4 * Fd + 12))
    freqs = (2.0 ** torch.arange(PE, device="cuda")).float()
    real = relight._gather_then_decode
    state = {"scheme": None, "rec": None}

    def pe(v):
        p = (v[..., None] * freqs).reshape(v.shape[0], -1)
        return torch.sin(p), torch.cos(p)

    def emulated(tensoIR, f, fh, rec_xyz, light_idx, rec_ray, light_div, dirs, dir_map, n_dirs, n_dev, full=False):
        feat = ops.vm_app(f, rec_xyz, light_idx, rec_ray, True, False, None, light_div, n_dev)[0][:, :Fd]
        nv = int(n_dev.item()) if n_dev is not None else feat.shape[0]
        out = torch.zeros((feat.shape[0], 3), dtype=torch.float32, device=feat.device)
        l1, l2 = SCHEMES[state["scheme"]]
        for lo in range(0, nv, 1 << 19):
            hi = min(nv, lo + (1 << 19))
            ft = feat[lo:hi]
            aux = dirs[(rec_ray[lo:hi].long() % n_dirs)]
            s, c = pe(ft)
            sa, ca = pe(aux)
            x_main = torch.cat([ft, s, c], dim=1)
            x_aux = torch.cat([aux, sa, ca], dim=1)
            z1 = x_aux @ W0[:, c_aux].t() + b0 + prod(x_main, W0[:, c_feat], l1)
            h1 = torch.relu(z1)
            h2 = torch.relu(prod(h1, W1, l2) + b1)
            out[lo:hi] = torch.sigmoid(h2 @ W2.t() + b2)
        if state["rec"] is not None:
            state["rec"][state["scheme"]] = out[:nv].clone()
        return out

    res, maps, recs = {}, {}, {}
    with P.policy(False, None, None):
        _, maps["kernels_full"] = P.render(m, rays, lidx, noise, S)
    relight._gather_then_decode = emulated
    try:
        state["rec"] = recs
        for name in SCHEMES:
            state["scheme"] = name
            with P.policy(False, None, None):
                _, maps[name] = P.render(m, rays, lidx, noise, S)
    finally:
        relight._gather_then_decode = real
    for name in maps:
        if name == "exact":
            continue
        res[name] = {"map_vs_exact": stats(maps[name], maps["exact"])}
        if name in recs:
            d = (recs[name] - recs["exact"]).double()
            res[name]["records_vs_exact"] = {"bias": float(d.mean(0).abs().max()), "rms": float(d.pow(2).mean().sqrt()), "max": float(d.abs().max()), "records": int(d.shape[0])}
        print(name, json.dumps(res[name]), flush=True)
    rep = {"iterations": a.iters, "grids": r.grids, "rays": n, "library_source_hash": open(os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so.srchash")).read().strip(),
           "what": "rgb_with_brdf_map (and the decoded records) under emulated decoder operand schemes against the exact fp32 decoder, full-precision gather, trained 300^3 checkpoint",
           "schemes": res}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()

"""Cases of the indirect-light precision policy, shared by tests/test_gpu_precision_policy.py and
tools/precision_sweep.py / tools/precision_300.py (which write profiles/r0N_precision_*.json).  Reference stage: models/relight_utils.py:777-834
(compute_radiance: compute_appfeature -> renderModule on the secondary-ray records, fp32 throughout).

  * trained(): a checkpoint TRAINED through the product API (tests/train_sequence.py: 450 iterations incl. updateAlphaMask /
    shrink / upsample, 330 of them with the relighting losses), rendered with the default policy and compared with the oracle;
  * sweep(): the synthetic scene with its appearance planes / radiance-decoder weights / light rows scaled up (decoder weight
    norms and feature magnitudes grow in training) -- what the f16 kernels would do unguarded, what the policy decides."""
import contextlib
import io
import types

import torch

MAPS = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map", "acc_map", "normals_diff_map",
        "normals_orientation_loss_map"]
NAMES = MAPS + ["acc_mask", "albedo_smoothness_loss", "roughness_smoothness_loss"]
ARGS = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)


@contextlib.contextmanager
def policy(guard, mlp="f16", app="h16"):
    """Temporarily select the indirect-light policy: guard=True -> auto, guard=False + (f16, h16) -> forced f16,
    (None, None) -> full."""
    from tensoir_amd import ops
    old = ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL
    ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = guard, mlp, app
    try:
        yield
    finally:
        ops.INDIRECT_GUARD, ops.SECONDARY_MLP_IMPL, ops.SECONDARY_APP_IMPL = old


@torch.no_grad()
def render(model, rays, lidx, noise, n_samples):
    """-> (12-tuple of the primary pass, rgb_with_brdf rows [B,3]) through the checked route of the parity tests."""
    from tensoir_amd import relight
    out, maps = model(rays, lidx, N_samples=n_samples, _brdf_jitter_dense=noise, _return_maps=True)
    brdf = relight.shade_from_maps(model, maps, rays, lidx, "fixed_envirmap", ARGS, acc_thres=0.5)
    return out, brdf


def three_policies(model, rays, lidx, noise, n_samples):
    """rgb_with_brdf under forced f16, full and auto (+ what auto decided and measured)."""
    res = {}
    with policy(False):
        out, res["f16"] = render(model, rays, lidx, noise, n_samples)
    with policy(False, None, None):
        _, res["full"] = render(model, rays, lidx, noise, n_samples)
    model.__dict__.pop("_indirect_state", None)
    with policy(True):
        _, res["auto"] = render(model, rays, lidx, noise, n_samples)
        res["decision"] = model.indirect_precision()
    d16 = (res["f16"] - res["full"]).double()
    res["f16_vs_full"] = {"max_abs": float(d16.abs().max()), "rms": float(d16.pow(2).mean().sqrt()), "mean_signed": float(d16.mean()),
                          "finite": bool(torch.isfinite(res["f16"]).all())}
    res["auto_vs_full_max_abs"] = float((res["auto"] - res["full"]).abs().max())
    return out, res


def oracle_compare(sc, out, brdf, rays, lidx, noise, n_samples, sel, fp64_floor=False):
    """Every map of the HIP render (rows `sel`) against oracle.renderer_train on the same rays -> {map: parity_metrics}."""
    from oracle import tensoir_oracle as O          # checker only
    from tests.helpers import ggx_flip_rays, parity_metrics
    with torch.no_grad():
        ref = O.renderer_train(sc, rays.cpu()[sel], lidx.cpu()[sel], n_samples=n_samples, brdf_jitter=noise[sel], second_n_sample=96)
    got = dict(zip(NAMES, out))
    rep = {n: parity_metrics(got[n].cpu()[sel], ref[n]) for n in MAPS}
    keep = ~ggx_flip_rays(ref["normal_map"], rays.cpu()[sel])       # the reference's GGX normal flip is discontinuous at N.V = 0
    rep["rgb_with_brdf_map"] = parity_metrics(brdf.cpu()[sel][keep], ref["rgb_with_brdf_map"][keep])
    # the per-pixel relative figure over pixels brighter than 0.2 (a trained scene has dark pixels, where an absolute 1e-5 --
    # the fp32 noise of a 150-term split-bf16 decoder, measured on every trained run -- is 1e-4 relative at brightness 0.1)
    for n in ("rgb_map", "normal_map"):
        rep[n]["max_rel_pixel_bright"] = parity_metrics(got[n].cpu()[sel], ref[n], 0.2)["max_rel_pixel"]
    rep["rgb_with_brdf_map"]["max_rel_pixel_bright"] = parity_metrics(brdf.cpu()[sel][keep], ref["rgb_with_brdf_map"][keep], 0.2)["max_rel_pixel"]
    rep["n_rays"] = int(ref["rgb_map"].shape[0])
    rep["n_hit"] = int((ref["acc_map"] > 0.5).sum())
    rep["ggx_normal_flip_rays"] = int((~keep).sum())
    if fp64_floor:
        # the reference's own fp32 noise floor on this checkpoint: the oracle's primary pass in fp64 against itself in fp32
        # (weight-threshold decisions, tensorBase_rotated_lights.py:924, flip on a sharp trained scene in ANY fp32 implementation)
        with torch.no_grad():
            o32 = O.forward_primary(sc, rays.cpu()[sel], lidx.cpu()[sel].to(torch.int32), n_samples, True, True, None, noise[sel], "aten")
            o64 = O.forward_primary(sc.to(torch.float64), rays.cpu()[sel].double(), lidx.cpu()[sel].to(torch.int32), n_samples, True, True, None,
                                    noise[sel].double(), "aten")
        rep["oracle_fp32_vs_fp64"] = {n: parity_metrics(o32[i], o64[i]) for i, n in enumerate(MAPS)}
    return rep


def trained(n_iters=450):
    from tests.train_sequence import reconstruct
    with contextlib.redirect_stdout(io.StringIO()):
        r = reconstruct("single_light", n_iters=n_iters, batch=1024, upsamp=(200, 300), mask_updates=(120, 250),
                        model_kw=dict(envmap_h=8, envmap_w=16))
    return r


def schedule_300(n_iters):
    """Mask updates / up-samplings of train_tensoIR.py's default schedule (10k / 15k masks, 10k .. 40k up-samplings of 80k iterations)
    compressed into n_iters, leaving the last 3/8 at the final grid."""
    f = n_iters / 2400.0
    return dict(mask_updates=(int(500 * f), int(800 * f)), upsamp=tuple(int(x * f) for x in (600, 900, 1200, 1500)))


def trained_300(n_iters=2400, batch=4096, views=12, res=160):
    """A checkpoint trained to 300^3 through the product API: 128^3 -> 300^3 in four up-samplings, two mask updates, shrink,
    relighting losses from the first mask update on (tools/precision_300.py records the full-length run)."""
    from tests.train_sequence import reconstruct
    with contextlib.redirect_stdout(io.StringIO()):
        return reconstruct("single_light", n_iters=n_iters, batch=batch, dataset=f"synthetic:views={views},res={res}", grid0=128, grid1=300,
                           model_kw=dict(envmap_h=8, envmap_w=16), **schedule_300(n_iters))


def trained_case(r, n_rays=2048, stride=2):
    """The trained model's own training rays (host tensors, as the script passes them) under the three policies + the oracle on every
    `stride`-th of them."""
    from tests.helpers import scene_from_model
    m = r.model
    rays = r.rays_f[:n_rays].cuda()
    lidx = r.lidx_f[:n_rays].cuda().to(torch.int32).reshape(-1, 1)
    S = int(m.nSamples)
    noise = torch.randn(rays.shape[0], S, 3, generator=torch.Generator().manual_seed(5))
    out, res = three_policies(m, rays, lidx, noise, S)
    to_cpu = lambda v: v.detach().cpu() if torch.is_tensor(v) else v
    ckpt = {"kwargs": {k: to_cpu(v) for k, v in m.get_kwargs().items()}, "state_dict": {k: to_cpu(v) for k, v in m.state_dict().items()}}
    sc = scene_from_model(ckpt, m, 8, 16)
    rep = oracle_compare(sc, out, res["auto"], rays, lidx, noise, S, slice(0, rays.shape[0], stride), fp64_floor=True)
    return res, rep


SWEEP = [dict(name="as initialised"),
         dict(name="planes x4", plane=4.0), dict(name="planes x16", plane=16.0), dict(name="planes x64", plane=64.0),
         dict(name="decoder x2", dec=2.0), dict(name="decoder x4", dec=4.0),
         dict(name="light rows x8", light=8.0),
         dict(name="planes x16, decoder x2, light x8", plane=16.0, dec=2.0, light=8.0),
         dict(name="planes x64, decoder x4, light x8", plane=64.0, dec=4.0, light=8.0),
         dict(name="planes x3000, lines x100 (fp16 range)", plane=3000.0, line=100.0, oracle=False)]


def sweep_case(cfg, grid=128, oracle_rays=128):
    """One adversarially scaled scene: forced-f16 / full / auto renders of the 4096-ray batch + oracle on a subsample."""
    import tensoir_amd
    from tensoir_amd import synth
    from tests.helpers import scene_from_model
    ck = synth.make_checkpoint(grid=(grid,) * 3, seed=20211202)
    sd = ck["state_dict"]
    for i in range(3):
        sd[f"app_plane.{i}"] = sd[f"app_plane.{i}"] * cfg.get("plane", 1.0)
        sd[f"app_line.{i}"] = sd[f"app_line.{i}"] * cfg.get("line", 1.0)
    for layer in (0, 2, 4):
        sd[f"renderModule.mlp.{layer}.weight"] = sd[f"renderModule.mlp.{layer}.weight"] * cfg.get("dec", 1.0)
    sd["light_line.weight"] = sd["light_line.weight"] * cfg.get("light", 1.0)
    model = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=8, envmap_w=16)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        model.updateAlphaMask((128, 128, 128))
    rays = synth.make_rays(64, 64).cuda()
    lidx = torch.zeros(4096, 1, dtype=torch.int32, device="cuda")
    S = 512 if grid >= 300 else int(model.nSamples)
    noise = torch.randn(4096, S, 3, generator=torch.Generator().manual_seed(7))
    out, res = three_policies(model, rays, lidx, noise, S)
    rep = None
    if oracle_rays:
        sc = scene_from_model(ck, model, 8, 16)
        rep = oracle_compare(sc, out, res["auto"], rays, lidx, noise, S, slice(0, 4096, 4096 // oracle_rays))
    rng = model.half_range()
    res["range_bound"] = None if rng is None else (rng.ok(), rng.bound)[1]
    return res, rep

"""CPU oracle for the TensoIR ray-march + PBR-shading hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tensoir_amd/`` may import this file;
it is imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the *checker* / CPU baseline, never as
the thing that is shipped or measured as the product.

This is a from-scratch functional restatement (plain torch CPU tensors, no
nn.Module) of the reference algorithm.  Every function cites the reference
``file:line`` it follows (paths relative to the TensoIR repository root).

Parity pin: the reference has no tests / golden vectors of its own (SURVEY.md
section 4), so this oracle is pinned against outputs of the *imported reference
itself* on seeded inputs: ``oracle/make_golden.py`` (run in the build container
where the reference checkout is present) writes ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks this file against them.

Two interpolation back-ends are provided:
  * ``"aten"``     -- F.grid_sample, the op the reference calls (fp32 speed path,
                      used for the CPU baseline timing);
  * ``"explicit"`` -- index/weight formulas written out (works in fp64, and is
                      the arithmetic the HIP kernels restate).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

# models/tensorBase_rotated_lights.py:398-399
MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


# --------------------------------------------------------------------------
# scene container
# --------------------------------------------------------------------------
class Scene(SimpleNamespace):
    """All tensors / scalars the hot path reads.

    Tensor shapes are the reference's state_dict shapes
    (models/tensoRF_rotated_lights.py:11-29, tensorBase_rotated_lights.py:405-488):
      density_plane[i] [1,Cd,grid[mat1],grid[mat0]]   density_line[i] [1,Cd,grid[vec],1]
      app_plane[i]     [1,Ca,...]                     app_line[i]     [1,Ca,grid[vec],1]
      basis_mat [app_dim, 3*Ca]   light_line [L, 3*Ca]
      mlp_rgb / mlp_brdf / mlp_normal : dict(w0,b0,w1,b1,w2,b2)
      lgtSGs [K,7]   light_rotation : list of degrees
      aabb [2,3]   grid : [gx,gy,gz]
      alpha_volume [Dz,Hy,Wx] float 0/1 or None, alpha_aabb [2,3]
    """

    def to(self, dtype):
        out = Scene(**self.__dict__)
        for k, v in self.__dict__.items():
            if torch.is_tensor(v) and v.is_floating_point():
                setattr(out, k, v.to(dtype))
            elif isinstance(v, list) and v and torch.is_tensor(v[0]):
                setattr(out, k, [t.to(dtype) for t in v])
            elif isinstance(v, dict) and v and torch.is_tensor(next(iter(v.values()))):
                setattr(out, k, {a: b.to(dtype) for a, b in v.items()})
        return out


def step_geometry(aabb, grid, step_ratio):
    """models/tensorBase_rotated_lights.py:608-619 (update_stepSize)."""
    aabb_size = aabb[1] - aabb[0]
    inv_aabb = 2.0 / aabb_size
    gs = torch.tensor(list(grid), dtype=torch.long)
    units = aabb_size / (gs - 1)
    step = torch.mean(units) * step_ratio
    diag = torch.sqrt(torch.sum(torch.square(aabb_size)))
    n_samples = int((diag / step).item()) + 1
    return SimpleNamespace(aabb_size=aabb_size, inv_aabb=inv_aabb, units=units,
                           step=step, diag=diag, n_samples=n_samples)


def scene_from_state_dict(sd, kwargs, alpha_volume=None, alpha_aabb=None,
                          envmap_h=16, envmap_w=32, fixed_fresnel=0.04):
    """Build a Scene from a reference checkpoint (state_dict + get_kwargs()).

    Parameter names: models/tensorBase_rotated_lights.py:675-692 (save/load),
    SURVEY.md section 5 (checkpoint row).
    """
    def mlp(prefix):
        return {"w0": sd[f"{prefix}.mlp.0.weight"], "b0": sd[f"{prefix}.mlp.0.bias"],
                "w1": sd[f"{prefix}.mlp.2.weight"], "b1": sd[f"{prefix}.mlp.2.bias"],
                "w2": sd[f"{prefix}.mlp.4.weight"], "b2": sd[f"{prefix}.mlp.4.bias"]}
    sd = {k: (v.detach().clone().float() if torch.is_tensor(v) and v.is_floating_point() else v)
          for k, v in sd.items()}
    sc = Scene(
        aabb=torch.as_tensor(kwargs["aabb"]).detach().clone().float().view(2, 3),
        grid=[int(g) for g in kwargs["gridSize"]],
        density_plane=[sd[f"density_plane.{i}"] for i in range(3)],
        density_line=[sd[f"density_line.{i}"] for i in range(3)],
        app_plane=[sd[f"app_plane.{i}"] for i in range(3)],
        app_line=[sd[f"app_line.{i}"] for i in range(3)],
        basis_mat=sd["basis_mat.weight"],
        light_line=sd["light_line.weight"],
        mlp_rgb=mlp("renderModule"),
        mlp_brdf=mlp("renderModule_brdf"),
        mlp_normal=mlp("renderModule_normal") if "renderModule_normal.mlp.0.weight" in sd else None,
        normals_kind=str(kwargs.get("normals_kind", "derived_plus_predicted")),
        lgtSGs=sd.get("lgtSGs"),
        light_kind=str(kwargs.get("light_kind", "sg")),
        light_rgbs_raw=sd.get("_light_rgbs"),         # light_kind == 'pixel': the [envmap_h * envmap_w, 3] map parameters
        light_rotation=[int(r) for r in kwargs["light_rotation"]],
        density_shift=float(kwargs["density_shift"]),
        distance_scale=float(kwargs["distance_scale"]),
        weight_thres=float(kwargs["rayMarch_weight_thres"]),
        near_far=[float(x) for x in kwargs["near_far"]],
        step_ratio=float(kwargs["step_ratio"]),
        pos_pe=int(kwargs["pos_pe"]), view_pe=int(kwargs["view_pe"]), fea_pe=int(kwargs["fea_pe"]),
        envmap_h=int(envmap_h), envmap_w=int(envmap_w),
        fixed_fresnel=float(fixed_fresnel),
        alpha_volume=None if alpha_volume is None else alpha_volume.detach().clone().float(),
        alpha_aabb=None if alpha_aabb is None else alpha_aabb.detach().clone().float().view(2, 3),
    )
    return sc


# --------------------------------------------------------------------------
# interpolation primitives (Appendix A of SURVEY.md; models/relight_utils.py:57-107)
# --------------------------------------------------------------------------
def _bilinear_explicit(img, u, v, want_grad=False):
    """img [C,H,W]; u indexes W, v indexes H; align_corners=True, zero padding.

    Index/weight formulas: models/relight_utils.py:64-79 (the reference's own
    restatement of F.grid_sample).  Out-of-range taps contribute zero
    (F.grid_sample padding_mode='zeros').
    Returns [C,N] (and d/du, d/dv [C,N] when want_grad).
    """
    C, H, W = img.shape
    ix = ((u + 1) / 2) * (W - 1)
    iy = ((v + 1) / 2) * (H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    tx = ix - x0
    ty = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    x1 = x0 + 1
    y1 = y0 + 1
    flat = img.reshape(C, H * W)

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))
        return flat[:, idx] * ok.to(img.dtype)

    a = tap(x0, y0)  # nw
    b = tap(x1, y0)  # ne
    c = tap(x0, y1)  # sw
    d = tap(x1, y1)  # se
    out = a * ((1 - tx) * (1 - ty)) + b * (tx * (1 - ty)) + c * ((1 - tx) * ty) + d * (tx * ty)
    if not want_grad:
        return out
    du = ((b - a) * (1 - ty) + (d - c) * ty) * ((W - 1) / 2)
    dv = ((c - a) * (1 - tx) + (d - b) * tx) * ((H - 1) / 2)
    return out, du, dv


def _linear_explicit(line, w, want_grad=False):
    """line [C,H]; grid_sample on a [1,C,H,1] image at (0,w): pure linear interp along H."""
    C, H = line.shape
    iy = ((w + 1) / 2) * (H - 1)
    y0 = torch.floor(iy)
    ty = iy - y0
    y0 = y0.long()
    y1 = y0 + 1

    def tap(yy):
        ok = (yy >= 0) & (yy < H)
        return line[:, yy.clamp(0, H - 1)] * ok.to(line.dtype)

    a = tap(y0)
    b = tap(y1)
    out = a * (1 - ty) + b * ty
    if not want_grad:
        return out
    return out, (b - a) * ((H - 1) / 2)


def sample_plane(plane, u, v, backend):
    """plane [1,C,H,W] sampled at (u->W, v->H): models/tensoRF_rotated_lights.py:104."""
    if backend == "aten":
        grid = torch.stack((u, v), dim=-1).view(1, -1, 1, 2)
        return F.grid_sample(plane, grid, align_corners=True).view(plane.shape[1], -1)
    return _bilinear_explicit(plane[0], u, v)


def sample_line(line, w, backend):
    """line [1,C,H,1] sampled at (0, w): models/tensoRF_rotated_lights.py:99-100,106."""
    if backend == "aten":
        grid = torch.stack((torch.zeros_like(w), w), dim=-1).view(1, -1, 1, 2)
        return F.grid_sample(line, grid, align_corners=True).view(line.shape[1], -1)
    return _linear_explicit(line[0, :, :, 0], w)


def sample_occupancy(sc, xyz, backend="aten"):
    """AlphaGridMask.sample_alpha: models/tensorBase_rotated_lights.py:112-119.

    Trilinear grid_sample (align_corners=True, zero padding) of the 0/1 volume
    [Dz,Hy,Wx] in the mask's own aabb; callers test ``> 0``.
    """
    aabb = sc.alpha_aabb
    inv = 1.0 / (aabb[1] - aabb[0]) * 2
    q = (xyz - aabb[0]) * inv - 1
    vol = sc.alpha_volume
    if backend == "aten":
        return F.grid_sample(vol.view(1, 1, *vol.shape), q.view(1, -1, 1, 1, 3),
                             align_corners=True).view(-1)
    D, H, W = vol.shape
    ix = ((q[:, 0] + 1) / 2) * (W - 1)
    iy = ((q[:, 1] + 1) / 2) * (H - 1)
    iz = ((q[:, 2] + 1) / 2) * (D - 1)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = torch.zeros_like(ix)
    flat = vol.reshape(-1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy, zz = x0 + dx, y0 + dy, z0 + dz
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
                idx = (zz.clamp(0, D - 1) * H + yy.clamp(0, H - 1)) * W + xx.clamp(0, W - 1)
                wgt = (tx if dx else 1 - tx) * (ty if dy else 1 - ty) * (tz if dz else 1 - tz)
                out = out + flat[idx] * ok.to(vol.dtype) * wgt
    return out


# --------------------------------------------------------------------------
# VM field (K2 / K4 / K6)
# --------------------------------------------------------------------------
def density_feature(sc, xyz, backend="aten"):
    """compute_densityfeature: models/tensoRF_rotated_lights.py:95-110.  xyz in [-1,1]^3."""
    f = torch.zeros(xyz.shape[0], dtype=xyz.dtype)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        p = sample_plane(sc.density_plane[i], xyz[:, m0], xyz[:, m1], backend)
        l = sample_line(sc.density_line[i], xyz[:, VEC_MODE[i]], backend)
        f = f + torch.sum(p * l, dim=0)
    return f


def feature2density(sc, f):
    """models/tensorBase_rotated_lights.py:813-817 (softplus branch, threshold 20)."""
    return F.softplus(f + sc.density_shift)


def density_grad(sc, xyz):
    """Analytic d sigma / d xyz (normalised coords) and the derived normal.

    Restates compute_derived_normals (models/tensorBase_rotated_lights.py:839-856)
    = autograd through compute_densityfeature_with_xyz_grad
    (models/tensoRF_rotated_lights.py:113-129) and the custom grid_sample
    (models/relight_utils.py:57-107), in closed form.
    """
    n = xyz.shape[0]
    f = torch.zeros(n, dtype=xyz.dtype)
    g = torch.zeros(n, 3, dtype=xyz.dtype)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        vi = VEC_MODE[i]
        p, dpu, dpv = _bilinear_explicit(sc.density_plane[i][0], xyz[:, m0], xyz[:, m1], True)
        l, dl = _linear_explicit(sc.density_line[i][0, :, :, 0], xyz[:, vi], True)
        f = f + torch.sum(p * l, dim=0)
        g[:, m0] = g[:, m0] + torch.sum(dpu * l, dim=0)
        g[:, m1] = g[:, m1] + torch.sum(dpv * l, dim=0)
        g[:, vi] = g[:, vi] + torch.sum(p * dl, dim=0)
    x = f + sc.density_shift
    dsig = torch.where(x > 20, torch.ones_like(x), torch.sigmoid(x))  # softplus'
    grad = dsig[:, None] * g
    normal = -grad / torch.clamp(torch.linalg.norm(grad, dim=-1, keepdim=True), min=1e-6)
    return F.softplus(x), grad, normal


def app_planeline(sc, xyz, backend="aten"):
    """The [3*Ca, N] plane*line product shared by compute_{app,both,intrin}feature
    (models/tensoRF_rotated_lights.py:132-224)."""
    ps, ls = [], []
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        ps.append(sample_plane(sc.app_plane[i], xyz[:, m0], xyz[:, m1], backend))
        ls.append(sample_line(sc.app_line[i], xyz[:, VEC_MODE[i]], backend))
    return torch.cat(ps) * torch.cat(ls)


def app_feature(sc, xyz, light_idx, backend="aten"):
    """compute_appfeature: models/tensoRF_rotated_lights.py:197-224."""
    pl = app_planeline(sc, xyz, backend)
    light = sc.light_line[light_idx.view(-1).long()].permute(1, 0)
    return F.linear((pl * light).T, sc.basis_mat)


def intrin_feature(sc, xyz, backend="aten"):
    """compute_intrinfeature: models/tensoRF_rotated_lights.py:167-195."""
    pl = app_planeline(sc, xyz, backend)
    mean_w = torch.mean(sc.light_line, dim=0).unsqueeze(-1)
    return F.linear((pl * mean_w).T, sc.basis_mat)


def both_feature(sc, xyz, light_idx, backend="aten"):
    """compute_bothfeature: models/tensoRF_rotated_lights.py:132-165."""
    pl = app_planeline(sc, xyz, backend)
    light = sc.light_line[light_idx.view(-1).long()].permute(1, 0)
    mean_w = torch.mean(sc.light_line, dim=0).unsqueeze(-1)
    return F.linear((pl * light).T, sc.basis_mat), F.linear((pl * mean_w).T, sc.basis_mat)


# --------------------------------------------------------------------------
# decoders (K5)
# --------------------------------------------------------------------------
def positional_encoding(x, freqs):
    """models/tensorBase_rotated_lights.py:12-17."""
    bands = (2 ** torch.arange(freqs)).to(x.dtype)
    y = (x[..., None] * bands).reshape(x.shape[:-1] + (freqs * x.shape[-1],))
    return torch.cat([torch.sin(y), torch.cos(y)], dim=-1)


def mlp_input(feat, aux, fea_pe, aux_pe):
    """[feat, aux, PE(feat), PE(aux)]: models/tensorBase_rotated_lights.py:137-142, 199-204."""
    parts = [feat, aux]
    if fea_pe > 0:
        parts.append(positional_encoding(feat, fea_pe))
    if aux_pe > 0:
        parts.append(positional_encoding(aux, aux_pe))
    return torch.cat(parts, dim=-1)


def mlp3(w, x):
    """Linear-ReLU-Linear-ReLU-Linear trunk: models/tensorBase_rotated_lights.py:129-133."""
    h = torch.relu(F.linear(x, w["w0"], w["b0"]))
    h = torch.relu(F.linear(h, w["w1"], w["b1"]))
    return F.linear(h, w["w2"], w["b2"])


def render_rgb(sc, viewdirs, feat):
    """MLPRender_Fea.forward: models/tensorBase_rotated_lights.py:136-146."""
    return torch.sigmoid(mlp3(sc.mlp_rgb, mlp_input(feat, viewdirs, sc.fea_pe, sc.view_pe)))


def render_brdf(sc, pts, feat):
    """MLPBRDF_PEandFeature (outc=4, sigmoid): :198-208, wired at :430-431."""
    return torch.sigmoid(mlp3(sc.mlp_brdf, mlp_input(feat, pts, sc.fea_pe, sc.pos_pe)))


def render_normal(sc, pts, feat):
    """MLPBRDF_PEandFeature (outc=3, tanh): :198-208, wired at :423-424."""
    return torch.tanh(mlp3(sc.mlp_normal, mlp_input(feat, pts, sc.fea_pe, sc.pos_pe)))


def render_normal_residue(sc, pts, derived, feat):
    """MLPNormal_normal_and_PExyz (outc=3, tanh): models/tensorBase_rotated_lights.py:253-262, wired at :426-428 -- the
    'residue_prediction' decoder; input order [pts, derived normal, features, PE(features), PE(pts)]."""
    parts = [pts, derived, feat]
    if sc.fea_pe > 0:
        parts.append(positional_encoding(feat, sc.fea_pe))
    if sc.pos_pe > 0:
        parts.append(positional_encoding(pts, sc.pos_pe))
    return torch.tanh(mlp3(sc.mlp_normal, torch.cat(parts, dim=-1)))


# --------------------------------------------------------------------------
# ray marching (K1, K3)
# --------------------------------------------------------------------------
def sample_ray(sc, rays_o, rays_d, n_samples, jitter=None):
    """TensorBase.sample_ray: models/tensorBase_rotated_lights.py:705-724.

    ``jitter`` [B,1] in [0,1) replaces the is_train torch.rand_like draw (:717).
    """
    geo = step_geometry(sc.aabb, sc.grid, sc.step_ratio)
    n = n_samples if n_samples > 0 else geo.n_samples
    near, far = sc.near_far
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (sc.aabb[1] - rays_o) / vec
    rate_b = (sc.aabb[0] - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
    rng = torch.arange(n)[None].to(rays_o.dtype)
    if jitter is not None:
        rng = rng.repeat(rays_d.shape[-2], 1)
        rng = rng + jitter
    step = geo.step.to(rays_o.dtype) * rng
    z = t_min[..., None] + step
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
    outside = ((sc.aabb[0] > pts) | (pts > sc.aabb[1])).any(dim=-1)
    return pts, z, ~outside


def normalize_coord(sc, xyz):
    """models/tensorBase_rotated_lights.py:640-641."""
    geo = step_geometry(sc.aabb, sc.grid, sc.step_ratio)
    return (xyz - sc.aabb[0]) * geo.inv_aabb.to(xyz.dtype) - 1


def raw2alpha(sigma, dist):
    """models/tensorBase_rotated_lights.py:21-28."""
    alpha = 1.0 - torch.exp(-sigma * dist)
    T = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1, dtype=alpha.dtype),
                                 1.0 - alpha + 1e-10], -1), -1)
    return alpha, alpha * T[:, :-1], T[:, -1:]


def march_sigma(sc, pts, valid, backend="aten"):
    """Occupancy cull + density for a [N,S,3] point set.

    models/tensorBase_rotated_lights.py:892-919 (and the identical blocks at
    models/relight_utils.py:683-695, 803-815).  Returns sigma [N,S], the refined
    valid mask and the normalised coordinates.
    """
    valid = valid.clone()
    if sc.alpha_volume is not None:
        occ = sample_occupancy(sc, pts[valid], backend) > 0
        invalid = ~valid
        invalid[valid] |= ~occ
        valid = ~invalid
    sigma = torch.zeros(pts.shape[:-1], dtype=pts.dtype)
    xyz_n = pts
    if valid.any():
        xyz_n = normalize_coord(sc, pts)
        sigma[valid] = feature2density(sc, density_feature(sc, xyz_n[valid], backend))
    return sigma, valid, xyz_n


def linear2srgb(x):
    """linear2srgb_torch: models/relight_utils.py:489-515 (after the [0,1] clip at :518-533)."""
    x = x.clamp(0, 1)
    lin = x * 12.92
    nonlin = 1.055 * torch.pow(x + 1e-6, 1 / 2.4) - (1.055 - 1)
    return torch.where(x <= 0.0031308, lin, nonlin)


def safe_l2_normalize(x, eps=1e-6):
    """models/relight_utils.py:13-14 / dataLoader/ray_utils.py:278-279."""
    return x / torch.clamp(torch.linalg.norm(x, dim=-1, keepdim=True), min=eps)


def relative_smoothness(a, b):
    """compute_relative_smoothness_loss: models/tensorBase_rotated_lights.py:858-863."""
    base = torch.maximum(a, b).clip(min=1e-6)
    return torch.sum(((a - b) / base) ** 2, dim=-1, keepdim=True)


def forward_primary(sc, rays, light_idx, n_samples=-1, white_bg=True, is_relight=True,
                    ray_jitter=None, brdf_jitter=None, backend="aten", return_aux=False):
    """TensorBase.forward: models/tensorBase_rotated_lights.py:868-1036.

    ``ray_jitter`` [B,1]: the is_train stratification draw (:717), None = eval.
    ``brdf_jitter``: dense [B,S,3] N(0,1) noise replacing torch.randn_like (:937);
    it is indexed with the app mask so every implementation sees the same noise
    for the same sample.  None -> torch.randn_like on the compacted points, i.e.
    exactly the reference's draw.
    Returns the reference's 12-tuple (:1033-1036 / :983-986).
    """
    dt = rays.dtype
    viewdirs = rays[:, 3:6]
    pts, z, valid = sample_ray(sc, rays[:, :3], viewdirs, n_samples, ray_jitter)
    B, S = z.shape[0] if z.shape[0] == rays.shape[0] else rays.shape[0], z.shape[1]
    z = z.expand(B, S)
    dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), dim=-1)
    vd = viewdirs.view(-1, 1, 3).expand(pts.shape)
    lidx = light_idx.view(-1, 1, 1).expand(B, S, 1)

    sigma, valid, xyz = march_sigma(sc, pts, valid, backend)
    alpha, weight, bg = raw2alpha(sigma, dists * sc.distance_scale)
    app_mask = weight > sc.weight_thres

    rgb = torch.zeros(B, S, 3, dtype=dt)
    normal = torch.zeros(B, S, 3, dtype=dt)
    albedo = torch.zeros(B, S, 3, dtype=dt)
    rough = torch.zeros(B, S, 1, dtype=dt)
    alb_cost = torch.zeros(B, S, 1, dtype=dt)
    rgh_cost = torch.zeros(B, S, 1, dtype=dt)
    ndiff = torch.zeros(B, S, 1, dtype=dt)
    norient = torch.zeros(B, S, 1, dtype=dt)
    if app_mask.any():
        xa = xyz[app_mask]
        rad_f, int_f = both_feature(sc, xa, lidx[app_mask], backend)
        rgb[app_mask] = render_rgb(sc, vd[app_mask], rad_f)
        if is_relight:
            brdf = render_brdf(sc, xa, int_f)
            va, vr = brdf[..., :3], brdf[..., 3:4] * 0.9 + 0.09
            albedo[app_mask] = va
            rough[app_mask] = vr
            noise = torch.randn_like(xa) if brdf_jitter is None else brdf_jitter[app_mask].to(dt)
            xj = xa + noise * 0.01
            brdf_j = render_brdf(sc, xj, intrin_feature(sc, xj, backend))
            vaj, vrj = brdf_j[..., :3], brdf_j[..., 3:4] * 0.9 + 0.09
            alb_cost[app_mask] = relative_smoothness(va, vaj)
            rgh_cost[app_mask] = relative_smoothness(vr, vrj)
            # the three normals kinds with kernels (:946-960); normals_diff / orientation are only filled in the
            # derived_plus_predicted branch and stay zero otherwise
            kind = getattr(sc, "normals_kind", "derived_plus_predicted")
            if kind == "purely_predicted":
                normal[app_mask] = render_normal(sc, xa, int_f)
            elif kind == "purely_derived":
                normal[app_mask] = density_grad(sc, xa)[2]
            elif kind == "derived_plus_predicted":
                _, _, derived = density_grad(sc, xa)
                pred = render_normal(sc, xa, int_f)
                normal[app_mask] = pred
                ndiff[app_mask] = torch.sum((pred - derived) ** 2, dim=-1, keepdim=True)
                norient[app_mask] = torch.sum(vd[app_mask] * pred, dim=-1, keepdim=True).clamp(min=0)
            elif kind == "residue_prediction":         # :962-968
                _, _, derived = density_grad(sc, xa)
                pred = render_normal_residue(sc, xa, derived, int_f)
                normal[app_mask] = pred
                ndiff[app_mask] = torch.sum((pred - derived) ** 2, dim=-1, keepdim=True)
                norient[app_mask] = torch.sum(vd[app_mask] * pred, dim=-1, keepdim=True).clamp(min=0)
            elif kind == "gt_normals":
                pass                                   # zeros: the caller substitutes the ground-truth normals (:951-952, renderer.py:82-83)
            else:
                raise ValueError(f"normals_kind {kind!r}")

    acc = torch.sum(weight, -1)
    depth = torch.sum(weight * z, -1)
    rgb_map = torch.sum(weight[..., None] * rgb, -2)
    aux = SimpleNamespace(weight=weight, sigma=sigma, valid=valid, app_mask=app_mask, z=z,
                          bg=bg, xyz=xyz)
    if not is_relight:
        if white_bg:
            depth = depth + (1.0 - acc) * rays[..., -1]
            rgb_map = rgb_map + (1.0 - acc[..., None])
        out = (rgb_map, depth, None, None, None, None, acc, None, None, None, None, None)
        return (out, aux) if return_aux else out

    normal_map = torch.sum(weight[..., None] * normal, -2)
    ndiff_map = torch.sum(weight[..., None] * ndiff, -2)
    norient_map = torch.sum(weight[..., None] * norient, -2)
    albedo_map = torch.sum(weight[..., None] * albedo, -2)
    rough_map = torch.sum(weight[..., None] * rough, -2)
    fresnel_map = torch.zeros_like(albedo_map).fill_(sc.fixed_fresnel)
    alb_loss = torch.mean(torch.sum(weight[..., None] * alb_cost, -2))
    rgh_loss = torch.mean(torch.sum(weight[..., None] * rgh_cost, -2))
    if white_bg:
        depth = depth + (1.0 - acc) * rays[..., -1]          # quirk: rays_d.z  (:1005)
        rgb_map = rgb_map + (1.0 - acc[..., None])
        normal_map = normal_map + (1 - acc[..., None]) * torch.tensor([0.0, 0.0, 1.0], dtype=dt)
        albedo_map = albedo_map + (1 - acc[..., None])
        rough_map = rough_map + (1 - acc[..., None])
        fresnel_map = fresnel_map + (1 - acc[..., None])
    rgb_map = rgb_map.clamp(0, 1)
    if rgb_map.shape[0] > 0:
        rgb_map = linear2srgb(rgb_map)
    albedo_map = albedo_map.clamp(0, 1)
    fresnel_map = fresnel_map.clamp(0, 1)
    rough_map = rough_map.clamp(0, 1)
    normal_map = safe_l2_normalize(normal_map)
    acc_mask = acc > 0.5
    out = (rgb_map, depth, normal_map, albedo_map, rough_map, fresnel_map, acc,
           ndiff_map, norient_map, acc_mask, alb_loss, rgh_loss)
    return (out, aux) if return_aux else out


# --------------------------------------------------------------------------
# environment light (a14, a15)
# --------------------------------------------------------------------------
def envmap_dirs(envmap_h, envmap_w, jitter=None):
    """generate_envir_map_dir / gen_light_incident_dirs('stratified_sampling'):
    models/tensorBase_rotated_lights.py:435-453, :511-526.

    jitter = (u_phi, u_theta) uniform [0,1) [H,W] tensors (the two rand_like draws of :520).
    Returns light_area_weight [H*W], dirs [H*W,3].
    """
    lat = np.pi / envmap_h
    lng = 2 * np.pi / envmap_w
    phi, theta = torch.meshgrid([
        torch.linspace(np.pi / 2 - 0.5 * lat, -np.pi / 2 + 0.5 * lat, envmap_h),
        torch.linspace(np.pi - 0.5 * lng, -np.pi + 0.5 * lng, envmap_w)], indexing="ij")
    sin_phi = torch.sin(torch.pi / 2 - phi)
    area = (4 * torch.pi * sin_phi / torch.sum(sin_phi)).to(torch.float32).reshape(-1)
    if jitter is not None:
        phi = phi + lat * (jitter[0] - 0.5)
        theta = theta + lng * (jitter[1] - 0.5)
    dirs = torch.stack([torch.cos(theta) * torch.cos(phi),
                        torch.sin(theta) * torch.cos(phi),
                        torch.sin(phi)], dim=-1).view(-1, 3)
    return area, dirs


def light_rotation_matrices(rot_deg):
    """models/tensorBase_rotated_lights.py:478-488."""
    mats = []
    for r in rot_deg:
        a = torch.tensor(r / 180 * torch.pi).to(torch.float32)
        mats.append(torch.tensor([[torch.cos(a), -torch.sin(a), 0],
                                  [torch.sin(a), torch.cos(a), 0],
                                  [0, 0, 1]]).to(torch.float32))
    return torch.stack(mats, dim=0)


def sg_radiance(lgtSGs, dirs):
    """render_envmap_sg: models/tensorBase_rotated_lights.py:70-86.  dirs [...,3] -> [...,3]."""
    v = dirs.unsqueeze(-2)
    lobes = lgtSGs[:, :3] / torch.norm(lgtSGs[:, :3], dim=-1, keepdim=True)
    lam = torch.abs(lgtSGs[:, 3:4])
    mu = torch.abs(lgtSGs[:, -3:])
    rgb = mu * torch.exp(lam * (torch.sum(v * lobes, dim=-1, keepdim=True) - 1.0))
    return torch.sum(rgb, dim=-2)


def light_rgbs(sc, dirs):
    """get_light_rgbs (light_kind == 'sg'): models/tensorBase_rotated_lights.py:577-606.
    dirs [D,3] -> [L,D,3].  With ``sc.lgtSGs_list`` set: the general multi-light variant, one SG set per light and
    no rotation (models/tensorBase_general_multi_lights.py:566-582)."""
    sg_list = getattr(sc, "lgtSGs_list", None)
    if sg_list is not None:
        d = dirs.reshape(-1, 3)
        return torch.stack([sg_radiance(sg.to(dirs.dtype), d) for sg in sg_list], dim=0)
    rot = light_rotation_matrices(sc.light_rotation).to(dirs.dtype)
    remapped = torch.matmul(dirs.reshape(1, -1, 3), rot).reshape(-1, 3)
    if getattr(sc, "light_kind", "sg") in ("pixel", "gt"):
        # models/tensorBase_rotated_lights.py:589-605: softplus(beta=5) map ('pixel') or the data set's probe as it is ('gt',
        # `sc.light_probe` [envmap_h * envmap_w, 3]), equirectangular lookup, align_corners=False
        img = F.softplus(sc.light_rgbs_raw.to(dirs.dtype), beta=5) if sc.light_kind == "pixel" else sc.light_probe.to(dirs.dtype)
        env = img.reshape(sc.envmap_h, sc.envmap_w, 3).permute(2, 0, 1).unsqueeze(0)
        phi = torch.arccos(remapped[:, 2]).reshape(-1) - 1e-6
        theta = torch.atan2(remapped[:, 1], remapped[:, 0]).reshape(-1)
        grid = torch.stack((-theta / math.pi, (phi / math.pi) * 2 - 1)).permute(1, 0).unsqueeze(0).unsqueeze(0)
        out = F.grid_sample(env, grid, align_corners=False).squeeze(0).squeeze(1).permute(1, 0)
        return out.reshape(len(sc.light_rotation), -1, 3)
    return sg_radiance(sc.lgtSGs.to(dirs.dtype), remapped).reshape(len(sc.light_rotation), -1, 3)


# --------------------------------------------------------------------------
# secondary rays (K7) and shading (K8)
# --------------------------------------------------------------------------
def sample_ray_equally(sc, rays_o, rays_d, n_sample, near, far):
    """models/relight_utils.py:707-722."""
    t = torch.linspace(0.0, 1.0, n_sample, dtype=rays_o.dtype)
    z = (near * (1.0 - t) + far * t).unsqueeze(0)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z.view(1, -1, 1)
    outside = ((sc.aabb[0] > pts) | (pts > sc.aabb[1])).any(dim=-1)
    return pts, z, ~outside


def compute_transmittance(sc, surf_pts, light_dir, n_sample=128, near=0.1, far=2.0,
                          backend="aten"):
    """models/relight_utils.py:657-705.  Returns (T_end, 1-acc)."""
    pts, z, valid = sample_ray_equally(sc, surf_pts, light_dir, n_sample, near, far)
    dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), dim=-1)
    sigma, valid, _ = march_sigma(sc, pts, valid, backend)
    _, weight, T = raw2alpha(sigma, dists * sc.distance_scale)
    return T.squeeze(-1), 1 - torch.sum(weight, -1)


def compute_radiance(sc, surf_pts, light_dir, light_idx, n_sample=128, near=0.05, far=1.5,
                     backend="aten"):
    """models/relight_utils.py:777-834.  Returns (T_end, 1-acc, indirect[N,3])."""
    pts, z, valid = sample_ray_equally(sc, surf_pts, light_dir, n_sample, near, far)
    dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), dim=-1)
    N, S = pts.shape[:2]
    lidx = light_idx.view(-1, 1, 1).expand(N, S, 1)
    vd = light_dir.view(-1, 1, 3).expand(pts.shape)
    sigma, valid, xyz = march_sigma(sc, pts, valid, backend)
    _, weight, T = raw2alpha(sigma, dists * sc.distance_scale)
    ind = torch.zeros(N, S, 3, dtype=pts.dtype)
    app_mask = weight > sc.weight_thres
    if app_mask.any():
        feat = app_feature(sc, xyz[app_mask], lidx[app_mask], backend)
        ind[app_mask] = render_rgb(sc, vd[app_mask], feat)
    acc = torch.sum(weight, -1)
    return T.squeeze(-1), 1 - acc, torch.sum(weight[..., None] * ind, -2)


def ggx_specular(normal, pts2c, pts2l, roughness, fresnel):
    """GGX_specular: models/relight_utils.py:17-50."""
    L = F.normalize(pts2l, dim=-1)
    V = F.normalize(pts2c, dim=-1)
    H = F.normalize((L + V[:, None, :]) / 2.0, dim=-1)
    N = F.normalize(normal, dim=-1)
    NoV = torch.sum(V * N, dim=-1, keepdim=True)
    N = N * NoV.sign()
    NoL = torch.sum(N[:, None, :] * L, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoV = torch.sum(N * V, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoH = torch.sum(N[:, None, :] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    VoH = torch.sum(V[:, None, :] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    alpha = roughness * roughness
    alpha2 = alpha * alpha
    k = (alpha + 2 * roughness + 1.0) / 8.0
    FMi = ((-5.55473) * VoH - 6.98316) * VoH
    frac0 = fresnel[:, None, :] + (1 - fresnel[:, None, :]) * torch.pow(2.0, FMi)
    frac = frac0 * alpha2[:, None, :]
    nom0 = NoH * NoH * (alpha2[:, None, :] - 1) + 1
    nom1 = NoV * (1 - k) + k
    nom2 = NoL * (1 - k[:, None, :]) + k[:, None, :]
    nom = (4 * np.pi * nom0 * nom0 * nom1[:, None, :] * nom2).clamp(1e-6, 4 * np.pi)
    return frac / nom


def render_with_brdf(sc, depth, normal, albedo, roughness3, fresnel, rays, light_idx,
                     n_sample=96, near=0.05, far=1.5, dir_jitter=None, backend="aten",
                     use_srgb=True, return_aux=False, chunk_size=15000):
    """render_with_BRDF: models/relight_utils.py:403-483 (sample_method
    'fixed_envirmap', or 'stratified_sampling' when dir_jitter is given)."""
    rays_o, rays_d = rays[..., :3], rays[..., 3:]
    surf = rays_o + depth.unsqueeze(-1) * rays_d
    area, _ = envmap_dirs(sc.envmap_h, sc.envmap_w)
    _, dirs = envmap_dirs(sc.envmap_h, sc.envmap_w, dir_jitter)
    dirs = dirs.to(rays.dtype)
    area = area.to(rays.dtype)
    M, D = surf.shape[0], dirs.shape[0]
    surf2l = dirs.reshape(1, -1, 3).repeat(M, 1, 1)
    surf2c = safe_l2_normalize(-rays_d)
    cosine = torch.einsum("ijk,ik->ij", surf2l, normal).clamp(min=0.0)
    cmask = cosine > 1e-6
    vis = torch.zeros(M, D, 1, dtype=rays.dtype)
    ind = torch.zeros(M, D, 3, dtype=rays.dtype)
    if cmask.any():
        with torch.no_grad():      # compute_secondary_shading_effects is @torch.no_grad (models/relight_utils.py:344)
            # ... and walks the masked (point, direction) pairs in chunks (:373-391; renderer.py:97 hands its chunk_size
            # down).  Every pair is independent, so the chunking changes no value -- only the size of the temporaries (the
            # whole 4096 x 128 pair set at once is ~2x slower on the host cores than the reference's chunks).
            p_all, l_all = surf.unsqueeze(1).expand(-1, D, -1)[cmask], surf2l[cmask]
            li_all = light_idx.view(-1, 1, 1).expand(M, D, 1)[cmask]
            v = torch.zeros(p_all.shape[0], dtype=rays.dtype)
            i = torch.zeros(p_all.shape[0], 3, dtype=rays.dtype)
            for c in torch.split(torch.arange(p_all.shape[0]), int(chunk_size)):
                v[c], _, i[c] = compute_radiance(sc, p_all[c], l_all[c], li_all[c], n_sample, near, far, backend)
        vis[cmask] = v.reshape(-1, 1)
        ind[cmask] = i
    spec = ggx_specular(normal, surf2c, surf2l, roughness3, fresnel)
    brdf = albedo.unsqueeze(1).expand(-1, D, -1) / np.pi + spec
    env = light_rgbs(sc, dirs)
    direct = torch.index_select(env, 0, light_idx.view(-1).long())
    light = vis * direct + ind
    rgb = torch.sum(brdf * light * cosine[:, :, None] * area[None, :, None], dim=1)
    rgb = rgb.clamp(0.0, 1.0)
    if use_srgb and rgb.shape[0] > 0:
        rgb = linear2srgb(rgb)
    if return_aux:
        return rgb, SimpleNamespace(vis=vis, indirect=ind, cosine=cosine, env=env, spec=spec)
    return rgb


def renderer_train(sc, rays, light_idx, n_samples=-1, white_bg=True, is_relight=True,
                   second_n_sample=96, second_near=0.05, second_far=1.5,
                   ray_jitter=None, brdf_jitter=None, dir_jitter=None, backend="aten", normal_gt=None, chunk_size=160000):
    """Renderer_TensoIR_train: renderer.py:57-127.  Returns the 12-key dict.  chunk_size: pairs per secondary-pass chunk
    (renderer.py:68 / :97; the training scripts pass args.relight_chunk_size = 160000, opt.py)."""
    light_idx = light_idx.to(torch.int32)
    (rgb_map, depth, normal, albedo, rough, fresnel, acc, ndiff, norient, acc_mask,
     alb_loss, rgh_loss) = forward_primary(sc, rays, light_idx, n_samples, white_bg, is_relight,
                                           ray_jitter, brdf_jitter, backend)
    if getattr(sc, "normals_kind", "") == "gt_normals" and normal_gt is not None:      # renderer.py:82-83
        normal = normal_gt.to(rgb_map.dtype)
    if is_relight:
        masked = render_with_brdf(sc, depth[acc_mask], normal[acc_mask], albedo[acc_mask],
                                  rough[acc_mask].repeat(1, 3), fresnel[acc_mask],
                                  rays[acc_mask], light_idx[acc_mask], second_n_sample,
                                  second_near, second_far, dir_jitter, backend, chunk_size=chunk_size)
        rgb_brdf = torch.ones_like(rgb_map)
        rgb_brdf[acc_mask] = masked
    else:
        rgb_brdf = torch.ones_like(rgb_map)
    return {"rgb_map": rgb_map, "depth_map": depth, "normal_map": normal,
            "albedo_map": albedo, "acc_map": acc, "roughness_map": rough,
            "fresnel_map": fresnel, "rgb_with_brdf_map": rgb_brdf,
            "normals_diff_map": ndiff, "normals_orientation_loss_map": norient,
            "albedo_smoothness_loss": alb_loss, "roughness_smoothness_loss": rgh_loss}


# --------------------------------------------------------------------------
# HDR-map relighting (K9): scripts/relight_importance.py:115-171
# --------------------------------------------------------------------------
def envlight_tables(hdr):
    """Environment_Light.__init__: models/relight_utils.py:110-148.  hdr [H,W,3].
    Returns pdf_sample [H*W], pdf_return [H*W], dirs [H*W,3]."""
    H, W, _ = hdr.shape
    inten = torch.sum(hdr, dim=2, keepdim=True)
    h_int = 1.0 / H
    sin_t = torch.sin(torch.linspace(0 + 0.5 * h_int, np.pi - 0.5 * h_int, H))
    pdf = inten * sin_t.view(-1, 1, 1)
    pdf = pdf / torch.sum(pdf)
    pdf_ret = pdf * H * W / (2 * np.pi * np.pi * sin_t.view(-1, 1, 1))
    _, dirs = envmap_dirs(H, W)
    return pdf.reshape(-1), pdf_ret.reshape(-1), dirs


def envlight_lookup(hdr, d):
    """Environment_Light.get_light: models/relight_utils.py:191-205 (align_corners=True)."""
    env = hdr.permute(2, 0, 1).unsqueeze(0)
    phi = torch.arccos(d[:, 2]).reshape(-1) - 1e-6
    theta = torch.atan2(d[:, 1], d[:, 0]).reshape(-1)
    qy = (phi / np.pi) * 2 - 1
    qx = -theta / np.pi
    grid = torch.stack((qx, qy)).permute(1, 0).unsqueeze(0).unsqueeze(0)
    return F.grid_sample(env, grid, align_corners=True).squeeze().permute(1, 0).reshape(-1, 3)


def relight_importance(sc, surf, normal, albedo, roughness1, fresnel, rays_d,
                       light_dir, light_rgb, light_pdf, n_sample=96, near=0.05, far=1.5,
                       backend="aten"):
    """Loop body of scripts/relight_importance.py:119-170 for one env map, given the
    importance samples (light_dir/rgb/pdf [M,Ns,*]) drawn by sample_light (:119)."""
    M, Ns = light_dir.shape[:2]
    surf2c = safe_l2_normalize(-rays_d)
    cosine = torch.einsum("ijk,ik->ij", light_dir, normal)
    cmask = cosine > 1e-6
    vis = torch.zeros(M, Ns, 1, dtype=surf.dtype)
    if cmask.any():
        v, _ = compute_transmittance(sc, surf[:, None, :].expand(M, Ns, 3)[cmask], light_dir[cmask],
                                     n_sample, near, far, backend)
        vis[cmask] = v.unsqueeze(-1)
    spec = ggx_specular(normal, surf2c, light_dir, roughness1, fresnel)
    brdf = albedo.unsqueeze(1).expand(-1, Ns, -1) / np.pi + spec
    contrib = brdf * (vis * light_rgb) * cosine[:, :, None] / light_pdf
    rgb = torch.mean(contrib, dim=1).clamp(0.0, 1.0)
    if rgb.shape[0] > 0:
        rgb = linear2srgb(rgb)
    return rgb


# --------------------------------------------------------------------------
# occupancy maintenance (used to build synthetic scenes)
# --------------------------------------------------------------------------
def dense_alpha(sc, grid_size, backend="aten"):
    """getDenseAlpha + compute_alpha: models/tensorBase_rotated_lights.py:737-753, 819-837."""
    geo = step_geometry(sc.aabb, sc.grid, sc.step_ratio)
    g = grid_size
    samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, g[0]), torch.linspace(0, 1, g[1]),
                                         torch.linspace(0, 1, g[2]), indexing="ij"), -1)
    dense = sc.aabb[0] * (1 - samples) + sc.aabb[1] * samples
    alpha = torch.zeros_like(dense[..., 0])
    for i in range(g[0]):
        loc = dense[i].view(-1, 3)
        if sc.alpha_volume is not None:
            m = sample_occupancy(sc, loc, backend) > 0
        else:
            m = torch.ones(loc.shape[0], dtype=torch.bool)
        sig = torch.zeros(loc.shape[0])
        if m.any():
            sig[m] = feature2density(sc, density_feature(sc, normalize_coord(sc, loc[m]), backend))
        alpha[i] = (1 - torch.exp(-sig * geo.step)).view(g[1], g[2])
    return alpha, dense


def update_alpha_mask(sc, grid_size=(200, 200, 200), thres=0.001, backend="aten"):
    """updateAlphaMask: models/tensorBase_rotated_lights.py:755-779.  Mutates sc."""
    alpha, dense = dense_alpha(sc, grid_size, backend)
    dense = dense.transpose(0, 2).contiguous()
    alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
    alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(tuple(grid_size)[::-1])
    alpha = (alpha >= thres).float()
    sc.alpha_volume = alpha
    sc.alpha_aabb = sc.aabb.clone()
    valid = dense[alpha > 0.5]
    return torch.stack((valid.amin(0), valid.amax(0)))


def filtering_rays(sc, all_rays, n_samples=256, bbox_only=False, backend="aten"):
    """filtering_rays: models/tensorBase_rotated_lights.py:781-811.  Returns the boolean keep mask [N]."""
    rays_o, rays_d = all_rays[..., :3], all_rays[..., 3:6]
    if bbox_only:
        vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
        rate_a = (sc.aabb[1] - rays_o) / vec
        rate_b = (sc.aabb[0] - rays_o) / vec
        return torch.maximum(rate_a, rate_b).amin(-1) > torch.minimum(rate_a, rate_b).amax(-1)
    pts, _, _ = sample_ray(sc, rays_o, rays_d, n_samples)
    occ = sample_occupancy(sc, pts.reshape(-1, 3), backend).view(pts.shape[:-1])
    return (occ > 0).any(-1)


# --------------------------------------------------------------------------
# training step: loss of train_tensoIR.py:262-311 and its parameter gradients (autograd on the
# functional restatement above) -- the checker for the HIP backward kernels
# --------------------------------------------------------------------------
TRAIN_WEIGHTS = dict(rgb_brdf=0.2, normals_diff=0.05, normals_orientation=0.1,
                     albedo_smoothness=0.1, roughness_smoothness=0.1)


def scene_parameters(sc):
    """name -> tensor for every trainable tensor of the scene, named like the reference's state_dict
    (models/tensorBase_rotated_lights.py:675-692)."""
    ps = {}
    for i in range(3):
        ps[f"density_plane.{i}"] = sc.density_plane[i]
        ps[f"density_line.{i}"] = sc.density_line[i]
        ps[f"app_plane.{i}"] = sc.app_plane[i]
        ps[f"app_line.{i}"] = sc.app_line[i]
    ps["basis_mat.weight"] = sc.basis_mat
    ps["light_line.weight"] = sc.light_line
    for prefix, m in (("renderModule", sc.mlp_rgb), ("renderModule_brdf", sc.mlp_brdf),
                      ("renderModule_normal", sc.mlp_normal)):
        if m is None:                     # normals_kind == 'purely_derived' has no normal decoder (:417-419)
            continue
        for j, k in ((0, "0"), (1, "2"), (2, "4")):
            ps[f"{prefix}.mlp.{k}.weight"] = m[f"w{j}"]
            ps[f"{prefix}.mlp.{k}.bias"] = m[f"b{j}"]
    if sc.lgtSGs is not None:
        ps["lgtSGs"] = sc.lgtSGs
    if getattr(sc, "light_rgbs_raw", None) is not None:
        ps["_light_rgbs"] = sc.light_rgbs_raw
    for i, sg in enumerate(getattr(sc, "lgtSGs_list", None) or []):     # general multi-light variant: a plain python list
        ps[f"lgtSGs_list.{i}"] = sg                                       # (models/tensorBase_general_multi_lights.py:463-479)
    return ps


def training_loss(ret, rgb_gt, is_relight, weights=None):
    """train_tensoIR.py:262-311 with the enhance ratios at 1 and the parameter regularisers off."""
    w = dict(TRAIN_WEIGHTS if weights is None else weights)
    loss = torch.mean((ret["rgb_map"] - rgb_gt) ** 2)
    if is_relight:
        loss = loss + w["rgb_brdf"] * torch.mean((ret["rgb_with_brdf_map"] - rgb_gt) ** 2)
        loss = loss + w["normals_diff"] * ret["normals_diff_map"].mean()
        loss = loss + w["normals_orientation"] * ret["normals_orientation_loss_map"].mean()
        loss = loss + w["roughness_smoothness"] * ret["roughness_smoothness_loss"]
        loss = loss + w["albedo_smoothness"] * ret["albedo_smoothness_loss"]
    return loss


def train_step_grads(sc, rays, light_idx, rgb_gt, is_relight=True, n_samples=-1, white_bg=True,
                     ray_jitter=None, brdf_jitter=None, dir_jitter=None, second_n_sample=96,
                     second_near=0.05, second_far=1.5, weights=None, backend="aten", normal_gt=None):
    """One training forward/backward on the oracle: returns (loss, {name: grad}, ret).  Leaves `sc` untouched."""
    work = Scene(**sc.__dict__)
    leaf = {}

    def mk(t):
        v = t.detach().clone().requires_grad_(True)
        return v
    for name in ("density_plane", "density_line", "app_plane", "app_line"):
        setattr(work, name, [mk(t) for t in getattr(sc, name)])
    work.basis_mat, work.light_line = mk(sc.basis_mat), mk(sc.light_line)
    work.lgtSGs = None if sc.lgtSGs is None else mk(sc.lgtSGs)
    if getattr(sc, "light_rgbs_raw", None) is not None:
        work.light_rgbs_raw = mk(sc.light_rgbs_raw)
    if getattr(sc, "lgtSGs_list", None) is not None:
        work.lgtSGs_list = [mk(t) for t in sc.lgtSGs_list]
    for name in ("mlp_rgb", "mlp_brdf", "mlp_normal"):
        if getattr(sc, name) is not None:
            setattr(work, name, {k: mk(v) for k, v in getattr(sc, name).items()})
    leaf = scene_parameters(work)
    with torch.enable_grad():
        ret = renderer_train(work, rays, light_idx, n_samples, white_bg, is_relight, second_n_sample,
                             second_near, second_far, ray_jitter, brdf_jitter, dir_jitter, backend, normal_gt)
        loss = training_loss(ret, rgb_gt, is_relight, weights)
    grads = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
    out = {n: (torch.zeros_like(p) if g is None else g) for (n, p), g in zip(leaf.items(), grads)}
    return loss.detach(), out, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ret.items()}

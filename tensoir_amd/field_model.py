"""Host-side mirror of the reference's field model for the hot path.

Same class names, constructor kwargs, parameter names/shapes (reference checkpoints load unchanged)
and method signatures as ``models/tensoRF_rotated_lights.py`` + ``models/tensorBase_rotated_lights.py``;
every per-sample computation is dispatched to libtensoir_hip.so (see tensoir_amd/ops.py).  What stays
in PyTorch is what SURVEY.md section 8 leaves there: parameter containers, regularisers,
up-sampling / shrinking, checkpoint I/O.
"""
from __future__ import annotations

import functools
import math
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import TensoirHipError, TirField

MAT_MODE = [[0, 1], [0, 2], [1, 2]]
VEC_MODE = [2, 1, 0]


def raw2alpha(sigma, dist):
    """models/tensorBase_rotated_lights.py:21-28 (kept for `from models... import raw2alpha` callers;
    the kernels implement the same recurrence with a wavefront prefix product)."""
    alpha = 1.0 - torch.exp(-sigma * dist)
    T = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1).to(alpha.device), 1.0 - alpha + 1e-10], -1), -1)
    weights = alpha * T[:, :-1]
    return alpha, weights, T[:, -1:]


def safe_l2_normalize(x, dim=None, eps=1e-6):
    return F.normalize(x, p=2, dim=dim, eps=eps)


def _host_values(owner, name, tensors, build):
    """Host copies of small device tensors (aabb, grid size, step size ...) without a device round trip per call:
    `.tolist()` / `float()` on a device tensor waits for everything queued before it, which stalls the launch queue
    once per step.  The copy is redone only when a tensor object, its storage or its version counter changes."""
    key = tuple((id(t), t.data_ptr(), t._version) for t in tensors)
    cache = owner.__dict__.get("_host_cache")
    if cache is None:
        cache = owner.__dict__["_host_cache"] = {}
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = cache[name] = (key, build(), tensors)      # keeps the tensors alive: ids stay unique
    return hit[1]


def channel_last(t):
    """[1,C,H,W] values stored as [H,W,C] in memory -- the layout the kernels gather from.  VM plane / line
    parameters are created like this, so a parameter IS its packed form: no shadow copy after every optimizer step,
    and the scatter kernels' channel-last gradients satisfy autograd's layout contract without a transposing copy."""
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def is_channel_last(p):
    if p.dim() != 4 or p.shape[0] != 1 or p.dtype != torch.float32:
        return False
    _, c, h, w = p.shape
    _, sc, sh, sw = p.stride()
    return (c == 1 or sc == 1) and (w == 1 or sw == c) and (h == 1 or sh == w * c)


def _no_grad_only(what, *tensors):
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            f"{what}: backward kernels are not implemented yet (SURVEY.md section 8f-1); "
            "call under torch.no_grad() -- tensoir_amd never falls back to eager PyTorch")


class _DensityL1Fn(torch.autograd.Function):
    """sum_i mean(|t_i|) and its gradient sign(t_i) / numel_i (zero where t_i is zero, as torch's abs backward) with
    multi-tensor kernels (see TensorVMSplit.density_L1)."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.save_for_backward(*ts)
        sums = torch._foreach_norm([t.detach() for t in ts], 1)            # sum |x| per tensor
        torch._foreach_div_(sums, [float(t.numel()) for t in ts])
        return torch.stack(sums).sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ts = ctx.saved_tensors
        grads = torch._foreach_sign(list(ts))
        torch._foreach_mul_(grads, [1.0 / float(t.numel()) for t in ts])
        torch._foreach_mul_(grads, g)
        return tuple(grads)


class AlphaGridMask(nn.Module):
    """models/tensorBase_rotated_lights.py:100-119."""

    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.aabb = aabb.to(self.device)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invgridSize = 1.0 / self.aabbSize * 2
        self.alpha_volume = alpha_volume.view(1, 1, *alpha_volume.shape[-3:])
        self.gridSize = torch.LongTensor([alpha_volume.shape[-1], alpha_volume.shape[-2],
                                          alpha_volume.shape[-3]]).to(self.device)
        self._bits = None

    def bits(self):
        if self._bits is None:
            self._bits = ops.pack_occupancy(self.alpha_volume.to(torch.float32))
        return self._bits

    def _descriptor(self):
        f = TirField()
        f.occ_nbr = self.bits().data_ptr()
        D, H, W = self.alpha_volume.shape[-3:]
        f.occ_dim[:] = [W, H, D]
        f.occ_aabb_min[:], f.occ_inv[:] = self.host_geometry()
        return f

    def host_geometry(self):
        return _host_values(self, "geom", (self.aabb, self.invgridSize),
                            lambda: (self.aabb[0].tolist(), self.invgridSize.tolist()))

    def occupied_box(self):
        """World-space box (lo[3], hi[3]) outside of which sample_alpha is 0: a point is reported occupied when one of
        the 8 voxels around it is (trilinear lookup > 0, :112-119), i.e. only within one cell of an occupied voxel; the box
        is the occupied voxels' index extent grown by 1.25 cells (the quarter cell covers fp32 rounding of positions many
        times over).  Read back once per mask (three tiny reductions); an empty mask gives an empty box."""
        def compute():
            vol = self.alpha_volume.reshape(self.alpha_volume.shape[-3:]) > 0          # [D, H, W]
            lo, hi = [], []
            amin, size = self.aabb[0].tolist(), self.aabbSize.tolist()
            for axis, dims in ((0, (0, 1)), (1, (0, 2)), (2, (1, 2))):              # x <- W, y <- H, z <- D
                n = vol.shape[2 - axis]
                idx = torch.nonzero(vol.any(dim=dims).reshape(-1))
                cell = size[axis] / max(n - 1, 1)
                if idx.numel() == 0:
                    lo.append(1.0); hi.append(-1.0)
                    continue
                i0, i1 = int(idx.min()), int(idx.max())
                lo.append(amin[axis] + (i0 - 1.25) * cell)
                hi.append(amin[axis] + (i1 + 1.25) * cell)
            return lo, hi
        return _host_values(self, "occ_box", (self.aabb, self.alpha_volume), compute)

    def sample_alpha(self, xyz_sampled):
        """Returns 1.0 where the reference's trilinear lookup is > 0 and 0.0 elsewhere (callers only
        ever test ``> 0``: :804, :823, :893-894)."""
        hit = ops.occupancy_query(self._descriptor(), xyz_sampled.reshape(-1, 3))
        return hit.to(torch.float32)

    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invgridSize - 1


class _Decoder(nn.Module):
    """Parameter container of one 150->128->128->out MLP with the reference's module layout
    (``.mlp.{0,2,4}``), evaluated by tir_mlp_fwd."""

    def __init__(self, in_chanel, pe_aux, feape, feature_c, outc, act, extra_in=0):
        super().__init__()
        self.in_mlpC = 2 * pe_aux * 3 + 2 * feape * in_chanel + 3 + in_chanel + extra_in
        self.feape = feape
        self.pe_aux = pe_aux
        self.outc = outc
        self.in_chanel = in_chanel
        self._act = act
        l1 = nn.Linear(self.in_mlpC, feature_c)
        l2 = nn.Linear(feature_c, feature_c)
        l3 = nn.Linear(feature_c, outc)
        self.mlp = nn.Sequential(l1, nn.ReLU(inplace=True), l2, nn.ReLU(inplace=True), l3)
        nn.init.constant_(self.mlp[-1].bias, 0)
        self._packed = None
        self._key = None

    def packed(self):
        mods = self.__dict__["_modules"]["mlp"]._modules            # (plain dictionary lookups: see _field_params)
        ps = [mods[i]._parameters[n] for i in ("0", "2", "4") for n in ("weight", "bias")]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._key:
            if self.feape != self.pe_aux:
                raise TensoirHipError("the gfx950 decoder kernel needs fea_pe == view_pe == pos_pe")
            self._packed = ops.PackedMlp(self.mlp, self.in_chanel, self.feape, self._act, w0=self.w0_std())
            self._key = key
        return self._packed

    def w0_std(self):
        """Layer-1 weights in the kernels' column order [feat, aux, PE(feat), PE(aux)] -- the module's own for these decoders."""
        return self.mlp[0].weight

    def run(self, feat, aux, aux_map=None):
        _no_grad_only(type(self).__name__, feat, aux, *self.parameters())
        return ops.mlp(self.packed(), feat, aux, aux_map)


class MLPRender_Fea(_Decoder):
    """models/tensorBase_rotated_lights.py:122-146."""

    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__(inChanel, viewpe, feape, featureC, 3, 0)
        self.viewpe = viewpe

    def forward(self, pts, viewdirs, features):
        return self.run(features, viewdirs)


class MLPBRDF_PEandFeature(_Decoder):
    """models/tensorBase_rotated_lights.py:182-208."""

    def __init__(self, inChanel, pospe=6, feape=6, featureC=128, outc=1, act_net=None):
        act = 1 if isinstance(act_net, nn.Tanh) else 0
        super().__init__(inChanel, pospe, feape, featureC, outc, act)
        self.pospe = pospe
        self.act_net = act_net if act_net is not None else nn.Sigmoid()

    def forward(self, pts, features):
        return self.run(features, pts)


class MLPNormal_normal_and_PExyz(_Decoder):
    """models/tensorBase_rotated_lights.py:236-262, the normal decoder of normals_kind == 'residue_prediction': its layer 1 sees
    [pts, derived normal, features, PE(features), PE(pts)] (153 columns).  Evaluated by the same kernels as the other decoders:
    the 150 columns they know are handed over in their order (w0_std), and the three normal columns -- like the bias and the
    position columns -- enter through the per-row start values of the layer-1 accumulators (ops.mlp_rows_table)."""

    def __init__(self, inChanel, pospe=6, feape=6, featureC=128, outc=1, act_net=None):
        act = 1 if isinstance(act_net, nn.Tanh) else 0
        super().__init__(inChanel, pospe, feape, featureC, outc, act, extra_in=3)
        self.pospe = pospe
        self.act_net = act_net if act_net is not None else nn.Sigmoid()
        F_, nf, na = inChanel, 2 * feape * inChanel, 2 * pospe * 3
        # module column of every kernel-order column: features, pts, PE(features), PE(pts)
        self.std_cols = list(range(6, 6 + F_)) + [0, 1, 2] + list(range(6 + F_, 6 + F_ + nf)) + list(range(6 + F_ + nf, 6 + F_ + nf + na))

    def std_cols_index(self, device):
        """std_cols as a device LongTensor (built once per device: a Python list index re-uploads 150 indices per use)."""
        cache = self.__dict__.setdefault("_std_cols_dev", {})
        key = str(device)
        if key not in cache:
            cache[key] = torch.tensor(self.std_cols, dtype=torch.long, device=device)
        return cache[key]

    def w0_std(self):
        """W0 gathered into the kernels' column order, once per parameter version (the forward pack and the backward pack of a
        training step share it: ADVICE r3)."""
        w = self.mlp[0].weight
        key = (w.data_ptr(), w._version)
        hit = self.__dict__.get("_w0_std")
        if hit is None or hit[0] != key:
            hit = (key, w.detach().index_select(1, self.std_cols_index(w.device)).contiguous())
            self.__dict__["_w0_std"] = hit
        return hit[1]

    def w0_normal(self):
        return self.mlp[0].weight.detach()[:, 3:6]

    def layer1_table(self, pts, normal):
        """[n, 128]: b0 + W0[:, pts and PE(pts) columns] x(pts) + W0[:, normal columns] normal."""
        table = ops.mlp_aux_table(self.packed(), pts)
        return table.addmm_(normal.to(torch.float32), self.w0_normal().t())

    def rows(self, pts, normal, features, n_dev=None, save_hidden=False):
        return ops.mlp_rows_table(self.packed(), features, self.layer1_table(pts, normal), n_dev, save_hidden)

    def forward(self, pts, normal, features):
        _no_grad_only(type(self).__name__, pts, normal, features, *self.parameters())
        return self.rows(pts, normal, features)


def fibonacci_sphere(samples=1):
    """models/tensorBase_rotated_lights.py:49-67."""
    i = np.arange(samples, dtype=np.float64)
    z = 1 - (i / float(samples - 1)) * 2
    r = np.sqrt(1 - z * z)
    th = np.pi * (3.0 - np.sqrt(5.0)) * i
    return np.stack([np.cos(th) * r, np.sin(th) * r, z], axis=-1)


def compute_energy(lgtSGs):
    lam = torch.abs(lgtSGs[:, 3:4])
    mu = torch.abs(lgtSGs[:, 4:])
    return mu * 2.0 * np.pi / lam * (1.0 - torch.exp(-2.0 * lam))


NORMAL_LOSS_KINDS = ("derived_plus_predicted", "residue_prediction")     # the kinds that fill normals_diff / orientation (:953-968)

# constructor arguments of the reference class that are kept verbatim as attributes (models/tensorBase_rotated_lights.py:343-403)
_PLAIN_CTOR_ARGS = ("app_dim", "alphaMask", "device", "density_shift", "alphaMask_thres", "distance_scale", "rayMarch_weight_thres",
                    "fea2denseAct", "near_far", "step_ratio", "shadingMode", "normals_kind", "pos_pe", "view_pe", "fea_pe",
                    "featureC", "envmap_w", "envmap_h", "dataset", "light_kind", "numLgtSGs", "fixed_fresnel")
# what a checkpoint's 'kwargs' entry carries (:646-673): checkpoint key -> attribute
_CKPT_KWARGS = {"density_n_comp": "density_n_comp", "appearance_n_comp": "app_n_comp", "app_dim": "app_dim",
                "density_shift": "density_shift", "alphaMask_thres": "alphaMask_thres", "distance_scale": "distance_scale",
                "rayMarch_weight_thres": "rayMarch_weight_thres", "fea2denseAct": "fea2denseAct", "near_far": "near_far",
                "step_ratio": "step_ratio", "shadingMode": "shadingMode", "pos_pe": "pos_pe", "view_pe": "view_pe",
                "fea_pe": "fea_pe", "featureC": "featureC", "normals_kind": "normals_kind", "light_num": "light_num",
                "light_kind": "light_kind", "numLgtSGs": "numLgtSGs", "light_rotation": "light_rotation"}


class TensorVMSplit(nn.Module):
    """TensorVMSplit (models/tensoRF_rotated_lights.py:6) on top of TensorBase
    (models/tensorBase_rotated_lights.py:343-403): same kwargs, same state_dict."""

    def __init__(self, aabb, gridSize, device, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0], density_shift=-10,
                 alphaMask_thres=0.001, distance_scale=25, rayMarch_weight_thres=0.0001,
                 pos_pe=2, view_pe=2, fea_pe=2, featureC=128, step_ratio=2.0, fea2denseAct="softplus",
                 normals_kind="purely_predicted", light_rotation=["000", "120", "240"],
                 envmap_w=32, envmap_h=16, light_kind="pixel", dataset=None, numLgtSGs=128,
                 fixed_fresnel=0.04, march_t_stop=1e-6, **kwargs):
        super().__init__()
        given = locals()
        for name in _PLAIN_CTOR_ARGS:               # stored under their own names (what get_kwargs() hands back)
            setattr(self, name, given[name])
        as_triple = lambda c: [c] * 3 if isinstance(c, int) else list(c)
        self.density_n_comp, self.app_n_comp = as_triple(density_n_comp), as_triple(appearance_n_comp)
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self.light_rotation = [int(r) for r in light_rotation]
        self.light_num = len(self.light_rotation)
        # transmittance below which a primary / secondary ray stops marching (error in acc, vis and gradients < this;
        # the reference marches every ray to the end: march_t_stop=0 reproduces that exactly)
        self.march_t_stop = float(march_t_stop)
        self.matMode, self.vecMode, self.comp_w = MAT_MODE, VEC_MODE, [1, 1, 1]
        self._field_cache = self._field_key = None
        self.update_stepSize(gridSize)
        self.init_svd_volume(gridSize[0], device)           # same construction order as the reference: seeded runs draw
        self.init_render_func(shadingMode, pos_pe, view_pe, fea_pe, featureC, device)      # the same initial parameters
        self.init_light()

    # ---- construction ----------------------------------------------------------------------------
    def init_svd_volume(self, res, device):
        """models/tensoRF_rotated_lights.py:11-29."""
        self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, self.gridSize, 0.1, device)
        self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, self.gridSize, 0.1, device)
        self.basis_mat = nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)
        self.light_line = nn.Embedding(self.light_num, sum(self.app_n_comp)).to(device)

    def init_one_svd(self, n_component, gridSize, scale, device):
        planes, lines = [], []
        for i in range(3):
            vec_id = self.vecMode[i]
            m0, m1 = self.matMode[i]
            planes.append(nn.Parameter(channel_last(scale * torch.randn((1, n_component[i], int(gridSize[m1]), int(gridSize[m0]))))))
            lines.append(nn.Parameter(channel_last(scale * torch.randn((1, n_component[i], int(gridSize[vec_id]), 1)))))
        return nn.ParameterList(planes).to(device), nn.ParameterList(lines).to(device)

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        """models/tensorBase_rotated_lights.py:405-431.  shadingMode 'MLP_Fea' (every shipped config) has gfx950 kernels; all five
        normals kinds of opt.py:198 do."""
        if shadingMode != "MLP_Fea":
            raise NotImplementedError(f"shadingMode={shadingMode!r}: only 'MLP_Fea' has gfx950 kernels")
        if self.normals_kind not in ("derived_plus_predicted", "purely_predicted", "purely_derived", "gt_normals", "residue_prediction"):
            raise NotImplementedError(f"normals_kind={self.normals_kind!r} is not one of the reference's (opt.py:198)")
        self.renderModule = MLPRender_Fea(self.app_dim, view_pe, fea_pe, featureC).to(device)
        if self.normals_kind in ("purely_predicted", "derived_plus_predicted"):
            self.renderModule_normal = MLPBRDF_PEandFeature(self.app_dim, pos_pe, fea_pe, featureC, outc=3,
                                                            act_net=nn.Tanh()).to(device)
        elif self.normals_kind == "residue_prediction":                      # :426-428
            self.renderModule_normal = MLPNormal_normal_and_PExyz(self.app_dim, pos_pe, fea_pe, featureC, outc=3,
                                                                  act_net=nn.Tanh()).to(device)
        self.renderModule_brdf = MLPBRDF_PEandFeature(self.app_dim, pos_pe, fea_pe, featureC, outc=4,
                                                      act_net=nn.Sigmoid()).to(device)

    @staticmethod
    @functools.lru_cache(maxsize=16)
    def _equirect_cells(rows, cols, row_hi, row_step):
        """Cell centres of a rows x cols equirect grid: the row coordinate runs row_hi - step/2 .. -row_hi + step/2, the azimuth
        pi - lng/2 .. -pi + lng/2 (the reference's orientation, models/tensorBase_rotated_lights.py:437-441).  Host tensors,
        the same for every call with the same grid (the stratified sampler asks once per training step): cached, never written to."""
        lng = 2 * np.pi / cols
        return torch.meshgrid([torch.linspace(row_hi - 0.5 * row_step, -row_hi + 0.5 * row_step, rows),
                               torch.linspace(np.pi - 0.5 * lng, -np.pi + 0.5 * lng, cols)], indexing="ij") + (lng,)

    @staticmethod
    def _unit_dirs(phi, theta):
        return torch.stack([torch.cos(theta) * torch.cos(phi), torch.sin(theta) * torch.cos(phi), torch.sin(phi)], dim=-1)

    def generate_envir_map_dir(self, envmap_h, envmap_w, is_jittor=False):
        """models/tensorBase_rotated_lights.py:435-453 (host side, tiny): solid-angle weights (sum 4 pi) and unit directions of the
        envmap_h x envmap_w cells, optionally jittered uniformly inside each cell (elevation drawn first, as the reference does)."""
        lat = np.pi / envmap_h
        phi, theta, lng = self._equirect_cells(envmap_h, envmap_w, np.pi / 2, lat)
        sin_phi = torch.sin(torch.pi / 2 - phi)
        weights = 4 * torch.pi * sin_phi / torch.sum(sin_phi)
        assert 0 not in weights, "every cell of the light grid must have a solid angle"
        if is_jittor:
            phi = phi + lat * (torch.rand_like(phi) - 0.5)
            theta = theta + lng * (torch.rand_like(theta) - 0.5)
        return weights.to(torch.float32).reshape(-1), self._unit_dirs(phi, theta).view(-1, 3)

    def init_light(self):
        """models/tensorBase_rotated_lights.py:455-488: the fixed direction grid, the z-rotation of every light and the trainable
        light -- 128 spherical Gaussians (lobe axis 3, sharpness 1, amplitude 3) or the pixel image."""
        self.light_area_weight, self.fixed_viewdirs = self.generate_envir_map_dir(self.envmap_h, self.envmap_w)
        if self.light_kind not in ("sg", "pixel", "gt"):
            raise NotImplementedError(f"light_kind={self.light_kind!r} is not one of the reference's ('sg', 'pixel', 'gt')")
        self._light_rotations()
        if self.light_kind == "gt":          # :592-593: nothing to train, the data set's probe is looked up as it is
            return
        if self.light_kind == "pixel":       # :459-460: a learnable envmap_h x envmap_w image behind softplus(beta=5)
            cells = self.envmap_w * self.envmap_h
            self._light_rgbs = nn.Parameter(torch.FloatTensor(cells, 3).uniform_(0, 3).to(torch.float32).to(self.device))
            return
        n = self.numLgtSGs
        sg = torch.randn(n, 7)                                   # the ONE random draw (seeded runs match the reference's)
        sg[:, 5:] = sg[:, 4:5].expand(-1, 2)                     # grey amplitudes
        sg[:, 3:4] = 10.0 + torch.abs(sg[:, 3:4] * 20.0)         # sharpness >= 10
        amplitude = torch.abs(sg[:, 4:])
        sg[:, 4:] = amplitude / torch.sum(compute_energy(sg), dim=0, keepdim=True) * 2.0 * np.pi * 0.8    # total energy 1.6 pi
        lobes = torch.from_numpy(fibonacci_sphere(n // 2).astype(np.float32))
        sg[:n // 2, :3] = lobes                                  # two Gaussians per Fibonacci direction
        sg[n // 2:, :3] = lobes
        self.lgtSGs = nn.Parameter(sg.to(self.device), requires_grad=True)

    def _light_rotations(self):
        """:479-487: one z-rotation matrix per light."""
        mats = []
        for i in range(self.light_num):
            a = torch.tensor(self.light_rotation[i] / 180 * torch.pi).to(torch.float32)
            mats.append(torch.tensor([[torch.cos(a), -torch.sin(a), 0], [torch.sin(a), torch.cos(a), 0],
                                      [0, 0, 1]]).to(torch.float32))
        self.light_rotation_matrix = torch.stack(mats, dim=0)

    def gen_light_incident_dirs(self, sample_number=-1, method="fixed_envirmap", device="cuda"):
        """models/tensorBase_rotated_lights.py:492-574 (host-side direction tables)."""
        if method == "fixed_envirmap":
            dirs = self.fixed_viewdirs
        elif method == "stratified_sampling":                   # one uniform draw inside every cell of the grid
            lat = np.pi / self.envmap_h
            phi, theta, lng = self._equirect_cells(self.envmap_h, self.envmap_w, np.pi / 2, lat)
            phi = phi + lat * (torch.rand_like(phi) - 0.5)
            dirs = self._unit_dirs(phi, theta + lng * (torch.rand_like(theta) - 0.5))
        elif method == "stratifed_sample_equal_areas":         # rows uniform in sin(elevation): equal-area cells
            band = 2 / self.envmap_h
            sin_phi, theta, lng = self._equirect_cells(self.envmap_h, self.envmap_w, 1.0, band)
            sin_phi = sin_phi + band * (torch.rand_like(sin_phi) - 0.5)
            dirs = self._unit_dirs(torch.asin(sin_phi), theta + lng * (torch.rand_like(theta) - 0.5))
        elif method == "importance_sample":
            _, view_dirs = self.generate_envir_map_dir(128, 256, is_jittor=True)
            envir_map = self.get_light_rgbs(view_dirs.reshape(-1, 3).to(device), device=device)[0]
            with torch.no_grad():
                envir_map = envir_map.reshape(128, 256, 3)
                inten = torch.sum(envir_map, dim=2, keepdim=True)
                h, w, _ = inten.shape
                sin_theta = torch.sin(torch.linspace(0 + 0.5 / h, np.pi - 0.5 / h, h)).to(device)
                pdf = inten * sin_theta.view(-1, 1, 1)
                pdf_s = pdf / torch.sum(pdf)
                pdf_c = pdf_s * h * w / (2 * np.pi * np.pi * sin_theta.view(-1, 1, 1))
                idx = torch.multinomial(pdf_s.view(-1), sample_number, replacement=True)
                d = view_dirs.view(-1, 3).to(device)[idx]
                return d, envir_map.view(-1, 3)[idx], pdf_c.view(-1, 1)[idx]
        else:
            raise ValueError(f"unknown light sampling method {method!r}")
        return dirs.reshape(-1, 3)

    def get_light_rgbs(self, incident_light_directions=None, device="cuda"):
        """models/tensorBase_rotated_lights.py:577-606 (sg): dirs [D,3] -> [L,D,3] via tir_env_sg_fwd."""
        dirs = ops.to_device(incident_light_directions, device, torch.float32).reshape(-1, 3).to(torch.float32).contiguous()
        rot = self.__dict__.get("_rot_dev")
        if rot is None or rot.device != dirs.device:
            rot = self.light_rotation_matrix.to(dirs.device).contiguous()
            self.__dict__["_rot_dev"] = rot
        if self.light_kind == "gt":              # :592-593 -- `dataset.lights_probes` [envmap_h * envmap_w, 3], no activation
            probe = getattr(self.dataset, "lights_probes", None)
            if probe is None:
                raise TensoirHipError("light_kind='gt' needs a dataset with lights_probes (the environment map of the scene)")
            probe = ops.to_device(probe, dirs.device, torch.float32).reshape(-1, 3).contiguous()
            return ops.env_pixel(probe, self.envmap_h, self.envmap_w, rot, dirs, softplus=False)
        if self.light_kind == "pixel":           # :585-605; tiny table, evaluated per call (no cache: the map trains)
            if torch.is_grad_enabled() and self._light_rgbs.requires_grad:
                from . import training
                return training.EnvPixelFn.apply(self._light_rgbs, rot, dirs, self.envmap_h, self.envmap_w)
            return ops.env_pixel(self._light_rgbs, self.envmap_h, self.envmap_w, rot, dirs)
        if torch.is_grad_enabled() and self.lgtSGs.requires_grad:
            from . import training
            return training.EnvSGFn.apply(self.lgtSGs, rot, dirs)
        # inference: the radiance table only changes with the SGs or the direction set -> cached per version
        key = (self.lgtSGs.data_ptr(), self.lgtSGs._version, dirs.data_ptr(), dirs._version, tuple(dirs.shape))
        cached = self.__dict__.get("_env_cache")
        if cached is None or cached[0] != key:
            cached = (key, ops.env_sg(self.lgtSGs.to(device), rot, dirs), dirs)
            self.__dict__["_env_cache"] = cached
        return cached[1]

    def update_stepSize(self, gridSize):
        """models/tensorBase_rotated_lights.py:608-619: voxel size, march step (step_ratio mean voxels) and the number of samples
        that covers the box diagonal."""
        extent = self.aabb[1] - self.aabb[0]
        self.aabbSize, self.invaabbSize = extent, 2.0 / extent
        grid_h = torch.LongTensor([int(g) for g in gridSize])
        self.gridSize = grid_h.to(self.device)
        # The two REDUCTIONS of this function are evaluated on the host (round 6).  The reference writes torch.mean(self.units) on
        # whatever device it runs on; for three equal voxel sizes ROCm's reduction returns a value ONE ULP from ATen-CPU's (the
        # platform the golden vectors and the oracle come from) -- the march's sample spacing is then an ulp off, a third of all
        # sample positions move by one to three ulps, and on a field trained to sharp surfaces sigma moves by up to 1e-3 relative
        # (profiles/r06_step_size_ulp.txt): the origin of the 1e-4 ... 1e-3 field-gradient distance from the oracle that rounds 3-6
        # attributed to other things.  Element-wise arithmetic is the same everywhere and stays on the device.
        extent_h = extent.detach().cpu()
        units_h = extent_h / (grid_h - 1)
        self.units = extent / (self.gridSize - 1)
        self.stepSize = (torch.mean(units_h) * self.step_ratio).to(self.device)
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(extent_h))).to(self.device)
        self.nSamples = int((self.aabbDiag / self.stepSize).item()) + 1
        self._field_key = None

    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invaabbSize - 1

    # ---- optimiser groups / regularisers (PyTorch, off the hot path) --------------------------------
    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        """models/tensoRF_rotated_lights.py:33-57."""
        g = [{"params": self.density_line, "lr": lr_init_spatialxyz},
             {"params": self.density_plane, "lr": lr_init_spatialxyz},
             {"params": self.app_line, "lr": lr_init_spatialxyz},
             {"params": self.app_plane, "lr": lr_init_spatialxyz},
             {"params": self.basis_mat.parameters(), "lr": lr_init_network},
             {"params": self.light_line.parameters(), "lr": 0.001}] + self._light_param_groups() + [
             {"params": self.renderModule.parameters(), "lr": lr_init_network},
             {"params": self.renderModule_brdf.parameters(), "lr": lr_init_network}]
        if hasattr(self, "renderModule_normal"):
            g.append({"params": self.renderModule_normal.parameters(), "lr": lr_init_network})
        return g

    def _light_param_groups(self):
        """models/tensoRF_rotated_lights.py:43-46."""
        if self.light_kind == "gt":
            return []
        return [{"params": self._light_rgbs if self.light_kind == "pixel" else self.lgtSGs, "lr": 0.001}]

    def light_parameters(self):
        """The environment light's trainable tensors (cache keys of captured graphs, train / eval routing)."""
        if self.light_kind == "pixel":
            return [self._light_rgbs]
        if self.light_kind == "gt":
            return []
        return list(getattr(self, "lgtSGs_list", None) or [self.lgtSGs])

    def vectorDiffs(self, vector_comps):
        """Mean |off-diagonal| of every line factor's Gram matrix (models/tensoRF_rotated_lights.py:59-69)."""
        total = 0
        for comp in vector_comps:
            v = comp.view(comp.shape[1], comp.shape[2])
            gram = v @ v.t()
            n = gram.shape[0]
            off_diag = gram.flatten()[1:].view(n - 1, n + 1)[:, :-1]         # drops the diagonal of an n x n matrix
            total = total + off_diag.abs().mean()
        return total

    def vector_comp_diffs(self):
        return self.vectorDiffs(self.density_line) + self.vectorDiffs(self.app_line)

    def density_L1(self):
        """:74-78: sum over the three density planes and lines of mean(|x|).  The training script adds it to the loss in EVERY
        iteration (train_tensoIR.py:283-286); written tensor by tensor it is ~40 framework launches per iteration (abs, mean,
        add per tensor, and their three backward nodes each) in a loop whose iteration is the SUM of its host and GPU segments.
        Here: multi-tensor norm / sign kernels over the six tensors, one autograd node -- 8 launches.  The six means are added
        by one reduction (association differs from the reference's left-to-right chain by at most a few ulp of the total)."""
        ts = [t for pair in zip(self.density_plane, self.density_line) for t in pair]
        return _DensityL1Fn.apply(*ts)

    def TV_loss_density(self, reg):
        return sum((reg(plane) * 1e-2 for plane in self.density_plane), 0)

    def TV_loss_app(self, reg):
        return sum((reg(plane) * 1e-2 for plane in self.app_plane), 0)

    # ---- packed shadow of the parameters -----------------------------------------------------------
    def _field_params(self):
        # (the ParameterLists' own dictionaries: iterating a ParameterList goes through nn.Module.__getattr__ and string index
        #  conversions per element -- 26 slow lookups per call, three calls per training step: 0.38 ms of host time per step)
        d = self.__dict__["_parameters"], self.__dict__["_modules"]
        out = []
        for name in ("density_plane", "density_line", "app_plane", "app_line"):
            out.extend(d[1][name]._parameters.values())
        out.append(d[1]["basis_mat"]._parameters["weight"])
        out.append(d[1]["light_line"]._parameters["weight"])
        return out

    def packed_field(self) -> TirField:
        """The TirField descriptor (+ the small derived tables: basis_mat^T, mean light row, occupancy bits); rebuilt
        whenever a parameter's storage or version changes (optimizer step, upsample, shrink, load).  The VM planes /
        lines are channel-last parameters and are referenced in place."""
        ps = self._field_params()
        mask = self.alphaMask
        key = (tuple((p.data_ptr(), p._version, tuple(p.shape)) for p in ps), id(mask),
               self.stepSize.data_ptr(), float(self.distance_scale), float(self.density_shift),
               float(self.rayMarch_weight_thres), float(self.near_far[0]), float(self.near_far[1]), self.fea2denseAct)
        if key == self._field_key and self._field_cache is not None:
            return self._field_cache["desc"]
        if len(set(self.density_n_comp)) != 1 or len(set(self.app_n_comp)) != 1:
            raise TensoirHipError("per-plane component counts must be equal for the gfx950 kernels")
        keep = {}
        f = TirField()
        g = _host_values(self, "geom", (self.aabb, self.invaabbSize, self.gridSize, self.stepSize),
                         lambda: (self.aabb[0].tolist(), self.aabb[1].tolist(), self.invaabbSize.tolist(),
                                  [int(v) for v in self.gridSize.tolist()], float(self.stepSize)))
        f.aabb_min[:], f.aabb_max[:], f.inv_aabb[:], f.grid[:], f.step_size = g
        f.distance_scale = float(self.distance_scale)
        f.density_shift = float(self.density_shift)
        f.weight_thres = float(self.rayMarch_weight_thres)
        f.near_, f.far_ = float(self.near_far[0]), float(self.near_far[1])
        f.act = {"softplus": 0, "relu": 1}[self.fea2denseAct]
        f.n_dcomp, f.n_acomp = self.density_n_comp[0], self.app_n_comp[0]
        f.app_dim, f.n_lights = self.app_dim, self.light_num
        for i in range(3):
            for name, src, dst in (("dp", self.density_plane, f.dplane), ("dl", self.density_line, f.dline),
                                   ("ap", self.app_plane, f.aplane), ("al", self.app_line, f.aline)):
                # a channel-last parameter is gathered in place; anything else (a tensor swapped in from outside)
                # gets a packed shadow copy
                t = src[i].detach() if is_channel_last(src[i]) else ops.pack_plane(src[i])
                keep[f"{name}{i}"] = t
                dst[i] = t.data_ptr()
        keep["basis"] = ops.pack_basis(self.basis_mat.weight)
        keep["ll"] = self.light_line.weight.detach().to(torch.float32).contiguous()
        keep["lm"] = ops.light_mean(keep["ll"])
        f.basis_t, f.light_line, f.light_mean = keep["basis"].data_ptr(), keep["ll"].data_ptr(), keep["lm"].data_ptr()
        if mask is not None:
            keep["bits"] = mask.bits()
            f.occ_nbr = keep["bits"].data_ptr()
            D, H, W = mask.alpha_volume.shape[-3:]
            f.occ_dim[:] = [W, H, D]
            f.occ_aabb_min[:], f.occ_inv[:] = mask.host_geometry()
            lo, hi = mask.occupied_box()
            if all(a < b for a, b in zip(lo, hi)):
                f.occ_lo[:], f.occ_hi[:] = lo, hi
            else:                                       # empty mask: nothing can be hit -- a degenerate box far away
                f.occ_lo[:], f.occ_hi[:] = [3.0e38] * 3, [3.4e38] * 3
        f.tune_lds_lines, f.tune_xcd_order = ops.TUNE["lds_lines"], ops.TUNE["xcd_order"]      # launch options (ops.TUNE)
        keep["desc"] = f
        self._field_cache, self._field_key = keep, key
        return f

    def packed_field_half(self, _fresh=False):
        """TirFieldHalf: fp16 shadow (saturating casts) of the appearance planes / lines for the indirect-light gather
        (ops.vm_app_h16 / ops.indirect_fused), or None when the field is not 48 components wide.  Built with one launch, cached
        with the packed field (same key: any optimizer step, upsample, shrink or load rebuilds it).  The same launch measures
        the abs-maxima the range guard needs (half_range()).  _fresh: the caller has just called packed_field()."""
        from ._lib import TirFieldHalf
        if not _fresh:
            self.packed_field()
        keep = self._field_cache
        if "half" not in keep:
            if self.app_n_comp[0] != 48:
                keep["half"] = None
            else:
                src = [keep[f"ap{i}"] for i in range(3)] + [keep[f"al{i}"] for i in range(3)]
                tabs, absmax = ops.pack_half(src, scan=(keep["ll"], keep["basis"]))
                fh = TirFieldHalf()
                for i in range(3):
                    fh.aplane[i], fh.aline[i] = tabs[i].data_ptr(), tabs[3 + i].data_ptr()
                keep["half"] = (fh, tabs)
                keep["half_range"] = ops.HalfRange(absmax)
        return keep["half"][0] if keep["half"] is not None else None

    def half_range(self, _fresh=False):
        """ops.HalfRange of the current fp16 shadow (None without one).  _fresh: the caller has just called packed_field() (the
        key walk over the parameters is not repeated)."""
        return self._field_cache.get("half_range") if self.packed_field_half(_fresh) is not None else None

    def indirect_precision(self):
        """What the indirect-light precision policy decided for this model so far (ops.INDIRECT_GUARD, relight._indirect_mode):
        {"policy": auto|f16|full, "mode": f16|full|None, "why": ..., "probe": {...}} -- also written into checkpoints."""
        st = self.__dict__.get("_indirect_state") or {}
        if ops.secondary_app_impl() is None and ops.secondary_mlp_impl() in (None, "hp"):
            pol = "hp" if ops.secondary_mlp_impl() == "hp" else "full"
        else:
            pol = "auto" if ops.INDIRECT_GUARD else "f16"
        return {"policy": pol, "mode": st.get("verdict") if pol == "auto" else pol, "why": st.get("why"), "probe": st.get("stats"),
                "probes_run": st.get("probes", 0), "fallbacks": st.get("fallbacks", 0)}

    # ---- per-point field functions (reference signatures) ---------------------------------------------
    def compute_densityfeature(self, xyz_sampled):
        """models/tensoRF_rotated_lights.py:95-110 -> tir_vm_density_fwd."""
        _no_grad_only("compute_densityfeature", *self._field_params())
        return ops.vm_density(self.packed_field(), xyz_sampled.reshape(-1, 3))[0]

    def feature2density(self, density_features):
        if self.fea2denseAct == "softplus":
            return F.softplus(density_features + self.density_shift)
        return F.relu(density_features)

    def compute_appfeature(self, xyz_sampled, light_idx):
        """models/tensoRF_rotated_lights.py:197-224 -> tir_vm_app_fwd."""
        _no_grad_only("compute_appfeature", *self._field_params())
        li = light_idx.reshape(-1).to(xyz_sampled.device, torch.int32)
        return ops.vm_app(self.packed_field(), xyz_sampled.reshape(-1, 3), li, None, True, False, ops.APP_IMPL)[0][:, :self.app_dim].contiguous()

    def compute_bothfeature(self, xyz_sampled, light_idx):
        """models/tensoRF_rotated_lights.py:132-165."""
        _no_grad_only("compute_bothfeature", *self._field_params())
        li = light_idx.reshape(-1).to(xyz_sampled.device, torch.int32)
        r, i = ops.vm_app(self.packed_field(), xyz_sampled.reshape(-1, 3), li, None, True, True, ops.APP_IMPL)
        return r[:, :self.app_dim].contiguous(), i[:, :self.app_dim].contiguous()

    def compute_intrinfeature(self, xyz_sampled):
        """models/tensoRF_rotated_lights.py:167-195."""
        _no_grad_only("compute_intrinfeature", *self._field_params())
        return ops.vm_app(self.packed_field(), xyz_sampled.reshape(-1, 3), None, None, False, True, ops.APP_IMPL)[1][:, :self.app_dim].contiguous()

    def compute_derived_normals(self, xyz_locs):
        """models/tensorBase_rotated_lights.py:839-856 -> tir_density_grad_fwd (closed form)."""
        _no_grad_only("compute_derived_normals", *self._field_params())
        return ops.density_grad(self.packed_field(), xyz_locs.reshape(-1, 3))[2]

    def compute_alpha(self, xyz_locs, length=1):
        """models/tensorBase_rotated_lights.py:819-837 (occupancy cull + density on the GPU)."""
        f = self.packed_field()
        xyz = xyz_locs.reshape(-1, 3).to(torch.float32)
        sigma = ops.vm_density(f, self.normalize_coord(xyz), False, True)[1]
        if self.alphaMask is not None:
            sigma = sigma * ops.occupancy_query(f, xyz).to(sigma.dtype)
        return (1 - torch.exp(-sigma * length)).view(xyz_locs.shape[:-1])

    # ---- occupancy maintenance (models/tensorBase_rotated_lights.py:737-811) ----------------------------
    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None):
        """models/tensorBase_rotated_lights.py:737-753 -> tir_dense_alpha (positions generated in the kernel from the
        linspace tables; the [g^3, 3] lattice is only materialised for the return value the reference promises)."""
        gridSize = self.gridSize if gridSize is None else gridSize
        alpha, lins = ops.dense_alpha(self.packed_field(), [int(g) for g in gridSize], float(self.stepSize))
        samples = torch.stack(torch.meshgrid(lins[0], lins[1], lins[2], indexing="ij"), -1)
        dense_xyz = self.aabb[0] * (1 - samples) + self.aabb[1] * samples
        return alpha, dense_xyz

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(200, 200, 200)):
        """models/tensorBase_rotated_lights.py:755-779, entirely on the device: dense alpha lattice -> clamp,
        3x3x3 max-pool, threshold (tir_alpha_pool) -> new AlphaGridMask; the new aabb is the position of the index
        bounding box of the occupied voxels (positions are separable and monotonic per axis, so the min / max over
        the occupied lattice points is attained at the min / max indices)."""
        gridSize = [int(g) for g in gridSize]
        alpha, lins = ops.dense_alpha(self.packed_field(), gridSize, float(self.stepSize))
        vol, bbox = ops.alpha_pool(alpha, self.alphaMask_thres)
        total_voxels = gridSize[0] * gridSize[1] * gridSize[2]
        self.alphaMask = AlphaGridMask(self.device, self.aabb, vol)
        self._field_key = None
        b = bbox.tolist()
        if b[3] < 0:
            raise TensoirHipError("updateAlphaMask: no voxel passes alphaMask_thres (empty scene)")
        lo = torch.stack([lins[a][b[a]] for a in range(3)])
        hi = torch.stack([lins[a][b[3 + a]] for a in range(3)])
        xyz_min = self.aabb[0] * (1 - lo) + self.aabb[1] * lo
        xyz_max = self.aabb[0] * (1 - hi) + self.aabb[1] * hi
        print(f"bbox: {xyz_min, xyz_max} alpha rest %%%f" % (torch.sum(vol) / total_voxels * 100))
        return torch.stack((xyz_min, xyz_max))

    @torch.no_grad()
    def filtering_rays(self, all_rays, N_samples=256, chunk=10240 * 5, bbox_only=False):
        """models/tensorBase_rotated_lights.py:781-811 -> tir_filter_rays (one wave per ray; the [chunk, N, 3] sample
        tensor of the reference is never built).  all_rays may live on the host (as in train_tensoIR.py:228)."""
        tt = time.time()
        flat = all_rays.reshape(-1, all_rays.shape[-1])
        N = flat.shape[0]
        f = self.packed_field()
        big = chunk * 64
        masks = []
        for i in range(0, N, big):
            rays_chunk = flat[i:i + big, :6].to(self.device, torch.float32).contiguous()
            masks.append(ops.filter_rays(f, rays_chunk, N_samples, bbox_only).to(all_rays.device))
        mask = torch.cat(masks).view(all_rays.shape[:-1]) if masks else torch.zeros(all_rays.shape[:-1], dtype=torch.bool)
        print(f"Ray filtering done! takes {time.time() - tt} s. ray mask ratio: {torch.sum(mask) / N}")
        from . import dist as tdist
        dp = tdist.launcher_dp()
        if dp is not None:      # the unmodified script under torchrun (tensoir_amd.run): this rank trains on its 1/world of the kept rays
            mask = tdist.shard_filter_mask(mask, *dp)
            print(f"[tensoir_amd] data-parallel launcher: rank {dp[0]} of {dp[1]} keeps {int(mask.sum())} training rays")
        return all_rays[mask], mask

    def sample_ray(self, rays_o, rays_d, is_train=True, N_samples=-1):
        """models/tensorBase_rotated_lights.py:705-724 (kept for filtering_rays; the march kernels
        generate the same samples on the fly and never materialise them)."""
        n = N_samples if N_samples > 0 else self.nSamples
        safe_d = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
        t_enter = torch.minimum((self.aabb[1] - rays_o) / safe_d, (self.aabb[0] - rays_o) / safe_d).amax(-1)
        t_enter = t_enter.clamp(min=self.near_far[0], max=self.near_far[1])
        k = torch.arange(n)[None].float()
        if is_train:                                            # one jitter per ray, shared by all of its samples
            k = k.repeat(rays_d.shape[-2], 1)
            k += torch.rand_like(k[:, [0]])
        z = t_enter[..., None] + self.stepSize * ops.to_device(k, rays_o.device)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
        inside = ~((self.aabb[0] > pts) | (pts > self.aabb[1])).any(dim=-1)
        return pts, z, inside

    # ---- resolution changes (models/tensoRF_rotated_lights.py:227-288) ---------------------------------
    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        """Bilinear (align_corners) resampling of every factor to res_target; the results are channel-last parameters."""
        def resized(t, size):
            return nn.Parameter(channel_last(F.interpolate(t.data, size=size, mode="bilinear", align_corners=True)))
        for i, ((m0, m1), v) in enumerate(zip(self.matMode, self.vecMode)):
            plane_coef[i] = resized(plane_coef[i], (res_target[m1], res_target[m0]))
            line_coef[i] = resized(line_coef[i], (res_target[v], 1))
        return plane_coef, line_coef

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(self.density_plane, self.density_line, res_target)
        self.update_stepSize(res_target)
        print(f"upsamping to {res_target}")

    @torch.no_grad()
    def shrink(self, new_aabb):
        """Crop every factor to the voxel range covering new_aabb (models/tensoRF_rotated_lights.py:254-288).  When the occupancy
        grid and the field grid differ, the box actually kept is the one spanned by the cropped voxel range (:277-285)."""
        first = torch.round(torch.round((new_aabb[0] - self.aabb[0]) / self.units)).long()
        last = torch.minimum(torch.round((new_aabb[1] - self.aabb[0]) / self.units).long() + 1, self.gridSize)      # exclusive
        keep = [slice(int(a), int(b)) for a, b in zip(first, last)]              # per world axis
        cropped = lambda t: nn.Parameter(channel_last(t))
        for i, ((m0, m1), v) in enumerate(zip(self.matMode, self.vecMode)):
            for lines, planes in ((self.density_line, self.density_plane), (self.app_line, self.app_plane)):
                lines[i] = cropped(lines[i].data[..., keep[v], :])                 # [1, C, R, 1]
                planes[i] = cropped(planes[i].data[..., keep[m1], keep[m0]])       # [1, C, grid[m1], grid[m0]]
        if not torch.all(self.alphaMask.gridSize == self.gridSize):
            lo_frac, hi_frac = first / (self.gridSize - 1), (last - 1) / (self.gridSize - 1)
            lo, hi = self.aabb[0], self.aabb[1]
            new_aabb = torch.stack([(1 - lo_frac) * lo + lo_frac * hi, (1 - hi_frac) * lo + hi_frac * hi]).to(new_aabb.dtype)
        self.aabb = new_aabb
        self.update_stepSize(tuple(last - first))

    # ---- checkpoint I/O (models/tensorBase_rotated_lights.py:646-692) ---------------------------------
    def get_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), **{k: getattr(self, a) for k, a in _CKPT_KWARGS.items()}}

    def save(self, path):
        """The reference's file layout: kwargs + state_dict + the occupancy volume as packed bits (:675-683)."""
        ckpt = {"kwargs": self.get_kwargs(), "state_dict": self.state_dict()}
        if self.alphaMask is not None:
            occupied = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt["alphaMask.shape"] = occupied.shape
            ckpt["alphaMask.mask"] = np.packbits(occupied.reshape(-1))
            ckpt["alphaMask.aabb"] = self.alphaMask.aabb.cpu()
        torch.save(ckpt, path)
        # the checkpoint file keeps exactly the reference's keys; which precision the indirect-light stage ran at for these
        # parameters (ops.INDIRECT_GUARD) goes into a sidecar next to it
        try:
            import json
            with open(str(path) + ".tensoir_amd.json", "w") as fh:
                json.dump({"indirect_precision": self.indirect_precision()}, fh, indent=1, default=str)
        except OSError:
            pass

    def load(self, ckpt):
        if "alphaMask.aabb" in ckpt:
            shape = tuple(ckpt["alphaMask.shape"])
            bits = np.unpackbits(ckpt["alphaMask.mask"])[:int(np.prod(shape))]
            self.alphaMask = AlphaGridMask(self.device, ckpt["alphaMask.aabb"].to(self.device),
                                           torch.from_numpy(bits.reshape(shape)).float().to(self.device))
        self.load_state_dict(ckpt["state_dict"])
        self._field_key = None
        self.__dict__.pop("_indirect_state", None)      # new parameters in the old storage: the precision verdict must be re-established

    # ---- the primary pass ----------------------------------------------------------------------------
    def forward(self, rays_chunk, light_idx, white_bg=True, is_train=False, ndc_ray=False, is_relight=True,
                N_samples=-1, _brdf_jitter_dense=None, _return_maps=False, _defer_check=False, _want_mask=True):
        """TensorBase.forward (models/tensorBase_rotated_lights.py:868-1036) as a chain of HIP launches:
        march -> scan -> compact -> appearance gather -> decoders -> analytic normals -> composite.

        Returns the reference's 12-tuple.  ``_brdf_jitter_dense`` ([B,S,3] N(0,1), tests only) replaces
        the torch.randn_like draw of :937 so different implementations see identical noise.
        """
        if ndc_ray:
            raise NotImplementedError("ndc_ray=True is not on the TensoIR hot path (no shipped config uses it)")
        dev = rays_chunk.device
        rays = rays_chunk.to(torch.float32).contiguous()
        B = rays.shape[0]
        S = N_samples if N_samples > 0 else self.nSamples
        f = self.packed_field()
        lidx = ops.to_device(light_idx.reshape(-1), dev, torch.int32).contiguous()
        # RNG draws in the reference's order and on the reference's devices (:717 CPU, :937 device, :1004 CPU)
        jitter = ops.to_device(torch.rand(B, 1), dev) if is_train else None
        from . import training
        if training.wants_grad(self):
            # training step: the same launches with the activations kept, and a hand-written backward
            # (tensoir_amd/training.py); torch.autograd only links the fused stages
            bg = bool(white_bg or (is_train and torch.rand((1,)) < 0.5))
            for attempt in range(2):
                try:
                    maps = training.PrimaryRenderFn.apply(self, rays, lidx, S, bg, bool(is_relight), jitter,
                                                          _brdf_jitter_dense, bool(_defer_check),
                                                          *training.field_param_list(self))
                    break
                except training._CapacityOverflow:     # hint dropped: the second attempt counts exactly
                    if attempt:
                        raise
            out = self.unpack_maps(maps, is_relight, want_mask=_want_mask)
            return (out, maps) if _return_maps else out
        # Record capacity: the number A of w > thres samples is only known on the device.  The first call per
        # (B, S) reads it back (one host sync in the middle of the pass); later inference calls size their buffers
        # from the previous count, bound every kernel by the device-side count (n_dev) and check for overflow
        # once everything -- incl. the caller's shading stage when _defer_check -- has been queued.
        hints = self.__dict__.setdefault("_app_cap_hints", {})
        cap = hints.get((B, S)) if (not is_train and _brdf_jitter_dense is None) else None
        words = viewdirs = None
        if cap is None:
            weight, acc, depth, _tend, cnt = ops.march_primary(f, rays, jitter, S, self.march_t_stop)
            offsets = ops.exclusive_scan(cnt)
            A = int(offsets[-1].item())                  # the one host sync of the pass
            n_dev = total_dev = None
        else:
            # hinted (sync-free) route: ONE launch marches, emits the view-direction table, re-arms the counters of the
            # later kernels of this pass and scans the record counts (its last workgroup)
            words = self._step_words(dev)
            weight, acc, depth, cnt, offsets, total_dev, viewdirs = ops.march_primary_fused(f, rays, S, self.march_t_stop,
                                                                                            cap, words)
            self.__dict__["_rec_counter_armed"] = words[1:3]
            A, n_dev = cap, offsets[B:]
            # eager calls: the count travels to the host while the rest of the pass is queued (no queue drain at the check)
            total_host = None if self.__dict__.get("_capture") is not None else ops.AsyncCount(total_dev)
        rec_ray, rec_k, rec_w, rec_xyz = ops.compact_primary(f, rays, jitter, weight, offsets, A)
        rgb = brdf = brdf_j = pred = derived = None
        rng_state = None
        if A > 0:
            if viewdirs is None:
                viewdirs = rays[:, 3:6].contiguous()
            merged = bool(is_relight) and _brdf_jitter_dense is None and f.n_acomp == 48 and ops.APP_IMPL == "mfma"
            if merged:
                # both appearance gathers of the stage in one launch (xyz + randn_like(xyz) * 0.01 of :937 drawn in the kernel:
                # Philox keyed by the framework's CUDA seed, device-side offset advanced once per pass by the compositing kernel)
                rng_state = self._jitter_rng(dev)
                rad, intr, xyz_j, intr_j = ops.vm_app_primary(f, rec_xyz, lidx, rec_ray, 0.01, rng_state, n_dev)
            else:
                rad, intr = ops.vm_app(f, rec_xyz, lidx, rec_ray, True, bool(is_relight), None, 0, n_dev)
            jobs = [(self.renderModule.packed(), rad, viewdirs, rec_ray)]
            if is_relight:
                pb = self.renderModule_brdf.packed()
                jobs.append((pb, intr, rec_xyz, None))
                if merged:
                    pass
                elif _brdf_jitter_dense is not None:
                    noise = _brdf_jitter_dense.to(dev, torch.float32)[rec_ray.long(), rec_k.long()]
                    xyz_j = torch.add(rec_xyz, noise, alpha=0.01)
                    intr_j = ops.vm_app(f, xyz_j, None, None, False, True, None, 0, n_dev)[1]
                else:
                    # xyz + randn_like(xyz) * 0.01 (:937): the N(0,1) triples are drawn inside the gather kernel (Philox keyed
                    # by the framework's CUDA seed, device-side offset advanced once per pass by the compositing kernel)
                    rng_state = self._jitter_rng(dev)
                    xyz_j, intr_j = ops.vm_app_jitter(f, rec_xyz, 0.01, 0, 0, rng_state, n_dev)
                jobs.append((pb, intr_j, xyz_j, None))
                if self.normals_kind in ("purely_predicted", "derived_plus_predicted"):
                    jobs.append((self.renderModule_normal.packed(), intr, rec_xyz, None))
            if ops.MLP_IMPL == "bf16x3":
                # the decoders of the primary stage run on the same records: ONE launch, the grid split between them
                outs = ops.mlp_multi(jobs, n_dev)
            else:
                outs = [ops.mlp(m, ft, ax, mp, None, 0, n_dev) for m, ft, ax, mp in jobs]
            rgb = outs[0]
            if is_relight:
                brdf, brdf_j = outs[1], outs[2]
                if self.normals_kind == "purely_derived":
                    pred = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
                elif self.normals_kind == "gt_normals":
                    pred = None                        # zeros (:951-952): Renderer_TensoIR_train substitutes normal_gt
                elif self.normals_kind == "residue_prediction":        # :962-968: the decoder also sees the derived normal
                    derived = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
                    pred = self.renderModule_normal.rows(rec_xyz, derived, intr, n_dev)
                else:
                    pred = outs[3]
                    if self.normals_kind == "derived_plus_predicted":
                        derived = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
        bg = bool(white_bg or (is_train and torch.rand((1,)) < 0.5))
        smooth = None
        if words is not None:
            maps, smooth = ops.composite_primary_fused(rays, offsets, rec_w, rgb, brdf, brdf_j, pred, derived, acc, depth,
                                                       bg, is_relight, self.fixed_fresnel, words, rng_state, 1)
        else:
            maps = ops.composite_primary(rays, offsets, rec_w, rgb, brdf, brdf_j, pred, derived, acc, depth,
                                         bg, is_relight, self.fixed_fresnel)
            if rng_state is not None:
                rng_state[1] += 1
        if self.normals_kind not in NORMAL_LOSS_KINDS and is_relight:
            maps[:, 16] = 0.0        # the orientation loss is only filled in the branches that predict AND derive (:953-968)

        def finish():
            """True when the pass is valid; False when the record capacity overflowed (the caller re-runs)."""
            total = A if total_dev is None else (total_host.get() if total_host is not None else int(total_dev.item()))
            if len(hints) > 64:
                hints.clear()
            hints[(B, S)] = min(max(int(total * 1.25) + 4096, 1 << 14, int(0.97 * hints.get((B, S), 0))), B * S)      # decays slowly (alternating light / heavy batches)
            if total_dev is not None and total > cap:
                hints.pop((B, S), None)               # next call takes the exact (synchronising) route
                return False
            return True
        capture = self.__dict__.get("_capture")
        if capture is not None:                       # HIP-graph capture (tensoir_amd/graph.py): no host reads here;
            if total_dev is None:                     # the owner of the graph checks the counters after each replay
                raise TensoirHipError("graph capture needs a warmed-up record-capacity hint (run one eager call first)")
            capture.append((total_dev, cap, ("primary", B, S)))
        elif _defer_check:
            self.__dict__["_pending_primary"] = finish
        elif not finish():
            return self.forward(rays_chunk, light_idx, white_bg, is_train, ndc_ray, is_relight, N_samples,
                                _brdf_jitter_dense, _return_maps, False, _want_mask)
        out = self.unpack_maps(maps, is_relight, want_mask=_want_mask, smooth=smooth)
        return (out, maps) if _return_maps else out

    def _step_words(self, dev):
        """Persistent int32[8] of device-side counters shared by the kernels of one pass: [0] active (point, direction)
        pairs, [1:3] secondary record counter {total, written prefix}, [3] / [4] last-workgroup tickets of the primary
        march / the compositing kernel, [5] primary record total.  Zero at creation; every word is re-armed on the device
        by the pass itself."""
        w = self.__dict__.get("_words")
        if w is None or w.device != torch.device(dev):
            w = self.__dict__["_words"] = torch.zeros((8,), dtype=torch.int32, device=dev)
            self.__dict__["_pair_counter"] = w[0:1]
        return w

    def _jitter_rng(self, dev):
        """Device-side {seed, offset} of the BRDF-jitter noise; re-keyed when the framework's CUDA seed changes."""
        seed = int(torch.cuda.initial_seed())
        st = self.__dict__.get("_jit_rng")
        if st is None or st[0] != seed or st[1].device != torch.device(dev):
            if self.__dict__.get("_capture") is not None:
                raise TensoirHipError("the jitter RNG state must exist before a graph capture (run one eager call first)")
            t = torch.tensor([seed & (2 ** 63 - 1), 0], dtype=torch.int64).to(dev)
            st = self.__dict__["_jit_rng"] = (seed, t)
        return st[1]

    def _finish_primary(self):
        """Deferred overflow check of the last forward(..., _defer_check=True); True = results are valid."""
        fin = self.__dict__.pop("_pending_primary", None)
        return True if fin is None else fin()

    @staticmethod
    def unpack_maps(maps, is_relight=True, want_mask=True, smooth=None):
        """[B,20] map rows -> the reference's 12-tuple (:1033-1036 / :983-986).  want_mask=False leaves acc_mask out
        (None): Renderer_TensoIR_train shades from the map rows and never looks at it -- one launch less per step."""
        if not is_relight:
            return (maps[:, 0:3], maps[:, 3], None, None, None, None, maps[:, 14], None, None, None, None, None)
        if maps.requires_grad:
            # training: ONE autograd node for all the column groups (split_with_sizes: its backward is a single concatenation
            # of the incoming gradients) instead of one slice node per map, whose backward each fills a zero [B,20] buffer,
            # copies into it and is then summed with the others -- ~20 small launches of a host-bound stretch of the step
            rgb, depth, normal, albedo, rough, fresnel, acc, ndiff, norient, sm, _pad = torch.split(
                maps, [3, 1, 3, 3, 1, 3, 1, 1, 1, 2, 1], dim=1)
            depth, acc = depth.squeeze(-1), acc.squeeze(-1)
            if smooth is None:
                smooth = torch.mean(sm, dim=0)
            return (rgb, depth, normal, albedo, rough, fresnel, acc, ndiff, norient, (acc > 0.5) if want_mask else None,
                    smooth[0], smooth[1])
        acc = maps[:, 14]
        if smooth is None:
            smooth = torch.mean(maps[:, 17:19], dim=0)    # both smoothness losses in one reduction launch
        return (maps[:, 0:3], maps[:, 3], maps[:, 4:7], maps[:, 7:10], maps[:, 10:11], maps[:, 11:14], acc,
                maps[:, 15:16], maps[:, 16:17], (acc > 0.5) if want_mask else None, smooth[0], smooth[1])

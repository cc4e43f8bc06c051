"""A dataset with the interface ``train_tensoIR.py`` expects from ``dataLoader.dataset_dict[...]`` entries
(dataLoader/tensoIR_rotation_setting.py:16-140), generated analytically: there is no dataset offline.

Scene: a Lambert sphere of radius 0.8 with a banded albedo under one directional light per light rotation
(rotated about z, as models/tensorBase_rotated_lights.py:478-488) plus ambient, composited on white.  Cameras sit on
a ring of radius 4 looking at the origin; ray directions are L2-normalised (tensoIR_rotation_setting.py:105-106).

Selected by ``datadir = synthetic:views=6,res=48`` in the config (tensoir_amd.run wraps every dataset_dict entry so
that such a datadir builds this class and anything else goes to the reference's own loader).
"""
from __future__ import annotations

import math

import torch


def parse_spec(root_dir):
    spec = {"views": 6, "res": 48, "radius": 0.8, "fov": 0.6911, "cam": 4.0}
    text = str(root_dir)
    if ":" in text:
        for item in text.split(":", 1)[1].split(","):
            if "=" in item:
                k, v = item.split("=", 1)
                spec[k.strip()] = float(v) if "." in v else int(v)
    return spec


def is_synthetic(root_dir):
    return str(root_dir).startswith("synthetic")


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, root_dir, hdr_dir=None, split="train", random_test=False, N_vis=-1, downsample=1.0, sub=0,
                 light_rotation=("000",), light_name="sunset", light_name_list=None, is_stack=False, light_names=None,
                 **unused):
        spec = parse_spec(root_dir)
        self.split = split
        self.N_vis = N_vis
        self.downsample = downsample
        self.white_bg = True
        self.near_far = [2.0, 6.0]
        self.scene_bbox = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]) * downsample
        self.light_name = light_name
        names = light_name_list if light_name_list else list(light_rotation or ["000"])
        if light_names:          # the relighting test split (dataLoader/tensoIR_relighting_test.py:15-25): one G.T. image per map
            names = self.light_names = list(light_names)
        self.light_rotation = [str(r) for r in names]
        self.light_num = len(self.light_rotation)
        self.lights_probes = None
        res = max(4, int(spec["res"] / downsample))
        self.img_wh = (res, res)
        n_views = int(spec["views"]) if split == "train" else max(1, int(spec["views"]) // 3)
        if sub > 0:
            n_views = min(n_views, sub)
        self.n_views = n_views
        self.radius, self.cam, self.fov = float(spec["radius"]), float(spec["cam"]), float(spec["fov"])
        frames = [self._frame(v, l) for v in range(n_views) for l in range(self.light_num)]
        self.frames = frames
        self.all_rays = torch.cat([f["rays"] for f in frames], 0)                 # [N*H*W, 6]
        self.all_rgbs = torch.cat([f["rgbs"] for f in frames], 0)                 # [N*H*W, 3]
        self.all_masks = []                                                       # as the reference leaves it (:133)
        self.all_light_idx = torch.cat([f["light_idx"] for f in frames], 0)       # [N*H*W, 1] int8 (:130)

    # ---- geometry --------------------------------------------------------------------------------
    def _camera(self, v):
        phase = 0.0 if self.split == "train" else 0.5
        az = 2 * math.pi * (v + phase) / max(self.n_views, 1)
        el = 0.35 * math.sin(1.7 * v + 0.3)
        eye = torch.tensor([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)]) * self.cam
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        return eye, fwd, right, up

    def _frame(self, v, l):
        W, H = self.img_wh
        eye, fwd, right, up = self._camera(v)
        focal = 0.5 * W / math.tan(0.5 * self.fov)
        j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        d = ((i - W / 2 + 0.5) / focal)[..., None] * right - ((j - H / 2 + 0.5) / focal)[..., None] * up + fwd
        d = (d / d.norm(dim=-1, keepdim=True)).reshape(-1, 3)
        o = eye.expand_as(d)
        # ray / sphere intersection
        b = (o * d).sum(-1)
        disc = b * b - ((o * o).sum(-1) - self.radius ** 2)
        hit = disc > 0
        t = -b - torch.sqrt(disc.clamp(min=0))
        p = o + t[:, None] * d
        n = p / self.radius
        ang = math.radians(float(int(self.light_rotation[l]) if self.light_rotation[l].isdigit() else 40 * l))
        light = torch.tensor([math.cos(ang) * 0.6, math.sin(ang) * 0.6, 0.8])
        light = light / light.norm()
        albedo = torch.stack([0.55 + 0.35 * torch.sin(6 * p[:, 2]), 0.5 + 0.3 * torch.sin(5 * p[:, 0] + 1.0),
                              0.45 + 0.25 * torch.cos(4 * p[:, 1])], -1)
        shade = (n * light).sum(-1).clamp(min=0)[:, None] * 0.8 + 0.2
        rgb = torch.where(hit[:, None], (albedo * shade).clamp(0, 1), torch.ones_like(albedo))
        normals = torch.where(hit[:, None], n, torch.tensor([0.0, 0.0, 1.0]).expand_as(n))
        return {"rays": torch.cat([o, d], -1).contiguous(), "rgbs": rgb.contiguous(),
                "light_idx": torch.full((d.shape[0], 1), l, dtype=torch.int8),
                "rgbs_mask": hit[:, None], "normals": normals, "albedo": torch.where(hit[:, None], albedo, torch.ones_like(albedo))}

    # ---- Dataset protocol (test-split consumers index whole frames) ----------------------------------
    def __len__(self):
        return self.n_views

    def __getitem__(self, idx):
        fs = self.frames[idx * self.light_num:(idx + 1) * self.light_num]
        return {"img_wh": self.img_wh, "light_idx": torch.stack([f["light_idx"] for f in fs]),
                "rgbs": torch.stack([f["rgbs"] for f in fs]), "rgbs_mask": fs[0]["rgbs_mask"],
                "rays": fs[0]["rays"], "normals": fs[0]["normals"], "albedo": fs[0]["albedo"]}


def _fits(ds, device, share=0.25):
    """The training table (rays, colours, light indices) against `share` of the device's free memory."""
    need = sum(t.numel() * t.element_size() for t in (getattr(ds, n, None) for n in ("all_rays", "all_rgbs", "all_light_idx")) if torch.is_tensor(t))
    try:
        free, _ = torch.cuda.mem_get_info(torch.device(device))
    except Exception:
        return False
    return need <= share * free


def _to_device(ds, device):
    """Training-loop residency: keep the training rays / colours / light indices of a dataset in HBM, so that
    the unmodified loop's ``rays_filtered[rays_idx]`` (train_tensoIR.py:239-242) gathers on the device and only the 32 KB
    index tensor crosses PCIe per step (the script builds its permutation on the host with numpy, :49)."""
    moved = 0
    for name in ("all_rays", "all_rgbs", "all_light_idx"):
        t = getattr(ds, name, None)
        if torch.is_tensor(t):
            setattr(ds, name, t.to(device))
            moved += t.numel() * t.element_size()
    print(f"[tensoir_amd] training table resident on {device}: {moved / 2 ** 20:.1f} MiB (rays, colours, light indices; "
          f"TENSOIR_DEVICE_DATASET=0 keeps it on the host)", flush=True)
    return ds


def _light_defaults(cls):
    import inspect
    try:
        params = inspect.signature(cls.__init__).parameters
    except (TypeError, ValueError):
        return {}
    return {k: p.default for k, p in params.items()
            if k in ("light_name_list", "light_rotation", "light_names", "light_name") and p.default is not inspect.Parameter.empty}


def wrap_dataset_dict(dataset_dict, device=None, only_if_fits=False):
    """Every entry keeps its reference class for real data directories and builds the analytic dataset when
    ``datadir`` starts with ``synthetic``.  device (e.g. 'cuda'): training splits are moved to that device once
    (only_if_fits: when the table takes less than a quarter of the free device memory)."""
    for name, cls in list(dataset_dict.items()):
        if getattr(cls, "__tensoir_wrapped__", False):
            continue

        def factory(root_dir, *a, _cls=cls, **k):
            if is_synthetic(root_dir):
                # a caller that leaves the light set to the class default (render_test() of the general multi-light script,
                # train_tensoIR_general_multi_lights.py:74) gets THAT reference class's default, not ours
                k = {**_light_defaults(_cls), **k}
            ds = SyntheticDataset(root_dir, *a, **k) if is_synthetic(root_dir) else _cls(root_dir, *a, **k)
            if device is not None and k.get("split", "train") == "train" and (not only_if_fits or _fits(ds, device)):
                _to_device(ds, device)
            return ds
        factory.__tensoir_wrapped__ = True
        factory.__name__ = getattr(cls, "__name__", name)
        dataset_dict[name] = factory
    return dataset_dict

"""General multi-light variant: ``models/tensoRF_general_multi_lights.py`` +
``models/tensorBase_general_multi_lights.py``.  Identical hot path; only the environment light differs:
one spherical-Gaussian set per light (``lgtSGs_list``) instead of one set seen under per-light rotations
(:463-479, :575-582).  The class keeps the reference's name so that
``from models.tensoRF_general_multi_lights import TensorVMSplit`` can be rebound to it (tensoir_amd.run)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .field_model import TensorVMSplit as _RotatedTensorVMSplit
from .field_model import compute_energy, fibonacci_sphere


class TensorVMSplit(_RotatedTensorVMSplit):
    def __init__(self, aabb, gridSize, device, light_name_list=("sunset", "snow", "courtyard"), **kargs):
        self.light_name_list = list(light_name_list)
        kargs.pop("light_rotation", None)
        # the base class sizes light_line / light_num from the rotation list: one (unused) entry per light
        super().__init__(aabb, gridSize, device, light_rotation=["000"] * len(self.light_name_list), **kargs)

    def init_light(self):
        """models/tensorBase_general_multi_lights.py:455-479: an independent SG set per light.  Like the reference
        the sets live in a plain python list (they are optimised -- get_optparam_groups -- but not part of the
        state_dict)."""
        self.light_area_weight, self.fixed_viewdirs = self.generate_envir_map_dir(self.envmap_h, self.envmap_w)
        if self.light_kind != "sg":
            raise NotImplementedError(f"light_kind={self.light_kind!r}: only 'sg' has gfx950 kernels")
        self.lgtSGs_list = []
        for _ in range(self.light_num):
            sg = nn.Parameter(torch.randn(self.numLgtSGs, 7), requires_grad=True)
            sg.data[:, -2:] = sg.data[:, -3:-2].expand((-1, 2))
            sg.data[:, 3:4] = 10.0 + torch.abs(sg.data[:, 3:4] * 20.0)
            energy = compute_energy(sg.data)
            sg.data[:, 4:] = torch.abs(sg.data[:, 4:]) / torch.sum(energy, dim=0, keepdim=True) * 2.0 * np.pi * 0.8
            lobes = fibonacci_sphere(self.numLgtSGs // 2).astype(np.float32)
            sg.data[:self.numLgtSGs // 2, :3] = torch.from_numpy(lobes)
            sg.data[self.numLgtSGs // 2:, :3] = torch.from_numpy(lobes)
            sg.data = sg.data.to(self.device)
            self.lgtSGs_list.append(sg)
        self.light_rotation_matrix = torch.eye(3, dtype=torch.float32)[None]

    @property
    def lgtSGs(self):
        """First light's set (callers that only test ``requires_grad`` / device)."""
        return self.lgtSGs_list[0]

    def get_light_rgbs(self, incident_light_directions=None, device="cuda"):
        """models/tensorBase_general_multi_lights.py:566-582: [light_num, D, 3], no rotation."""
        dirs = incident_light_directions.to(device).reshape(-1, 3).to(torch.float32).contiguous()
        eye = self.__dict__.get("_eye_dev")
        if eye is None or eye.device != dirs.device:
            eye = torch.eye(3, dtype=torch.float32, device=dirs.device)[None].contiguous()
            self.__dict__["_eye_dev"] = eye
        rows = []
        for sg in self.lgtSGs_list:
            if torch.is_grad_enabled() and sg.requires_grad:
                from . import training
                rows.append(training.EnvSGFn.apply(sg, eye, dirs)[0])
            else:
                rows.append(ops.env_sg(sg, eye, dirs)[0])
        return torch.stack(rows, dim=0)

    def _light_param_groups(self):
        """models/tensoRF_general_multi_lights.py:45-46."""
        return [{"params": sg, "lr": 0.001} for sg in self.lgtSGs_list]

    def get_kwargs(self):
        kw = super().get_kwargs()
        kw.pop("light_rotation", None)
        kw["light_name_list"] = self.light_name_list
        return kw

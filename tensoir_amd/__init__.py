"""tensoir_amd -- MI355X-native (gfx950) ray-march + PBR-shading hot path of TensoIR.

Mirrors the reference's call surface (TensorVMSplit / Renderer_TensoIR_train / render_with_BRDF ...);
all per-sample work runs in libtensoir_hip.so through the C ABI of include/tensoir_hip.h.
"""
from .field_model import AlphaGridMask, TensorVMSplit, raw2alpha  # noqa: F401
from .relight import (Environment_Light, GGX_specular, compute_radiance,  # noqa: F401
                      compute_secondary_shading_effects, compute_transmittance, render_with_BRDF,
                      relight_with_envmap)
from .renderer import Renderer_TensoIR_train  # noqa: F401
from . import general_multi_lights  # noqa: F401  (TensorVMSplit with one SG set per light)
from . import optim  # noqa: F401  (Adam with a single-launch step())

__version__ = "0.1.0"


def model_from_checkpoint(ckpt, device="cuda", **extra):
    """Rebuild a model the way train_tensoIR.py:163-168 does: TensorVMSplit(**kwargs).load(ckpt)."""
    kwargs = dict(ckpt["kwargs"])
    kwargs.pop("light_num", None)
    kwargs["light_rotation"] = [f"{int(r):03d}" for r in kwargs["light_rotation"]]
    kwargs.update({"device": device})
    kwargs.update(extra)
    model = TensorVMSplit(**kwargs)
    model.load(ckpt)
    return model

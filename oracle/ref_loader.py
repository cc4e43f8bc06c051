"""Import the *reference* TensoIR python modules (read-only checkout) on CPU.

TEST INFRASTRUCTURE ONLY -- used by oracle/make_golden.py and by the optional
``-m "not gpu"`` cross-checks that run only when the checkout is present (the
build container).  Nothing here is available on the GPU box and nothing in the
product imports it.  Recipe: SURVEY.md section 8(c).
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("TENSOIR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference modules (models.*, renderer)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    try:
        import cv2  # noqa: F401
    except Exception:
        _stub("cv2", COLORMAP_JET=2)
    try:
        import loguru  # noqa: F401
    except Exception:
        lg = types.SimpleNamespace(debug=lambda *a, **k: None, info=lambda *a, **k: None,
                                   warning=lambda *a, **k: None)
        _stub("loguru", logger=lg)
    try:
        import kornia  # noqa: F401
    except Exception:
        _stub("kornia", create_meshgrid=None)
    for name in ("torchvision", "imageio", "plyfile", "skimage", "lpips"):
        try:
            __import__(name)
        except Exception:
            _stub(name)
    if "torchvision" in sys.modules and not hasattr(sys.modules["torchvision"], "transforms"):
        tr = _stub("torchvision.transforms", Compose=None, ToTensor=None)
        ut = _stub("torchvision.utils")
        sys.modules["torchvision"].transforms = tr
        sys.modules["torchvision"].utils = ut
    if "skimage" in sys.modules and not hasattr(sys.modules["skimage"], "measure"):
        sys.modules["skimage"].measure = _stub("skimage.measure")
    if "plyfile" in sys.modules and not hasattr(sys.modules["plyfile"], "PlyData"):
        sys.modules["plyfile"].PlyData = None
        sys.modules["plyfile"].PlyElement = None

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    tensorf = importlib.import_module("models.tensoRF_rotated_lights")
    base = importlib.import_module("models.tensorBase_rotated_lights")
    RU = importlib.import_module("models.relight_utils")
    renderer = importlib.import_module("renderer")

    # CPU only: sample_ray_equally defaults device='cuda' (models/relight_utils.py:708) and
    # its callers never forward `device` (:672-678, :792-798).
    orig = RU.sample_ray_equally
    if not getattr(orig, "_cpu_patched", False):
        def sample_ray_equally_cpu(tensoIR, rays_o, rays_d, nSample=-1, vis_near=0.03,
                                   vis_far=1.5, device=None):
            return orig(tensoIR, rays_o, rays_d, nSample=nSample, vis_near=vis_near,
                        vis_far=vis_far, device=rays_o.device)
        sample_ray_equally_cpu._cpu_patched = True
        RU.sample_ray_equally = sample_ray_equally_cpu
    return types.SimpleNamespace(tensorf=tensorf, base=base, RU=RU, renderer=renderer,
                                 TensorVMSplit=tensorf.TensorVMSplit)

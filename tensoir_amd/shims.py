"""Stand-ins for the third-party modules the reference scripts import but this image does not ship
(SURVEY.md section 8c): cv2, loguru, kornia, torchvision, imageio, plyfile, skimage, lpips, configargparse and
tensorboard.  Product side (used by ``python -m tensoir_amd.run``): a module is only replaced when the real one
cannot be imported.  None of them is on the hot path -- they are image I/O, logging, metrics and CLI parsing --
so the stand-ins either implement the few functions the training script actually calls (``configargparse``,
``SummaryWriter``, ``kornia.create_meshgrid``, ``imageio.imwrite`` through PIL) or raise a clear error on use.
scripts/relight_importance.py additionally writes videos, reads its own PNGs back, colour-maps depth and asks for LPIPS:
``imageio.mimsave`` / ``imageio.v2.imread`` go through PIL, ``cv2.applyColorMap`` is a small numpy jet ramp, and ``lpips.LPIPS``
-- whose pretrained networks cannot exist offline -- reports NaN (the metric is then visibly unmeasured, nothing else changes).
"""
from __future__ import annotations

import argparse
import importlib
import re
import sys
import types


def _missing(name):
    try:
        importlib.import_module(name)
        return False
    except Exception:
        return True


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__tensoir_shim__ = True
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


def _unavailable(mod, fn):
    def raiser(*a, **k):
        raise RuntimeError(f"{mod}.{fn} is not available in this environment (tensoir_amd.shims stands in for {mod}); "
                           f"install {mod} to use this code path")
    raiser.__name__ = fn
    return raiser


class _Lazy(types.ModuleType):
    """A module whose unknown attributes are functions that raise on CALL (so `import x; x.foo` at import time works)."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _unavailable(self.__name__, item)


def _lazy(name, **attrs):
    m = _Lazy(name)
    m.__dict__.update(attrs)
    m.__tensoir_shim__ = True
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


# ----------------------------------------------------------------------------------------------- configargparse
_LIST = re.compile(r"^\[(.*)\]$")


def parse_config_text(text):
    """``key = value`` lines (``#`` / ``;`` comments, blank lines, optional quotes, ``[a, b]`` lists) -> [(key, value)]
    where value is a string or a list of strings -- the subset of configargparse's default syntax the TensoIR
    configs use (configs/single_light/armadillo.txt)."""
    items = []
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line or line.startswith(";") or line.startswith("---"):
            continue
        if "=" in line:
            key, val = line.split("=", 1)
        elif ":" in line:
            key, val = line.split(":", 1)
        else:
            key, val = line, "true"
        key, val = key.strip().lstrip("-"), val.strip()
        m = _LIST.match(val)
        if m:
            val = [v.strip().strip("'\"") for v in m.group(1).split(",") if v.strip()]
        else:
            val = val.strip("'\"")
        items.append((key, val))
    return items


class ConfigArgumentParser(argparse.ArgumentParser):
    """argparse + ``is_config_file=True`` arguments: values from the config file act as defaults that the command
    line overrides (configargparse's precedence: command line > config file > add_argument defaults)."""

    def __init__(self, *a, **k):
        k.pop("default_config_files", None)
        k.pop("config_file_parser_class", None)
        super().__init__(*a, **k)
        self._config_dests = []

    def add_argument(self, *names, **kw):
        is_cfg = kw.pop("is_config_file", False) or kw.pop("is_config_file_arg", False)
        kw.pop("env_var", None)
        action = super().add_argument(*names, **kw)
        if is_cfg:
            self._config_dests.append(action.dest)
        return action

    add = add_argument

    def _config_argv(self, path, cli):
        given = {a.split("=", 1)[0].lstrip("-") for a in cli if a.startswith("--")}
        by_name = {}
        for act in self._actions:
            for opt in act.option_strings:
                by_name[opt.lstrip("-")] = act
        out = []
        with open(path) as fh:
            for key, val in parse_config_text(fh.read()):
                act = by_name.get(key)
                if act is None:
                    self.error(f"unrecognized config-file key: {key}")
                if key in given or any(o.lstrip("-") in given for o in act.option_strings):
                    continue                                     # the command line wins
                opt = act.option_strings[0]
                if isinstance(act, (argparse._StoreTrueAction, argparse._StoreFalseAction, argparse._StoreConstAction)):
                    truth = str(val).lower() in ("1", "true", "yes", "on")
                    if truth:
                        out.append(opt)
                elif isinstance(val, list):
                    for v in val:
                        out += [opt, v]
                else:
                    out += [opt, val]
        return out

    def parse_known_args(self, args=None, namespace=None):
        cli = list(sys.argv[1:] if args is None else args)
        if self._config_dests:
            probe = argparse.ArgumentParser(add_help=False)
            for act in self._actions:
                if act.dest in self._config_dests:
                    probe.add_argument(*act.option_strings, dest=act.dest, default=act.default)
            known, _ = probe.parse_known_args(cli)
            extra = []
            for dest in self._config_dests:
                path = getattr(known, dest, None)
                if path:
                    extra += self._config_argv(path, cli)
            cli = extra + cli
        return super().parse_known_args(cli, namespace)


ArgParser = ConfigArgumentParser


# ----------------------------------------------------------------------------------------------- tensorboard
class SummaryWriter:
    """No-op ``torch.utils.tensorboard.SummaryWriter``: keeps the last value of every tag (handy in tests)."""

    def __init__(self, log_dir=None, *a, **k):
        self.log_dir = log_dir
        self.scalars = {}

    def add_scalar(self, tag, value, global_step=None, *a, **k):
        try:
            value = float(value)
        except Exception:
            pass
        self.scalars[tag] = (global_step, value)

    def __getattr__(self, item):
        if item.startswith("add_") or item in ("flush", "close"):
            return lambda *a, **k: None
        raise AttributeError(item)


# ----------------------------------------------------------------------------------------------- small real pieces
def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    """kornia.create_meshgrid: [1,H,W,2] grid of (x, y) pixel coordinates (dataLoader/ray_utils.py:37)."""
    import torch
    dtype = dtype or torch.float32
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / max(width - 1, 1) - 0.5) * 2
        ys = (ys / max(height - 1, 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).unsqueeze(0)


def _imwrite(path, img, *a, **k):
    import numpy as np
    from PIL import Image
    arr = np.asarray(img)
    if arr.dtype != np.uint8:
        arr = (np.clip(arr, 0, 1) * 255).astype(np.uint8)
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[:, :, 0]
    Image.fromarray(arr).save(path)


def _imread(path, *a, **k):
    import numpy as np
    from PIL import Image
    return np.asarray(Image.open(path))


def _mimsave(path, frames, *a, **k):
    """No video encoder offline: the frames go into an animated PNG at the requested path."""
    import numpy as np
    from PIL import Image
    imgs = [Image.fromarray(np.asarray(f).astype(np.uint8)) for f in frames]
    if imgs:
        if not _mimsave.__dict__.get("warned"):
            _mimsave.warned = True
            import warnings
            warnings.warn(f"tensoir_amd.shims: no video encoder in this environment -- {path} and every later imageio.mimsave "
                          "output is written as an ANIMATED PNG (APNG container, whatever the file extension says); players that "
                          "expect a video stream will not open it", stacklevel=2)
        with open(path, "wb") as fh:
            imgs[0].save(fh, format="PNG", save_all=True, append_images=imgs[1:])


def _apply_color_map(x, cmap=2):
    """cv2.applyColorMap for utils.visualize_depth_numpy (:30): uint8 [H,W] -> BGR uint8 [H,W,3], piecewise-linear jet."""
    import numpy as np
    v = np.asarray(x, dtype=np.float32).reshape(np.asarray(x).shape[:2]) / 255.0
    r = np.clip(1.5 - np.abs(4 * v - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * v - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * v - 1), 0, 1)
    return (np.stack([b, g, r], -1) * 255).astype(np.uint8)


class _NoLPIPS:
    """lpips.LPIPS needs pretrained AlexNet / VGG weights; offline the metric is reported as NaN."""

    def __init__(self, *a, **k):
        import warnings
        warnings.warn("lpips is not installed: LPIPS metrics are reported as NaN (tensoir_amd.shims)")

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, gt, im, *a, **k):
        import torch
        return torch.full((1,), float("nan"))


def _make_grid(tensor, nrow=8, padding=2, normalize=False, value_range=None, scale_each=False, pad_value=0.0, **k):
    """torchvision.utils.make_grid for the evaluation loop's tensorboard images (renderer.py:443-452): [N,C,H,W] -> [C,H',W']."""
    import torch
    t = torch.stack(list(tensor)) if isinstance(tensor, (list, tuple)) else tensor
    t = t.float()
    if t.dim() == 3:
        t = t[None]
    if normalize:
        lo, hi = value_range if value_range is not None else (float(t.min()), float(t.max()))
        t = (t.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
    n, c, h, w = t.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = t.new_full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), pad_value)
    for i in range(n):
        y, x = (i // cols) * (h + padding) + padding, (i % cols) * (w + padding) + padding
        grid[:, y:y + h, x:x + w] = t[i]
    return grid


class _ToTensor:
    def __call__(self, pic):
        import numpy as np
        import torch
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1)
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class _Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


# ----------------------------------------------------------------------------------------------- install
def install():
    """Install the stand-ins for whatever is missing; returns the list of module names that were shimmed."""
    done = []
    if _missing("cv2"):
        _lazy("cv2", COLORMAP_JET=2, IMREAD_UNCHANGED=-1, COLOR_BGR2RGB=4, COLOR_RGB2BGR=4, INTER_AREA=3, INTER_LINEAR=1,
              applyColorMap=_apply_color_map)
        done.append("cv2")
    if _missing("loguru"):
        log = types.SimpleNamespace(**{k: (lambda *a, **kw: None) for k in
                                       ("debug", "info", "warning", "error", "success", "critical", "add", "remove")})
        _module("loguru", logger=log)
        done.append("loguru")
    if _missing("kornia"):
        _module("kornia", create_meshgrid=create_meshgrid)
        done.append("kornia")
    if _missing("torchvision"):
        tv = _lazy("torchvision")
        _lazy("torchvision.transforms", Compose=_Compose, ToTensor=_ToTensor)
        _lazy("torchvision.utils", make_grid=_make_grid)
        tv.__path__ = []
        done.append("torchvision")
    if _missing("imageio"):
        io = _lazy("imageio", imwrite=_imwrite, imsave=_imwrite, imread=_imread, mimsave=_mimsave, mimwrite=_mimsave)
        io.__path__ = []
        _lazy("imageio.v2", imwrite=_imwrite, imread=_imread, mimsave=_mimsave)
        done.append("imageio")
    if _missing("plyfile"):
        _lazy("plyfile", PlyData=None, PlyElement=None)
        done.append("plyfile")
    if _missing("skimage"):
        sk = _lazy("skimage")
        sk.__path__ = []
        _lazy("skimage.measure")
        _lazy("skimage.metrics")
        done.append("skimage")
    if _missing("lpips"):
        _lazy("lpips", LPIPS=_NoLPIPS)
        done.append("lpips")
    if _missing("configargparse"):
        _module("configargparse", ArgumentParser=ConfigArgumentParser, ArgParser=ConfigArgumentParser,
                ArgumentDefaultsHelpFormatter=argparse.ArgumentDefaultsHelpFormatter, Namespace=argparse.Namespace)
        done.append("configargparse")
    if _missing("torch.utils.tensorboard"):
        import torch.utils
        _module("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
        done.append("torch.utils.tensorboard")
    return done

"""Launcher that runs an UNMODIFIED TensoIR script on top of the HIP path:

    python -m tensoir_amd.run /path/to/TensoIR/train_tensoIR.py --config configs/single_light/armadillo.txt --render_only 1 --ckpt ...

It puts the TensoIR checkout on sys.path, installs stand-ins for the third-party modules this image lacks
(tensoir_amd/shims.py: cv2, loguru, kornia, torchvision, imageio, plyfile, skimage, lpips, configargparse,
tensorboard -- image I/O / logging / CLI, none on the hot path), imports the reference modules, rebinds the
hot-path symbols to the tensoir_amd implementations and then runs the script with runpy.
The script's own imports (`from renderer import *`, `from models.tensoRF_rotated_lights import raw2alpha,
TensorVMSplit, AlphaGridMask`, train_tensoIR.py:6-14) then resolve to ours.  With `datadir = synthetic:views=6,res=48`
in the config the dataset is generated analytically (tensoir_amd/synth_dataset.py; no dataset exists offline).
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys

PATCHES = {
    "models.tensoRF_rotated_lights": ["TensorVMSplit", "AlphaGridMask", "raw2alpha"],
    "models.tensorBase_rotated_lights": ["AlphaGridMask", "raw2alpha"],
    "models.tensoRF_general_multi_lights": ["TensorVMSplit:general", "AlphaGridMask", "raw2alpha"],
    "models.tensorBase_general_multi_lights": ["AlphaGridMask", "raw2alpha"],
    "models.relight_utils": ["render_with_BRDF", "compute_radiance", "compute_transmittance",
                             "compute_secondary_shading_effects", "GGX_specular", "brdf_specular",
                             "Environment_Light"],
    "renderer": ["Renderer_TensoIR_train", "render_with_BRDF"],
}


def install(reference_root: str):
    """Import the reference modules from `reference_root` and rebind the hot path.  Returns the
    {module: [symbols]} actually patched."""
    import tensoir_amd
    from tensoir_amd import field_model, general_multi_lights, relight, renderer, shims, synth_dataset
    shims.install()
    ours = {}
    for mod in (field_model, relight, renderer):
        ours.update({k: getattr(mod, k) for k in dir(mod) if not k.startswith("_")})
    ours["brdf_specular"] = relight.GGX_specular
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    done = {}
    _host_threads()
    if os.environ.get("TENSOIR_LAUNCH_MODE", "hip") == "reference":
        # A/B aid (TENSOIR_LAUNCH_MODE=reference): the same launcher, stand-ins and analytic dataset, but NOTHING rebound --
        # the script runs the reference's own PyTorch implementation (on the GPU through PyTorch-ROCm, or on the host)
        synth_dataset.wrap_dataset_dict(importlib.import_module("dataLoader").dataset_dict, device=None)
        _allow_numpy_in_checkpoints()
        return done
    for name, symbols in PATCHES.items():
        m = importlib.import_module(name)
        for s in symbols:
            if s.endswith(":general"):          # the per-light-SG variant of the same class name
                s = s.split(":")[0]
                setattr(m, s, getattr(general_multi_lights, s))
            else:
                setattr(m, s, ours[s])
        done[name] = [s.split(":")[0] for s in symbols]
    # optimizer.step() of the training loop (train_tensoIR.py:197, :317) on one launch; TENSOIR_TORCH_ADAM=1 keeps torch's
    if os.environ.get("TENSOIR_TORCH_ADAM", "0") != "1":
        import torch
        from tensoir_amd import optim
        torch.optim.Adam = optim.LauncherAdam       # the script's optimizer: one launch; anything unsupported: torch's own step
        done["torch.optim"] = ["Adam"]
    # The training rays stay resident in HBM (batches are gathered on the device: only the 32 KB index tensor crosses PCIe per
    # step) whenever the ray table fits -- TENSOIR_DEVICE_DATASET = auto (default: rays + colours + light indices below a quarter
    # of the free HBM; 100 views of 800 x 800 are 2.6 GB of 288), 1 (always), 0 (the script's host tensors)
    dev = None
    mode = os.environ.get("TENSOIR_DEVICE_DATASET", "auto")
    if mode in ("1", "auto"):
        import torch
        # the device the script will select (`cuda:{args.local_rank}`, train_tensoIR.py:25-29; main() passes LOCAL_RANK on as
        # --local_rank), not whatever is current while the dataset is built: several ranks must not all allocate on GPU 0
        dev = f"cuda:{_local_rank()}" if torch.cuda.is_available() else None
    synth_dataset.wrap_dataset_dict(importlib.import_module("dataLoader").dataset_dict, device=dev, only_if_fits=(mode == "auto"))
    # data-parallel training of the unmodified script under torchrun (tensoir_amd/dist.py LAUNCHER_DP); TENSOIR_LAUNCHER_DP=0 keeps
    # the reference's behaviour (N identical trainers, train_tensoIR.py:22-27)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("TENSOIR_LAUNCHER_DP", "1") != "0":
        from tensoir_amd import dist as tdist
        tdist.LAUNCHER_DP["on"] = True
        done["tensoir_amd.dist"] = ["LAUNCHER_DP"]
    _allow_numpy_in_checkpoints()
    return done


def _local_rank(argv=None) -> int:
    """--local_rank of the script's command line if given, else LOCAL_RANK (torchrun), else 0."""
    argv = sys.argv if argv is None else argv
    for i, a in enumerate(argv):
        if a == "--local_rank" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--local_rank="):
            return int(a.split("=", 1)[1])
    return int(os.environ.get("LOCAL_RANK", "0"))


def _host_threads():
    """The training loop's host side is a handful of tiny CPU tensor ops per iteration (three 4096-row gathers from the ray
    table, train_tensoIR.py:240-242).  PyTorch sizes its OpenMP team to the core count; on a 128-core / 256-thread MI355X host
    every such op then costs ~2 ms of team start-up (measured: 36.8 ms per iteration of the unmodified script against 8.7 ms
    with 8 threads, 7.6 ms of it GPU work; profiles/r03_script_head_to_head.json).  Unless the user chose a thread count
    (OMP_NUM_THREADS / TENSOIR_HOST_THREADS=0 keeps PyTorch's default) the launcher uses 8."""
    if "OMP_NUM_THREADS" in os.environ:
        return
    n = int(os.environ.get("TENSOIR_HOST_THREADS", "8"))
    if n > 0:
        import torch
        n = min(n, os.cpu_count() or n)
        torch.set_num_threads(n)
        # a process-wide side effect of install(): say so once (CPU-heavy phases of an unmodified script -- real-dataset ray
        # generation, image metrics -- run on this many threads too; INTEGRATION.md "Host threads")
        print(f"[tensoir_amd.run] PyTorch intra-op threads set to {n} for this process (TENSOIR_HOST_THREADS=N or OMP_NUM_THREADS "
              f"choose another count, TENSOIR_HOST_THREADS=0 keeps PyTorch's default of {os.cpu_count()})", file=sys.stderr, flush=True)


def _allow_numpy_in_checkpoints():
    """The scripts call ``torch.load(args.ckpt, map_location=device)`` (train_tensoIR.py:54, :77, :164); TensoIR checkpoints
    carry the occupancy mask as ``np.packbits`` bytes (tensorBase_rotated_lights.py:681), which torch >= 2.6 refuses under its
    new ``weights_only=True`` default.  Allow-list numpy's array reconstruction so that --ckpt / --render_only keep working."""
    import numpy as np
    import torch
    try:
        import numpy._core.multiarray as ma
    except ImportError:                          # numpy 1.x
        import numpy.core.multiarray as ma
    if hasattr(torch.serialization, "add_safe_globals"):
        torch.serialization.add_safe_globals([ma._reconstruct, np.ndarray, np.dtype] + sorted(
            {type(np.dtype(t)) for t in (np.uint8, np.bool_, np.float32, np.float64, np.int32, np.int64)}, key=repr))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    root = os.path.dirname(script)
    if os.path.basename(root) == "scripts":
        root = os.path.dirname(root)
    rest = argv[1:]
    # torchrun exports LOCAL_RANK; the scripts read args.local_rank (opt.py:21, the older torch.distributed.launch convention)
    if "LOCAL_RANK" in os.environ and not any(a == "--local_rank" or a.startswith("--local_rank=") for a in rest):
        rest = rest + ["--local_rank", os.environ["LOCAL_RANK"]]
    sys.argv = [script] + rest
    install(root)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()

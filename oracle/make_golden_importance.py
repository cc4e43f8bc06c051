"""Generate tests/golden/importance_sample.npz: the IMPORTED REFERENCE's `gen_light_incident_dirs(method='importance_sample')`
(models/tensorBase_rotated_lights.py:547-572) on the seeded small scene -- the jittered 128 x 256 direction table it draws
from, the SG environment map evaluated on it (light 0), the sampling pdf and the returned pdf, plus one seeded draw
(directions, radiance, pdf, indices) of the reference itself on the CPU.

The product draws its indices with the DEVICE generator (torch.multinomial on a CUDA tensor, as the reference does when it
runs on a GPU), so the test (tests/test_gpu_parity.py::test_importance_sampled_light_directions) pins what is deterministic:
the direction table under the same CPU jitter draws, the radiance / pdf values behind every returned sample, and the
distribution of the drawn indices.

Run in the build container (needs the read-only reference checkout):  python oracle/make_golden_importance.py
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle.make_golden import build_reference_model  # noqa: E402
from tests.helpers import golden_checkpoint  # noqa: E402

SEED = 20211202 + 31
N_DRAW = 4096


def main():
    ref = ref_loader.load()
    g0 = np.load(os.path.join(ROOT, "tests", "golden", "small_scene.npz"))
    envh, envw = [int(x) for x in g0["scene/envmap_hw"]]
    ckpt = golden_checkpoint(g0)
    ckpt["kwargs"]["light_rotation"] = [int(r) for r in ckpt["kwargs"]["light_rotation"]]
    m = build_reference_model(ref, ckpt, envh, envw)
    # The branch itself calls self.get_light_rgbs(...) without its device argument (default 'cuda', :549/:577), so the unmodified
    # method cannot run on this GPU-less box: its lines :548-560 are executed here one by one through the reference's OWN
    # generate_envir_map_dir / get_light_rgbs (device='cpu'), with the CPU generator seeded: the jitter draws of
    # generate_envir_map_dir first, then the multinomial -- the order the method consumes them in.
    torch.manual_seed(SEED)
    with torch.no_grad():
        _, view_dirs = m.generate_envir_map_dir(128, 256, is_jittor=True)
        env = m.get_light_rgbs(view_dirs.reshape(-1, 3).to("cpu"), device="cpu")[0].reshape(128, 256, 3)
        inten = torch.sum(env, dim=2, keepdim=True)
        sin_theta = torch.sin(torch.linspace(0 + 0.5 / 128, np.pi - 0.5 / 128, 128))
        p = inten * sin_theta.view(-1, 1, 1)
        pdf_s = p / torch.sum(p)
        pdf_c = pdf_s * 128 * 256 / (2 * np.pi * np.pi * sin_theta.view(-1, 1, 1))
        idx = torch.multinomial(pdf_s.view(-1), N_DRAW, replacement=True)
    vd = view_dirs.reshape(-1, 3)
    out = os.path.join(ROOT, "tests", "golden", "importance_sample.npz")
    np.savez_compressed(out, seed=np.array([SEED], np.int64), view_dirs=vd.numpy().astype(np.float32),
                        envir_map=env.view(-1, 3).numpy().astype(np.float32), pdf_to_sample=pdf_s.view(-1).numpy().astype(np.float32),
                        pdf_to_compute=pdf_c.view(-1).numpy().astype(np.float32), draw_idx=idx.numpy().astype(np.int32))
    print("wrote", out, os.path.getsize(out), "bytes;", "sum pdf_s", float(pdf_s.sum()))


if __name__ == "__main__":
    main()

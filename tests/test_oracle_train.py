"""Pins oracle.train_step_grads (autograd through the functional restatement) against the parameter
gradients of one training step of the imported reference (tests/golden/train_grads.npz, written by
oracle/make_golden_train.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tensoir_oracle as O
from tests.helpers import T, golden_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20211202


@pytest.fixture(scope="module")
def tg():
    return np.load(os.path.join(ROOT, "tests", "golden", "train_grads.npz"))


def grad_err(a, b):
    """max |a-b| / max |b| over the tensor (scale-relative: gradients span many magnitudes)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def run_oracle(golden, tg, tag, relight, stratified=False):
    sc = golden_scene(golden)
    rays, lidx = T(golden, "rays/rays"), T(golden, "rays/light_idx")
    n2, near, far = tg["train/second"]
    S = int(tg["train/n_samples"][0])
    gt = T(tg, "train/rgb_gt")
    dir_jitter = None
    if stratified:   # replay: rand [B,1], randn [A,3], then two rand [envh,envw]
        torch.manual_seed(SEED + 12)
        jit = torch.rand(rays.shape[0], 1)
        O.forward_primary(sc, rays, lidx.int(), n_samples=S, ray_jitter=jit)
        dir_jitter = (torch.rand(sc.envmap_h, sc.envmap_w), torch.rand(sc.envmap_h, sc.envmap_w))
    torch.manual_seed(SEED + 12)
    jit = torch.rand(rays.shape[0], 1)
    assert np.array_equal(jit.numpy(), tg[f"train_{tag}/ray_jitter"])
    return O.train_step_grads(sc, rays, lidx, gt, is_relight=relight, n_samples=S, ray_jitter=jit,
                              dir_jitter=dir_jitter, second_n_sample=int(n2), second_near=near, second_far=far)


@pytest.mark.parametrize("tag,relight,strat", [("A", False, False), ("B", True, False), ("C", True, True)])
def test_train_grads_vs_reference(golden, tg, tag, relight, strat):
    loss, grads, ret = run_oracle(golden, tg, tag, relight, strat)
    assert abs(float(loss) - float(tg[f"train_{tag}/loss"][0])) < 2e-6
    for k in ("rgb_map", "acc_map", "rgb_with_brdf_map"):
        assert float((ret[k] - T(tg, f"train_{tag}/out/{k}")).abs().max()) < 3e-5, k
    checked = 0
    for name, gr in grads.items():
        ref = tg[f"train_{tag}/grad/{name}"]
        if np.abs(ref).max() == 0:
            assert float(gr.abs().max()) == 0, name
            continue
        assert grad_err(gr, ref) < 2e-3, (name, grad_err(gr, ref))
        checked += 1
    assert checked >= (18 if not relight else 30)

"""Pins the oracle's normals_kind branches ('purely_predicted' = the reference's class default, 'purely_derived')
against tests/golden/normals_kinds.npz (imported reference, oracle/make_golden_kinds.py): forward maps and the
parameter gradients of one training step.  In both kinds normals_diff / normals_orientation_loss are zero
(models/tensorBase_rotated_lights.py:946-960).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tensoir_oracle as O
from tests.helpers import T, golden_checkpoint, scene_from_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20211202
NAMES = ["rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map",
         "acc_map", "normals_diff_map", "normals_orientation_loss_map", "acc_mask",
         "albedo_smoothness_loss", "roughness_smoothness_loss"]


@pytest.fixture(scope="module")
def kg():
    return np.load(os.path.join(ROOT, "tests", "golden", "normals_kinds.npz"))


def kind_checkpoint(golden, kind, kg=None):
    ck = golden_checkpoint(golden)
    ck["kwargs"]["normals_kind"] = kind
    if kind in ("purely_derived", "gt_normals"):
        ck["state_dict"] = {k: v for k, v in ck["state_dict"].items() if not k.startswith("renderModule_normal")}
    if kind == "residue_prediction":          # MLPNormal_normal_and_PExyz: a 153-wide layer 1 (the fixture's seeded weights)
        kg = np.load(os.path.join(ROOT, "tests", "golden", "normals_kinds.npz")) if kg is None else kg
        ck["state_dict"]["renderModule_normal.mlp.0.weight"] = T(kg, "residue_prediction/w0_normal_decoder")
    return ck


def kind_scene(golden, kind):
    ck = kind_checkpoint(golden, kind)
    eh, ew = [int(x) for x in golden["scene/envmap_hw"]]
    return scene_from_checkpoint(ck, eh, ew)


KINDS = ["purely_predicted", "purely_derived", "gt_normals", "residue_prediction"]
DERIVING = ("purely_derived", "residue_prediction")          # kinds whose forward differentiates the density (autograd in the oracle)


@pytest.mark.parametrize("kind", KINDS)
def test_forward_kinds_vs_reference(golden, kg, kind):
    sc = kind_scene(golden, kind)
    rays, lidx = T(golden, "rays/rays"), T(golden, "rays/light_idx")
    torch.manual_seed(SEED + 3)
    with torch.set_grad_enabled(kind in DERIVING):
        out = O.forward_primary(sc, rays, lidx.int())
    for n, v in zip(NAMES, out):
        ref = kg[f"{kind}/fwd/{n}"]
        if n == "acc_mask":
            assert np.array_equal(v.numpy(), ref)
        else:
            assert float((v.detach() - torch.from_numpy(ref)).abs().max()) < 3e-5, n
    if kind == "residue_prediction":          # the second kind that fills the two normal losses (:962-968)
        assert float(np.abs(kg[f"{kind}/fwd/normals_diff_map"]).max()) > 0.0
    else:
        assert float(np.abs(kg[f"{kind}/fwd/normals_orientation_loss_map"]).max()) == 0.0
        assert float(np.abs(kg[f"{kind}/fwd/normals_diff_map"]).max()) == 0.0


@pytest.mark.parametrize("kind", KINDS)
def test_eval_render_kinds_vs_reference(golden, kg, kind):
    """The boundary call per normals_kind; with 'gt_normals' the ground-truth normals replace the (zero) normal map before the
    shading stage and in the returned dict (renderer.py:82-83; models/tensorBase_rotated_lights.py:951-952)."""
    sc = kind_scene(golden, kind)
    rays, lidx = T(golden, "rays/rays"), T(golden, "rays/light_idx")
    ngt = T(kg, "normal_gt") if kind == "gt_normals" else None
    torch.manual_seed(SEED + 3)
    with torch.set_grad_enabled(kind in DERIVING):
        ret = O.renderer_train(sc, rays, lidx, n_samples=-1, second_n_sample=24, second_near=0.05, second_far=1.5, normal_gt=ngt)
    for k in ("rgb_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "acc_map", "rgb_with_brdf_map"):
        assert float((ret[k].detach() - T(kg, f"{kind}/eval_render/{k}")).abs().max()) < 3e-5, (kind, k)
    if kind == "gt_normals":
        assert float((T(kg, f"{kind}/eval_render/normal_map") - T(kg, "normal_gt")).abs().max()) == 0.0


@pytest.mark.parametrize("kind", KINDS)
def test_train_grads_kinds_vs_reference(golden, kg, kind):
    sc = kind_scene(golden, kind)
    rays, lidx = T(golden, "rays/rays"), T(golden, "rays/light_idx")
    S = int(kg["n_samples"][0])
    torch.manual_seed(SEED + 12)
    jit = torch.rand(rays.shape[0], 1)
    assert np.array_equal(jit.numpy(), kg[f"{kind}/train/ray_jitter"])
    loss, grads, ret = O.train_step_grads(sc, rays, lidx, T(kg, "rgb_gt"), is_relight=True, n_samples=S,
                                          ray_jitter=jit, second_n_sample=24, second_near=0.05, second_far=1.5,
                                          normal_gt=(T(kg, "normal_gt") if kind == "gt_normals" else None))
    assert abs(float(loss) - float(kg[f"{kind}/train/loss"][0])) < 2e-6
    checked = 0
    for name, gr in grads.items():
        ref = torch.from_numpy(kg[f"{kind}/train/grad/{name}"]).double()
        if float(ref.abs().max()) == 0:
            assert float(gr.abs().max()) == 0, name
            continue
        err = float((gr.double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-3, (name, err)
        checked += 1
    assert checked >= (19 if kind == "gt_normals" else 25)

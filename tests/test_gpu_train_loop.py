"""The call sequence of ``reconstruction()`` (train_tensoIR.py:110-461) driven through the PRODUCT API only -- the
reference checkout does not exist on the GPU box, so this is the on-hardware evidence for the script-level drop-in
(tests/test_launcher.py runs the unmodified script itself wherever a checkout is present):

  analytic dataset -> TensorVMSplit(aabb, reso, device, **script kwargs) (:170-192) -> get_optparam_groups + Adam (:196,
  :206) -> filtering_rays(bbox_only=True) on the host ray tensor (:228) -> iterations (:237-317): host batch indexing,
  Renderer_TensoIR_train(is_train=True, is_relight=flag, sample_method='stratified_sampling'), image loss + L1 / TV
  regularisers + relighting losses, backward, Adam step, lr decay -> updateAlphaMask + shrink + relighting on (:385-398)
  -> second mask update + ray re-filtering (:401-406) -> upsample_volume_grid + fresh optimizer (:409-422) -> save /
  reload the checkpoint (:424, :163-168)."""
import os

import numpy as np
import pytest
import torch

from tests.train_sequence import reconstruct

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["single_light", "rotated_multi_lights", "general_multi_lights"])
def test_reconstruction_call_sequence(tmp_path, variant):
    """variant = which of the reference's three training scripts is mirrored: train_tensoIR.py (one light),
    train_tensoIR_rotated_multi_lights.py (one SG set seen under three rotations) or
    train_tensoIR_general_multi_lights.py (one SG set per light, models/tensoRF_general_multi_lights.py).
    The loop itself: tests/train_sequence.py."""
    from tensoir_amd import Renderer_TensoIR_train
    if variant == "general_multi_lights":
        from tensoir_amd.general_multi_lights import TensorVMSplit
    else:
        from tensoir_amd import TensorVMSplit
    r = reconstruct(variant, n_iters=150, batch=1024, upsamp=(100, 130), mask_updates=(60, 110))
    m, dev, args, rays_f, lidx_f, losses, grids = r.model, r.device, r.args, r.rays_f, r.lidx_f, r.losses, r.grids
    assert len(grids) == 3 and grids[-1][0] > grids[0][0]                 # two up-samplings happened
    assert m.alphaMask is not None
    psnr = -10 * np.log10(np.mean(losses[-10:]))
    assert psnr > 20.0, (psnr, losses[:4], losses[-10:])           # it learns the sphere (reference CPU path: 26.8 dB)
    # checkpoint round trip (train_tensoIR.py:424 / :163-168): same render from the reloaded model
    path = os.path.join(str(tmp_path), "synth.th")
    m.save(path)
    ck = torch.load(path, map_location=dev, weights_only=False)
    kw = ck["kwargs"]
    kw.update({"device": dev})
    m2 = TensorVMSplit(**kw)
    m2.load(ck)
    probe, pl = rays_f[:512], lidx_f[:512]
    with torch.no_grad():
        a = Renderer_TensoIR_train(probe, None, pl, m, N_samples=-1, device=dev, args=args)
        b = Renderer_TensoIR_train(probe, None, pl, m2, N_samples=-1, device=dev, args=args)
    # (the per-light SG sets of the general variant live outside the state_dict, as in the reference: a reloaded model
    # has fresh ones, so its physically-based re-render differs by design)
    keys = ("rgb_map", "depth_map", "acc_map") + (() if variant == "general_multi_lights" else ("rgb_with_brdf_map",))
    for k in keys:
        assert torch.equal(a[k], b[k]), k

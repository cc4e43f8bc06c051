"""Host-side mirror of the reference interface: construction, checkpoint compatibility, light tables,
grid maintenance.  CPU only (no kernel launches)."""
import io
import types

import numpy as np
import pytest
import torch

import tensoir_amd
from tensoir_amd import synth
from tensoir_amd.field_model import TensorVMSplit
from tests.helpers import T, golden_checkpoint

SEED = 20211202


@pytest.fixture(scope="module")
def model(golden):
    eh, ew = [int(x) for x in golden["scene/envmap_hw"]]
    return tensoir_amd.model_from_checkpoint(golden_checkpoint(golden), "cpu", envmap_h=eh, envmap_w=ew)


def test_state_dict_is_reference_compatible(golden, model):
    ref_keys = sorted(k[3:] for k in golden.files if k.startswith("sd/"))
    sd = model.state_dict()
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == golden["sd/" + k].shape, k
        assert torch.equal(sd[k], T(golden, "sd/" + k))


def test_step_geometry_matches_reference(golden, model):
    assert model.nSamples == int(golden["scene/nSamples"][0])
    assert float(model.stepSize) == float(golden["scene/stepSize"][0])
    assert model.alphaMask.alpha_volume.shape[-3:] == golden["scene/alpha_volume"].shape


def test_kwargs_roundtrip(tmp_path, model):
    path = tmp_path / "ckpt.th"
    model.save(str(path))
    ckpt = torch.load(str(path), weights_only=False)
    assert set(ckpt) == {"kwargs", "state_dict", "alphaMask.shape", "alphaMask.mask", "alphaMask.aabb"}
    m2 = tensoir_amd.model_from_checkpoint(ckpt, "cpu", envmap_h=model.envmap_h, envmap_w=model.envmap_w)
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert torch.equal(m2.alphaMask.alpha_volume, model.alphaMask.alpha_volume)
    kw = model.get_kwargs()
    assert kw["light_rotation"] == [0, 120, 240] and kw["light_num"] == 3 and kw["gridSize"] == [20, 24, 28]


def test_light_tables_match_reference(golden, model):
    area, dirs = model.generate_envir_map_dir(model.envmap_h, model.envmap_w)
    assert torch.allclose(area, T(golden, "env/area"), atol=1e-7)
    assert torch.allclose(dirs, T(golden, "env/dirs"), atol=1e-7)
    assert torch.equal(model.gen_light_incident_dirs(method="fixed_envirmap"), model.fixed_viewdirs)
    torch.manual_seed(SEED + 4)
    strat = model.gen_light_incident_dirs(method="stratified_sampling")
    assert torch.allclose(strat, T(golden, "env/strat_dirs"), atol=1e-6)
    assert model.light_rotation_matrix.shape == (3, 3, 3)
    with pytest.raises(ValueError):
        model.gen_light_incident_dirs(method="nope")


def test_importance_sample_direction_table_matches_reference(model):
    """The jittered 128 x 256 table `gen_light_incident_dirs(method='importance_sample')` draws from
    (models/tensorBase_rotated_lights.py:548: generate_envir_map_dir(128, 256, is_jittor=True)): same CPU generator draws as the
    reference -> the reference's table (tests/golden/importance_sample.npz).  The radiance / pdf / sampling side runs on the
    device: tests/test_gpu_parity.py::test_importance_sampled_light_directions."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "importance_sample.npz"))
    torch.manual_seed(int(g["seed"][0]))
    _, dirs = model.generate_envir_map_dir(128, 256, is_jittor=True)
    assert torch.allclose(dirs.reshape(-1, 3), torch.from_numpy(g["view_dirs"]), atol=1e-6)
    assert abs(float(g["pdf_to_sample"].sum()) - 1.0) < 1e-5 and g["draw_idx"].shape == (4096,)


def test_sample_ray_matches_reference(golden, model):
    rays = T(golden, "rays/rays")
    pts, z, valid = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False)
    assert torch.equal(pts, T(golden, "march/pts")) and torch.equal(z, T(golden, "march/z"))
    assert torch.equal(valid, T(golden, "march/valid"))
    torch.manual_seed(SEED + 2)
    _, zt, vt = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=True, N_samples=40)
    assert torch.equal(zt, T(golden, "march/train_z")) and torch.equal(vt, T(golden, "march/train_valid"))


def test_upsample_and_optimizer_groups():
    ck = synth.make_checkpoint(grid=(12, 14, 16), seed=3)
    m = tensoir_amd.model_from_checkpoint(ck, "cpu", envmap_h=4, envmap_w=8)
    groups = m.get_optparam_groups(0.02, 0.001)
    n_group_params = sum(len(list(g["params"])) if not isinstance(g["params"], torch.nn.Parameter) else 1 for g in groups)
    assert n_group_params == len(list(m.parameters()))
    m.upsample_volume_grid([20, 22, 24])
    assert m.gridSize.tolist() == [20, 22, 24]
    assert tuple(m.density_plane[0].shape) == (1, 16, 22, 20) and tuple(m.app_line[0].shape) == (1, 48, 24, 1)
    assert m._field_key is None                       # packed shadow invalidated
    # VM parameters live in the kernels' channel-last layout (the parameter is its own packed form) ...
    from tensoir_amd.field_model import is_channel_last
    vm = list(m.density_plane) + list(m.density_line) + list(m.app_plane) + list(m.app_line)
    assert all(is_channel_last(p) for p in vm)
    # ... and stay there through load_state_dict from ordinary (NCHW-contiguous) reference tensors and Adam state
    sd = {k: v.contiguous() for k, v in m.state_dict().items()}
    assert not is_channel_last(sd["app_plane.0"])
    vals = sd["app_plane.0"].clone()
    m.load_state_dict(sd)
    assert all(is_channel_last(p) for p in vm) and torch.equal(m.app_plane[0].detach(), vals)
    assert is_channel_last(torch.zeros_like(m.app_plane[0], memory_format=torch.preserve_format))
    assert m.app_plane[0].detach().permute(0, 2, 3, 1).is_contiguous()
    reg = lambda x: (x[..., 1:, :] - x[..., :-1, :]).pow(2).mean()
    for v in (m.vector_comp_diffs(), m.density_L1(), m.TV_loss_density(reg), m.TV_loss_app(reg)):
        assert torch.isfinite(v)


def test_unsupported_configurations_fail_loudly():
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3])
    with pytest.raises(NotImplementedError):
        TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="SH", light_kind="sg", density_n_comp=[16] * 3,
                      appearance_n_comp=[48] * 3)
    with pytest.raises(NotImplementedError):
        TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="MLP_Fea", light_kind="no_such_light", density_n_comp=[16] * 3,
                      appearance_n_comp=[48] * 3)
    # light_kind='gt' (:592-593) constructs without a light parameter
    gtm = TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="MLP_Fea", light_kind="gt", density_n_comp=[16] * 3,
                        appearance_n_comp=[48] * 3)
    assert gtm.light_parameters() == [] and not hasattr(gtm, "lgtSGs") and len(gtm.get_optparam_groups()) == 9
    with pytest.raises(NotImplementedError):
        TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="MLP_Fea", light_kind="sg", normals_kind="no_such_kind",
                      density_n_comp=[16] * 3, appearance_n_comp=[48] * 3)
    # 'residue_prediction' (:426-428): a 153-column layer 1 in the reference's own column order
    r = TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="MLP_Fea", light_kind="sg", normals_kind="residue_prediction",
                      density_n_comp=[16] * 3, appearance_n_comp=[48] * 3)
    dec = r.renderModule_normal
    assert dec.mlp[0].weight.shape == (128, 153) and sorted(dec.std_cols) == [0, 1, 2] + list(range(6, 153))
    assert torch.equal(dec.w0_std()[:, 27:30], dec.mlp[0].weight[:, 0:3]) and torch.equal(dec.w0_std()[:, :27], dec.mlp[0].weight[:, 6:33])
    # the other kinds of the reference construct: the learnable pixel environment map (:459-460), ground-truth normals (:951-952)
    m = TensorVMSplit(aabb, [8, 8, 8], "cpu", shadingMode="MLP_Fea", light_kind="pixel", normals_kind="gt_normals",
                      envmap_h=4, envmap_w=8, density_n_comp=[16] * 3, appearance_n_comp=[48] * 3)
    assert m._light_rgbs.shape == (32, 3) and not hasattr(m, "lgtSGs") and not hasattr(m, "renderModule_normal")
    assert float(m._light_rgbs.detach().min()) >= 0.0 and float(m._light_rgbs.detach().max()) <= 3.0
    assert m.light_parameters()[0] is m._light_rgbs and "_light_rgbs" in m.state_dict()


def test_no_silent_fallbacks(model, golden):
    rays, lidx = T(golden, "rays/rays"), T(golden, "rays/light_idx")
    with pytest.raises(NotImplementedError):           # stand-alone per-point calls have no autograd bridge
        model.compute_densityfeature(rays[:, :3])
    if not torch.cuda.is_available():
        from tensoir_amd._lib import TensoirHipError
        with torch.no_grad(), pytest.raises(TensoirHipError):
            model(rays, lidx)                          # CPU tensors never run (inference path) ...
        with pytest.raises(TensoirHipError):
            model(rays, lidx)                          # ... nor on the training path: no eager fallback


def test_synth_rays():
    r = synth.make_rays(4, 6)
    assert r.shape == (24, 6) and torch.allclose(r[:, 3:].norm(dim=-1), torch.ones(24), atol=1e-6)
    assert torch.all(r[:, 2] == 4.0)


def test_launcher_rebinds_reference_symbols():
    """tensoir_amd.run.install: the reference's own modules end up pointing at our hot path."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference checkout not present (GPU box)")
    ref_loader.load()                                    # stubs for cv2/loguru/... + sys.path
    from tensoir_amd import run
    done = run.install(ref_loader.REF_ROOT)
    import importlib
    r = importlib.import_module("renderer")
    ru = importlib.import_module("models.relight_utils")
    tf = importlib.import_module("models.tensoRF_rotated_lights")
    assert r.Renderer_TensoIR_train is tensoir_amd.Renderer_TensoIR_train
    assert ru.render_with_BRDF is tensoir_amd.render_with_BRDF and ru.compute_radiance is tensoir_amd.compute_radiance
    assert tf.TensorVMSplit is tensoir_amd.TensorVMSplit and tf.AlphaGridMask is tensoir_amd.AlphaGridMask
    from tensoir_amd import optim
    try:
        assert set(done) == set(run.PATCHES) | {"torch.optim"} and torch.optim.Adam is optim.LauncherAdam
    finally:
        torch.optim.Adam = optim._TorchAdam              # this process goes on to run other tests on the CPU


def test_general_multi_light_variant_host_side():
    """models/tensoRF_general_multi_lights.py mirror: per-light SG sets in a plain list (optimised, not in the
    state_dict -- as in the reference), light_name_list round-trips through get_kwargs."""
    from tensoir_amd.general_multi_lights import TensorVMSplit as General
    m = General(torch.tensor([[-1.5] * 3, [1.5] * 3]), [16, 16, 16], "cpu", density_n_comp=[16] * 3,
                appearance_n_comp=[48] * 3, shadingMode="MLP_Fea", normals_kind="derived_plus_predicted",
                light_kind="sg", step_ratio=0.5, light_name_list=["sunset", "snow"])
    assert m.light_num == 2 and len(m.lgtSGs_list) == 2 and m.light_line.weight.shape[0] == 2
    assert "lgtSGs" not in m.state_dict()
    groups = m.get_optparam_groups()
    flat = [p for g in groups for p in (g["params"] if isinstance(g["params"], (list, tuple)) else [g["params"]])]
    assert sum(1 for p in flat if any(p is sg for sg in m.lgtSGs_list)) == 2
    kw = m.get_kwargs()
    assert kw["light_name_list"] == ["sunset", "snow"] and "light_rotation" not in kw


def test_host_value_cache_follows_tensor_changes():
    """Geometry constants are mirrored on the host once per change (no `.tolist()` per call: it would drain the
    launch queue); replacing the tensor, or writing into it, refreshes the mirror."""
    from tensoir_amd.field_model import _host_values

    class Owner:
        pass
    o, calls = Owner(), []
    a = torch.tensor([1.0, 2.0])

    def build():
        calls.append(1)
        return a.tolist()
    assert _host_values(o, "k", (a,), build) == [1.0, 2.0] and len(calls) == 1
    assert _host_values(o, "k", (a,), build) == [1.0, 2.0] and len(calls) == 1          # cached
    a.mul_(2.0)                                                                           # in-place: version bump
    assert _host_values(o, "k", (a,), build) == [2.0, 4.0] and len(calls) == 2
    a = torch.tensor([5.0, 6.0])                                                          # new tensor object
    assert _host_values(o, "k", (a,), build) == [5.0, 6.0] and len(calls) == 3


def test_to_device_is_a_plain_to_on_cpu():
    from tensoir_amd import ops
    t = torch.arange(6).reshape(2, 3)
    out = ops.to_device(t, "cpu", torch.int32)
    assert out.dtype == torch.int32 and torch.equal(out.long(), t)


def test_adam_is_constructor_and_state_compatible_and_loud_on_cpu():
    """tensoir_amd.optim.Adam: same constructor / param_groups / state_dict layout as torch.optim.Adam
    (train_tensoIR.py:197 builds it over per-tensor groups and rescales group['lr'], :321-322); no CPU path."""
    import pytest
    from tensoir_amd import optim
    from tensoir_amd._lib import TensoirHipError
    ps = [torch.nn.Parameter(torch.randn(1, 4, 5, 3)), torch.nn.Parameter(torch.randn(7))]
    groups = [{"params": ps[0], "lr": 0.02}, {"params": ps[1], "lr": 0.001}]
    o = optim.Adam(groups, betas=(0.9, 0.99))
    assert isinstance(o, torch.optim.Adam) and [g["lr"] for g in o.param_groups] == [0.02, 0.001]
    ref = torch.optim.Adam([{"params": [p.detach().clone().requires_grad_()], "lr": g["lr"]} for p, g in zip(ps, groups)], betas=(0.9, 0.99))
    assert {k for k in o.state_dict()["param_groups"][0]} == {k for k in ref.state_dict()["param_groups"][0]}
    for p in ps:
        p.grad = torch.ones_like(p)
    with pytest.raises(TensoirHipError):
        o.step()
    # a refused step mutates nothing (validation comes first): no state entry, no advanced step count -- also when the
    # offending parameter is the LAST one of the list
    assert all(len(o.state[p]) == 0 for p in ps)
    o2 = optim.Adam([{"params": ps[0], "lr": 0.02}, {"params": ps[1], "lr": 0.001, "weight_decay": 0.1}], betas=(0.9, 0.99))
    with pytest.raises(NotImplementedError):
        o2.step()
    assert all(len(o2.state[p]) == 0 for p in ps)
    # the class the launcher binds leaves foreign optimizers alone: CPU parameters / weight decay take torch's own step
    q = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(5))]
    q[1].data.copy_(q[0].data)
    for t in q:
        t.grad = torch.ones_like(t)
    la = optim.LauncherAdam([q[0]], lr=0.01, weight_decay=0.1)
    ta = optim._TorchAdam([q[1]], lr=0.01, weight_decay=0.1)
    la.step(); ta.step()
    assert isinstance(la, torch.optim.Adam) and torch.equal(q[0].data, q[1].data)
    assert optim._dense_key(torch.empty(1, 4, 5, 3).permute(0, 2, 3, 1)) is not None
    assert optim._dense_key(torch.empty(8, 8)[:, ::2]) is None


def test_graph_lane_state_swap_is_exception_safe():
    """GraphedRenderer._own_state: while open, the model's per-pass device words are the renderer's own; on exit (also
    through an exception) the model's come back and whatever the pass created stays with the renderer."""
    import types
    from tensoir_amd import graph
    model = types.SimpleNamespace()
    model.__dict__.update({"_words": "model-words", "_jit_rng": "model-rng", "other": 1})
    r = types.SimpleNamespace(model=model, _own={})
    with graph._OwnState(r):
        assert "_words" not in model.__dict__ and "_jit_rng" not in model.__dict__ and model.other == 1
        model.__dict__["_words"] = "lane-words"               # what model._step_words() would create on first use
        model.__dict__["_pair_counter"] = "lane-counter"
    assert model._words == "model-words" and model._jit_rng == "model-rng" and "_pair_counter" not in model.__dict__
    assert r._own == {"_words": "lane-words", "_pair_counter": "lane-counter"}
    try:
        with graph._OwnState(r):
            assert model._words == "lane-words" and model._pair_counter == "lane-counter" and "_jit_rng" not in model.__dict__
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert model._words == "model-words" and model._jit_rng == "model-rng" and "_pair_counter" not in model.__dict__
    assert r._own == {"_words": "lane-words", "_pair_counter": "lane-counter"}


# ---- host-side grid maintenance and regularisers against the IMPORTED reference (build container only) ---------------------
def _reference_twin(golden):
    """(reference model, our model) on the CPU with the golden scene's parameters and occupancy mask."""
    import contextlib
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference checkout not present (GPU box)")
    ref = ref_loader.load()
    ckpt = golden_checkpoint(golden)
    eh, ew = [int(x) for x in golden["scene/envmap_hw"]]
    kw = dict(ckpt["kwargs"])
    kw.pop("light_num", None)
    aabb, grid = kw.pop("aabb"), kw.pop("gridSize")
    kw["light_rotation"] = [f"{int(r):03d}" for r in kw["light_rotation"]]
    with contextlib.redirect_stdout(io.StringIO()):
        rm = ref.TensorVMSplit(aabb.clone(), list(grid), "cpu", envmap_h=eh, envmap_w=ew, **kw)
        rm.load(ckpt)
    ours = tensoir_amd.model_from_checkpoint(ckpt, "cpu", envmap_h=eh, envmap_w=ew)
    return rm, ours


def _same_field(rm, ours):
    for name in ("density_plane", "density_line", "app_plane", "app_line"):
        for a, b in zip(getattr(rm, name), getattr(ours, name)):
            assert a.shape == b.shape and torch.equal(a.detach(), b.detach().contiguous()), name
    assert torch.equal(rm.aabb, ours.aabb) and torch.equal(rm.gridSize, ours.gridSize)
    assert float(rm.stepSize) == float(ours.stepSize) and rm.nSamples == ours.nSamples
    assert torch.equal(rm.units, ours.units) and torch.equal(rm.invaabbSize, ours.invaabbSize)


def test_upsample_and_shrink_match_the_reference_bit_for_bit(golden):
    """upsample_volume_grid (models/tensoRF_rotated_lights.py:227-252) and shrink (:254-288) are host-side PyTorch on both sides:
    same planes / lines, aabb, step geometry after each -- incl. the aabb correction when the mask grid differs from the field's."""
    import contextlib
    rm, ours = _reference_twin(golden)
    _same_field(rm, ours)
    target = [int(g) + 5 for g in rm.gridSize]
    with contextlib.redirect_stdout(io.StringIO()):
        rm.upsample_volume_grid(target)
        ours.upsample_volume_grid(target)
    _same_field(rm, ours)
    lo, hi = rm.aabb[0], rm.aabb[1]
    new_aabb = torch.stack([lo + 0.21 * (hi - lo), hi - 0.17 * (hi - lo)])
    assert not torch.all(rm.alphaMask.gridSize == rm.gridSize)            # the :277-285 correction branch is taken
    with contextlib.redirect_stdout(io.StringIO()):
        rm.shrink(new_aabb.clone())
        ours.shrink(new_aabb.clone())
    _same_field(rm, ours)


def test_regularisers_match_the_reference(golden):
    """density_L1, vector_comp_diffs, TV_loss_* (models/tensoRF_rotated_lights.py:59-90) act on the raw parameters in PyTorch."""
    from oracle import ref_loader
    rm, ours = _reference_twin(golden)
    sys_utils = __import__("importlib").import_module("utils")
    tv = sys_utils.TVLoss()
    for name, args in (("density_L1", ()), ("vector_comp_diffs", ()), ("TV_loss_density", (tv,)), ("TV_loss_app", (tv,))):
        a, b = getattr(rm, name)(*args).detach(), getattr(ours, name)(*args).detach()
        assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(a))), name
    ga = [g["lr"] for g in rm.get_optparam_groups(0.02, 0.001)]
    gb = [g["lr"] for g in ours.get_optparam_groups(0.02, 0.001)]
    assert ga == gb
    na = [sum(p.numel() for p in (g["params"] if not isinstance(g["params"], torch.Tensor) else [g["params"]])) for g in rm.get_optparam_groups()]
    nb = [sum(p.numel() for p in (g["params"] if not isinstance(g["params"], torch.Tensor) else [g["params"]])) for g in ours.get_optparam_groups()]
    assert na == nb


@pytest.mark.parametrize("light_kind,normals_kind", [("sg", "derived_plus_predicted"), ("pixel", "derived_plus_predicted"),
                                                     ("sg", "residue_prediction")])
def test_seeded_construction_draws_the_reference_parameters(golden, light_kind, normals_kind):
    """Same torch seed -> the same initial planes, lines, decoders, SGs / pixel light as the reference constructor (the order and
    shapes of the random draws are part of the drop-in: seeded experiments reproduce)."""
    import contextlib
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference checkout not present (GPU box)")
    ref = ref_loader.load()
    aabb = torch.tensor([[-1.5, -1.4, -1.3], [1.5, 1.4, 1.3]])
    kw = dict(density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27, shadingMode="MLP_Fea", step_ratio=0.5,
              normals_kind=normals_kind, light_rotation=["000", "120"], envmap_w=8, envmap_h=4,
              light_kind=light_kind, numLgtSGs=128)
    torch.manual_seed(77)
    with contextlib.redirect_stdout(io.StringIO()):
        rm = ref.TensorVMSplit(aabb.clone(), [12, 14, 16], "cpu", **kw)
    torch.manual_seed(77)
    ours = TensorVMSplit(aabb.clone(), [12, 14, 16], "cpu", **kw)
    a, b = rm.state_dict(), ours.state_dict()
    assert sorted(a) == sorted(b)
    for k in a:
        assert torch.equal(a[k], b[k].contiguous()), k
    assert torch.equal(rm.light_area_weight, ours.light_area_weight) and torch.equal(rm.fixed_viewdirs, ours.fixed_viewdirs)
    assert torch.equal(rm.light_rotation_matrix, ours.light_rotation_matrix)
    torch.manual_seed(5)
    da = rm.gen_light_incident_dirs(method="stratified_sampling")
    torch.manual_seed(5)
    db = ours.gen_light_incident_dirs(method="stratified_sampling")
    assert torch.equal(da, db)
    torch.manual_seed(6)
    da = rm.gen_light_incident_dirs(method="stratifed_sample_equal_areas")
    torch.manual_seed(6)
    db = ours.gen_light_incident_dirs(method="stratifed_sample_equal_areas")
    assert torch.equal(da, db)


def test_cdf_guide_tables_restrict_the_search_without_changing_it():
    """ops.cdf_guide_tables (the inverse-CDF importance sampler of models/relight_utils.py:150-188 on the device): the search
    started inside [guide[k], guide[k + 1]], k = floor(u G), returns the index the full search returns -- emulated here with
    the kernel's loop on a map with a sun, empty rows, and sizes that are not powers of two; thresholds and their
    predecessors included."""
    from tensoir_amd import ops
    gen = torch.Generator().manual_seed(0)
    H, W = 100, 300
    pdf = torch.rand(H, W, generator=gen).double() ** 4
    pdf[10, 20:30] *= 1000
    pdf[40:44] = 0
    rows = pdf.sum(1)
    row_cdf = (torch.cumsum(rows, 0) / rows.sum()).float()
    col_cdf = (torch.cumsum(pdf, 1) / rows.clamp(min=1e-300).unsqueeze(1)).float()
    row_cdf[-1] = 1.0
    col_cdf[:, -1] = 1.0
    rg, packed, gr, gc = ops.cdf_guide_tables(row_cdf, col_cdf)
    assert (gr, gc) == (128, 512) and rg.shape == (gr + 1,) and packed.shape == (H, gc // 2 + 1) and packed.dtype == torch.int32
    cg = np.ascontiguousarray(packed.numpy()).view(np.uint16).reshape(H, gc + 2)         # what the kernel reads (little endian)
    assert int(rg[-1]) == H - 1 and bool((cg[:, gc] == W - 1).all())
    assert bool((np.diff(rg.numpy()) >= 0).all()) and bool((np.diff(cg[:, :gc + 1].astype(np.int64), axis=1) >= 0).all())

    def upper(cdf, u, lo, hi):
        while lo < hi:
            mid = (lo + hi) >> 1
            if cdf[mid] > u:
                hi = mid
            else:
                lo = mid + 1
        return lo
    rng = np.random.default_rng(1)
    tr = (np.arange(gr + 1, dtype=np.float32) / np.float32(gr))
    us = np.concatenate([rng.random(3000).astype(np.float32), np.float32([0.0, 1.0 - 2.0 ** -24, 0.5]), tr[:-1],
                         np.nextafter(tr[1:], np.float32(0)).astype(np.float32)])
    rc, cc = row_cdf.numpy(), col_cdf.numpy()
    for u in us:
        row = upper(rc, u, 0, H - 1)
        k = min(int(np.float32(u) * np.float32(gr)), gr - 1)
        assert upper(rc, u, int(rg[k]), int(rg[k + 1])) == row
        k = min(int(np.float32(u) * np.float32(gc)), gc - 1)
        assert upper(cc[row], u, int(cg[row, k]), int(cg[row, k + 1])) == upper(cc[row], u, 0, W - 1)


def test_c5_pair_order_switches(monkeypatch):
    """TENSOIR_C5_PAIRS / _BINS / _BLOCK_PAIRS (ops.c5_pair_order): how the importance-sampled pairs of
    scripts/relight_importance.py:127-131 reach the visibility march; unknown values fail loudly."""
    from tensoir_amd import ops
    for k in ("TENSOIR_C5_PAIRS", "TENSOIR_C5_BINS", "TENSOIR_C5_BLOCK_PAIRS"):
        monkeypatch.delenv(k, raising=False)
    assert ops.c5_pair_order() == ("binned", (15, 17), 512)
    monkeypatch.setenv("TENSOIR_C5_PAIRS", "compact")
    assert ops.c5_pair_order() == ("compact", (1, 1), 512)
    monkeypatch.setenv("TENSOIR_C5_PAIRS", "mask")
    assert ops.c5_pair_order()[0] == "mask"
    monkeypatch.setenv("TENSOIR_C5_PAIRS", "Binned")
    monkeypatch.setenv("TENSOIR_C5_BINS", "8x8")
    monkeypatch.setenv("TENSOIR_C5_BLOCK_PAIRS", "4096")
    assert ops.c5_pair_order() == ("binned", (8, 8), 4096)
    monkeypatch.setenv("TENSOIR_C5_BINS", "8")
    with pytest.raises(ValueError):
        ops.c5_pair_order()
    monkeypatch.delenv("TENSOIR_C5_BINS")
    monkeypatch.setenv("TENSOIR_C5_PAIRS", "sorted")
    with pytest.raises(ValueError):
        ops.c5_pair_order()


def test_density_l1_multi_tensor_node_matches_the_plain_expression():
    """TensorVMSplit.density_L1 (one autograd node over multi-tensor kernels) against the reference's per-tensor expression
    (models/tensoRF_rotated_lights.py:74-78): value to a few ulp, gradients sign(x) / numel (zero at zero), parameter layout kept."""
    from tensoir_amd import field_model as fm
    gen = torch.Generator().manual_seed(4)
    shapes = [(1, 16, 30, 31), (1, 16, 29, 1), (1, 16, 28, 30), (1, 16, 31, 1), (1, 16, 31, 29), (1, 16, 30, 1)]
    ts = [torch.nn.Parameter(torch.randn(s, generator=gen).contiguous(memory_format=torch.channels_last)) for s in shapes]
    with torch.no_grad():
        ts[0][0, 0, 0, :5] = 0.0
    a = fm._DensityL1Fn.apply(*ts)
    b = 0
    for i in range(0, 6, 2):
        b = b + ts[i].abs().mean() + ts[i + 1].abs().mean()
    # (on the CPU the norm of a channel-last tensor is summed sequentially: 5e-6 relative; the device kernels reduce in trees)
    assert abs(float(a.detach()) - float(b.detach())) <= 1e-5 * float(b.detach())
    (0.37 * a).backward()
    ga = [t.grad.clone() for t in ts]
    for t in ts:
        t.grad = None
    (0.37 * b).backward()
    for x, t in zip(ga, ts):
        assert x.stride() == t.stride() and float((x - t.grad).abs().max()) <= 1e-9
    assert float(ga[0][0, 0, 0, :5].abs().max()) == 0.0


def test_grouped_output_clone_copies_each_buffer_once():
    """graph._clone_grouped (what the boundary's cached graph hands out): views of one block stay views of ONE fresh block, a
    small view of a large buffer is copied alone, non-tensors pass through, nothing aliases the graph's own buffers."""
    from tensoir_amd.graph import _clone_grouped
    maps, big = torch.randn(100, 20), torch.randn(1000000)
    out = {"rgb": maps[:, 0:3], "depth": maps[:, 3], "acc": maps[:, 14], "loss": big[5:6].squeeze(), "x": 3, "other": torch.randn(100, 3)}
    c = _clone_grouped(out)
    for k, v in out.items():
        if torch.is_tensor(v):
            assert torch.equal(c[k], v) and c[k].shape == v.shape and c[k].stride() == v.stride() and c[k].data_ptr() != v.data_ptr(), k
    assert c["x"] == 3
    assert c["rgb"].untyped_storage().data_ptr() == c["depth"].untyped_storage().data_ptr() == c["acc"].untyped_storage().data_ptr()
    assert c["loss"].untyped_storage().nbytes() == 4
    maps.zero_()
    assert float(c["rgb"].abs().sum()) > 0

"""Host logic of round 5 that needs no GPU: the verdict state machine of the indirect-light precision policy
(relight._indirect_mode / _set_verdict), the range guard's bound (ops.HalfRange.judge), the shard layout used by the image
all-gather (dist._layout / gather_records at world 1) and the arithmetic of bench.simulate_ranks."""
import math
import os
import sys
import time
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Dec:
    def __init__(self):
        self._key = ((100, 0),)

    def packed(self):
        return None


class _Model:
    """The three attributes the state machine reads: the packed-field key (first element: one (ptr, version, shape) per parameter),
    the radiance decoder's key, and a __dict__ for the state."""

    def __init__(self):
        self._field_key = (((1000, 0, (1, 48, 8, 8)), (2000, 0, (1, 48, 8, 1))), "rest")
        self.renderModule = _Dec()

    def step(self):                      # an optimizer step: same storage, next version
        self._field_key = (tuple((p, v + 1, s) for p, v, s in self._field_key[0]), "rest")

    def realloc(self):                   # upsample / shrink: new storage
        self._field_key = (tuple((p + 64, 0, s) for p, v, s in self._field_key[0]), "rest")

    def packed_field(self):
        return None


@pytest.fixture
def auto_policy(monkeypatch):
    from tensoir_amd import ops
    monkeypatch.setattr(ops, "INDIRECT_GUARD", True)
    monkeypatch.setattr(ops, "SECONDARY_MLP_IMPL", "f16")
    monkeypatch.setattr(ops, "SECONDARY_APP_IMPL", "h16")
    monkeypatch.setattr(ops, "MLP_IMPL", "bf16x3")
    return ops


def test_verdict_state_machine(auto_policy):
    from tensoir_amd import relight
    ops, m = auto_policy, _Model()
    assert relight._indirect_mode(m) == "probe"                          # nothing known yet
    relight._set_verdict(m, "f16", "probe", {"map_max_abs": 1e-6})
    assert relight._indirect_mode(m) == "f16" and relight._indirect_mode(m, training=True) == "f16"
    # inference: a verdict belongs to exactly one parameter version
    m.step()
    assert relight._indirect_mode(m) == "probe"
    # training: carried over for `interval` versions of the same storage, then re-established
    for _ in range(ops.INDIRECT_PROBE["interval"]):
        assert relight._indirect_mode(m, training=True) == "f16"
        assert relight._indirect_mode(m, training=True) == "f16"         # (asking twice for one version does not age it twice)
        m.step()
    assert relight._indirect_mode(m, training=True) == "probe"
    # ... and an inference pass never inherits a CARRIED verdict: strict verdict at v0, five training steps, inference at v5 probes
    m2 = _Model()
    relight._set_verdict(m2, "f16", "probe", {"map_max_abs": 1e-6})
    for _ in range(5):
        m2.step()
        assert relight._indirect_mode(m2, training=True) == "f16"
    assert relight._indirect_mode(m2) == "probe"
    assert relight._indirect_mode(m2, training=True) == "f16"            # (the training loop still rides on it)
    relight._set_verdict(m, "full", "probe", {"map_max_abs": 9e-5})
    assert relight._indirect_state(m)["fallbacks"] == 1
    m.step()
    assert relight._indirect_mode(m, training=True) == "full"
    # new storage (upsample, shrink, a model rebuilt from a checkpoint): at once
    m.realloc()
    assert relight._indirect_mode(m, training=True) == "probe"
    # a verdict taken with the TRAINING limit never serves an inference pass of the same version
    relight._set_verdict(m, "f16", "probe", {"map_max_abs": 6e-5}, train_limit=True)
    assert relight._indirect_mode(m, training=True) == "f16" and relight._indirect_mode(m) == "probe"
    relight._set_verdict(m, "full", "probe", {"map_max_abs": 6e-5})
    assert relight._indirect_mode(m) == "full"


def test_forced_policies_bypass_the_state_machine(monkeypatch):
    from tensoir_amd import ops, relight
    m = _Model()
    monkeypatch.setattr(ops, "MLP_IMPL", "bf16x3")
    monkeypatch.setattr(ops, "INDIRECT_GUARD", False)
    monkeypatch.setattr(ops, "SECONDARY_MLP_IMPL", "f16")
    monkeypatch.setattr(ops, "SECONDARY_APP_IMPL", "h16")
    assert relight._indirect_mode(m) == "f16"
    monkeypatch.setattr(ops, "SECONDARY_MLP_IMPL", None)
    monkeypatch.setattr(ops, "SECONDARY_APP_IMPL", None)
    assert relight._indirect_mode(m) == "full"
    monkeypatch.setattr(ops, "SECONDARY_MLP_IMPL", "f16")
    monkeypatch.setattr(ops, "SECONDARY_APP_IMPL", "h16")
    monkeypatch.setattr(ops, "MLP_IMPL", "mfma")                        # the exact decoder modes stay exact end to end
    assert relight._indirect_mode(m) == "full"


def test_half_range_bound():
    from tensoir_amd.ops import HalfRange
    lim = 6.0e4
    ok, b = HalfRange.judge([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 0.5, 0.1], lim)
    assert ok and b == 18.0                                              # a light row below 1 does not shrink the bound
    ok, b = HalfRange.judge([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0, 0.1], lim)
    assert ok and b == 144.0
    assert not HalfRange.judge([300.0, 1.0, 1.0, 300.0, 1.0, 1.0, 1.0, 0.1], lim)[0]          # plane x line alone leaves the range
    assert not HalfRange.judge([100.0, 1.0, 1.0, 100.0, 1.0, 1.0, 7.0, 0.1], lim)[0]          # ... or with the light row
    assert not HalfRange.judge([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 7.0e4], lim)[0]            # basis_mat is cast to fp16 too
    assert not HalfRange.judge([float("nan"), 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.1], lim)[0]     # a NaN anywhere fails
    assert not HalfRange.judge([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, float("nan"), 0.1], lim)[0]


@pytest.mark.parametrize("tile", [0, 5, 16])
def test_layout_reassembles_every_sharding(tile):
    from tensoir_amd import dist as tdist
    n, world = 103, 4
    cap, index = tdist._layout(n, world, tile, torch.device("cpu"))
    assert cap == tdist.shard_capacity(n, world, tile) and index.shape == (n,)
    # what the all-gather would deliver: rank r's rows, padded to cap
    rows = torch.arange(n, dtype=torch.float32).view(-1, 1).repeat(1, 3)
    gathered = torch.full((world * cap, 3), -1.0)
    for r in range(world):
        mine = tdist.shard_rows(n, r, world, tile)
        gathered[r * cap: r * cap + mine.numel()] = rows[mine]
    assert torch.equal(gathered.index_select(0, index), rows)
    one = tdist.gather_records(rows, n, 0, 1, tile)
    assert torch.equal(one, rows) and one.data_ptr() != rows.data_ptr()


def test_simulate_ranks_arithmetic(monkeypatch):
    import bench
    n, chunk = 16000, 1000
    cost = torch.ones(n)
    cost[:4000] = 0.0                                                    # the first quarter of the "image" is background: free
    # a clock that only the "render" advances (plus a fixed call overhead): the arithmetic is what is tested, and a sleep-based
    # version of this test failed once on a loaded host
    clock = {"t": 0.0}

    def render_shard(mine):
        clock["t"] += float(cost[mine].sum()) * 2e-6 + 5e-5
    monkeypatch.setattr(bench.time, "perf_counter", lambda: clock["t"])
    t1 = float(cost.sum()) * 2e-6 + 5e-5
    sim = bench.simulate_ranks(render_shard, n, chunk, t1, 96, 4, passes=2, tiles=[0, chunk])
    assert [c["world"] for c in sim["configs"]] == [2, 2, 4, 4]
    by = {(c["world"], c["tile"]): c for c in sim["configs"]}
    # row tiles: rank 0 of 4 holds only background, the others a full quarter each -> max / mean = 4 / 3
    assert by[(4, 0)]["imbalance_max_over_mean"] == pytest.approx(4 / 3, rel=0.25)
    # tiles of one chunk dealt round-robin: every rank gets exactly one of the four background tiles
    assert by[(4, chunk)]["imbalance_max_over_mean"] == pytest.approx(1.0, abs=1e-3)
    assert by[(4, chunk)]["predicted_speedup"] > by[(4, 0)]["predicted_speedup"]
    for c in sim["configs"]:
        assert len(c["per_shard_ms"]) == c["world"] and c["predicted_ms"] >= c["max_ms"]
        wire = (c["world"] - 1) / c["world"] * n * 96 / (0.6 * 153e9 * (c["world"] - 1))
        assert c["exchange_model_ms"] == pytest.approx(1e3 * wire, abs=1e-3)
    assert set(sim["best_per_world"]) == {"2", "4"}


def test_single_ray_bisect_finds_the_ray():
    """bench.single_ray_bisect / grad_deviation (the follow-up of a strict miss of the train workload's gradient check): per-ray
    gradients whose mean is the batch gradient, one ray deviating -> that ray is found, the others agree."""
    import bench
    gen = torch.Generator().manual_seed(3)
    n = 128
    per_ray = {"renderModule.mlp.0.weight": torch.randn(n, 16, 10, generator=gen), "app_plane.0": torch.randn(n, 1, 4, 9, 9, generator=gen),
               "basis_mat.weight": torch.randn(n, 5, 12, generator=gen)}
    bad = {k: v.clone() for k, v in per_ray.items()}
    bad["renderModule.mlp.0.weight"][37] *= 1.6                      # one unit's share of one ray
    bad["app_plane.0"][37] += 0.3 * torch.randn(1, 4, 9, 9, generator=gen)
    calls = []

    def dev_of(idx):
        calls.append(int(idx.numel()))
        return bench.grad_deviation({k: v[idx].mean(0) for k, v in bad.items()}, {k: v[idx].mean(0) for k, v in per_ray.items()})
    full = dev_of(torch.arange(n))
    assert full["dense"] > 2e-3 and full["l2"] > 3e-3
    ray, alone, rest = bench.single_ray_bisect(n, dev_of)
    assert ray == 37
    assert alone["dense"] > 0.3 and rest["dense"] == 0.0 and rest["l2"] == 0.0 and rest["abs"] == 0.0
    assert sum(calls[1:]) == 2 * (n - 1) + (n - 1)                   # rays evaluated: every level's two halves + all rays but one
    assert bench.single_ray_bisect(1, dev_of)[0] == 0

"""cpu_baseline bookkeeping: bench.py times the oracle ("port") on the GPU box because the reference
checkout cannot travel there; profiles/port_over_reference.json (oracle/calibrate_port.py) relates the two on the same inputs.
Here: the committed calibration is well-formed and says the port is a fair stand-in (within 2x either way, identical maps), and
-- where the checkout exists -- a small fresh run of both implementations on the bench scene agrees map for map."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_calibration():
    d = json.load(open(os.path.join(ROOT, "profiles", "port_over_reference.json")))
    for k in ("port_over_reference", "reference_rays_per_s", "port_rays_per_s", "port_vs_reference_max_rel_floor1", "sample", "threads"):
        assert k in d, k
    assert 0.5 < d["port_over_reference"] < 2.0
    assert abs(d["port_over_reference"] - d["reference_rays_per_s"] / d["port_rays_per_s"]) < 0.02
    assert d["port_vs_reference_max_rel_floor1"] < 1e-5 and "4096 rays" in d["sample"]


@pytest.mark.timeout(900)
def test_port_equals_reference_on_the_bench_scene():
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("no reference checkout on this box")
    from oracle.calibrate_port import measure
    rep = measure(n_rays=32, calls=1, grid=96)
    assert rep["port_vs_reference_max_rel_floor1"] < 1e-5
    assert rep["reference_rays_per_s"] > 0 and rep["port_rays_per_s"] > 0


def test_bench_reports_the_relation():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    v = bench.port_vs_reference(200.0)
    d = json.load(open(os.path.join(ROOT, "profiles", "port_over_reference.json")))
    assert v["port_over_reference_time"] == d["port_over_reference"]
    assert abs(v["reference_equivalent_rays_per_s"] - 200.0 * d["port_over_reference"]) < 0.01

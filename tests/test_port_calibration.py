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
    # in a process of its own: tests/test_launcher.py rebinds the reference's modules in THIS interpreter (tensoir_amd.run.install)
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "cal.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "calibrate_port.py"), "--rays", "32", "--calls", "1", "--grid", "96",
                            "--out", out], capture_output=True, text=True, timeout=850, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        rep = json.load(open(out))
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

"""The bench.py contract the driver depends on, checked on the CPU box: the CLI parses with no flags (N = 1, small K / W), refuses
a scaling line from fewer ranks than it claims, fails loudly without a GPU (no CPU fallback), and the committed evidence lines
(profiles/r03_v9_*bench.json and profiles/r04*_bench.json, written by bench.py on an MI355X) carry every field of the contract incl.
the `roofline` and `cpu_baseline` objects; the round-4 lines additionally: every roofline fraction <= 1 against an independent
peak, the side workloads inside the headline line, the full-batch CPU baseline and its relation to the reference."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench(*args, env=None):
    e = dict(os.environ, **(env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=e)


def test_defaults_and_help():
    r = _bench("--help")
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in r.stdout
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
    finally:
        sys.argv = old
    assert a.gpus == 1 and 1 <= a.steps <= 200 and 0 <= a.warmup <= 20 and a.workload == "batch"


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box check")
    r = _bench("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and not any(l.startswith('{"metric"') for l in r.stdout.splitlines())


def test_scaling_line_needs_its_ranks():
    """A launcher environment that disagrees with --gpus is refused (a 2-GPU line must not come from one rank)."""
    r = _bench("--gpus", "2", "--steps", "1", env={"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "2" in (r.stderr + r.stdout)
    assert "starting 2 ranks" not in r.stderr


def test_bare_multi_gpu_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE / RANK in the environment re-executes itself under
    torch.distributed.run with one rank per GPU.  On this box (no GPU) both ranks come up, say so, and fail loudly -- the
    launch itself is what is checked here; tests/test_gpu_multirank.py and tools/two_rank_gloo.sh run it on hardware."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box check")
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr and "--master-addr 127.0.0.1" in r.stderr
    # (the launcher tears the other rank down as soon as one has failed: on a loaded box the second "up" line may not appear)
    assert "rank 0/2 up" in r.stderr or "rank 1/2 up" in r.stderr
    assert r.stderr.count("bench.py needs a GPU") >= 1 and not any(l.startswith('{"metric"') for l in r.stdout.splitlines())


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_v9_*bench.json"))))
def test_committed_bench_lines_carry_the_contract(path):
    d = json.load(open(path))
    missing = [k for k in REQUIRED if k not in d]
    assert not missing, (os.path.basename(path), missing)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert isinstance(d["config"], dict) and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf is None or all(k in rf for k in ("bound", "achieved", "peak", "unit", "frac"))
    if os.path.basename(path) == "r03_v9_bench.json":
        assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"]
        cb = d["cpu_baseline"]
        assert all(k in cb for k in ("value", "unit", "cores", "kind")) and cb["kind"] in ("reference", "port")
        assert d["parity"]["ok"] is True and d["metric"].startswith("primary+secondary rays/sec")


def _rooflines(d):
    out = [(k, v) for k, v in d.items() if k.startswith("roofline") and isinstance(v, dict) and "frac" in v]
    for wl, v in (d.get("workloads") or {}).items():
        if isinstance(v, dict) and isinstance(v.get("roofline"), dict):
            out.append((f"workloads.{wl}", v["roofline"]))
    return out


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r04*_bench.json"))))
def test_round4_bench_lines(path):
    """no roofline fraction above 1, no self-referential peak; the headline line carries the three
    side workloads with parity and roofline each, a full-batch CPU baseline and the port-vs-reference relation."""
    d = json.load(open(path))
    missing = [k for k in REQUIRED if k not in d]
    assert not missing, (os.path.basename(path), missing)
    rls = _rooflines(d)
    assert rls, path
    for name, rf in rls:
        assert all(k in rf for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")), (name, rf)
        assert rf["bound"] in ("hbm", "mfma", "l2", "valu", "l2-atomics") and 0.0 < rf["frac"] <= 1.0, (name, rf)
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3, (name, rf)            # achieved / peak, nothing else
        assert "peak = achieved / frac" not in json.dumps(rf), name                        # round 3's circular VALU peak is gone
    assert d["parity"] is None or d["parity"]["ok"] is True
    if d["metric"].startswith("primary+secondary rays/sec"):
        assert d["settle_steps"] == 300 and d["library"]["source_hash"] and "stale" in d["pmc"]
        assert d["precision_policy"]["indirect"] in ("f16", "full")
        cb = d["cpu_baseline"]
        assert cb["kind"] in ("port", "reference") and "full batch" in cb["sample"] and "vs_reference" in cb
        assert d["parity"]["max_rel"] < 1e-4 and d["parity"]["max_rel_floor1"] < 1e-4          # both metrics asserted by the run
        wl = d["workloads"]
        assert set(wl) == {"image", "relight", "train"}
        for k, v in wl.items():
            assert "error" not in v, (k, v)
            assert v["value"] > 0 and v["roofline"] and v["cpu_baseline"] and v["parity"]["ok"] in (True, False)
        vm = d["roofline_vm_sample"]
        assert vm["bound"] == "valu" and "valu" in vm and vm["valu"]["fma_floor_per_pass"] == 84.0

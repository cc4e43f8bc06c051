"""Multi-rank runs on real kernels: tests/multirank_worker.py under torch.distributed.run.

* ``test_two_ranks_rccl`` -- wherever >= 2 GPUs are visible: two ranks over RCCL (backend "nccl"), one GPU each.
  ``dist.render_sharded`` (eager and HIP-graph chunk renderers, row tiles and interleaved tiles) must equal the unsharded
  render bit for bit on every rank, and the data-parallel training step's all-reduced gradients must equal the same step
  over gloo (the collective adds two fp32 numbers either way: only the backward's atomics differ between runs) and the
  single-process full-batch step.  Skipped on one-GPU boxes.
* ``test_two_ranks_share_one_gpu_gloo`` -- everywhere a GPU exists: the same worker with both ranks on cuda:0 over gloo,
  so the sharded flows run on real kernels on the builder's / driver's one-GPU boxes too."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GTOL = 2e-3          # gradients: max |a - b| / max |b| per tensor (fp32 atomics reorder the sums between two runs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(backend, out, world=2, shared=False):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_worker.py"), "--backend", backend, "--out", str(out)]
    if shared:
        cmd.append("--shared-gpu")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return [json.load(open(os.path.join(out, f"rank{k}_{backend}.json"))) for k in range(world)]


def _check_ranks(res, backend, world=2):
    assert [r["rank"] for r in res] == list(range(world)) and all(r["world"] == world and r["backend"] == backend for r in res)
    for r in res:
        assert set(r["image"]) == {"eager/tile0", "eager/tile49", "graphed/tile0", "graphed/tile49"}
        for route, bad in r["image"].items():
            assert bad == [], (r["rank"], route, bad)              # every map of the gathered image bit-identical
        assert r["buckets"] >= 3 and r["max_abs_diff_vs_rank0"] == 0.0


def _full_batch_grads():
    """The same step in THIS process on all 48 rays, no collective: what the mean of the two shard gradients must equal."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multirank_worker as W
    import tensoir_amd
    from tests.helpers import golden_checkpoint
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_scene.npz"))
    tg = np.load(os.path.join(ROOT, "tests", "golden", "train_grads.npz"))
    eh, ew = [int(x) for x in g["scene/envmap_hw"]]
    model = tensoir_amd.model_from_checkpoint(golden_checkpoint(g), "cuda", envmap_h=eh, envmap_w=ew)
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    rays, lidx = (t.cuda() for t in W.build_inputs(g))
    B = 48
    gt = torch.from_numpy(np.array(tg["train/rgb_gt"]))[:B].cuda()
    jitter = torch.rand(B, 1, generator=torch.Generator().manual_seed(5))
    noise = torch.randn(B, 64, 3, generator=torch.Generator().manual_seed(6))
    grads, _ = W.dp_step(model, rays[:B].contiguous(), lidx[:B].contiguous(), gt, jitter, noise, args, torch.arange(B), lambda ps: 0)
    return {k: v.cpu() for k, v in grads.items()}


def _gerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-20))


@pytest.mark.timeout(900)
@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL with one rank per GPU)")
def test_two_ranks_rccl(tmp_path):
    res = _launch("nccl", tmp_path)
    _check_ranks(res, "nccl")
    assert res[0]["rccl_version"] and {r["device"] for r in res} == {"cuda:0", "cuda:1"}
    res_g = _launch("gloo", tmp_path)
    _check_ranks(res_g, "gloo")
    gn, gg = torch.load(os.path.join(tmp_path, "grads_nccl.pt")), torch.load(os.path.join(tmp_path, "grads_gloo.pt"))
    full = _full_batch_grads()
    assert set(gn) == set(gg) == set(full) and len(gn) >= 30
    for k in gn:
        assert _gerr(gn[k], gg[k]) < GTOL, k                    # RCCL all-reduce == gloo all-reduce
        assert _gerr(gn[k], full[k]) < GTOL, k                  # mean of the shard gradients == the full-batch gradient
        assert gn[k].stride() == full[k].stride(), k


@pytest.mark.timeout(900)
@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL with one rank per GPU)")
@pytest.mark.parametrize("workload", ["batch", "image"])
def test_two_ranks_rccl_bench_path(workload):
    """The first multi-GPU box exercises the driver's command too: `bench.py --gpus 2` (self-launched ranks, RCCL) prints one
    JSON line of a 2-rank job -- rccl_version set, world_size 2, one per-rank time each."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--workload", workload]
    if workload == "batch":
        cmd += ["--no-side-workloads", "--no-sharp-scene", "--no-exact-pass", "--boundary-calls", "3", "--sustained-steps", "5", "--profile-steps", "1"]
    else:
        cmd += ["--image-side", "256"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["value"] > 0
    assert line.get("rccl_version") or line.get("backend") == "nccl"
    per_rank = line.get("per_rank_ms_per_step") or line.get("per_rank_render_ms")
    assert per_rank is not None and len(per_rank) == 2


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu_gloo(tmp_path):
    assert torch.cuda.is_available()
    res = _launch("gloo", tmp_path, shared=True)
    _check_ranks(res, "gloo")
    gg = torch.load(os.path.join(tmp_path, "grads_gloo.pt"))
    full = _full_batch_grads()
    assert set(gg) == set(full) and len(gg) >= 30
    for k in gg:
        assert _gerr(gg[k], full[k]) < GTOL, k
        assert gg[k].stride() == full[k].stride(), k

"""Drop-in A/B of an UNMODIFIED reference training script on one MI355X: the same launcher, config and analytic dataset, once
with the hot path rebound to tensoir_amd (`hip`) and once with nothing rebound (`reference`: the reference's own PyTorch code on
the GPU through PyTorch-ROCm).  Needs a checkout named by TENSOIR_REFERENCE (tools/stage_reference.sh).  Iteration times come
from timestamping the script's own progress lines (tools/stamp.py); phases = the compressed armadillo schedule of
tests/data/armadillo_compressed.txt.  Usage: python tools/script_head_to_head.py --out gpurun_out/h2h.json [--modes hip reference]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(mode, ref, cfg, script, timeout, host_threads=None):
    tmp = tempfile.mkdtemp(prefix=f"h2h_{mode}_")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), TENSOIR_LAUNCH_MODE=mode,
               PYTHONUNBUFFERED="1")
    if host_threads is not None:
        env["TENSOIR_HOST_THREADS"] = str(host_threads)
    log = os.path.join(tmp, "log.txt")
    cmd = (f"{sys.executable} -u -m tensoir_amd.run {os.path.join(ref, script)} --config {cfg} --basedir {tmp} 2>&1 | "
           f"{sys.executable} {os.path.join(ROOT, 'tools', 'stamp.py')} > {log}")
    t0 = time.time()
    rc = subprocess.run(["bash", "-c", f"set -o pipefail; timeout {timeout} {cmd}"], env=env, cwd=tmp).returncode
    wall = time.time() - t0
    marks = {}
    for line in open(log, errors="replace"):
        m = re.match(r"\s*([0-9.]+) .*Iteration (\d+) PSNR: train_rgb = ([0-9.]+) train_rgb_brdf = ([0-9.]+)", line)
        if m:
            marks.setdefault(int(m.group(2)), (float(m.group(1)), float(m.group(3)), float(m.group(4))))
    tail = open(log, errors="replace").read()[-1500:]
    return {"mode": mode, "rc": rc, "wall_s": round(wall, 2), "marks": marks, "log_tail": tail if rc else "", "dir": tmp}


def render(mode, ref, cfg, script, ckpt, timeout, host_threads=None):
    """`--render_only 1 --render_test 1 --ckpt <ckpt>` (render_test(), train_tensoIR.py:62-108): the reference's evaluation loop
    over the test split; seconds per image from its own "test i / N" lines, metrics from its final prints."""
    tmp = tempfile.mkdtemp(prefix=f"h2h_render_{mode}_")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), TENSOIR_LAUNCH_MODE=mode,
               PYTHONUNBUFFERED="1")
    if host_threads is not None:
        env["TENSOIR_HOST_THREADS"] = str(host_threads)
    log = os.path.join(tmp, "log.txt")
    cmd = (f"{sys.executable} -u -m tensoir_amd.run {os.path.join(ref, script)} --config {cfg} --basedir {tmp} --ckpt {ckpt} "
           f"--render_only 1 --render_test 1 2>&1 | {sys.executable} {os.path.join(ROOT, 'tools', 'stamp.py')} > {log}")
    rc = subprocess.run(["bash", "-c", f"set -o pipefail; timeout {timeout} {cmd}"], env=env, cwd=tmp).returncode
    text = open(log, errors="replace").read()
    marks = [(float(m.group(1)), int(m.group(2))) for m in re.finditer(r"^\s*([0-9.]+) test (\d+) / \d+", text, re.M)]
    out = {"mode": mode, "rc": rc, "images": len(marks)}
    if len(marks) >= 2:
        out["s_per_image"] = round((marks[-1][0] - marks[0][0]) / (marks[-1][1] - marks[0][1]), 4)
    for key in ("PSNRs_test", "PSNRs_rgb_brdf_test", "MAE_test", "PSNR_albedo_three"):
        m = re.search(rf"{key}: (.*)", text)
        if m:
            nums = [float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", m.group(1))]
            out[key] = round(sum(nums) / len(nums), 4) if nums else None
    if rc:
        out["log_tail"] = text[-1500:]
    return out


def phase(marks, a, b):
    if a in marks and b in marks and b > a:
        return round((marks[b][0] - marks[a][0]) / (b - a) * 1e3, 2)
    return None


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--out", default="gpurun_out/h2h.json")
    p.add_argument("--config", default=os.path.join(ROOT, "tests", "data", "armadillo_compressed.txt"))
    p.add_argument("--script", default="train_tensoIR.py")
    p.add_argument("--modes", nargs="+", default=["hip", "reference"])
    p.add_argument("--timeout", type=int, default=900)
    p.add_argument("--render", type=int, default=1, help="also run --render_only on the hip run's checkpoint in every mode")
    p.add_argument("--host-threads", default=None, help="TENSOIR_HOST_THREADS for both modes (0 = PyTorch's default team)")
    p.add_argument("--set", nargs="*", default=[], metavar="KEY=VALUE", help="config edits (VALUE 'none' drops the key), e.g. "
                   "light_rotation='[000, 120, 240]' for train_tensoIR_rotated_multi_lights.py")
    a = p.parse_args()
    if a.set:
        edits = dict(kv.split("=", 1) for kv in a.set)
        lines = [l for l in open(a.config).read().splitlines() if l.split("=")[0].strip() not in edits]
        lines += [f"{k} = {v}" for k, v in edits.items() if v.lower() != "none"]
        fd, path = tempfile.mkstemp(prefix="h2h_cfg_", suffix=".txt")
        with os.fdopen(fd, "w") as fh:
            fh.write("\n".join(lines) + "\n")
        a.base_config, a.config = a.config, path
    ref = os.environ.get("TENSOIR_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, a.script)):
        raise SystemExit("no reference checkout (TENSOIR_REFERENCE)")
    res = {"config": os.path.relpath(getattr(a, "base_config", a.config), ROOT), "config_edits": a.set, "script": a.script,
           "phases": {"radiance_only_128^3": [10, 40], "relight_210^3..260^3": [110, 190], "relight_300^3": [210, 390]},
           "host_threads": a.host_threads if a.host_threads is not None else "launcher default (8)", "host_cpus": os.cpu_count(),
           "note": "ms per training iteration of the UNMODIFIED script (data sampling, forward, losses, backward, Adam, "
                   "regularisers), from timestamps of its own progress lines; batch 4096 rays; analytic dataset"}
    for mode in a.modes:
        r = run(mode, ref, a.config, a.script, a.timeout, a.host_threads)
        marks = r.pop("marks")
        r["ms_per_iteration"] = {k: phase(marks, *v) for k, v in res["phases"].items()}
        last = max(marks) if marks else None
        r["last_progress"] = {"iteration": last, "train_rgb_psnr": marks[last][1], "train_rgb_brdf_psnr": marks[last][2]} if marks else None
        res[mode] = r
        print(json.dumps({mode: r})[:1200], flush=True)
    if a.render and "hip" in res:
        # the checkpoint the HIP run wrote, rendered by BOTH implementations: same file format, comparable metrics
        expname = re.search(r"^expname\s*=\s*(\S+)", open(a.config).read(), re.M).group(1)
        ckpt = os.path.join(res["hip"]["dir"], expname, f"{expname}.th")
        res["render_test"] = {"checkpoint": "written by the hip training run above", "note": "seconds per image of the reference's "
                              "evaluation loop (renderer.py:135-519: chunked renderer calls, 10 D2H copies per chunk, SSIM, PNG "
                              "dumps), test split of the analytic dataset"}
        for mode in a.modes:
            res["render_test"][mode] = render(mode, ref, a.config, a.script, ckpt, a.timeout, a.host_threads)
            print(json.dumps(res["render_test"][mode])[:800], flush=True)
    for mode in a.modes:
        res.get(mode, {}).pop("dir", None)
    if "hip" in res and "reference" in res:
        res["speedup"] = {k: (round(res["reference"]["ms_per_iteration"][k] / res["hip"]["ms_per_iteration"][k], 2)
                              if res["reference"]["ms_per_iteration"][k] and res["hip"]["ms_per_iteration"][k] else None)
                          for k in res["phases"]}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res.get("speedup")))


if __name__ == "__main__":
    main()

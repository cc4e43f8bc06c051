"""Board power and shader clock while the step graph replays with one and with two batches in flight (rocm-smi polled
during the replays).  GPU box: python tools/lanes_power.py"""
import os, subprocess, sys, threading, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd.graph import GraphedRenderer

a = types.SimpleNamespace(rays=4096, samples=512, grid=300, env_h=8, env_w=16, second_samples=96)
dev = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, dev, 0)
model.march_t_stop = 1e-6
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
grs = []
for i in range(2):
    g = GraphedRenderer(model, rays.shape[0], N_samples=a.samples, args=args, device=dev)
    g(rays, lidx)
    grs.append(g)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
samples, stop = [], False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(o.strip().splitlines()[-1] if o.strip() else "")
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.05)
def run(k, n):
    cur = torch.cuda.current_stream()
    for s in streams[:k]:
        s.wait_stream(cur)
    for i in range(n):
        with torch.cuda.stream(streams[i % k]):
            grs[i % k].graph.replay()
    for s in streams[:k]:
        cur.wait_stream(s)
for k in (1, 2, 1, 2):
    run(k, 300); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    t0 = time.perf_counter()
    n = 2500
    run(k, n); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print(f"{k} in flight: {1e3 * dt / n:.4f} ms per step")
    for s in samples[len(samples) // 2: len(samples) // 2 + 3]:
        print("   ", s)

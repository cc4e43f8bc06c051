"""Do two independent batches in flight (two captured graphs of the same model replayed on two HIP streams) raise the
whole-job rate?  Every kernel of a step has a tail in which CUs idle; a second, independent step can fill it.
GPU box: python tools/two_stream_probe.py"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tensoir_amd.graph import GraphedRenderer

a = types.SimpleNamespace(rays=4096, samples=512, grid=300, env_h=8, env_w=16, second_samples=96)
dev = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, dev, 0)
model.march_t_stop = 1e-6
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
grs, keep = [], []
NG = int(os.environ.get("NG", 4))
for i in range(NG):
    g = GraphedRenderer(model, rays.shape[0], N_samples=a.samples, args=args, device=dev)
    out = g(rays, lidx)
    grs.append(g)
    keep.append((model.__dict__.pop("_words", None), model.__dict__.pop("_pair_counter", None), model.__dict__.pop("_jit_rng", None)))
torch.cuda.synchronize()
o0, o1 = grs[0](rays, lidx), grs[1](rays, lidx)
same = all(torch.equal(o0[k], o1[k]) for k in o0 if torch.is_tensor(o0[k]) and "smoothness" not in k)
print("outputs of the two graphs identical:", same)
def run_one(n):
    for _ in range(n):
        grs[0].graph.replay()
def run_k(k):
    def f(n):
        cur = torch.cuda.current_stream()
        s = [torch.cuda.Stream() for _ in range(k)]
        for x in s:
            x.wait_stream(cur)
        for i in range(n):
            with torch.cuda.stream(s[i % k]):
                grs[i % k].graph.replay()
        for x in s:
            cur.wait_stream(x)
    return f
for label, fn in (("one stream", run_one), ("two streams", run_k(2)), ("three streams", run_k(3)), ("four streams", run_k(4)), ("one stream", run_one), ("two streams", run_k(2))):
    fn(300); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(400); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 400
    print(f"{label}: {ms:.4f} ms per step, {rays.shape[0] / ms * 1e3 / 1e6:.3f} M rays/s")
o0b = {k: v.clone() for k, v in grs[0].out.items() if torch.is_tensor(v)}
ok = all(torch.equal(o0[k], o0b[k]) for k in o0b if "smoothness" not in k)
print("outputs after concurrent replays identical:", ok, "| overflow flags:", int(grs[0]._host[8]), int(grs[1]._host[8]))

"""Is the decoder kernel clock/power limited?  Times it on random and on all-zero data (same instruction stream) while
polling rocm-smi for sclk / power (GPU box)."""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensoir_amd
from tensoir_amd import ops, synth
n = int(os.environ.get("N", 2_000_000))
ck = synth.make_checkpoint(grid=(32, 32, 32), seed=5)
m = tensoir_amd.model_from_checkpoint(ck, "cuda", envmap_h=4, envmap_w=8)
g = torch.Generator().manual_seed(0)
feat = torch.zeros(n, 32); feat[:, :27] = torch.randn(n, 27, generator=g) * 1.5
aux = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
samples = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o.strip().splitlines()[-1] if o.strip() else ""))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.05)
for label, f, a, zero_w in (("random", feat, aux, False), ("zeros", torch.zeros_like(feat), torch.zeros_like(aux), True)):
    mod = m.renderModule_brdf
    if zero_w:
        with torch.no_grad():
            for p in mod.parameters():
                p.zero_()
    pk = mod.packed()
    fc, ac = f.cuda(), a.cuda()
    with torch.no_grad():
        for _ in range(5):
            ops.mlp(pk, fc, ac, None, "bf16x3")
        torch.cuda.synchronize()
        samples.clear(); stop = False
        th = threading.Thread(target=poll); th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 3000
        e0.record()
        for _ in range(iters):
            ops.mlp(pk, fc, ac, None, "bf16x3")
        e1.record(); torch.cuda.synchronize()
        stop = True; th.join()
    ms = e0.elapsed_time(e1) / iters
    print(f"{label}: {ms:.4f} ms per launch ({n} rows)")
    for t, s in samples[len(samples)//2: len(samples)//2 + 3]:
        print("   ", s)

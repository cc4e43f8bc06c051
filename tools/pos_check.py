"""Are the sample positions of the HIP primary march bit-identical to the oracle's sample_ray?  (GPU box, scratch check.)"""
import os, sys, types
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import bench
from oracle import tensoir_oracle as O
from tests.helpers import scene_from_model
from tensoir_amd import ops
sys.argv = ["bench.py"]; a = bench.parse()
device = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
S = a.samples
r = rays[::32].contiguous(); B = r.shape[0]
gen = torch.Generator().manual_seed(21)
torch.set_grad_enabled(False)
for name, jit in (("no jitter", None), ("jitter", torch.rand(B, 1, generator=gen))):
    f = model.packed_field()
    w, sig, acc, dep, tend, cnt = ops.march_primary_train(f, r, None if jit is None else jit.to(device), S, 0.0)
    keep = torch.ones_like(w, dtype=torch.bool)            # every sample as a "record": offsets = k S
    # use compact_primary with a weight array of ones so that all samples are emitted in (ray, k) order
    ones = torch.ones_like(w)
    offsets = (torch.arange(B + 1, device=device, dtype=torch.int32) * S).contiguous()
    thr = model.rayMarch_weight_thres
    rec_ray, rec_k, rec_w, rec_xyz = ops.compact_primary(f, r, None if jit is None else jit.to(device), ones, offsets, B * S)
    sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
    pts, z, valid = O.sample_ray(sc, r[:, :3].cpu(), r[:, 3:6].cpu(), S, jit)
    xyz_o = O.normalize_coord(sc, pts).reshape(-1, 3)
    xyz_h = rec_xyz.cpu()
    ok = (rec_ray.cpu().long() * S + rec_k.cpu().long() == torch.arange(B * S)).all()
    d = (xyz_h - xyz_o)
    neq = (d != 0).any(1)
    ulp = (xyz_h.view(torch.int32) - xyz_o.contiguous().view(torch.int32)).abs()
    print(name, "order ok", bool(ok), "samples", B * S, "positions differing", int(neq.sum()), "max |d|", float(d.abs().max()), "max ulp", int(ulp.max()),
          "in-box differing", int((neq & valid.reshape(-1)).sum()), flush=True)
    # sigma of the march vs sigma at the oracle's coordinates (HIP gather)
    fh = model.compute_densityfeature(xyz_o[valid.reshape(-1)].to(device)).cpu()
    sg = torch.zeros(B * S); sg[valid.reshape(-1)] = torch.nn.functional.softplus(fh + float(sc.density_shift))
    sm = sig.cpu().reshape(-1)
    m = (sm > 0) & (sg > 0)
    rel = ((sm - sg).abs() / sg)[m]
    print("   march sigma vs gather at the oracle's coordinates:", int(m.sum()), "samples, rel rms", float(rel.pow(2).mean().sqrt()), "max", float(rel.max()), flush=True)

"""Every map of given rays of the headline batch on HIP and on the oracle, with N.V (GPU box).  Usage: python tools/ray_diag.py 2083 733"""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from oracle import tensoir_oracle as O
from tests.helpers import scene_from_model
from tensoir_amd import Renderer_TensoIR_train
ids = [int(v) for v in sys.argv[1:]] or [0]
sys.argv = ["bench.py"]; a = bench.parse()
device = torch.device("cuda", 0)
args = types.SimpleNamespace(second_nSample=a.second_samples, second_near=0.05, second_far=1.5)
ckpt, model, rays, lidx = bench.build_scene(a, device, 0)
sc = scene_from_model(ckpt, model, a.env_h, a.env_w)
sel = torch.tensor(ids)
with torch.no_grad():
    ref = O.renderer_train(sc, rays.cpu()[sel], lidx.cpu()[sel], n_samples=a.samples, second_n_sample=a.second_samples)
    ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=False, is_relight=True,
                                 sample_method="fixed_envirmap", chunk_size=160000, device=device, args=args)
for j, i in enumerate(ids):
    rd = rays[i, 3:6].cpu(); v = -rd / rd.norm()
    print(f"ray {i}:")
    for k in ("acc_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "fresnel_map", "rgb_map", "rgb_with_brdf_map"):
        h, o = ret[k][i].reshape(-1).cpu(), ref[k][j].reshape(-1)
        print(f"   {k:20s} hip {[float(f'{x:.8g}') for x in h.tolist()]}  oracle {[float(f'{x:.8g}') for x in o.tolist()]}  |d| {float((h - o).abs().max()):.2e}")
    nh, no = ret["normal_map"][i].cpu(), ref["normal_map"][j]
    print(f"   N.V hip {float((nh * v).sum()):+.4e}  oracle {float((no * v).sum()):+.4e}   |N| hip {float(nh.norm()):.7f} oracle {float(no.norm()):.7f}")

# per light direction: visibility and indirect radiance of the ray's surface point, HIP against the oracle, from the SAME surface point (the oracle's)
from tensoir_amd import relight
with torch.no_grad():
    dirs = model.gen_light_incident_dirs(method="fixed_envirmap").cuda().contiguous()
    D = dirs.shape[0]
    for j, i in enumerate(ids):
        o, d = rays[i, :3].cpu(), rays[i, 3:6].cpu()
        for name, dep in (("oracle's surface point", ref["depth_map"][j].reshape(-1)[0]), ("HIP's surface point", ret["depth_map"][i].reshape(-1)[0].cpu())):
            x = (o + d * dep).reshape(1, 3)
            P = x.expand(D, 3).contiguous()
            vo, _, io = O.compute_radiance(sc, P, dirs.cpu(), lidx.cpu()[i:i + 1].expand(D, 1), n_sample=a.second_samples)
            vh, _, ih = relight.compute_radiance(model, P.to(device), dirs, lidx[i:i + 1].expand(D, 1), nSample=a.second_samples)
            dv, di = (vh.cpu() - vo).abs(), (ih.cpu() - io).abs().max(-1).values
            k = int(dv.argmax())
            print(f"ray {i}, {name}: visibility max |hip - oracle| {float(dv.max()):.2e} at direction {k} (hip {float(vh[k]):.6f} oracle {float(vo[k]):.6f}, n.l {float((ref['normal_map'][j] * dirs[k].cpu()).sum()):+.3e});"
                  f" directions with |d vis| > 1e-5: {int((dv > 1e-5).sum())}; indirect max |d| {float(di.max()):.2e}")

# the oracle's shading on the oracle's maps and on HIP's maps of the same ray
with torch.no_grad():
    for j, i in enumerate(ids):
        rr, ll = rays.cpu()[i:i + 1], lidx.cpu()[i:i + 1]
        for name, src, jj in (("oracle maps", ref, j), ("HIP maps", {k: v.cpu() for k, v in ret.items() if torch.is_tensor(v) and v.dim() > 0}, i)):
            g = lambda k: src[k][jj:jj + 1]
            out, aux = O.render_with_brdf(sc, g("depth_map").reshape(1), g("normal_map"), g("albedo_map"), g("roughness_map").reshape(1, -1).expand(1, 3), g("fresnel_map"), rr, ll,
                                          n_sample=a.second_samples, return_aux=True)
            print(f"ray {i}: oracle shading on {name}: {[float(f'{x:.8g}') for x in out.reshape(-1).tolist()]}   (oracle's own {[float(f'{x:.8g}') for x in ref['rgb_with_brdf_map'][j].tolist()]}, HIP's {[float(f'{x:.8g}') for x in ret['rgb_with_brdf_map'][i].cpu().tolist()]})"
                  f"  active directions {int((aux.cosine > 1e-6).sum())}, smallest positive cosine {float(aux.cosine[aux.cosine > 0].min()):.3e}")

# how discontinuous is the reference's own shading in the depth of these rays?  (the oracle on its own maps, depth moved by whole ulps)
import numpy as np
with torch.no_grad():
    for j, i in enumerate(ids):
        rr, ll = rays.cpu()[i:i + 1], lidx.cpu()[i:i + 1]
        g = lambda k: ref[k][j:j + 1]
        dep0 = g("depth_map").reshape(1)
        outs = []
        for u in (-2, -1, 0, 1, 2):
            dep = torch.from_numpy(np.nextafter(dep0.numpy(), np.float32(np.inf if u > 0 else -np.inf))) if abs(u) == 1 else dep0.clone()
            if abs(u) == 2:
                dep = torch.from_numpy(np.nextafter(np.nextafter(dep0.numpy(), np.float32(np.inf if u > 0 else -np.inf)), np.float32(np.inf if u > 0 else -np.inf)))
            o_ = O.render_with_brdf(sc, dep, g("normal_map"), g("albedo_map"), g("roughness_map").reshape(1, -1).expand(1, 3), g("fresnel_map"), rr, ll, n_sample=a.second_samples)
            outs.append(float(o_.reshape(-1)[0]))
        print(f"ray {i}: oracle rgb_with_brdf[0] with its own depth moved by -2, -1, 0, +1, +2 ulps: {[float(f'{x:.8g}') for x in outs]}")

"""Diagnostics (GPU box): per-sample d loss / d density-feature of the HIP march backward vs fp64 autograd, for the
structured cotangents of an is_relight=False training step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tensoir_oracle as O
from tests.helpers import golden_scene, golden_checkpoint, T
import tensoir_amd
from tensoir_amd import ops, training

g = np.load(os.path.join(ROOT, "tests/golden/small_scene.npz")); tg = np.load(os.path.join(ROOT, "tests/golden/train_grads.npz"))
sc = golden_scene(g)
eh, ew = [int(x) for x in g["scene/envmap_hw"]]
m = tensoir_amd.model_from_checkpoint(golden_checkpoint(g), "cuda", envmap_h=eh, envmap_w=ew); m.march_t_stop = 0.0
rays, lidx = T(g, "rays/rays"), T(g, "rays/light_idx")
S = 64; gt = T(tg, "train/rgb_gt"); B = rays.shape[0]
gen = torch.Generator().manual_seed(21)
jit = torch.rand(B, 1, generator=gen)
out, aux = O.forward_primary(sc, rays, lidx.int(), n_samples=S, is_relight=False, ray_jitter=jit, return_aux=True)
z = aux.z; dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), -1) * sc.distance_scale
xyz = aux.xyz; app = aux.app_mask
rgb = torch.zeros(B, S, 3)
rf, _ = O.both_feature(sc, xyz[app], lidx.view(-1, 1, 1).expand(B, S, 1)[app].int())
vd = rays[:, 3:6].view(-1, 1, 3).expand(B, S, 3)
rgb[app] = O.render_rgb(sc, vd[app], rf)
feat = torch.zeros(B, S); feat[aux.valid] = O.density_feature(sc, xyz[aux.valid])
f64 = feat.double().clone().requires_grad_(True)
sig = torch.where(aux.valid, torch.nn.functional.softplus(f64 - 10.0), torch.zeros_like(f64))
a, w, bg = O.raw2alpha(sig, dists.double())
rm = (w[..., None] * rgb.double()).sum(-2) + (1 - w.sum(-1)[..., None])
loss = ((rm - gt.double()) ** 2).mean(); loss.backward()
ref = f64.grad
# HIP: same cotangents
gc = (2 * (rm.detach() - gt.double()) / (B * 3)).float()
gw = torch.where(app, (rgb * gc[:, None, :]).sum(-1), torch.zeros(B, S))
ga = -gc.sum(-1); gd = torch.zeros(B)
f = m.packed_field()
r = rays.cuda()
weight, sigma, acc, depth, _t, _c = ops.march_primary_train(f, r, jit.cuda(), S, 0.0)
print("fwd weight err", float((weight.cpu() - w.float()).abs().max()), "sigma err", float((sigma.cpu() - sig.float().detach()).abs().max()))
bufs = training._grad_buffers(m, f)
gf = ops.march_primary_bwd(f, bufs["desc"], r, jit.cuda(), sigma, weight, gw.cuda(), ga.cuda(), gd.cuda(), True).cpu()
d = (gf.double() - ref).abs()
print("per-sample df: max ref", float(ref.abs().max()), "max err", float(d.max()), "rel", float(d.max() / ref.abs().max()))
i = int(d.argmax()); b, k = i // S, i % S
print("worst at ray", b, "k", k, "hip", float(gf[b, k]), "ref", float(ref[b, k]), "sigma", float(sig[b, k]), "w", float(w[b, k]), "valid", bool(aux.valid[b, k]))
print("row ref ", ref[b].numpy()[max(0,k-4):k+5])
print("row hip ", gf[b].numpy()[max(0,k-4):k+5])
# total gradient check via oracle scatter of ref df

"""Training-step benchmark (GPU box): forward + hand-written backward + Adam on the synthetic C2 scene.

    python tools/train_bench.py [--steps 20] [--rays 4096] [--samples 512] [--grid 300] [--relight 1]

Reports iterations/s, rays/s and a per-entry-point breakdown (events around every C call).  One step =
Renderer_TensoIR_train(is_train=True) + the loss of train_tensoIR.py:262-311 + backward() + optimizer.step()."""
import argparse, json, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tensoir_amd import Renderer_TensoIR_train, ops

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--samples", type=int, default=512)
ap.add_argument("--grid", type=int, default=300)
ap.add_argument("--relight", type=int, default=1)
ap.add_argument("--cprofile", type=int, default=0, help="print the N hottest host functions of 10 steps (stderr)")
ap.add_argument("--census", default="", help="write a census of the framework (ATen) operators of one step to this JSON file: which "
                                             "phase issues them, and whether product code or the caller's loss expression / autograd does")
a = ap.parse_args()
a.env_h, a.env_w, a.second_samples = 8, 16, 96
dev = torch.device("cuda", 0)
ckpt, model, rays, lidx = bench.build_scene(a, dev, 0)
model.march_t_stop = 1e-6
args = types.SimpleNamespace(second_nSample=96, second_near=0.05, second_far=1.5)
gt = torch.rand(rays.shape[0], 3, device=dev)
# the optimizer the launcher binds (tensoir_amd.optim.Adam: one launch per step); TENSOIR_TORCH_ADAM=1: torch's multi-tensor Adam
from tensoir_amd import optim as _optim
_Adam = _optim._TorchAdam if os.environ.get("TENSOIR_TORCH_ADAM", "0") == "1" else _optim.Adam
opt = _Adam(model.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99))
W = dict(rgb_brdf=0.2, normals_diff=0.0005, normals_orientation=0.001, albedo_smoothness=0.001, roughness_smoothness=0.001)


PHASE = None        # host seconds per phase (forward incl. its end-of-pass count read, loss, backward, optimizer)


def step():
    t0 = time.perf_counter()
    ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=True,
                                 is_relight=bool(a.relight), sample_method="stratified_sampling", device=dev, args=args)
    loss = torch.mean((ret["rgb_map"] - gt) ** 2)
    if a.relight:
        loss = loss + W["rgb_brdf"] * torch.mean((ret["rgb_with_brdf_map"] - gt) ** 2) \
            + W["normals_diff"] * ret["normals_diff_map"].mean() \
            + W["normals_orientation"] * ret["normals_orientation_loss_map"].mean() \
            + W["roughness_smoothness"] * ret["roughness_smoothness_loss"] + W["albedo_smoothness"] * ret["albedo_smoothness_loss"]
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if PHASE is not None:
        PHASE[0] += t1 - t0; PHASE[1] += t2 - t1; PHASE[2] += t3 - t2
    return loss


l0 = step()
for _ in range(100):                       # untimed, a fixed count (the scene trains while it is timed: same trajectory every
    step()                                 # run): brings the GPU out of its idle power state -- a fresh box ramps its clocks
    torch.cuda.synchronize()               # over the first ~100 ms, 13 ms per step instead of 6 measured right after start
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
PHASE = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(a.steps):
    l1 = step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
phase, PHASE = [round(1e3 * v / a.steps, 3) for v in PHASE], None
if a.cprofile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(a.cprofile)
    st.sort_stats("cumtime").print_stats(a.cprofile)
census = None
if a.census:
    # Every operator that reaches the dispatcher during 4 steps, by phase.  "product" = issued from tensoir_amd code (the forward
    # call, the backward of its autograd functions, the optimizer); "caller" = the loss expression the training script writes
    # (train_tensoIR.py:262-311) and what autograd derives from it (slice / mean / pow backward, gradient accumulation).
    from torch.utils._python_dispatch import TorchDispatchMode
    from tensoir_amd import training as _tr
    torch.autograd.set_multithreading_enabled(False)       # backward on this thread: the dispatch mode is thread-local
    NO_LAUNCH = {"empty", "empty_like", "empty_strided", "view", "_unsafe_view", "reshape", "as_strided", "slice", "select", "expand",
                 "permute", "t", "transpose", "squeeze", "unsqueeze", "detach", "alias", "split", "split_with_sizes", "unbind", "narrow",
                 "_local_scalar_dense", "is_pinned", "record_stream", "is_same_size", "sym_size", "stride", "_to_copy_noop", "lift_fresh",
                 "result_type", "can_cast", "unsafe_split", "chunk", "unsafe_chunk", "resize_", "set_", "_reshape_alias", "new_empty",
                 "new_empty_strided", "view_as", "expand_as", "contiguous_noop", "_pin_memory"}
    cur = {"phase": "forward", "product": 1}
    hist = {}

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = getattr(getattr(func, "overloadpacket", func), "__name__", str(func))
            key = f"{cur['phase']}/{'product' if cur['product'] else 'caller'}"
            h = hist.setdefault(key, {})
            h[name] = h.get(name, 0) + 1
            return func(*args, **(kwargs or {}))

    def flagged(fn):
        def wrapped(*x, **k):
            old, cur["product"] = cur["product"], 1
            try:
                return fn(*x, **k)
            finally:
                cur["product"] = old
        return staticmethod(wrapped)
    saved = {c: c.backward for c in (_tr.PrimaryRenderFn, _tr.EnvSGFn, _tr.EnvPixelFn, _tr.ShadeFn)}
    for c, b in saved.items():
        c.backward = flagged(b)

    def census_step():
        cur.update(phase="forward", product=1)
        ret = Renderer_TensoIR_train(rays, None, lidx, model, N_samples=a.samples, white_bg=True, is_train=True,
                                     is_relight=bool(a.relight), sample_method="stratified_sampling", device=dev, args=args)
        cur.update(phase="loss", product=0)
        loss = torch.mean((ret["rgb_map"] - gt) ** 2)
        if a.relight:
            loss = loss + W["rgb_brdf"] * torch.mean((ret["rgb_with_brdf_map"] - gt) ** 2) \
                + W["normals_diff"] * ret["normals_diff_map"].mean() \
                + W["normals_orientation"] * ret["normals_orientation_loss_map"].mean() \
                + W["roughness_smoothness"] * ret["roughness_smoothness_loss"] + W["albedo_smoothness"] * ret["albedo_smoothness_loss"]
        cur.update(phase="backward", product=0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        cur.update(phase="optimizer", product=1)
        opt.step()
    N_CENSUS = 4
    with Census():
        for _ in range(N_CENSUS):
            census_step()
    torch.cuda.synchronize()
    for c, b in saved.items():
        c.backward = staticmethod(b)
    torch.autograd.set_multithreading_enabled(True)
    rows_c = {}
    for key, h in sorted(hist.items()):
        launching = {n: round(v / N_CENSUS, 2) for n, v in sorted(h.items(), key=lambda kv: -kv[1]) if n not in NO_LAUNCH}
        rows_c[key] = {"operators_that_launch_per_step": round(sum(launching.values()), 2), "by_operator": launching,
                       "views_and_allocations_per_step": round(sum(v for n, v in h.items() if n in NO_LAUNCH) / N_CENSUS, 2)}
    census = {"steps": N_CENSUS, "note": "operators seen by a TorchDispatchMode (backward on the calling thread); an operator that launches "
              "is one ATen kernel or copy in almost every case; product = issued by tensoir_amd code, caller = the script's loss expression "
              "and the autograd nodes derived from it", "by_phase": rows_c,
              "launching_operators_per_step": {"product": round(sum(r["operators_that_launch_per_step"] for k, r in rows_c.items() if k.endswith("product")), 2),
                                               "caller": round(sum(r["operators_that_launch_per_step"] for k, r in rows_c.items() if k.endswith("caller")), 2)}}
    with open(a.census, "w") as fh:
        json.dump(census, fh, indent=1)
ops.TIMING = []
for _ in range(3):
    step()
torch.cuda.synchronize()
agg = {}
for name, e0, e1 in ops.TIMING:
    k = agg.setdefault(name, [0.0, 0])
    k[0] += e0.elapsed_time(e1); k[1] += 1
ops.TIMING = None
rows = sorted(((n, v[0] / 3, v[1] / 3) for n, v in agg.items()), key=lambda r: -r[1])
out = {"it_per_s": round(a.steps / el, 2), "ms_per_step": round(1e3 * el / a.steps, 3),
       "rays_per_s": round(a.steps * rays.shape[0] / el, 1), "loss_first": float(l0), "loss_last": float(l1),
       "config": {"rays": rays.shape[0], "samples": a.samples, "grid": a.grid, "relight": bool(a.relight)},
       "hip_ms_per_step": round(sum(r[1] for r in rows), 3),
       "host_ms_per_step": {"forward_incl_count_read": phase[0], "loss_and_backward": phase[1], "optimizer": phase[2]},
       "entry_points": [{"name": n, "ms_per_step": round(ms, 4), "launches": c} for n, ms, c in rows[:16]]}
print(json.dumps(out))

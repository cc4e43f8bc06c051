"""One rank of the multi-rank hardware checks (launched by tests/test_gpu_multirank.py and tools/ under
``python -m torch.distributed.run --nproc-per-node N tests/multirank_worker.py --backend nccl|gloo --out DIR``).

What every rank does, on real kernels:

1. **image**: the golden scene's rays (tiled to 343 with small pose changes) rendered unsharded on this rank, then through
   ``dist.render_sharded`` with the eager renderer and with the HIP-graph chunk renderer (captured BEFORE the process group
   exists, as bench.py does), row tiles and interleaved tiles: every map of the gathered image must equal the unsharded
   render bit for bit (rays are independent).
2. **dp step**: one data-parallel training step -- rays[rank::world], backward, ``dist.allreduce_gradients`` (0.25 MB
   buckets) -- and rank 0 writes the averaged gradients; the test compares them with the single-process full-batch step and,
   across backends, nccl against gloo.

Not a pytest file (no test_ prefix): it needs the launcher's RANK / WORLD_SIZE / LOCAL_RANK."""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_inputs(g, n_tiles=7):
    """[49 * n_tiles, 6] rays: the golden camera batch under small rotations about y (every copy still sees the object)."""
    from tests.helpers import T
    rays, lidx = T(g, "rays/rays"), T(g, "rays/light_idx")
    out = []
    for k in range(n_tiles):
        a = 0.04 * k
        R = torch.tensor([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]], dtype=torch.float32)
        d = rays[:, 3:] @ R.T
        out.append(torch.cat([rays[:, :3] @ R.T, d / d.norm(dim=-1, keepdim=True)], dim=-1))
    return torch.cat(out).contiguous(), lidx.repeat(n_tiles, 1).contiguous()


def dp_step(model, rays, lidx, gt, jitter, noise, args, mine, reduce_fn):
    """Forward + backward on rays[mine] with the given jitter draws; reduce_fn(params) -> bucket count.  Returns the
    gradients by parameter name (after the reduction)."""
    from oracle import tensoir_oracle as O          # the reference's loss formula (checker side of the test)
    from tensoir_amd import Renderer_TensoIR_train
    Bm = int(mine.numel())
    jm, nm = jitter[mine], noise[mine]
    orig_rand, orig_fwd = torch.rand, type(model).forward

    def fake_rand(*a, **k):
        return jm.clone() if tuple(a) == (Bm, 1) else orig_rand(*a, **k)

    def fwd(self, r, l, **k):
        return orig_fwd(self, r, l, _brdf_jitter_dense=nm, **k)
    torch.rand, type(model).forward = fake_rand, fwd
    try:
        ret = Renderer_TensoIR_train(rays[mine], None, lidx[mine], model, N_samples=64, white_bg=True, is_train=True,
                                     is_relight=True, sample_method="fixed_envirmap", device=rays.device, args=args)
    finally:
        torch.rand, type(model).forward = orig_rand, orig_fwd
    model.zero_grad(set_to_none=True)
    O.training_loss(ret, gt[mine], True).backward()
    params = [p for p in model.parameters() if p.requires_grad]
    buckets = reduce_fn(params)
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, buckets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", required=True)
    ap.add_argument("--shared-gpu", action="store_true", help="every rank on cuda:0 (gloo plumbing on a one-GPU box)")
    a = ap.parse_args()
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    import torch.distributed as dist
    import tensoir_amd
    from tensoir_amd import Renderer_TensoIR_train, _lib
    from tensoir_amd import dist as tdist
    from tests.helpers import golden_checkpoint
    assert torch.cuda.is_available(), "multirank_worker needs a GPU"
    dev_idx = 0 if a.shared_gpu else local
    assert a.shared_gpu or torch.cuda.device_count() >= world, "one GPU per rank (or --shared-gpu with gloo)"
    torch.cuda.set_device(dev_idx)
    device = torch.device("cuda", dev_idx)
    assert _lib.lib().tir_device_check() == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_scene.npz"))
    tg = np.load(os.path.join(ROOT, "tests", "golden", "train_grads.npz"))
    eh, ew = [int(x) for x in g["scene/envmap_hw"]]
    model = tensoir_amd.model_from_checkpoint(golden_checkpoint(g), device, envmap_h=eh, envmap_w=ew)
    args = types.SimpleNamespace(second_nSample=24, second_near=0.05, second_far=1.5)
    rays, lidx = (t.to(device) for t in build_inputs(g))
    n = rays.shape[0]
    chunk = 49                       # the golden camera batch

    def eager(r, l):
        with torch.no_grad():
            return Renderer_TensoIR_train(r, None, l, model, N_samples=64, white_bg=True, is_train=False, is_relight=True,
                                          sample_method="fixed_envirmap", device=device, args=args)
    full = {k: v.clone() for k, v in eager(rays, lidx).items() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n}
    graphed = tdist.GraphedChunkRenderer(model, chunk, args, N_samples=64, device=device, lanes=2)
    with torch.no_grad():          # capture + capacity learning before any RCCL thread exists
        for _ in range(3):
            tdist._render_chunks(graphed, rays, lidx, torch.arange(n, device=device), chunk)
            if graphed.validate():
                break
    dist.init_process_group(a.backend, **({"device_id": device} if a.backend == "nccl" else {}))
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "device": str(device), "image": {}}
    try:
        assert dist.get_world_size() == world
        for name, fn in (("eager", eager), ("graphed", graphed)):
            for tile in (0, chunk):
                with torch.no_grad():
                    img = tdist.render_sharded(fn, rays, lidx, chunk=chunk, tile=tile)
                bad = [k for k in tdist.unpack_records(torch.zeros(1, tdist.RECORD)) if not torch.equal(img[k].reshape(full[k].shape), full[k])]
                res["image"][f"{name}/tile{tile}"] = bad                      # [] = every map bit-identical
        # ---- data-parallel training step ----
        B = 48                       # equal shards: the mean of the shard gradients is the full-batch gradient
        r64, l64 = rays[:B].contiguous(), lidx[:B].contiguous()
        gt = torch.from_numpy(np.array(tg["train/rgb_gt"]))[:B].to(device)
        jitter = torch.rand(B, 1, generator=torch.Generator().manual_seed(5))
        noise = torch.randn(B, 64, 3, generator=torch.Generator().manual_seed(6))
        mine = tdist.shard_batch(B, rank, world)
        grads, buckets = dp_step(model, r64, l64, gt, jitter, noise, args, mine,
                                 lambda ps: tdist.allreduce_gradients(ps, bucket_mb=0.25))
        torch.cuda.synchronize()
        res["buckets"] = buckets
        # every rank must hold the same averaged gradients: compare rank r's with rank 0's through one more collective
        worst = 0.0
        for nme in sorted(grads):
            ref = grads[nme].clone()
            dist.broadcast(ref, src=0)
            worst = max(worst, float((ref - grads[nme]).abs().max()))
        res["max_abs_diff_vs_rank0"] = worst
        if rank == 0:
            torch.save({k: v.cpu() for k, v in grads.items()}, os.path.join(a.out, f"grads_{a.backend}.pt"))
        if a.backend == "nccl":
            res["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    finally:
        dist.barrier()
        dist.destroy_process_group()
    with open(os.path.join(a.out, f"rank{rank}_{a.backend}.json"), "w") as fh:
        json.dump(res, fh)


if __name__ == "__main__":
    main()

"""Ray-sharded rendering across the GPUs of one node.

Rays are independent units: the field (70-190 MB) is replicated on every GPU, each rank renders its
shard with the same kernels, and ONE collective at the end -- an all-gather of the per-ray output
records over RCCL/xGMI -- assembles the image on every rank.  The reference has no data-path
collective at all (it only calls init_process_group + barrier, train_tensoIR.py:22-27); this is the
MI355X-native replacement for its sequential per-chunk loop (renderer.py:225-249).

One process per GPU; ``backend='nccl'`` is RCCL on ROCm; the same code runs on ``gloo`` for CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

RECORD = 24          # floats per ray in the exchanged record: 20 map floats + rgb_with_brdf 3 + pad


def shard_rows(n_rays: int, rank: int, world: int, tile: int = 0):
    """Index tensor (int64, ascending) of the rays rank `rank` renders.

    tile == 0: contiguous row-tiles (rank r gets [r*ceil(n/world), ...)), the 800x800 -> 8 x 100-row
    layout of SURVEY 8e.  tile > 0: interleaved tiles of `tile` rays (ray i -> rank (i // tile) % world)
    for load balance when background rows terminate early.
    """
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if tile <= 0:
        per = (n_rays + world - 1) // world
        lo, hi = min(rank * per, n_rays), min((rank + 1) * per, n_rays)
        return torch.arange(lo, hi, dtype=torch.int64)
    idx = torch.arange(n_rays, dtype=torch.int64)
    return idx[(idx // tile) % world == rank]


def shard_capacity(n_rays: int, world: int, tile: int = 0) -> int:
    """Largest shard over ranks (every rank pads its record buffer to this for the all-gather)."""
    return max(int(shard_rows(n_rays, r, world, tile).numel()) for r in range(world))


def pack_records(ret: dict) -> torch.Tensor:
    """12-key dict of Renderer_TensoIR_train -> [n, RECORD] record rows (one concatenation kernel)."""
    rgb = ret["rgb_map"]
    n = rgb.shape[0]
    col = lambda t: t.reshape(n, -1).to(torch.float32)
    pad = rgb.new_zeros((n, 3), dtype=torch.float32)
    return torch.cat([col(rgb), col(ret["depth_map"]), col(ret["normal_map"]), col(ret["albedo_map"]),
                      col(ret["roughness_map"]), col(ret["fresnel_map"]), col(ret["acc_map"]),
                      col(ret["normals_diff_map"]), col(ret["normals_orientation_loss_map"]), pad,
                      col(ret["rgb_with_brdf_map"]), pad[:, :1]], dim=1)


def unpack_records(rec: torch.Tensor) -> dict:
    return {"rgb_map": rec[:, 0:3], "depth_map": rec[:, 3], "normal_map": rec[:, 4:7],
            "albedo_map": rec[:, 7:10], "roughness_map": rec[:, 10:11], "fresnel_map": rec[:, 11:14],
            "acc_map": rec[:, 14], "normals_diff_map": rec[:, 15:16],
            "normals_orientation_loss_map": rec[:, 16:17], "rgb_with_brdf_map": rec[:, 20:23]}


_LAYOUT = {}


def _layout(n_rays: int, world: int, tile: int, device):
    """(capacity, gather index [n_rays]) of a sharding, built once per (image size, world, tile, device) and kept on the device:
    row i of the image is row `index[i]` of the all-gathered [world * capacity] record block.  (Rebuilding the per-rank index
    tensors on the host for every image cost 0.5 - 4.7 ms of host arithmetic and pageable copies per image.)"""
    key = (int(n_rays), int(world), int(tile), str(device))
    hit = _LAYOUT.get(key)
    if hit is None:
        cap = shard_capacity(n_rays, world, tile)
        index = torch.empty((n_rays,), dtype=torch.int64)
        for r in range(world):
            idx = shard_rows(n_rays, r, world, tile)
            index[idx] = r * cap + torch.arange(idx.numel(), dtype=torch.int64)
        if len(_LAYOUT) > 16:
            _LAYOUT.clear()
        hit = _LAYOUT[key] = (cap, index.to(device))
    return hit


def gather_records(local: torch.Tensor, n_rays: int, rank: int, world: int, tile: int = 0, group=None):
    """All-gather the per-rank record rows and put them back in image order.  local: [n_local, RECORD]."""
    cap, index = _layout(n_rays, world, tile, local.device)
    if world == 1 and local.shape[0] == cap:
        gathered = local
    else:
        buf = torch.zeros((cap, local.shape[1]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
        if world == 1:
            gathered = buf
        else:
            gathered = torch.empty((world * cap, local.shape[1]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(gathered, buf, group=group)
    return gathered.index_select(0, index)


def _render_chunks(render_fn, rays, light_idx, mine, chunk):
    """This rank's chunks -> list of packed record blocks.  A renderer with ``render_packed`` (GraphedChunkRenderer) runs
    indexing, replay and packing of chunk i on lane (i mod lanes)'s stream -- several chunks in flight -- and is joined
    before the blocks are used; any other callable is called chunk by chunk on the current stream."""
    chunks = [c for c in torch.split(mine, chunk) if c.numel()]
    packed = getattr(render_fn, "render_packed", None)
    if packed is None:
        return [pack_records(render_fn(rays[c], light_idx[c])) for c in chunks]
    render_fn.fork()
    parts = [packed(rays, light_idx, c) for c in chunks]
    render_fn.join()
    return parts


def render_sharded(render_fn, rays, light_idx, rank=None, world=None, chunk=4096, tile=0, group=None):
    """Render `rays` ([N,6], identical on every rank) data-parallel over ranks.

    render_fn(rays_chunk, light_idx_chunk) -> dict with the keys of Renderer_TensoIR_train.
    Every rank returns the full-image dict (one all-gather per image, not per chunk).  A render_fn with a
    ``validate()`` method (GraphedChunkRenderer) is asked once per image whether its deferred capacity checks held;
    if not, this rank's chunks are rendered again (the renderer re-captures with room).
    """
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays.shape[0]
    mine = shard_rows(n, rank, world, tile).to(rays.device)
    for attempt in range(4):
        parts = _render_chunks(render_fn, rays, light_idx, mine, chunk)
        validate = getattr(render_fn, "validate", None)
        if validate is None or validate():
            break
    else:
        raise RuntimeError("render_sharded: the renderer's record capacity kept overflowing")
    if parts:
        local = torch.cat(parts, dim=0)
    else:
        local = torch.zeros((0, RECORD), dtype=torch.float32, device=rays.device)
    return unpack_records(gather_records(local, n, rank, world, tile, group))


def render_sharded_timed(render_fn, rays, light_idx, rank=None, world=None, chunk=4096, tile=0, group=None):
    """render_sharded with the two phases timed on the host (device drained at the phase boundaries when the rays live
    on a GPU): returns (image dict, local_render_s, exchange_s).  Used by ``bench.py --workload image``."""
    import time
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    on_gpu = rays.is_cuda

    def drain():
        if on_gpu:
            torch.cuda.synchronize(rays.device)
    n = rays.shape[0]
    mine = shard_rows(n, rank, world, tile).to(rays.device)
    drain()
    t0 = time.perf_counter()
    for attempt in range(4):
        parts = _render_chunks(render_fn, rays, light_idx, mine, chunk)
        validate = getattr(render_fn, "validate", None)
        if validate is None or validate():
            break
    else:
        raise RuntimeError("render_sharded_timed: the renderer's record capacity kept overflowing")
    local = torch.cat(parts, dim=0) if parts else torch.zeros((0, RECORD), dtype=torch.float32, device=rays.device)
    drain()
    t1 = time.perf_counter()
    img = unpack_records(gather_records(local, n, rank, world, tile, group))
    drain()
    t2 = time.perf_counter()
    return img, t1 - t0, t2 - t1


class GraphedChunkRenderer:
    """render_fn for render_sharded: full chunks replay a captured HIP graph with no host wait per chunk (inputs go into
    the graph's static buffers, outputs are packed straight from them, the record-capacity check of all replays is made
    once per image by validate()); a ragged last chunk takes the eager renderer.

    ``lanes`` chunks are in flight at once: lane l owns a captured graph (own buffers, own device-side pass state) and a
    HIP stream; chunk i is indexed, replayed and packed on lane (i mod lanes).  A step's kernels have tails in which CUs
    idle; the other lane's chunk fills them (+10-14 % whole-image rate with two lanes, tools/two_stream_probe.py)."""

    def __init__(self, tensoIR, chunk, args, N_samples=-1, white_bg=True, is_relight=True, device="cuda", lanes=2):
        from .graph import GraphedRenderer
        from .renderer import Renderer_TensoIR_train
        self.grs = [GraphedRenderer(tensoIR, chunk, N_samples=N_samples, white_bg=white_bg, is_relight=is_relight,
                                    args=args, device=device) for _ in range(max(1, int(lanes)))]
        self.gr = self.grs[0]
        self.streams = None
        self.device = torch.device(device)
        self._i = 0
        self._eager = lambda r, l: Renderer_TensoIR_train(r, None, l, tensoIR, N_samples=N_samples, white_bg=white_bg,
                                                          is_train=False, is_relight=is_relight,
                                                          sample_method="fixed_envirmap", device=device, args=args)

    def _lane_streams(self):
        if self.streams is None:
            self.streams = [torch.cuda.Stream(device=self.device) for _ in self.grs]
        return self.streams

    def fork(self):
        """The lanes start behind everything queued on the caller's stream (the ray tensors, the shard indices)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self._lane_streams():
            st.wait_stream(cur)

    def join(self):
        """The caller's stream waits for every lane."""
        cur = torch.cuda.current_stream(self.device)
        for st in self._lane_streams():
            cur.wait_stream(st)

    @torch.no_grad()
    def render_packed(self, rays, light_idx, idx):
        """Chunk rays[idx] -> packed records [len(idx), RECORD], everything queued on the next lane's stream.  Call fork()
        before the first chunk and join() before using the blocks (render_sharded does)."""
        lane = self._i % len(self.grs)
        self._i += 1
        gr, st = self.grs[lane], self._lane_streams()[lane]
        caller = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            if idx.numel() != gr.n_rays:
                rec = pack_records(self._eager(rays[idx], light_idx[idx]))
            else:
                torch.index_select(rays, 0, idx, out=gr.rays)
                gr.lidx.copy_(light_idx[idx].reshape(-1, 1), non_blocking=True)
                rec = pack_records(gr(clone_outputs=False, defer_check=True))
            rec.record_stream(caller)          # allocated on the lane's stream, consumed (after join) on the caller's
        return rec

    @torch.no_grad()
    def __call__(self, rays, light_idx):
        """One chunk on the caller's stream through lane 0's graph (the single-lane interface)."""
        if rays.shape[0] != self.gr.n_rays:
            return self._eager(rays, light_idx)
        self.gr.rays.copy_(rays, non_blocking=True)
        self.gr.lidx.copy_(light_idx.reshape(-1, 1), non_blocking=True)
        return self.gr(clone_outputs=False, defer_check=True)

    def validate(self):
        if self.streams is not None:
            self.join()
        return all([gr.validate() for gr in self.grs])


# ---- data-parallel training (the reference itself never all-reduces: section 2.1) ----------
# The unmodified training scripts under `torchrun` / WORLD_SIZE > 1 (train_tensoIR.py:22-27 create the process group and then
# run N identical trainers: same seeds, same batches, no gradient exchange).  `python -m tensoir_amd.run` switches LAUNCHER_DP
# on: TensorVMSplit.filtering_rays then hands every rank a disjoint 1/world of the kept training rays (the k-th kept ray goes to
# rank k mod world -- the mask it returns selects the same rows of the colour / light-index tables, train_tensoIR.py:228-231) and
# LauncherAdam.step() averages the gradients over the ranks (bucketed all-reduce below) before the one-launch update: batch_size
# rays per rank and step, `world * batch_size` per optimizer step, identical parameters on every rank.
LAUNCHER_DP = {"on": False}


def launcher_dp():
    """(rank, world) when the launcher's data-parallel mode is active in this process (enabled by tensoir_amd.run, process group
    initialised by the script, more than one rank), else None."""
    if not LAUNCHER_DP["on"] or not dist.is_available() or not dist.is_initialized():
        return None
    world = dist.get_world_size()
    return (dist.get_rank(), world) if world > 1 else None


def shard_filter_mask(mask: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rows of a boolean keep-mask that belong to `rank`: the k-th True goes to rank k mod world (shards differ by at most
    one ray; their union is the mask, they are pairwise disjoint)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    flat = mask.reshape(-1)
    kept = flat.nonzero(as_tuple=False).reshape(-1)
    out = torch.zeros_like(flat)
    out[kept[rank::world]] = True
    return out.view(mask.shape)


def shard_batch(n_rays: int, rank: int, world: int):
    """Training batches are split `rays[rank::world]`: every rank marches batch/world rays."""
    return torch.arange(rank, n_rays, world, dtype=torch.int64)


def _memory_order(p):
    """Dimension permutation that walks `p` in memory order (so that flattening a gradient that shares p's layout
    -- e.g. the channel-last VM planes -- is a view, not a transposing copy).  Depends only on the parameter, hence
    identical on every rank."""
    return sorted(range(p.dim()), key=lambda d: (-p.stride(d), d))


def allreduce_gradients(params, group=None, bucket_mb: float = 64.0, average: bool = True, force: bool = False):
    """Bucketed all-reduce (RCCL over xGMI with backend 'nccl') of the .grad of `params`, in place.

    Buckets are sized for per-link-bound ring collectives on point-to-point xGMI: 64 MB keeps the 17.4 M (300^3) /
    30.8 M (400^3) parameter gradients to 2 collectives instead of ~40 per-tensor ones.  Buckets are launched
    asynchronously in reverse parameter order (decoders first, the large planes last, matching the order the
    hand-written backward produces them) and awaited together.  Parameters without a gradient contribute zeros so
    that every rank issues the same collectives.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ps = [p for p in params if p.requires_grad]
    if (world == 1 and not force) or not ps:          # force: issue the collectives on a 1-rank group too (RCCL path tests)
        return 0
    cap = max(1, int(bucket_mb * (1 << 20) / 4))
    buckets, cur, cur_n = [], [], 0
    for p in reversed(ps):
        if cur and cur_n + p.numel() > cap:
            buckets.append(cur)
            cur, cur_n = [], 0
        cur.append(p)
        cur_n += p.numel()
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        dev = b[0].device
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).permute(_memory_order(p)).reshape(-1)
                          .to(torch.float32) for p in b])
        flat = flat.to(dev)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending.append((b, flat, work))
    for b, flat, work in pending:
        work.wait()
        if average:
            flat.div_(world)
        off = 0
        for p in b:
            perm = _memory_order(p)
            if p.grad is None:
                p.grad = torch.empty_like(p)          # preserve_format: the parameter's own layout
            dst = p.grad.permute(perm)
            dst.copy_(flat[off:off + p.numel()].view(dst.shape))
            off += p.numel()
    return len(buckets)

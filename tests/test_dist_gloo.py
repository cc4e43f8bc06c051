"""Ray sharding + record all-gather on the gloo backend (world_size 2 and 3, CPU processes)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tensoir_amd import dist as tdist


def fake_render(rays, light_idx):
    """Deterministic per-ray function standing in for Renderer_TensoIR_train (rays are independent)."""
    n = rays.shape[0]
    base = rays.sum(-1, keepdim=True) + light_idx.float().view(-1, 1)
    f = lambda k, c: torch.sin(base * (k + 1)).repeat(1, c)
    return {"rgb_map": f(0, 3), "depth_map": f(1, 1)[:, 0], "normal_map": f(2, 3), "albedo_map": f(3, 3),
            "roughness_map": f(4, 1), "fresnel_map": f(5, 3), "acc_map": f(6, 1)[:, 0],
            "normals_diff_map": f(7, 1), "normals_orientation_loss_map": f(8, 1), "rgb_with_brdf_map": f(9, 3)}


@pytest.mark.parametrize("tile", [0, 5])
def test_partition_covers_every_ray_once(tile):
    for n in (0, 1, 7, 64, 801):
        for world in (1, 2, 3, 8):
            idx = torch.cat([tdist.shard_rows(n, r, world, tile) for r in range(world)])
            assert sorted(idx.tolist()) == list(range(n))
            assert tdist.shard_capacity(n, world, tile) >= (n + world - 1) // world or n == 0


def test_single_process_sharding_is_bit_exact():
    torch.manual_seed(0)
    rays, li = torch.randn(103, 6), torch.randint(0, 3, (103, 1), dtype=torch.int32)
    full = fake_render(rays, li)
    for world in (1, 2, 4):
        for tile in (0, 4):
            parts = [tdist.pack_records(fake_render(rays[i], li[i])) for i in
                     (tdist.shard_rows(103, r, world, tile) for r in range(world))]
            out = torch.empty(103, tdist.RECORD)
            for r in range(world):
                out[tdist.shard_rows(103, r, world, tile)] = parts[r]
            got = tdist.unpack_records(out)
            for k in got:
                assert torch.equal(got[k], full[k]), k


def _worker(rank, world, port, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        rays, li = torch.randn(61, 6), torch.randint(0, 3, (61, 1), dtype=torch.int32)
        got = tdist.render_sharded(fake_render, rays, li, chunk=9, tile=tile)
        full = fake_render(rays, li)
        ok = all(torch.equal(got[k], full[k]) for k in got)
        # the timed variant bench.py --workload image uses: same image, two non-negative phase times
        got2, t_local, t_exch = tdist.render_sharded_timed(fake_render, rays, li, chunk=9, tile=tile)
        ok = ok and all(torch.equal(got2[k], full[k]) for k in got2) and t_local >= 0 and t_exch >= 0
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile", [(2, 0), (2, 4), (3, 0)])
def test_gloo_all_gather(world, tile):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok for _, ok in res)


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(s)) for s in ((3, 5), (7,), (2, 2, 4), (50,))]
        for i, p in enumerate(ps):
            if i != 1 or rank == 0:                      # one parameter has no gradient on rank 1
                p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        # a channel-last VM plane (values differ per element, so a wrong flattening order would show): its gradient
        # is flattened in memory order -- a view -- and comes back in the parameter's own layout
        from tensoir_amd.field_model import channel_last, is_channel_last
        cl = torch.nn.Parameter(channel_last(torch.randn(1, 4, 3, 5)))
        base = torch.arange(60, dtype=torch.float32).reshape(1, 4, 3, 5)
        cl.grad = channel_last(base * (rank + 1))
        ps.append(cl)
        n_buckets = tdist.allreduce_gradients(ps, bucket_mb=60 * 4 / (1 << 20))       # 60-float buckets
        want = [sum((r + 1) * (i + 1) for r in range(world)) / world for i in range(len(ps) - 1)]
        want[1] = 2.0 / world                           # only rank 0 contributed (value 1*2)
        ok = all(torch.allclose(p.grad, torch.full_like(p, w)) for p, w in zip(ps[:-1], want))
        ok = ok and torch.allclose(cl.grad, base * sum(r + 1 for r in range(world)) / world) and is_channel_last(cl.grad)
        q.put((rank, ok and n_buckets >= 2))
    finally:
        dist.destroy_process_group()


def test_gloo_gradient_allreduce_buckets():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res)
    idx = torch.cat([tdist.shard_batch(10, r, 3) for r in range(3)])
    assert sorted(idx.tolist()) == list(range(10))


def _spawn(target, world, *args, timeout=240):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, *_ in res) == list(range(world))
    return sorted(res)


def _worker8(rank, world, port, n_rays, chunk, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(4)
        rays, li = torch.randn(n_rays, 6), torch.randint(0, 3, (n_rays, 1), dtype=torch.int32)
        calls = []

        def render(r, l):
            calls.append(int(r.shape[0]))
            return fake_render(r, l)
        got = tdist.render_sharded(render, rays, li, chunk=chunk, tile=tile)
        full = fake_render(rays, li)
        ok = all(torch.equal(got[k], full[k]) for k in got)
        mine = int(tdist.shard_rows(n_rays, rank, world, tile).numel())
        # this rank rendered exactly its shard, in chunks of `chunk` with one ragged tail at most
        ok = ok and sum(calls) == mine and all(c == chunk for c in calls[:-1]) and (not calls or 0 < calls[-1] <= chunk)
        q.put((rank, ok, mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile", [0, 64])
def test_gloo_eight_ranks_ragged_image(tile):
    """SURVEY 8e at the node's rank count: 8 ranks, an image whose ray count is not divisible by 8 x chunk (ragged last chunk
    on some ranks, and with row tiles a last rank with fewer rays than the others), row tiles and interleaved chunk-sized
    tiles -- every rank ends up with the identical full image."""
    n_rays, chunk = 8 * 64 * 3 + 37, 64
    res = _spawn(_worker8, 8, n_rays, chunk, tile)
    assert all(ok for _, ok, _ in res), res
    assert sum(m for *_, m in res) == n_rays
    if tile:
        assert max(m for *_, m in res) - min(m for *_, m in res) <= tile


def _launcher_dp_worker(rank, world, port, q):
    """The launcher's data-parallel step (tensoir_amd.run under torchrun): shard_filter_mask partitions the kept rays,
    LauncherAdam.step() averages the gradients before the update.  The 'model' is a least-squares toy whose per-ray gradient is
    known, so the two-rank run can be compared with ONE process stepping on the concatenated batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensoir_amd import optim
        tdist.LAUNCHER_DP["on"] = True
        assert tdist.launcher_dp() == (rank, world)
        torch.manual_seed(3)
        keep = torch.rand(41) > 0.3                             # what filtering_rays found inside the box (same on every rank)
        mine = tdist.shard_filter_mask(keep, rank, world)
        x, y = torch.randn(41, 5), torch.randn(41, 2)

        def make():
            torch.manual_seed(9)
            return [torch.nn.Parameter(torch.randn(5, 2)), torch.nn.Parameter(torch.zeros(2))]

        def loss_of(ps, rows):
            return ((x[rows] @ ps[0] + ps[1] - y[rows]) ** 2).mean()
        ps = make()
        opt = optim.LauncherAdam([{"params": [ps[0]], "lr": 0.02}, {"params": [ps[1]], "lr": 0.001}], betas=(0.9, 0.99))
        for _ in range(3):
            opt.zero_grad()
            loss_of(ps, mine).backward()
            opt.step()
        # one process, the same three steps on the mean of the per-rank losses (= what averaging the gradients computes)
        tdist.LAUNCHER_DP["on"] = False
        ref = make()
        ropt = torch.optim.Adam([{"params": [ref[0]], "lr": 0.02}, {"params": [ref[1]], "lr": 0.001}], betas=(0.9, 0.99))
        shards = [tdist.shard_filter_mask(keep, r, world) for r in range(world)]
        for _ in range(3):
            ropt.zero_grad()
            (sum(loss_of(ref, s) for s in shards) / world).backward()
            ropt.step()
        ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(ps, ref))
        union = torch.stack(shards).sum(0)
        ok = ok and torch.equal(union.bool(), keep) and int(union.max()) == 1
        ok = ok and max(int(s.sum()) for s in shards) - min(int(s.sum()) for s in shards) <= 1
        q.put((rank, ok, [p.detach().reshape(-1).tolist() for p in ps]))
    finally:
        dist.destroy_process_group()


def test_gloo_launcher_data_parallel_step():
    res = _spawn(_launcher_dp_worker, 2)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]                                # identical parameters on both ranks after the steps


def test_launcher_dp_is_off_without_the_launcher():
    assert tdist.LAUNCHER_DP["on"] is False and tdist.launcher_dp() is None
    m = torch.zeros(7, dtype=torch.bool)
    m[[1, 2, 5, 6]] = True
    assert tdist.shard_filter_mask(m, 0, 2).tolist() == [False, True, False, False, False, True, False]
    assert tdist.shard_filter_mask(m, 1, 2).tolist() == [False, False, True, False, False, False, True]
    assert tdist.shard_filter_mask(m.view(1, 7), 0, 1).shape == (1, 7)

"""world_size-2 gloo test of the scene sharding used by bench.py --gpus N (no data-path collective;
only bookkeeping collectives: max-over-ranks time, result gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dualdiff_amd.parallel import gather_scenes, max_over_ranks, shard_scenes


def test_shard_scenes_partition():
    for n in (1, 2, 5, 8, 13):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_scenes(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def _worker(rank, world, port, n_scenes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_scenes(n_scenes, rank, world)
    # stand-in for "denoise my scenes": every scene's latents are a function of its global index only
    local = torch.stack([torch.full((6, 4, 2, 3), float(i)) for i in range(lo, hi)]) if hi > lo \
        else torch.zeros((0, 6, 4, 2, 3))
    full = gather_scenes(local, n_scenes)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        out.put((full[:, 0, 0, 0, 0].tolist(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_and_timing():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_scenes = 5                                         # ragged: ranks get 3 and 2 scenes
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenes, q)) for r in range(2)]
    for p in procs:
        p.start()
    vals, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert vals == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert t == 2.0                                      # slowest rank


def _cfg_worker(rank, world, port, out):
    from dualdiff_amd.parallel import cfg_all_gather, cfg_pair_groups
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = cfg_pair_groups(world)
    mine = groups[rank // 2]
    # stand-in for "noise prediction of my CFG half": value = 10 * scene + half
    eps_half = torch.full((6, 4, 2, 3), 10.0 * (rank // 2) + (rank % 2))
    eps2 = cfg_all_gather(eps_half, mine)
    # classifier-free guidance on the gathered halves (pipeline_bev_controlnet.py:487-490)
    guided = eps2[0] + 2.0 * (eps2[1] - eps2[0])
    out.put((rank, tuple(eps2.shape), eps2[0, 0, 0, 0, 0].item(), eps2[1, 0, 0, 0, 0].item(), guided[0, 0, 0, 0].item()))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_split_exchange_gloo():
    """CFG split over rank pairs: 4 ranks = 2 scenes x {uncond, cond}; every rank of a pair
    must see [uncond, cond] in that order and compute the same guided prediction."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 4
    procs = [ctx.Process(target=_cfg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape, u, c, g in got:
        scene = rank // 2
        assert shape == (2, 6, 4, 2, 3)
        assert (u, c) == (10.0 * scene, 10.0 * scene + 1.0)
        assert g == u + 2.0 * (c - u)

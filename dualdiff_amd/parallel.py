"""Multi-GPU sharding of the denoising workload (one process per GPU, torch.distributed / RCCL).

The unit of work is a scene (6 views x CFG = 12 view-instances).  Scenes are independent — the
reference itself only ever shards its val set over ranks (`tools/downstream_v3_batched.py:120,157`,
`val_set_gen.py:121`) — so ranks take disjoint scene slices and there is NO data-path collective.
Collectives are used only for bookkeeping: max-over-ranks timing and gathering result tensors.

For single-scene LATENCY the classifier-free-guidance halves can additionally be split over a pair of
GPUs (SURVEY.md §8e "CFG split"): each rank runs the 6 view-instances of its half and the two noise
predictions (67 KB each in bf16) are exchanged once per step — `cfg_all_gather`, the one real
exchange step on that path (pipeline_bev_controlnet.py:487-490 combines the halves).
"""
import torch
import torch.distributed as dist


def shard_scenes(n_scenes, rank, world):
    """Contiguous, balanced slice [lo, hi) of scene indices for `rank` (first ranks take the remainder)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(n_scenes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (the throughput denominator of bench.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scenes(local, n_scenes):
    """All-gather per-rank result tensors (scene-major, dim 0) back into scene order on every rank.
    Ragged shards (n_scenes % world != 0) are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_scenes(n_scenes, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def cfg_pair_groups(world):
    """Process groups [2g, 2g+1] (rank 2g = unconditional half, 2g+1 = conditional half).  Every rank
    must call this (new_group is collective); returns the list of groups."""
    if world % 2:
        raise ValueError("CFG split needs an even number of ranks, got %d" % world)
    return [dist.new_group([2 * g, 2 * g + 1]) for g in range(world // 2)]


def cfg_all_gather(eps_half, group=None):
    """(b*n, 4, h, w) noise prediction of this rank's CFG half -> (2, b*n, 4, h, w) in
    [unconditional, conditional] order = rank order inside the pair group."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("cfg_all_gather needs an initialised process group")
    if dist.get_world_size(group) != 2:
        raise ValueError("a CFG pair group has exactly 2 ranks")
    eps_half = eps_half.contiguous()
    out = [torch.empty_like(eps_half), torch.empty_like(eps_half)]
    dist.all_gather(out, eps_half, group=group)
    return torch.stack(out)

"""Multi-GPU sharding of the denoising workload (one process per GPU, torch.distributed / RCCL).

The unit of work is a scene (6 views x CFG = 12 view-instances).  Scenes are independent — the
reference itself only ever shards its val set over ranks (`tools/downstream_v3_batched.py:120,157`,
`val_set_gen.py:121`) — so ranks take disjoint scene slices and there is NO data-path collective.
Collectives are used only for bookkeeping: max-over-ranks timing and gathering result tensors.
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

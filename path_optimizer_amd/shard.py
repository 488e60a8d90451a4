"""Multi-GPU batch split.  Paths are independent QPs: rank r of W takes the contiguous range of path
ids [r*B/W, (r+1)*B/W) — no data-path collective.  torch.distributed (RCCL on GPUs, gloo on CPU) is used only for the
timing barrier and for reducing a handful of statistics."""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced split: the first `total % world` ranks get one extra path."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(iters_sum: float, unsolved: float, iters_max: float, elapsed: float, device=None):
    """(sum, sum, max, max) over ranks; identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return iters_sum, unsolved, iters_max, elapsed
    s = torch.tensor([iters_sum, unsolved], dtype=torch.float64, device=device)
    m = torch.tensor([iters_max, elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(s[0]), float(s[1]), float(m[0]), float(m[1])


def gather_to_root(t, world: int, rank: int):
    """SURVEY.md §8e collective (1): every rank's [B/G, ...] block to rank 0 (torch.distributed gather = ncclGather-style point-to-point
    sends under RCCL, plain gather under gloo), concatenated in rank order = global path order for the contiguous split.  Returns the
    [B, ...] tensor on rank 0 and None elsewhere; identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist

    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, parts, dst=0)
    return torch.cat(parts, dim=0) if rank == 0 else None

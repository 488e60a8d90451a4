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

"""Frame sharding across ranks (one process per GPU) and the final metric reduction.

Frames are independent (they share only the read-only model), so the data path has no collective:
each rank optimises its own contiguous shard.  The only communication is one small all-reduce of
run statistics at the end (RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""
import torch


def shard_bounds(n_total, world, rank):
    """Contiguous, balanced shard [lo, hi) of n_total frames for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(dist, device, max_stats=(), sum_stats=()):
    """all_reduce(MAX) of max_stats and all_reduce(SUM) of sum_stats; returns two lists of floats.
    `dist` is torch.distributed (initialised) or None for a single process."""
    mx = torch.tensor(list(max_stats), dtype=torch.float64, device=device)
    sm = torch.tensor(list(sum_stats), dtype=torch.float64, device=device)
    if dist is not None:
        if mx.numel():
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if sm.numel():
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return [float(x) for x in mx.cpu()], [float(x) for x in sm.cpu()]

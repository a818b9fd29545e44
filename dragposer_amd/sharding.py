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


def pick_kernel(optimizer, n_total, world):
    """The kernel every rank passes to `LatentOptimizer.optimize(kernel=...)` when ONE batch of n_total frames is cut into `world`
    shards: the library's own choice (dp_auto_kernel) for the LARGEST shard.  kernel="auto" decides per launch from the launch's
    frame count, and the two kernels differ in the last bits of their arithmetic -- left to itself, a rank whose shard falls on the
    other side of the threshold (32 frames per CU) than its neighbour's would compute its frames in other arithmetic than they do.
    What this does NOT promise: that the sharded result equals the one-GPU result of the whole batch bit for bit -- that batch, if it
    is larger than one round of dp_w4, runs dp_w16 on one GPU and (shards being smaller) dp_w4 when sharded; both are held to the same
    reference runs, and a caller that needs the bits to agree pins kernel="w16" (or "w4") on both sides."""
    largest = shard_bounds(n_total, world, 0)[1]
    return optimizer.auto_kernel(largest)


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

"""CPU, world_size 2 over gloo: the N>1 path of the benchmark -- contiguous frame shards with no
data-path collective plus ONE small metric reduction -- gives the same numbers as one process."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from dragposer_amd.sharding import reduce_stats, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 64, 4096, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pick_kernel_is_the_choice_for_the_largest_shard():
    """every rank of a sharded batch gets the same kernel, the library's choice for the LARGEST shard (the GPU counterpart, across
    the real threshold: tests/test_hip_w4.py::test_shards_of_one_batch_run_one_arithmetic)"""
    from dragposer_amd.sharding import pick_kernel

    class Fake:  # LatentOptimizer.auto_kernel on a 256-CU device: dp_w16 beyond 16 frames per CU
        def auto_kernel(self, n):
            return "w16" if n > 4096 else "w4"

    assert pick_kernel(Fake(), 8192, 2) == "w4"      # 4096 + 4096
    assert pick_kernel(Fake(), 8193, 2) == "w16"     # 4097 + 4096: "auto" per rank would have split the arithmetic
    assert pick_kernel(Fake(), 8193, 1) == "w16" and pick_kernel(Fake(), 32768, 8) == "w4" and pick_kernel(Fake(), 32769, 8) == "w16"


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import ref_torch as R
    from oracle.analytic import AnalyticOracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = R.load_golden(os.path.join(ROOT, "tests", "golden", "s1.npz"))
    lo, hi = shard_bounds(len(g["z0"]), world, rank)
    a = [g[k][lo:hi] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    o = AnalyticOracle().optimize(*a, 50)  # stands in for the rank-local HIP launch
    err = np.linalg.norm(o["pos"] - g["pos"][lo:hi], axis=-1) * 1000.0
    mx, sm = reduce_stats(dist, torch.device("cpu"), max_stats=[err.max(), 1.0 + rank], sum_stats=[err.sum(), err.size, hi - lo])
    if rank == 0:
        q.put((mx, sm))
    dist.destroy_process_group()


def test_two_rank_gloo_reduction_equals_single_process():
    from oracle import ref_torch as R
    from oracle.analytic import AnalyticOracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mx, sm = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = R.load_golden(os.path.join(ROOT, "tests", "golden", "s1.npz"))
    a = [g[k] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    o = AnalyticOracle().optimize(*a, 50)
    err = np.linalg.norm(o["pos"] - g["pos"], axis=-1) * 1000.0
    assert mx[0] == pytest.approx(err.max(), rel=1e-12) and mx[1] == 2.0
    assert sm[0] == pytest.approx(err.sum(), rel=1e-9) and sm[1] == err.size and sm[2] == len(g["z0"])

"""GPU, two or more devices (skipped on the one-GPU lease; the driver's 8-GPU node runs it): the frame-sharded path of
DESIGN.md section 7 on real hardware -- two contexts on two devices at once against one context, and the very command the
driver launches for N = 2 (RCCL barriers and the one metric reduction)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


def _need_two():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")


def test_two_devices_shard_one_batch_bit_for_bit():
    """contiguous shards of one batch on cuda:0 and cuda:1, launched back to back (both kernels in flight together: the
    contexts' DeviceGuard selects each device whatever torch's current device is) = the rows of the whole batch on cuda:0"""
    _need_two()
    from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
    from dragposer_amd.sharding import shard_bounds

    b = R.synth_inputs(R.OracleModel(), 4096)
    devs = [torch.device("cuda:0"), torch.device("cuda:1")]
    opts = [LatentOptimizer(device=d) for d in devs]
    whole = {k: v.cpu().numpy() for k, v in opts[0].optimize(**to_device_batch(b, devs[0]), n_iter=50).items()}
    parts = []
    torch.cuda.set_device(0)  # (deliberately not the second context's device)
    for r, (o, d) in enumerate(zip(opts, devs)):
        lo, hi = shard_bounds(4096, 2, r)
        parts.append((lo, hi, o.optimize(**to_device_batch({k: b[k][lo:hi] for k in KEYS}, d), n_iter=50)))
    for d in devs:
        torch.cuda.synchronize(d)
    for lo, hi, out in parts:
        for k, v in out.items():
            np.testing.assert_array_equal(v.cpu().numpy(), whole[k][lo:hi], err_msg=k)


@pytest.mark.parametrize("sizing", [["--frames", "4096"], ["--total-frames", "8192"]])
def test_bench_under_torch_distributed_run_with_rccl(sizing):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` over RCCL, as the driver launches it (a child
    process: nothing here re-execs a process that has touched the GPU)."""
    _need_two()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"] + sizing
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == ("weak" if sizing[0] == "--frames" else "strong")
    assert line["config"]["frames_total"] == 8192 and line["parity_p99_mm_vs_oracle"] < 0.05
    assert p.stderr.count("kernel_ms") == 2  # every rank reported its own launch time
    if sizing[0] == "--frames":  # the weak pass carries north_star's strong-scaling numbers with it (bench.py: "strong")
        assert [b["frames_total"] for b in line["strong"]] == [4096, 8192] and all(len(b["kernel_ms_per_rank"]) == 2 for b in line["strong"])


def test_rccl_process_group_of_one_runs_the_metric_reduction():
    """ONE GPU is enough for this one (it does not skip): the `nccl` branch of bench.py -- torch.distributed over RCCL, initialised with a
    device id, the reduction of dragposer_amd.sharding.reduce_stats on DEVICE tensors, the barrier -- executed in-process with a world of
    one rank.  No data moves between GPUs; what is exercised is that RCCL initialises in this process next to the library's HIP runtime and
    that the collectives the N > 1 path issues are the ones the backend accepts."""
    import torch.distributed as dist

    from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
    from dragposer_amd.sharding import pick_kernel, reduce_stats, shard_bounds

    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        opt = LatentOptimizer(device=dev)
        b = R.synth_inputs(R.OracleModel(), 256)
        lo, hi = shard_bounds(256, 1, 0)
        out = opt.optimize(**to_device_batch({k: b[k][lo:hi] for k in KEYS}, dev), n_iter=10, kernel=pick_kernel(opt, 256, 1))
        dist.barrier()
        torch.cuda.synchronize()
        err = float(out["loss"].sum())
        (mx0, mx1), (sm0, sm1) = reduce_stats(dist, dev, max_stats=[err, 2.5], sum_stats=[err, 1.0])
        assert mx0 == pytest.approx(err) and mx1 == 2.5 and sm0 == pytest.approx(err) and sm1 == 1.0
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """the driver's N = 2 command with gloo in place of RCCL and both ranks on cuda:0 (the one-GPU lease has no second device): the whole
    N > 1 code path of bench.py -- shards, pinned kernel, barriers, the metric reduction, the strong block with every rank's kernel time"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--backend", "gloo",
           "--all-ranks-on-device0", "--precondition-ms", "5"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_total"] == 8192 and line["parity_p99_mm_vs_oracle"] < 0.05
    st = line["strong"]
    assert [b["frames_total"] for b in st] == [4096, 8192] and [b["frames_per_gpu"] for b in st] == [2048, 4096]
    assert all(b["kernel"] == "w4" and len(b["kernel_ms_per_rank"]) == 2 and min(b["kernel_ms_per_rank"]) > 0 for b in st)
    assert 0.5 < line["roofline"]["sclk_ghz"] < 2.6

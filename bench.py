#!/usr/bin/env python3
"""Benchmark of the hot path: frames/s of the per-frame latent optimisation (6 trackers, 50 Adam
iterations per frame) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one dp_optimize launch: all 50 iterations of all frames)
over the rank's batch of synthetic frames (recipe S of SURVEY.md 8d; targets are produced on the
device with dp_forward).  Inputs are resident in HBM before the timed region.  Frames shard
across ranks with no data-path collective; RCCL is used only for the barriers and the final metric
reduction.  Rank 0 prints ONE JSON line.

Two ways to size the job:
  --frames F        F frames PER GPU (default 4096: at N = 1 exactly BASELINE's workload) -> "scaling": "weak"
  --total-frames T  ONE batch of T frames cut into contiguous shards (dragposer_amd.sharding) -> "scaling": "strong"
                    (north_star's 4096-frame batch at 1/2/4/8 GPUs: --total-frames 4096; BASELINE config 3: 8192)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_FRAME_ITER = 35520  # folded decoder 24->40->60->92, forward + backward-to-input
PEAK_F32_MFMA = 157.3e12     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak per GPU
TRACK6 = [0, 3, 7, 13, 17, 21]
W6 = {0: (10.0, 10.0), 3: (5.0, 0.01), 7: (5.0, 0.01), 13: (5.0, 0.01), 17: (5.0, 0.01), 21: (5.0, 0.01)}


def synth_on_device(opt, B, seed, device, lo=0, hi=None, mixed=False):
    """Recipe S, targets = FK(decode(Zs), CR) computed by the product's own forward kernel.  [lo, hi): this rank's
    shard of the B-frame batch (every rank draws the same batch and keeps its rows).  mixed: recipe S4, every
    frame tracks its own 1..6 of the six joints."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    hi = B if hi is None else hi
    Ball = B
    Zs = (torch.randn(B, 24, generator=g) * 0.3)[lo:hi]
    Z0 = (torch.randn(B, 24, generator=g) * 0.3)[lo:hi]
    ZT = Z0 + (0.05 * torch.randn(B, 24, generator=g))[lo:hi]
    CR = torch.randn(B, 4, generator=g)[lo:hi]
    CR = CR / torch.linalg.norm(CR, dim=-1, keepdim=True)
    B = hi - lo
    w = torch.zeros(B, 22, 2)
    tracked = torch.zeros(B, 22, dtype=torch.uint8)
    if mixed:  # recipe S4's own draws (oracle.ref_torch.synth_inputs and the goldens draw the same): Eb, then a randperm per frame
        Eb = torch.randint(1, 7, (Ball,), generator=g)
        keep = torch.zeros(Ball, 6, dtype=torch.bool)
        for b in range(Ball):
            keep[b, torch.randperm(6, generator=g)[: int(Eb[b])]] = True
        keep = keep[lo:hi]
        for k, j in enumerate(TRACK6):
            tracked[:, j] = keep[:, k].to(torch.uint8)
            w[:, j] = torch.tensor(W6[j]) * keep[:, k, None]
    for j, wj in (() if mixed else W6.items()):
        w[:, j] = torch.tensor(wj)
        tracked[:, j] = 1
    Zs, Z0, ZT, CR, w, tracked = (t.to(device).contiguous() for t in (Zs, Z0, ZT, CR, w, tracked))
    fk = opt.forward(Zs, CR, outputs=("pos", "rot"))
    m = tracked.to(torch.float32).unsqueeze(-1)
    return dict(z0=Z0, z_tgt=ZT, cur_rot=CR, tgt_pos=(fk["pos"] * m).contiguous(), tgt_rot=(fk["rot"] * m).contiguous(),
                w=w, tracked=tracked)


PMC_FILE = "profiles/r06_pmc_per_launch.json"


def pmc_traffic_bytes(frames, iters):
    """HBM bytes per launch, NOT measured in this run: read from the committed rocprofv3 PMC passes of this very command
    (separate --pmc runs, tools/collect_profiles.sh).  FETCH_SIZE under-counts reads 2x on gfx950
    (MI355X_MICROARCH.md, HBM); both counters are in KB.  Only valid for the configuration the counters were collected on."""
    path = os.path.join(ROOT, PMC_FILE)
    if frames != 4096 or iters != 50 or not os.path.exists(path):
        return None
    with open(path) as f:
        c = json.load(f)
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


def measure_traffic(argv_tail, kernel_substr):
    """HBM bytes per launch of the dominant kernel MEASURED BY THIS COMMAND: two child runs of this very script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (the TCC counters do not fit one pass; the program directly after
    `--`), same workload, 6 launches each; counters summed over the XCDs per dispatch, averaged over the full-size launches (the
    first dispatch of the kernel is the forward-only launch that makes the synthetic targets).  Corrected as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests at 64 bytes), both in KB.  Called BEFORE this process touches
    the GPU (a process that has initialised HIP must not start other programs).  Returns (bytes, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import Counter, defaultdict

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    got = {}
    work = tempfile.mkdtemp(prefix="dp_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--traffic", "none", "--precondition-ms", "0"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            # (its own session: on a timeout the whole group goes -- rocprofv3 AND the profiled child, which would otherwise keep the GPU busy
            #  while this process takes its timings)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=180)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f"rocprofv3 --pmc {counter} pass timed out (killed with its process group)"
            if rc != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {rc})"
            per = defaultdict(float)
            rows = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [x for x in csv.DictReader(open(path)) if kernel_substr in x["Kernel_Name"] and x["Counter_Name"] == counter]
            if not rows:
                return None, f"no {counter} rows for {kernel_substr}"
            grid = Counter(x["Grid_Size"] for x in rows).most_common(1)[0][0]
            first = min(int(x["Dispatch_Id"]) for x in rows)
            for x in rows:
                if x["Grid_Size"] == grid and int(x["Dispatch_Id"]) != first:
                    per[x["Dispatch_Id"]] += float(x["Counter_Value"])
            if not per:
                return None, f"no full-size dispatch in the {counter} pass"
            got[counter] = (sum(per.values()) / len(per), len(per))
    except Exception as e:  # (a profiler that is absent, refused or broken must not take the benchmark down)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    note = (f"measured by this command: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same workload as child processes "
            f"(means over {got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches); bytes = 2 x FETCH_SIZE (gfx950 read under-count) + WRITE_SIZE, both KB "
            f"-> {got['FETCH_SIZE'][0]:.1f} KB fetched (uncorrected), {got['WRITE_SIZE'][0]:.1f} KB written")
    return (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0, note


def usable_cpus():
    """the cores this process may actually run on: the affinity mask and the cgroup quota, not os.cpu_count() (the GPU box shows all 256 of
    the host's and hands a one-GPU job 16 of them)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(batch_np, n_iter, budget_s=20.0):
    """The oracle ("port") timed on this host, as SURVEY.md 8(d) words it: (i) the reference-shaped B = 1 autograd + torch.optim.Adam loop on a
    bounded sample of the same workload, with 1 thread (the headline `value`) and with every usable core; (ii) the same mathematics batched
    over the 4096 frames on every usable core; plus the scalar C restatement on one core.  `batch_np`: the first frames of the GPU's batch."""
    from oracle import ref_torch as R
    from oracle.analytic import AnalyticOracle

    model = R.OracleModel()
    ncpu = usable_cpus()
    torch.set_num_threads(1)
    args = [batch_np[k] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    R.optimize_reference_shaped(model, *[a[:1] for a in args], n_iter)  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 64 and time.perf_counter() - t0 < budget_s:
        R.optimize_reference_shaped(model, *[a[n:n + 1] for a in args], n_iter)
        n += 1
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit="frames/s", cores=1, kind="port",
               sample=f"{n} frames x {n_iter} iters, batch-1 PyTorch autograd + torch.optim.Adam loop (as the reference runs), 1 thread")
    # (i) again with every usable core: torch's intra-op threads on 24..92-wide tensors (slower than one thread; stated, as SURVEY 8d asks)
    torch.set_num_threads(ncpu)
    R.optimize_reference_shaped(model, *[a[:1] for a in args], n_iter)
    n, t0 = 0, time.perf_counter()
    while n < 16 and time.perf_counter() - t0 < 6.0:
        R.optimize_reference_shaped(model, *[a[n:n + 1] for a in args], n_iter)
        n += 1
    out["reference_shaped_all_cores_frames_per_s"] = n / (time.perf_counter() - t0)
    out["reference_shaped_all_cores_sample"] = f"{n} frames, torch.set_num_threads({ncpu})"
    # the scalar C restatement of the same math (analytic backward), one core
    A = AnalyticOracle(precision="f32")
    nb = min(512, len(args[0]))
    t0 = time.perf_counter()
    A.optimize(*[a[:nb] for a in args], n_iter)
    out["c_port_1core_frames_per_s"] = nb / (time.perf_counter() - t0)
    # (ii) the torch restatement batched over frames (what a CPU user gets by vectorising the reference), B = 4096 on every usable core
    nb = min(4096, len(args[0]))
    R.optimize(model, *[a[:64] for a in args], 2)  # warm-up
    t0 = time.perf_counter()
    R.optimize(model, *[a[:nb] for a in args], n_iter)
    out["batched_torch_frames_per_s"] = nb / (time.perf_counter() - t0)
    out["batched_torch_frames"] = nb
    out["batched_torch_threads"] = ncpu
    torch.set_num_threads(1)
    out["host_cpus"] = os.cpu_count()
    out["usable_cpus"] = ncpu
    out["cpu_model"] = cpu_model_name()
    return out


PRECONDITION_MS = 60.0  # of the measured launch, back to back, before the warm-up steps (see timed_pass)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU (weak scaling)")
    ap.add_argument("--total-frames", type=int, default=0, help="one batch of this many frames sharded over the GPUs (strong scaling)")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--config", default="s1", choices=["s1", "s4"],
                    help="s1: BASELINE's headline workload (6 trackers, fp32); s4: BASELINE config 5 (1-6 trackers per frame, bf16-rounded decoder weights)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "w4", "w16"], help="include/dragposer.h: DP_KERNEL_*")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "file", "none"],
                    help="roofline.traffic: measure = two rocprofv3 --pmc child passes of this workload (N = 1 only, adds about a minute); file = the committed "
                         "PMC passes (profiles/); auto = measure when rocprofv3 is there and N = 1, else file")
    ap.add_argument("--precondition-ms", type=float, default=PRECONDITION_MS,
                    help="GPU-milliseconds of the measured launch run back to back before the warm-up steps, so that the timed steps run at the shader "
                         "clock a busy GPU holds (DVFS: 2.08 GHz from idle, 2.39 GHz after ~25 ms of load, profiles/r05_clock_ramp.txt); 0 = off")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling block (north_star's 4096-frame batch, config 3's 8192) after the main pass")
    ap.add_argument("--lib", default=None, help="diagnostic builds (tools/ab.sh): path of the library to load instead of dragposer_amd/lib/libdragposer_hip.so")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal on a 1-GPU box (with --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    traffic, traffic_note = None, None
    under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if args.traffic == "auto" and under_profiler:
        args.traffic = "file"  # (this process is itself being profiled: no profiler inside the profiler)
    if world == 1 and args.traffic in ("auto", "measure"):  # before anything touches the GPU in this process
        tail = ["--frames", str(args.frames), "--iters", str(args.iters), "--config", args.config, "--kernel", args.kernel]
        if args.total_frames > 0:
            tail += ["--total-frames", str(args.total_frames)]
        traffic, traffic_note = measure_traffic(tail, "dp_w")
        if traffic is None:
            print(f"[bench] HBM traffic not measured ({traffic_note}); falling back to the committed PMC passes", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product has no CPU path")
    if args.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.lib:
        from dragposer_amd import _lib

        _lib.LIB_PATH = os.path.abspath(args.lib)
    from dragposer_amd.optimizer import LatentOptimizer

    from dragposer_amd.sharding import pick_kernel, reduce_stats, shard_bounds

    s4 = args.config == "s4"
    opt = LatentOptimizer(device=device, weight_dtype="bf16" if s4 else "fp32")
    N = args.iters
    if args.total_frames > 0:  # strong scaling: contiguous shards of ONE batch
        lo, hi = shard_bounds(args.total_frames, world, rank)
        if hi <= lo:
            raise SystemExit(f"--total-frames {args.total_frames} leaves rank {rank} of {world} without frames")
        batch = synth_on_device(opt, args.total_frames, 1234, device, lo, hi, mixed=s4)
        B, total_per_step = hi - lo, args.total_frames
        if args.kernel == "auto":  # one batch, several launches: every shard in the same arithmetic (sharding.pick_kernel)
            args.kernel = pick_kernel(opt, args.total_frames, world)
    else:  # weak scaling: every rank its own batch of --frames
        batch = synth_on_device(opt, args.frames, 1234 + rank, device, mixed=s4)
        B, total_per_step = args.frames, args.frames * world
    names = ("z", "z_pre", "pose", "disp", "world_disp", "world_rot", "pos", "loss", "iters", "status", "clock")
    from dragposer_amd.optimizer import sclk_ghz

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_pass(batch, kernel, steps, warmup, from_idle=False):
        """W untimed warm-up steps, then EXACTLY `steps` timed steps between barrier + synchronize on both sides.  Ahead of the warm-up the
        device is PRECONDITIONED: the same launch back to back for --precondition-ms of GPU time.  Why: the shader clock is a DVFS state --
        2.08 GHz on the first launches after idle, 2.14 after ~3 ms, the 2.38-2.40 GHz plateau after ~25 ms of sustained load, back down
        after a 10 ms gap (profiles/r05_clock_ramp.txt) -- and 20 timed steps are 2.7 ms: without it the line reports the idle clock.
        The semantics of `steps` / `warmup` are untouched; the line states the preconditioning (config.precondition_ms) and the clock the
        last timed launch ran at (roofline.sclk_ghz).  Returns (wall seconds, kernel ms per step by HIP events, GHz, results)."""
        out = opt.optimize(**batch, n_iter=N, outputs=names, kernel=kernel)
        torch.cuda.synchronize()
        # (the step as a caller that reuses its buffers issues it: arguments marshalled once, LatentOptimizer.plan -- one dp_optimize call per step)
        step = opt.plan(**batch, n_iter=N, outputs=names, out=out, kernel=kernel)
        # what a caller who submits ONE batch to an idle GPU gets: after 150 ms without work, one launch between two events (three times: the median), with the
        # clock it ran at (the DVFS floor, 2.08-2.10 GHz).  Reported beside the steady figures (roofline.*_from_idle); never part of `value`.
        idle = None
        if from_idle:
            tries = []
            for _ in range(3):  # (the median of three: the first launch after a pause also pays whatever the runtime let go idle)
                time.sleep(0.15)
                i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                i0.record()
                step()
                i1.record()
                torch.cuda.synchronize()
                tries.append((i0.elapsed_time(i1), sclk_ghz(out["clock"])))
            idle = sorted(tries)[1]
        if args.precondition_ms > 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                step()
            e1.record()
            torch.cuda.synchronize()
            per = max(e0.elapsed_time(e1) / 3.0, 1e-3)
            for _ in range(min(20000, int(args.precondition_ms / per) + 1)):  # (no synchronisation from here to the timed region's own)
                step()
        for _ in range(warmup):
            step()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()  # torch's current stream == the stream the kernel is launched on
        for _ in range(steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0  # this rank's K steps, device work complete; the job's time is the MAX over the ranks (reduce_stats below) --
        barrier()                      # which is what the closing barrier would make every rank read anyway, plus the barrier's own latency
        return dt, ev0.elapsed_time(ev1) / steps, sclk_ghz(out["clock"]), out, idle

    dt, kern_ms, sclk, out, idle = timed_pass(batch, args.kernel, args.steps, args.warmup, from_idle=True)
    fpb, tpb, lds_bytes = opt.kernel_geometry()

    # ---- parity on a sample (rank-local), reduced with the timing
    err_mm = float("nan")
    if not args.no_parity:
        from oracle.analytic import AnalyticOracle

        nb = min(256, B)
        cpu = {k: v[:nb].cpu().numpy() for k, v in batch.items()}
        ref = AnalyticOracle(precision="f32", weight_rounding="bf16" if s4 else "none").optimize(cpu["z0"], cpu["z_tgt"], cpu["cur_rot"], cpu["tgt_pos"],
                                                       cpu["tgt_rot"], cpu["w"], cpu["tracked"], N)
        e = np.linalg.norm(out["pos"][:nb].cpu().numpy() - ref["pos"], axis=-1) * 1000.0
        err_mm = float(np.percentile(e, 99))

    # the only collective of the job: one MAX-reduction of three doubles (RCCL over xGMI; gloo in the rehearsal)
    kern_ms_rank = kern_ms
    (dt, kern_ms, err_mm), _ = reduce_stats(dist, device, max_stats=[dt, kern_ms, err_mm if err_mm == err_mm else -1.0])
    if world > 1:  # every rank's own launch time, on stderr (rank 0's JSON line carries the maximum)
        print(f"[rank {rank}] kernel_ms {kern_ms_rank:.4f} frames {B}", file=sys.stderr, flush=True)

    # ---- N > 1: the numbers north_star names, in the same process group, without the driver changing its command: ONE 4096-frame batch
    #      (north_star) and ONE 8192-frame batch (BASELINE config 3) cut into contiguous shards, every shard in the kernel the largest one
    #      gets (sharding.pick_kernel).  Expectation stated in DESIGN.md section 8: flat -- one GPU already runs 4096 frames at one wave per
    #      SIMD, fewer frames per GPU leave SIMDs idle without shortening any wave's work.
    strong = []
    if world > 1 and not args.no_strong and args.total_frames == 0:
        for total in (4096, 8192):
            lo, hi = shard_bounds(total, world, rank)
            kern = pick_kernel(opt, total, world)
            sb = synth_on_device(opt, total, 1234, device, lo, hi, mixed=s4)
            sdt, sk, sclk_s, _, _ = timed_pass(sb, kern, args.steps, args.warmup)
            # every rank's own launch time: a one-hot vector summed over the ranks (the same small all-reduce as everything else here)
            (sdt, sk_max), per_rank = reduce_stats(dist, device, max_stats=[sdt, sk], sum_stats=[sk if r == rank else 0.0 for r in range(world)])
            sfpb = opt.kernel_geometry()[0]
            strong.append({"frames_total": total, "frames_per_gpu": shard_bounds(total, world, 0)[1], "kernel": kern,
                           "value": total * args.steps / sdt, "unit": "frames/s", "ms_per_step": sdt / args.steps * 1e3,
                           "kernel_ms_max": sk_max, "kernel_ms_per_rank": per_rank, "sclk_ghz_rank0": sclk_s,
                           "workgroup_frames": sfpb, "scaling": "strong"})

    if rank == 0:
        value = total_per_step * args.steps / dt
        # roofline of the dominant kernel: the slowest rank's launch (the largest shard; its frames set the FLOP count)
        Bk = -(-total_per_step // world)
        achieved = Bk * N * FLOP_PER_FRAME_ITER / (kern_ms * 1e-3)
        res = {
            "metric": "frames/sec (6 trackers, 50 iters/frame)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_frames > 0 else "weak",
            "vs_baseline": None,
            # dp_w16: every fp32 operand as the exact sum of three bf16 terms, six term products per block on the bf16 MFMA, fp32 accumulate
            "dtype": ((("bf16-rounded weights" if s4 else "f32 weights") + " and f32 activations, each as three bf16 terms on the bf16 MFMA (f32-equivalent), f32 accumulate")
                      if fpb >= 64 else ("f32 (bf16-rounded weights)" if s4 else "f32")),
            "data": "synthetic",
            "config": {"workload": ((f"{'S4' if s4 else 'S1'}: ONE batch of {total_per_step} synthetic frames" if args.total_frames > 0 else
                                     f"{'S4' if s4 else 'S1'}: {B} synthetic frames per GPU") +
                                    (f", 1-6 of the trackers [0,3,7,13,17,21] per frame (masked loss), bf16-rounded decoder weights, {N} Adam iters/frame (BASELINE config 5)"
                                     if s4 else f", 6 trackers [0,3,7,13,17,21], {N} Adam iters/frame, fp32 (BASELINE north_star: 4096-frame batch)")),
                       "frames_per_gpu": Bk, "frames_total": total_per_step, "iters": N, "precondition_ms": args.precondition_ms,
                       "parallelism": f"frames sharded x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA,
                         "traffic": traffic if traffic is not None else (pmc_traffic_bytes(Bk, N) if args.traffic != "none" else None),
                         "traffic_note": (traffic_note if traffic is not None else
                                          "traffic not collected (--traffic none)" if args.traffic == "none" else
                                          f"not measured in this run: HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE of the committed PMC passes ({PMC_FILE}, same command)"
                                          if pmc_traffic_bytes(Bk, N) is not None else "no committed counters for this configuration (they exist for 4096 frames x 50 iterations)")
                                         + f"; algorithmic {Bk * 2326:.3g}",
                         # the roofline's 157.3 TF is the fp32 matrix rate AT 2.4 GHz; the chip holds less than that with every SIMD on the matrix pipe
                         # (profiles/r05_clock_ramp.txt).  frac above stays against the full 157.3; this is the same work against the clock actually held
                         "sclk_ghz": sclk, "frac_at_held_clock": achieved / (PEAK_F32_MFMA * sclk / 2.4) if sclk == sclk and sclk > 0 else None,
                         # the same launch from an idle GPU (>= 150 ms without work, ONE launch, rank 0's): what a caller who submits one batch gets
                         "kernel_ms_from_idle": idle[0] if idle else None, "sclk_ghz_from_idle": idle[1] if idle else None,
                         "frac_from_idle": (Bk * N * FLOP_PER_FRAME_ITER / (idle[0] * 1e-3) / PEAK_F32_MFMA) if idle else None,
                         "kernel": {16: "dp_w4_kernel<4, false>", 64: "dp_w16_kernel<4, 1>", 128: "dp_w16_kernel<8, 2>"}.get(fpb, "?"), "kernel_ms": kern_ms,
                         "frac_of_bf16_mfma_peak_2.5PF": (achieved / 2.5e15 if fpb >= 64 else None),
                         "workgroup": {"frames": fpb, "threads": tpb, "lds_bytes": lds_bytes},
                         "flop_per_launch": Bk * N * FLOP_PER_FRAME_ITER,
                         "hbm_algorithmic_GBps": Bk * 2326 / (kern_ms * 1e-3) / 1e9},
            "parity_p99_mm_vs_oracle": err_mm,  # 256 frames x 22 joints vs the C oracle (fp32)
        }
        if strong:
            res["strong"] = strong
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            cpu = {k: v[:4096].cpu().numpy() for k, v in batch.items()}
            res["cpu_baseline"] = cpu_baseline(cpu, N)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

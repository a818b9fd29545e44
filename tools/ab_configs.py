#!/usr/bin/env python3
"""A/B of library builds on BASELINE's tracker configurations (GPU box): kernel time by HIP events of the 4096-frame launch under the 6-tracker
set x 50 iterations (S1), the 3-tracker set x 100 (S3, config 4) and the reference's 4-tracker set x 50, at the steady shader clock (each
measurement preceded by 60 ms of the same launch), interleaved rounds.  Each library in its own child process (one library per process).
usage: python tools/ab_configs.py LIB [LIB ...]   (LIB = path of a libdragposer_hip.so build)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"S1 6 trackers x 50": dict(trackers=6, n_iter=50, lam=0.02), "S3 3 trackers x 100": dict(trackers=3, n_iter=100, lam=0.15),
           "4 trackers x 50": dict(trackers=4, n_iter=50, lam=0.125), "3 trackers x 50": dict(trackers=3, n_iter=50, lam=0.15),
           "4 trackers x 100": dict(trackers=4, n_iter=100, lam=0.125),
           "S1 x 50, while-condition kernel": dict(trackers=6, n_iter=50, lam=0.02, early=True), "S1 1024 frames x 50": dict(trackers=6, n_iter=50, lam=0.02, frames=1024)}


def child(lib):
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from dragposer_amd import _lib

    _lib.LIB_PATH = os.path.abspath(lib)
    from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
    from oracle import ref_torch as R

    dev = torch.device("cuda:0")
    opt = LatentOptimizer(device=dev)
    res = {}
    for name, c in CONFIGS.items():
        if c["trackers"] == 4:  # pelvis + head + hands (python/config/4_trackers_config.json): the 6-tracker recipe with the feet's trackers off
            b = R.synth_inputs(R.OracleModel(), 4096, trackers=6)
            for j in (3, 7):
                b["tracked"][:, j] = 0; b["w"][:, j] = 0; b["tgt_pos"][:, j] = 0; b["tgt_rot"][:, j] = 0
        else:
            b = R.synth_inputs(R.OracleModel(), c.get("frames", 4096), trackers=c["trackers"])
        d = to_device_batch(b, dev)
        names = ("z", "z_pre", "pose", "disp", "world_disp", "world_rot", "pos", "loss", "iters")
        kw = dict(stop_eps_pos=1e-30, stop_eps_rot=1e-30, min_loss_incr=-1e30) if c.get("early") else {}  # (the while-condition never fails: all iterations, through the early-stop instantiation)
        out = opt.optimize(**d, n_iter=c["n_iter"], lambda_tmp=c["lam"], outputs=names, kernel="w4", **kw)
        for _ in range(int(60.0 / (0.14 * c["n_iter"] / 50)) + 1):
            opt.optimize(**d, n_iter=c["n_iter"], lambda_tmp=c["lam"], outputs=names, out=out, kernel="w4", **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            opt.optimize(**d, n_iter=c["n_iter"], lambda_tmp=c["lam"], outputs=names, out=out, kernel="w4", **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 50
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    libs = sys.argv[1:]
    got = {l: {c: [] for c in CONFIGS} for l in libs}
    for rep in range(int(os.environ.get("REPS", "3"))):
        for l in libs:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True, cwd=ROOT)
            line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
            if not line:
                print(l, "FAILED", p.stderr[-500:])
                continue
            for c, v in json.loads(line[-1][7:]).items():
                got[l][c].append(v)
    for c in CONFIGS:
        for l in libs:
            v = got[l][c]
            if v:
                print(f"{c:22s} {l:40s} median {statistics.median(v):.5f} ms  (min {min(v):.5f}, rounds {[round(x, 5) for x in v]})")

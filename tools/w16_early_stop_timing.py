#!/usr/bin/env python3
"""Batches with the reference's per-frame while-condition (early stop) through both optimise kernels: kernel time by HIP events.
Usage: tools/w16_early_stop_timing.py > gpurun_out/w16_early_stop.txt"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import ref_torch as R
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
m = R.OracleModel()
base = R.synth_inputs(m, 4096, seed=9)
kw = dict(n_iter=50, lambda_tmp=0.02, stop_eps_pos=1e-4, stop_eps_rot=1e-2, min_loss_incr=1e-5)
for B in (4096, 8192, 16384, 65536):
    b = {k: np.concatenate([v] * (B // 4096)) for k, v in base.items()}
    d = to_device_batch(b, dev)
    row = []
    for kern in ("w4", "w16"):
        out = opt.optimize(**d, kernel=kern, outputs=("z", "pos", "loss", "iters"), **kw)
        for _ in range(2): opt.optimize(**d, kernel=kern, outputs=("z", "pos", "loss", "iters"), out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): opt.optimize(**d, kernel=kern, outputs=("z", "pos", "loss", "iters"), out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row.append(f"{kern} {ms:.4f} ms {B / ms / 1e3:.2f} M frames/s (mean {out['iters'].float().mean().item():.1f} iterations)")
    print(f"early stop (1e-4, 1e-2, 1e-5), max 50 iterations, {B} frames: " + " | ".join(row), flush=True)

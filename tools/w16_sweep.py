#!/usr/bin/env python3
"""Batch sweep of the two optimise kernels on BASELINE config 5's workload (mixed 1-6 trackers per frame, bf16-rounded
weights) and on S1 (6 trackers, fp32 weights): kernel time by HIP events, frames/s, fraction of the fp32 / bf16 MFMA peaks.
Usage: tools/w16_sweep.py [sizes...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536, 131072]
dev = torch.device("cuda:0")
opts = {"bf16": LatentOptimizer(device=dev, weight_dtype="bf16"), "fp32": LatentOptimizer(device=dev)}
m = R.OracleModel()
base = {"bf16": R.synth_inputs(m, 4096, mixed=True), "fp32": R.synth_inputs(m, 4096)}
for wd in ("bf16", "fp32"):  # throw-away launches: the first timed loop of a process has shown a one-off 90 ms stall
    d = to_device_batch(base[wd], dev)
    for kern in ("w4", "w16"):
        for _ in range(5):
            opts[wd].optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"))
torch.cuda.synchronize()
for wd in ("bf16", "fp32"):
    for B in sizes:
        b = {k: np.concatenate([v] * ((B + 4095) // 4096))[:B] for k, v in base[wd].items()}
        d = to_device_batch(b, dev)
        row = []
        for kern in ("w4", "w16"):
            o = opts[wd]
            out = o.optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"))
            for _ in range(2):
                o.optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"), out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 if B <= 32768 else 4
            e0.record()
            for _ in range(reps):
                o.optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"), out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = B * 50 * 35520 / (ms * 1e-3)
            row.append(f"{kern} {ms:8.4f} ms {B / ms / 1e3:7.2f} M frames/s frac(157.3 TF) {fl / 157.3e12:.3f}" + (f" frac(2.5 PF) {fl / 2.5e15:.4f}" if kern == "w16" else ""))
        print(f"{'config 5: mixed trackers, bf16 weights' if wd == 'bf16' else 'S1: 6 trackers, fp32 weights':40s} {B:7d} frames | " + " | ".join(row), flush=True)

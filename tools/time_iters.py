#!/usr/bin/env python3
"""Diagnostic: kernel time (HIP events) of one dp_optimize launch against the iteration count -- separates the per-launch
cost (staging, per-frame setup, epilogue) from the per-iteration cost.  Usage: tools/time_iters.py [frames]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
d = to_device_batch(R.synth_inputs(R.OracleModel(), B), dev)
out = opt.allocate_outputs(B)
res = {}
for n in (1, 2, 10, 50, 100):
    for _ in range(3):
        opt.optimize(**d, n_iter=n, out=out)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); opt.optimize(**d, n_iter=n, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    res[n] = np.median(ts)
    print(f"B={B} n_iter={n:4d}: {res[n]:8.1f} us")
per = (res[100] - res[50]) / 50
print(f"per iteration {per:.3f} us; launch + setup + epilogue {res[50] - 50 * per:.1f} us (of {res[50]:.1f} at 50 iterations)")

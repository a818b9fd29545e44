#!/usr/bin/env python3
"""Runs only launches of one optimise kernel (for rocprofv3): tools/w16_run.py KERNEL FRAMES [REPS] [WEIGHTS]"""
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

kern, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wd = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dev = torch.device("cuda:0")
o = LatentOptimizer(device=dev, weight_dtype=wd)
base = R.synth_inputs(R.OracleModel(), min(B, 4096), mixed=(wd == "bf16"))
b = {k: np.concatenate([v] * (B // 4096)) if B >= 4096 else v for k, v in base.items()}
d = to_device_batch(b, dev)
out = o.optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"))
for _ in range(reps):
    o.optimize(**d, n_iter=50, kernel=kern, outputs=("z", "pos", "loss"), out=out)
torch.cuda.synchronize()
print("done", kern, B, reps)

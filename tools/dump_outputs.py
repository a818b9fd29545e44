#!/usr/bin/env python3
"""Diagnostic: run the S1 and S3 workloads through the library DRAGPOSER_LIB names and save z / pos / loss, so that two builds
can be compared bit for bit ([KERNEL=w16] [FRAMES=n] tools/dump_outputs.py OUT.npz; then np.array_equal on the two files)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
out = {}
for name, kw, n_iter, lam in (("s1", dict(), 50, 0.02), ("s3", dict(trackers=3), 100, 0.15), ("s4", dict(mixed=True), 50, 0.02)):
    b = R.synth_inputs(R.OracleModel(), int(os.environ.get("FRAMES", "2048")), **kw)
    o = opt.optimize(**to_device_batch(b, dev), n_iter=n_iter, lambda_tmp=lam, kernel=os.environ.get("KERNEL", "auto"))
    for k in ("z", "pos", "loss"):
        out[f"{name}_{k}"] = o[k].cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])

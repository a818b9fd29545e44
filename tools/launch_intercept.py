#!/usr/bin/env python3
"""Diagnostic (GPU box): kernel time of the headline launch against n_iter AT THE STEADY CLOCK (each size preceded by ~80 ms of itself; tools/time_iters.py
times from idle) -- slope = cycles per iteration of the product kernel, intercept = set-up + epilogue + what stands between two launches.
Usage: tools/launch_intercept.py LIB"""
# kernel time of the headline launch against n_iter (steady clock): slope = cycles per iteration, intercept = set-up + epilogue
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from dragposer_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch, sclk_ghz
from oracle import ref_torch as R
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
d = to_device_batch(R.synth_inputs(R.OracleModel(), 4096), dev)
names = ("z", "z_pre", "pose", "disp", "world_disp", "world_rot", "pos", "loss", "iters", "status", "clock")
res = {}
for N in (1, 10, 25, 50, 100):
    out = opt.optimize(**d, n_iter=N, outputs=names, kernel="w4")
    step = opt.plan(**d, n_iter=N, outputs=names, out=out, kernel="w4")
    for _ in range(int(80 / (0.003 * N + 0.01))): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): step()
    e1.record(); torch.cuda.synchronize()
    res[N] = (e0.elapsed_time(e1) / 200, sclk_ghz(out["clock"]))
    print(f"n_iter {N:4d}: {res[N][0]*1000:8.2f} us  sclk {res[N][1]:.3f} GHz  = {res[N][0]*1e-3*res[N][1]*1e9:9.0f} cycles", flush=True)
slope = (res[100][0] - res[50][0]) / 50
print(f"{sys.argv[1]}: per iteration {slope*1000:.3f} us = {slope*1e-3*res[100][1]*1e9:.0f} cycles; intercept (set-up + epilogue + launch) {(res[50][0]-50*slope)*1000:.2f} us = {(res[50][0]-50*slope)*1e-3*res[50][1]*1e9:.0f} cycles")

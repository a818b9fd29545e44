#!/usr/bin/env python3
"""Diagnostic: one frame of ONE sequence (the Unity budget: 10 iterations per frame, DragPoser.cs:34) -- host wall time per
frame of dp_optimize + dp_sequence_advance launched eagerly and replayed from a captured HIP graph, plus a result check.
Usage: tools/graph_latency.py [n_sequences] [iterations]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
d = to_device_batch(R.synth_inputs(R.OracleModel(), S), dev)
out = opt.allocate_outputs(S)
eager = {k: v.clone() for k, v in opt.optimize(**d, n_iter=N, out=out).items()}
torch.cuda.synchronize()

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    opt.optimize(**d, n_iter=N, out=out)
for v in out.values():
    v.zero_()
g.replay()
torch.cuda.synchronize()
for k in eager:
    assert torch.equal(out[k], eager[k]), k
print(f"graph replay reproduces the eager launch bit for bit ({S} sequence(s), {N} iterations)")


def wall(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"eager  : {wall(lambda: opt.optimize(**d, n_iter=N, out=out)):8.1f} us per frame (back-to-back launches, host wall)")
print(f"graph  : {wall(g.replay):8.1f} us per frame")
# latency of ONE frame (the interactive case: the host waits for the pose)
def one(fn):
    fn(); torch.cuda.synchronize()
print(f"eager, synchronised every frame: {wall(lambda: one(lambda: opt.optimize(**d, n_iter=N, out=out)), 500):8.1f} us")
print(f"graph, synchronised every frame: {wall(lambda: one(g.replay), 500):8.1f} us")

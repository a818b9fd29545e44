#!/usr/bin/env python3
"""Diagnostic: wall time of one drag_pose() call of the native Unity plug-in (libDragPoserDLL.so) -- upload, temporal
target block when due, optimise loop, epilogue, download, synchronisation -- at the Unity budget of 10 iterations
(Core/DragPoser.cs:34), with a reference-size temporal predictor (random weights: temporal.pt is not distributed).
Usage: tools/plugin_latency.py"""
import ctypes as C, os, shutil, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import export_temporal_bin as X
import test_unity_abi as U
from oracle import ref_torch as R
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.temporal import TemporalPredictor

tmp = tempfile.mkdtemp()
shutil.copy(os.path.join(U.DATA, "dragposer_model.bin"), os.path.join(tmp, "dragposer_model.bin"))
torch.manual_seed(0)
sd = {k: v.numpy() for k, v in TemporalPredictor().state_dict().items()}
sd["means_latent"], sd["stds_latent"], sd["sample_step"] = np.zeros(24, np.float32), np.ones(24, np.float32), np.array([4.0], np.float32)
X.write(sd, os.path.join(tmp, "temporal.bin"))
lib = U._load()
h = lib.init_drag_poser()
lib.set_reference_skeleton(h, U.CLIP.encode())
lib.load_models(h, tmp.encode())
assert lib.drag_poser_last_error(h) == b"", lib.drag_poser_last_error(h)
mask = np.zeros(22, np.float32); mask[R.TRACK6] = 1
w = np.ones((22, 2), np.float32)
for j, wj in R.W6.items():
    w[j] = wj
lib.set_mask_and_weights(h, mask.ctypes.data_as(C.POINTER(C.c_float)), w.ctypes.data_as(C.POINTER(U.F2)))
lib.set_optim_params(h, 1e-4, 1e-2, 10, 1e-2)
b = R.synth_inputs(R.OracleModel(), 64, seed=5)
idx = np.array(R.TRACK6)
frames = []
for t in range(64):
    tq = np.stack([U._mat_to_quat(M) for M in b["tgt_rot"][t, idx].reshape(6, 3, 3)])
    frames.append(((U.F3 * 6)(*[U.F3(*map(float, p)) for p in b["tgt_pos"][t, idx]]), (U.Qt * 6)(*[U.Qt(*map(float, q)) for q in tq])))
res_pose, res_pos = (U.Qt * 22)(), (U.F3 * 1)()
for lam, window, label in ((0.0, 0, "pull term off"), (0.02, 0, "temporal term on, window 0 (a prediction every frame)"), (0.15, 16, "temporal term on, window 16")):
    lib.set_lambdas(h, 1.0, lam, window)
    lib.init_drag_model(h, U.F3(0.0, 0.0, 0.0), U.Qt(*[float(v) for v in b["cur_rot"][0]]))
    for k in range(32):
        lib.drag_pose(h, 6, frames[k % 64][0], frames[k % 64][1], res_pose, res_pos)
    t0 = time.perf_counter()
    n = 320
    for k in range(n):
        lib.drag_pose(h, 6, frames[k % 64][0], frames[k % 64][1], res_pose, res_pos)
    dt = (time.perf_counter() - t0) / n
    assert lib.drag_poser_last_error(h) == b""
    print(f"drag_pose(), 6 trackers, maxIter 10, {label}: {dt * 1e6:.0f} us per call", flush=True)
lib.destroy_drag_poser(h)
# and without a predictor (no temporal.bin in the model folder): the optimise loop alone
os.remove(os.path.join(tmp, "temporal.bin"))
h = lib.init_drag_poser()
lib.set_reference_skeleton(h, U.CLIP.encode())
lib.load_models(h, tmp.encode())
lib.set_mask_and_weights(h, mask.ctypes.data_as(C.POINTER(C.c_float)), w.ctypes.data_as(C.POINTER(U.F2)))
lib.set_optim_params(h, 1e-4, 1e-2, 10, 1e-2)
lib.set_lambdas(h, 1.0, 0.0, 0)
lib.init_drag_model(h, U.F3(0.0, 0.0, 0.0), U.Qt(*[float(v) for v in b["cur_rot"][0]]))
for k in range(32):
    lib.drag_pose(h, 6, frames[k % 64][0], frames[k % 64][1], res_pose, res_pos)
t0 = time.perf_counter()
for k in range(320):
    lib.drag_pose(h, 6, frames[k % 64][0], frames[k % 64][1], res_pose, res_pos)
print(f"drag_pose(), 6 trackers, maxIter 10, no predictor loaded: {(time.perf_counter() - t0) / 320 * 1e6:.0f} us per call", flush=True)
lib.destroy_drag_poser(h)
shutil.rmtree(tmp)

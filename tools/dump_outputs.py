#!/usr/bin/env python3
"""Diagnostic: run the S1, S3 and S4 workloads (fixed count and with the reference's while-condition) and a whole-sequence launch
through the library DRAGPOSER_LIB names and save z / pos / loss / iters, so that two builds can be compared bit for bit
([KERNEL=w16] [FRAMES=n] [SEEDS=1234,1,2] tools/dump_outputs.py OUT.npz; then np.array_equal on the two files)."""
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

dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
kernel = os.environ.get("KERNEL", "auto")
B = int(os.environ.get("FRAMES", "2048"))
out = {}
for seed in [int(s) for s in os.environ.get("SEEDS", "1234").split(",")]:
    for name, kw, n_iter, lam in (("s1", dict(), 50, 0.02), ("s3", dict(trackers=3), 100, 0.15), ("s4", dict(mixed=True), 50, 0.02)):
        b = R.synth_inputs(R.OracleModel(), B, seed=seed, **kw)
        d = to_device_batch(b, dev)
        o = opt.optimize(**d, n_iter=n_iter, lambda_tmp=lam, kernel=kernel)
        for k in ("z", "pos", "loss"):
            out[f"{name}_{seed}_{k}"] = o[k].cpu().numpy()
        e = opt.optimize(**d, n_iter=n_iter, lambda_tmp=lam, kernel=kernel, stop_eps_pos=1e-4, stop_eps_rot=1e-2, min_loss_incr=1e-5)  # the while-condition
        for k in ("z", "pos", "iters"):
            out[f"{name}_{seed}_es_{k}"] = e[k].cpu().numpy()
if kernel in ("auto", "w4"):  # a whole-sequence launch (dp_w4_kernel<4, true, true>): 64 sequences x 40 steps of drifting targets
    from dragposer_amd.drag_pose import DragPose
    S, T = 64, 40
    b = R.synth_inputs(R.OracleModel(), S * T, seed=7)
    dp = DragPose(opt, None, np.zeros(24), np.ones(24), n_sequences=S)
    dp.set_initial_state(b["z0"][:S], np.zeros((S, 3)), b["cur_rot"][:S], np.zeros((S, 6)))
    idx = R.TRACK6
    tp = torch.tensor(b["tgt_pos"].reshape(T, S, 22, 3)[:, :, idx])
    tR = torch.tensor(b["tgt_rot"].reshape(T, S, 22, 9)[:, :, idx])
    poses, gpos, iters = dp.run_frames(tp, tR, idx, np.array([R.W6[j] for j in idx], np.float32), max_iter=30, learning_rate=1e-2,
                                       lambda_temporal=0.0, temporal_future_window=0, stop_eps_pos=1e-4, stop_eps_rot=1e-2, joint_adjustment_indices=(0, 0),
                                       joint_adjustment_weight=0.5)
    out["seq_poses"], out["seq_gpos"], out["seq_iters"] = poses.cpu().numpy(), gpos.cpu().numpy(), iters.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], len(out), "arrays")

#!/usr/bin/env python3
"""Diagnostic: the whole per-frame operator (temporal target block, early-stopped optimise loop, epilogue with history
buffers) for S sequences advancing in lock-step -- frames/s of the sequence state machine, not of the bare kernel.
The sequences are 400-frame windows of tests/data/_local/example.bvh at different offsets; the temporal predictor has
the reference's size with seeded random weights (temporal.pt is not distributed).
Usage: tools/throughput_sequences.py [config: 6|3]"""
import argparse, os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd import eval_drag as E
from dragposer_amd.encoder import PoseEncoder
from dragposer_amd.optimizer import LatentOptimizer
from dragposer_amd.temporal import TemporalPredictor

which = sys.argv[1] if len(sys.argv) > 1 else "6"
cfg = dict(E.DEFAULT_CONFIG) if which == "6" else json.load(open(os.path.join(ROOT, "dragposer_amd", "config", "3_trackers_config.json")))
clip = os.path.join(ROOT, "tests", "data", "_local", "example.bvh")
args = argparse.Namespace(max_frames=None, max_iter=100, verbose=False, torch_temporal=False)
raw = np.load(E.DEFAULT_MODEL)
opt = LatentOptimizer(E.DEFAULT_MODEL, device="cuda:0")
enc = PoseEncoder().to(opt.device)
base = E.prepare_file(args, clip, opt, enc, cfg, raw)
torch.manual_seed(0)
temporal = TemporalPredictor().eval()
pack = (temporal, np.zeros(24, np.float32), np.ones(24, np.float32))
L = 400
for S in (1, 64, 256, 1024):
    seqs = []
    for k in range(S):
        o = (k * 37) % (base["n_frames"] - L)
        q = dict(base)
        q["n_frames"] = L
        q["tp_rel"], q["tR"], q["gpos"] = base["tp_rel"][o:o + L], base["tR"][o:o + L], base["gpos"][o:o + L]
        m = dict(base["m"])
        for key in ("global_pos", "global_rot", "heights"):
            m[key] = base["m"][key][o:]
        q["m"] = m
        seqs.append(q)
    for native, per_frame in ((True, False), (True, True), (False, True)):
        args.torch_temporal, args.per_frame = not native, per_frame
        res, elapsed, lam, _ = E.run_sequences(args, seqs, opt, pack, cfg)
        it = np.mean([r["iters"].mean() for r in res])
        print(f"{which} trackers, lambda_temporal {lam}, window {cfg['temporal_future_window']}: S={S:5d} sequences x {L} frames, "
              f"{'native' if native else 'torch '} temporal block, {'host-driven frame loop' if per_frame else 'frame loop on the device '}: "
              f"{elapsed:7.3f} s = {S * L / elapsed:10.0f} frames/s ({it:.1f} iterations per frame)", flush=True)

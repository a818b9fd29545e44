#!/usr/bin/env python3
"""Why ONE 3-tracker sequence averages 19.8 iterations per frame with the native temporal block and 27.3 with the PyTorch one
(profiles/r03_sequence_throughput.txt) while 64 sequences and more average the same with either.

The same 400-frame sequence (tools/throughput_sequences.py's first one: random-weight predictor of the reference's size, 3-tracker
config, window 16), four ways: native / PyTorch temporal block, each from the encoder's initial latent and from that latent moved by
1e-7 (one fp32 ulp).  Printed per run: mean iterations per frame; per pair of runs: the first frame whose iteration count differs and
the distance of the returned poses there and at the end.  If the two predictors differed in substance, the perturbed twin of a run
would stay with it and the other predictor's run would not; if the loop is chaotic (an untrained, expansive Transformer feeding its
own history), a 1e-7 twin leaves as early and ends as far away as the other predictor's run does.
Also: the two predictors on IDENTICAL inputs (the state before the first prediction): max |z_tgt native - z_tgt torch|."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd import eval_drag as E
from dragposer_amd.drag_pose import DragPose
from dragposer_amd.encoder import PoseEncoder
from dragposer_amd.optimizer import LatentOptimizer
from dragposer_amd.temporal import TemporalPredictor

cfg = json.load(open(os.path.join(ROOT, "dragposer_amd", "config", "3_trackers_config.json")))
clip = os.path.join(ROOT, "tests", "data", "_local", "example.bvh")
args = argparse.Namespace(max_frames=None, max_iter=100, verbose=False, torch_temporal=False, per_frame=True)
raw = np.load(E.DEFAULT_MODEL)
opt = LatentOptimizer(E.DEFAULT_MODEL, device="cuda:0")
enc = PoseEncoder().to(opt.device)
base = E.prepare_file(args, clip, opt, enc, cfg, raw)
torch.manual_seed(0)
temporal = TemporalPredictor().eval()
pack = (temporal, np.zeros(24, np.float32), np.ones(24, np.float32))
L = 400
q = dict(base)
q["n_frames"] = L
q["tp_rel"], q["tR"], q["gpos"] = base["tp_rel"][:L], base["tR"][:L], base["gpos"][:L]
runs = {}
for name, native, eps in (("native", True, 0.0), ("native + 1e-7", True, 1e-7), ("torch", False, 0.0), ("torch + 1e-7", False, 1e-7)):
    s = dict(q)
    s["z0"] = q["z0"] + eps * torch.tensor([1.0, -1.0] * 12, device=q["z0"].device)
    args.torch_temporal = not native
    res, _, lam, _ = E.run_sequences(args, [s], opt, pack, cfg)
    runs[name] = res[0]
    print(f"{name:14s}: {res[0]['iters'].mean():5.1f} iterations per frame over {L} frames (lambda_temporal {lam}, window {cfg['temporal_future_window']})", flush=True)
mu4 = raw["means.dqs"].reshape(22, 8)[:, :4].reshape(88); sd4 = raw["stds.dqs"].reshape(22, 8)[:, :4].reshape(88)
for a, b in (("native", "torch"), ("native", "native + 1e-7"), ("torch", "torch + 1e-7"), ("native + 1e-7", "torch + 1e-7")):
    ia, ib = runs[a]["iters"], runs[b]["iters"]
    diff = np.nonzero(ia != ib)[0]
    first = int(diff[0]) if len(diff) else -1
    dq = np.abs((runs[a]["poses"] - runs[b]["poses"]) * sd4).max(axis=1)  # de-normalised quaternion components
    print(f"{a:14s} vs {b:14s}: iteration counts differ first at frame {first:3d} ({(ia == ib).mean():.2f} equal overall); quaternion components max |d| at frames "
          f"0 / 15 / 31 / 63 / 127 / 255 / 399: " + " / ".join(f"{dq[t]:.1e}" for t in (0, 15, 31, 63, 127, 255, 399)))
# the two predictors on identical inputs: the state of a fresh sequence
dp_n = DragPose(opt, temporal, pack[1], pack[2], n_sequences=1, native_temporal=True)
dp_t = DragPose(opt, temporal, pack[1], pack[2], n_sequences=1, native_temporal=False)
for dp in (dp_n, dp_t):
    dp.set_initial_state(q["z0"], q["m"]["global_pos"][0], q["m"]["global_rot"][0], q["m"]["heights"][0])
    dp._temporal_targets(16)
d = (dp_n.target_latent_buffer - dp_t.target_latent_buffer).abs().max().item()
print(f"identical inputs (fresh sequence, window 16): max |z_tgt native - z_tgt torch| = {d:.1e} (|z_tgt| max {dp_t.target_latent_buffer.abs().max().item():.2f})")

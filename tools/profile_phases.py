#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of dp_optimize_kernel from the -DDP_PROFILE build
(in-kernel s_memtime stamps taken by wave 0 of every workgroup).  Shares only -- the stamped
build's run time is never quoted.  Run on the GPU box:
    hipcc ... -DDP_PROFILE -o gpurun_out/libdp_prof.so ...; DRAGPOSER_LIB=gpurun_out/libdp_prof.so python tools/profile_phases.py
"""
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

NAMES = ["L0 mfma", "bar1", "L1 load+mfma", "bar2", "L2 load+mfma", "bar3", "P3a normalise/bones", "P3b tracker terms",
         "P3c gather/backward/out", "bar4", "bL2", "bar5", "bL1", "bar6", "bL0", "bar7", "Adam"]
NAMES_W4 = ["L0 (transpose, 24 steps, lrelu)", "L1 (40 steps)", "L2 (120 steps, 2 transposes)", "P3 stage 1 normalise/bones", "P3 stage 2 tracker terms",
            "P3 stage 3 gather/backward", "bL2 (104 steps)", "bL1 (60 steps)", "bL0 (40 steps)", "Adam"]
if os.environ.get("PHASE_KERNEL", "w4") == "w4":
    NAMES = NAMES_W4
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MAXT = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # tracker hint: > 0 selects dp_kernel4 (DP_KERNEL=4x1|4x2 forces a variant)
N = 50
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
b = R.synth_inputs(R.OracleModel(), B)
d = to_device_batch(b, dev)
dbg = torch.zeros(((B + 7) // 8) * 40 + 64, device=dev)
for _ in range(3):
    opt.optimize(**d, n_iter=N, max_trackers=MAXT, _debug=dbg)
torch.cuda.synchronize()
fpb, tpb, _ = opt.kernel_geometry()
grid = (B + fpb - 1) // fpb
print(f"kernel: {fpb} frames / {tpb} threads per workgroup, {grid} workgroups")
raw = dbg.cpu().numpy()[: grid * 40].view(np.uint64).reshape(grid, 20).astype(np.float64) / N
p = raw[:, :len(NAMES)]
L0SUB = os.environ.get("PHASE_L0") == "1"  # build with EXTRA_DEFS=-DW4_STAMP_L0: slots 12..15 = the sub-phases of L0, slot 0 empty
if L0SUB:
    p = np.concatenate([p, raw[:, 12:16]], axis=1)
    NAMES = list(NAMES) + ["  L0: Adam's tail (new latent arrives)", "  L0: D -> X transpose", "  L0: 24 K-steps + chain end", "  L0: LeakyReLU factor"]
tot = p.sum(1)
print(f"B={B}: cycles/iteration (s_memtime ticks) mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(NAMES):
    print(f"  {n:26s} {p[:, i].mean():8.0f}  {100 * p[:, i].mean() / tot.mean():5.1f}%")
if os.environ.get("PHASE_KERNEL", "w4") == "w4":
    print(f"  per launch: entry -> first iteration {raw[:, 10].mean() * N:.0f} cycles, last iteration -> stores done {raw[:, 11].mean() * N:.0f} cycles "
          f"(max over workgroups {raw[:, 10].max() * N:.0f} / {raw[:, 11].max() * N:.0f})")
    print(f"  shader clock held over the loop: {raw[:, 17].mean() / raw[:, 16].mean() * 0.1:.3f} GHz (s_memtime / s_memrealtime)")
    if not L0SUB:
      print("  setup split (time of arrival, no drain): image into LDS + flags + tracker loads issued {:.0f}, registers from LDS + frame blocks + trackers {:.0f}, barrier {:.0f}, loop entry {:.0f}".format(
        *[raw[:, i].mean() * N for i in (12, 13, 14, 15)]))
    print(f"  first iteration {raw[:, 18].mean() * N:.0f} cycles, second {raw[:, 19].mean() * N:.0f} (mean over all: {raw[:, 17].mean():.0f})")
elif raw[:, 17:].any():  # ad-hoc sub-stamps (slots 17..19) of an experiment
    print("  extra stamps 16..19:", " ".join(f"{raw[:, i].mean():.0f}" for i in range(16, 20)))

#!/usr/bin/env python3
"""Where and why a kernel leaves the fp32 oracle on GIVEN frames of a full-size reference fixture (tests/golden/full_s*_1024.npz).

For every frame: the kernel's and the fp32 C oracle's latent after 1, 2, ... n_iter iterations (the kernel is re-run with n_iter = n on
those frames only), the first iteration t* at which they differ by more than `thr`, and what sits at the last common point
z(t* - 1): the smallest |pre-activation| of the two LeakyReLU layers on the way, whether a unit has OPPOSITE SIGNS for the two
latents, the smallest |dL/dz| component, and how the difference grows afterwards (|dz| every 10 iterations).
Usage: tools/frame_divergence.py FIXTURE KERNEL FRAME[,FRAME...]      e.g.  full_s4_1024 w16 962,502"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
from test_hip_configs_at_size import CASES, load_case

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")
name, kernel, frames = sys.argv[1], sys.argv[2], [int(f) for f in sys.argv[3].split(",")]
c, ref, mt, b = load_case(os.path.join(ROOT, "tests", "golden"), name)
wd, N, lam = c["wd"], mt["n_iter"], mt["lambda_tmp"]
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev, weight_dtype="bf16" if wd == "bf16" else "fp32")
A32, A64 = AnalyticOracle(precision="f32", weight_rounding=wd), AnalyticOracle(precision="f64", weight_rounding=wd)
F = {k: v.astype(np.float64) for k, v in A64.folded().items()}
mm = lambda a, b_: np.linalg.norm(a - b_, axis=-1) * 1000.0


def pre_acts(z):
    p0 = F["A0"] @ z.astype(np.float64) + F["c0"]
    p1 = F["A1"] @ np.maximum(p0, 0.2 * p0) + F["b1"]
    return np.concatenate([p0, p1])


sub = {k: b[k][frames] for k in KEYS}
d = to_device_batch(sub, dev)
zg = [sub["z0"]] + [opt.optimize(**d, n_iter=n, lambda_tmp=lam, kernel=kernel, outputs=("z",))["z"].cpu().numpy() for n in range(1, N + 1)]
zo = [sub["z0"]] + [A32.optimize(*[sub[k] for k in KEYS], n, lam_tmp=lam)["z_final"] for n in range(1, N + 1)]
z64 = [sub["z0"]] + [A64.optimize(*[sub[k] for k in KEYS], n, lam_tmp=lam)["z_final"] for n in range(1, N + 1)]
fin = opt.optimize(**d, n_iter=N, lambda_tmp=lam, kernel=kernel)
for i, f in enumerate(frames):
    for thr in (1e-6, 1e-5):
        dz = np.array([np.abs(zg[n][i] - zo[n][i]).max() for n in range(N + 1)])
        d64 = np.array([np.abs(z64[n][i] - zo[n][i]).max() for n in range(N + 1)])
        ts = np.nonzero(dz > thr)[0]
        t = int(ts[0]) if len(ts) else N
        pg, po = pre_acts(zg[t - 1][i]), pre_acts(zo[t - 1][i])
        flip = np.nonzero(np.sign(pg) != np.sign(po))[0]
        near = min(np.abs(pre_acts(zo[n][i])).min() for n in range(t))
        _, g = A64.grad(*[sub[k][i:i + 1] if k != "z0" else zo[t - 1][i:i + 1] for k in KEYS], 1.0, lam)
        print(f"{name} {kernel} frame {f} (trackers {np.nonzero(sub['tracked'][i])[0].tolist()}): final {mm(fin['pos'].cpu().numpy()[i], ref['pos'][f]).max():.3f} mm from the reference; "
              f"first |dz| > {thr:g} vs the fp32 oracle at iteration {t} (before {dz[t - 1]:.1e}, at {dz[t]:.1e}); smallest |pre-activation| on the way {near:.1e}; "
              f"sign-flipped units at z(t* - 1): {flip.tolist()} (|pre| kernel {np.abs(pg[flip]).tolist()}, oracle {np.abs(po[flip]).tolist()}); "
              f"smallest |dL/dz_k| there {np.abs(g).min():.1e}")
    print("   |dz| kernel vs fp32 oracle every 5 iterations:", " ".join(f"{x:.1e}" for x in dz[::5]))
    print("   |dz| fp64  vs fp32 oracle every 5 iterations:", " ".join(f"{x:.1e}" for x in d64[::5]), flush=True)

"""Test helpers (GPU parity tests): the two mechanisms by which two CORRECT implementations of the optimise loop part ways on
a frame (tests/test_hip_parity.py module docstring, profiles/r03_soak_divergence.txt), evaluated on the fp64 oracle's trajectory of
given frames.  Only tests import this (it uses oracle/)."""
import numpy as np

from oracle.analytic import AnalyticOracle

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


def kink_distance(b, frames, n_iter, lam):
    """smallest |pre-activation| of the two LeakyReLU layers along the fp64 oracle's trajectory of each given frame"""
    A = AnalyticOracle(precision="f64")
    F = {k: v.astype(np.float64) for k, v in A.folded().items()}
    out = []
    for f in frames:
        a = [b[k][f:f + 1] for k in KEYS]
        mk = np.inf
        for t in range(n_iter):
            z = (a[0] if t == 0 else A.optimize(*a, t, lam_tmp=lam)["z_final"])[0].astype(np.float64)
            p0 = F["A0"] @ z + F["c0"]
            p1 = F["A1"] @ np.maximum(p0, 0.2 * p0) + F["b1"]
            mk = min(mk, np.abs(p0).min(), np.abs(p1).min())
        out.append(mk)
    return np.array(out)


def tiny_gradient(b, frames, lam, n_first=6):
    """The second, rarer mechanism (profiles/r03_soak_divergence*.txt: 10 of 60 missed frames over 36 seeds, one of them 2.1 mm): Adam's first steps
    move every component by about lr * sign(g) whatever |g| is, so a component of dL/dz within rounding of zero (typical smallest
    component: 1e-4) gives two correct implementations different steps.  Smallest |dL/dz_k| over the first iterations of the fp64
    oracle's trajectory of each given frame."""
    A = AnalyticOracle(precision="f64")
    out = []
    for f in frames:
        a = [b[k][f:f + 1] for k in KEYS]
        mg = np.inf
        for t in range(n_first):
            z = a[0] if t == 0 else A.optimize(*a, t, lam_tmp=lam)["z_final"]
            _, g = A.grad(z, *a[1:], 1.0, lam)
            mg = min(mg, float(np.abs(g).min()))
        out.append(mg)
    return np.array(out)


def explained(b, frames, n_iter, lam, flagged=()):
    """per given frame: is it flagged by a reference pair (fp32 vs fp64 runs of the same code), within fp32 rounding of a LeakyReLU
    kink (|pre-activation| < 5e-6; typical frames 3e-4), or moved by a gradient component within rounding of zero (< 1e-5)?"""
    k = kink_distance(b, frames, n_iter, lam)
    t = tiny_gradient(b, frames, lam)
    fl = set(int(f) for f in flagged)
    return np.array([int(f) in fl or kk < 5e-6 or tt < 1e-5 for f, kk, tt in zip(frames, k, t)], bool), k, t

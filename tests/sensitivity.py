"""Test helpers (GPU parity tests): the two mechanisms by which two CORRECT implementations of the optimise loop part ways on
a frame (tests/test_hip_parity.py module docstring, profiles/r03_soak_divergence.txt), evaluated on the fp64 oracle's trajectory of
given frames.  Only tests import this (it uses oracle/)."""
import numpy as np

from oracle.analytic import AnalyticOracle

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


def kink_distance(b, frames, n_iter, lam, weight_rounding="none"):
    """smallest |pre-activation| of the two LeakyReLU layers along the oracle's trajectory of each given frame -- along BOTH its fp64
    and its fp32 trajectory: a kernel follows the fp32 one to a few 1e-7, and a pre-activation that the fp32 path takes to 1e-6 of zero
    can sit 1e-4 away on a path that has meanwhile gone another way (profiles/r04_divergence_s4.txt, frame 962)"""
    out = []
    for f in frames:
        a = [b[k][f:f + 1] for k in KEYS]
        mk = np.inf
        for prec in ("f64", "f32"):
            A = AnalyticOracle(precision=prec, weight_rounding=weight_rounding)
            F = {k: v.astype(np.float64) for k, v in A.folded().items()}
            for t in range(n_iter):
                z = (a[0] if t == 0 else A.optimize(*a, t, lam_tmp=lam)["z_final"])[0].astype(np.float64)
                p0 = F["A0"] @ z + F["c0"]
                p1 = F["A1"] @ np.maximum(p0, 0.2 * p0) + F["b1"]
                mk = min(mk, np.abs(p0).min(), np.abs(p1).min())
        out.append(mk)
    return np.array(out)


def tiny_gradient(b, frames, lam, n_first=6, weight_rounding="none"):
    """The second, rarer mechanism (profiles/r03_soak_divergence*.txt: 10 of 60 missed frames over 36 seeds, one of them 2.1 mm): Adam's first steps
    move every component by about lr * sign(g) whatever |g| is, so a component of dL/dz within rounding of zero (typical smallest
    component: 1e-4) gives two correct implementations different steps.  Smallest |dL/dz_k| over the first iterations of the fp64
    oracle's trajectory of each given frame."""
    A = AnalyticOracle(precision="f64", weight_rounding=weight_rounding)
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


def explained(b, frames, n_iter, lam, flagged=(), weight_rounding="none"):
    """per given frame: is it flagged by a reference pair (fp32 vs fp64 runs of the same code), within fp32 rounding of a LeakyReLU
    kink (|pre-activation| < 5e-6; typical frames 3e-4), or moved by a gradient component within rounding of zero (< 1e-5)?
    `weight_rounding`: the decoder weights the frames were run with ("bf16": BASELINE config 5)."""
    k = kink_distance(b, frames, n_iter, lam, weight_rounding)
    t = tiny_gradient(b, frames, lam, weight_rounding=weight_rounding)
    fl = set(int(f) for f in flagged)
    return np.array([int(f) in fl or kk < 5e-6 or tt < 1e-5 for f, kk, tt in zip(frames, k, t)], bool), k, t

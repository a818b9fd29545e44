#!/usr/bin/env python3
"""Cost of a step of dp_optimize_sequence: per-step overhead and per-iteration cost (fixed iteration counts, one sequence and many)."""
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
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
g = R.load_golden(os.path.join(ROOT, "tests", "golden", "es.npz"))
T = 500
for S in (1, 4, 64, 1024):
    rep = (S + 63) // 64
    d = to_device_batch({k: np.concatenate([g[k]] * rep)[:S] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")}, dev)
    tp = d["tgt_pos"][None].repeat(T, 1, 1, 1).contiguous(); tr = d["tgt_rot"][None].repeat(T, 1, 1, 1).contiguous()
    res = []
    for n_iter in (1, 5, 10, 20):
        lat = d["z0"].clone()
        gp, gr = torch.zeros(S, 3, device=dev), d["cur_rot"].clone()
        lb, db, hb = torch.zeros(S, 60, 24, device=dev), torch.zeros(S, 60, 3, device=dev), torch.zeros(S, 60, 6, device=dev)
        kw = dict(n_iter=n_iter, lambda_tmp=0.0, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=float("-inf"))
        args = (lat, tp, tr, None, d["w"], d["tracked"], torch.zeros(S, 24, device=dev), (0, 24), gp, gr, lb, db, hb, (0, 4, 8, 13, 17, 21))
        opt.optimize_sequence(*args, **kw)
        torch.cuda.synchronize()
        t0 = time.time()
        opt.optimize_sequence(*args, **kw)
        torch.cuda.synchronize()
        res.append((n_iter, (time.time() - t0) / T * 1e6))
    per_iter = (res[-1][1] - res[0][1]) / (res[-1][0] - res[0][0])
    if S == 1:
        fixed_s1 = res[0][1] - per_iter
    print(f"S = {S:5d}: " + ", ".join(f"{n} it: {us:.1f} us/step" for n, us in res) + f" -> {per_iter:.2f} us per iteration, {res[0][1] - per_iter:.1f} us per step besides")

# long single-sequence runs: fixed count, then the reference's early-stop settings on fresh targets every step
T = 5052
S = 1
m = R.OracleModel()
b = R.synth_inputs(m, T)
d1 = to_device_batch({k: g[k][:1] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")}, dev)
tp = torch.from_numpy(b["tgt_pos"]).to(dev).reshape(T, 1, 22, 3).contiguous(); tr = torch.from_numpy(b["tgt_rot"]).to(dev).reshape(T, 1, 22, 9).contiguous()
for label, kw, same in (("fixed 8 iterations, same targets", dict(n_iter=8, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=float("-inf")), True),
                        ("fixed 8 iterations, fresh targets", dict(n_iter=8, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=float("-inf")), False),
                        ("early stop (1e-4, 1e-2, 1e-5), max 100, fresh targets", dict(n_iter=100, stop_eps_pos=1e-4, stop_eps_rot=1e-2, min_loss_incr=1e-5), False)):
    tpp = d1["tgt_pos"][None].repeat(T, 1, 1, 1).contiguous() if same else tp
    trr = d1["tgt_rot"][None].repeat(T, 1, 1, 1).contiguous() if same else tr
    for rep in range(2):
        lat = d1["z0"].clone()
        gp, gr = torch.zeros(S, 3, device=dev), d1["cur_rot"].clone()
        lb, db, hb = torch.zeros(S, 60, 24, device=dev), torch.zeros(S, 60, 3, device=dev), torch.zeros(S, 60, 6, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        r = opt.optimize_sequence(lat, tpp, trr, None, d1["w"], d1["tracked"], torch.zeros(S, 24, device=dev), (0, 24), gp, gr, lb, db, hb, (0, 4, 8, 13, 17, 21), lambda_tmp=0.0, **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
    it = r["iters"].float().mean().item()
    print(f"{label}: {T} steps in {dt * 1e3:.1f} ms = {dt / T * 1e6:.1f} us/step, mean iterations {it:.1f} -> {(dt / T * 1e6 - fixed_s1) / it:.2f} us per iteration after the {fixed_s1:.1f} us per step measured above")

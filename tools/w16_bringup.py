#!/usr/bin/env python3
"""GPU bring-up of the 16-frames-per-wave kernel: first-iteration loss / gradient against the fp64 analytic oracle, golden
parity, agreement with the 4-frames-per-wave kernel, and a first timing.  Usage: tools/w16_bringup.py [frames]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")
dev = torch.device("cuda:0")
gold = os.path.join(ROOT, "tests", "golden")
opts = {"none": LatentOptimizer(device=dev), "bf16": LatentOptimizer(device=dev, weight_dtype="bf16")}
mm = lambda a, b: np.linalg.norm(a - b, axis=-1) * 1000.0
for name in ("s1", "s4", "s3"):
    g = R.load_golden(os.path.join(gold, f"{name}.npz"))
    mt = g["meta"]
    o = opts[mt["weight_rounding"]]
    d = to_device_batch(g, dev)
    dbg = torch.zeros(len(g["z0"]), 240, device=dev)
    out = o.optimize(**d, n_iter=1, lambda_tmp=mt["lambda_tmp"], kernel="w16", _debug=dbg)
    torch.cuda.synchronize()
    A = AnalyticOracle(precision="f64", weight_rounding=mt["weight_rounding"])
    lo, gr = A.grad(*[g[k] for k in KEYS], 1.0, mt["lambda_tmp"])
    gz = dbg.cpu().numpy()[:, 208:232]
    l0 = out["loss"].cpu().numpy()
    print(f"{name}: iteration 0: max |dL/dz - oracle| {np.abs(gz - gr).max():.3e} (|g| max {np.abs(gr).max():.3f}); loss rel err "
          f"{np.abs(l0 - g['loss_hist'][:, 0]).max() / np.abs(g['loss_hist'][:, 0]).max():.3e}", flush=True)
    if np.abs(gz - gr).max() > 1e-4:
        b = int(np.abs(gz - gr).max(1).argmax())
        print("  worst frame", b, "\n  gpu", gz[b], "\n  ref", gr[b], "\n  loss gpu", l0[b], "ref", g["loss_hist"][b, 0])
    n = mt["n_iter"]
    o16 = {k: v.cpu().numpy() for k, v in o.optimize(**d, n_iter=n, lambda_tmp=mt["lambda_tmp"], kernel="w16").items()}
    o4 = {k: v.cpu().numpy() for k, v in o.optimize(**d, n_iter=n, lambda_tmp=mt["lambda_tmp"], kernel="w4").items()}
    e16, e4 = mm(o16["pos"], g["pos"]).max(1), mm(o4["pos"], g["pos"]).max(1)
    print(f"{name}: {n} iterations vs golden: w16 max {e16.max():.4f} mm (mean {e16.mean():.5f}), w4 max {e4.max():.4f} mm (mean {e4.mean():.5f}); "
          f"w16 vs w4 max {mm(o16['pos'], o4['pos']).max():.4f} mm; |z - golden| w16 {np.abs(o16['z'] - g['z_final']).max():.2e}", flush=True)
    for k in ("pose", "rot", "world_rot", "world_disp", "disp", "z_pre", "loss"):
        print(f"    {k}: w16 vs w4 max abs diff {np.abs(o16[k] - o4[k]).max():.3e}")
    assert (o16["iters"] == n).all()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = R.OracleModel()
b = R.synth_inputs(m, B, mixed=True)
d = to_device_batch(b, dev)
for wd in ("bf16", "none"):
    o = opts[wd]
    res = {}
    for kern in ("w4", "w16"):
        for _ in range(2):
            o.optimize(**d, n_iter=50, kernel=kern)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            r = o.optimize(**d, n_iter=50, kernel=kern)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        res[kern] = r["pos"].cpu().numpy()
        print(f"weights {wd}: {kern}: {B} frames x 50 iterations (mixed 1-6 trackers): {dt * 1e3:.3f} ms per launch = {B / dt / 1e6:.2f} M frames/s "
              f"(fp32-MFMA roofline fraction {B * 50 * 35520 / dt / 157.3e12:.3f})", flush=True)
    e = mm(res["w4"], res["w16"]).max(1)
    print(f"weights {wd}: w16 vs w4 over {B} frames: max {e.max():.4f} mm, p99.9 {np.percentile(e, 99.9):.5f}, mean {e.mean():.6f}, above 0.05 mm: {(e > 0.05).sum()}")

#!/usr/bin/env python3
"""GPU bring-up: stage-by-stage comparison of the HIP kernel with the oracle (run on the GPU box)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), flush=True)
opt = LatentOptimizer(device="cuda:0")
print("geometry", opt.frames_per_block, opt.threads_per_block, opt.lds_bytes, flush=True)
A = AnalyticOracle(precision="f32")
A64 = AnalyticOracle(precision="f64")
g = R.load_golden(os.path.join(ROOT, "tests/golden/s1.npz")); lam = g["meta"]["lambda_tmp"]
B = 64
# ---- stage 1: forward
z = torch.from_numpy(g["z0"]).to(dev); cr = torch.from_numpy(g["cur_rot"]).to(dev)
o = opt.forward(z, cr); torch.cuda.synchronize()
f = A64.forward(g["z0"], g["cur_rot"])
for k in ("pose", "world_disp", "world_rot", "pos", "rot"):
    print(f"forward {k:10s} maxdiff {np.abs(o[k].cpu().numpy().reshape(f[k].shape) - f[k]).max():.3e}", flush=True)
# ---- stage 2: one iteration with debug dump
db = to_device_batch(g, dev)
dbg = torch.zeros(B, 240, device=dev)
o1 = opt.optimize(**db, n_iter=1, lambda_tmp=lam, _debug=dbg); torch.cuda.synchronize()
dbg = dbg.cpu().numpy()
lo, gr = A64.grad(g["z0"], g["z_tgt"], g["cur_rot"], g["tgt_pos"], g["tgt_rot"], g["w"], g["tracked"], 1.0, lam)
import kernel_emu as KE
from dragposer_amd.model import HostModel
tabs = KE.host_tables(HostModel())
for blk in range(0, B, 16):
    sl = slice(blk, blk + 16)
    y, gy, gz, loss = KE.emulate_iteration(tabs, g["z0"][sl], g["z_tgt"][sl], g["cur_rot"][sl].astype(np.float64), g["tgt_pos"][sl], g["tgt_rot"][sl], g["w"][sl].astype(np.float64), g["tracked"][sl], 1.0, lam)
    print(f"blk {blk}: y diff {np.abs(dbg[sl, 0:92] - y[:, :92]).max():.3e}  gy diff {np.abs(dbg[sl, 104:208] - gy).max():.3e} (|gy| {np.abs(gy).max():.2e})  gz diff {np.abs(dbg[sl, 208:232] - gz).max():.3e}", flush=True)
print(f"iter-1 grad vs oracle {np.abs(dbg[:, 208:232] - gr).max():.3e}  loss diff {np.abs(o1['loss'].cpu().numpy() - lo).max():.3e}", flush=True)
print("z after 1 step diff vs golden-anchored oracle:", np.abs(o1["z"].cpu().numpy() - A.optimize(g["z0"], g["z_tgt"], g["cur_rot"], g["tgt_pos"], g["tgt_rot"], g["w"], g["tracked"], 1, lam_tmp=lam)["z_final"]).max(), flush=True)
# ---- stage 3: full runs vs goldens
for name in ("s1", "s3", "s4"):
    g = R.load_golden(os.path.join(ROOT, f"tests/golden/{name}.npz")); mt = g["meta"]
    op = opt if mt["weight_rounding"] == "none" else LatentOptimizer(device="cuda:0", weight_dtype="bf16")
    o = op.optimize(**to_device_batch(g, dev), n_iter=mt["n_iter"], lambda_tmp=mt["lambda_tmp"]); torch.cuda.synchronize()
    err = np.linalg.norm(o["pos"].cpu().numpy() - g["pos"], axis=-1) * 1000
    print(f"{name}: pos mm max {err.max():.5f} p99 {np.percentile(err, 99):.5f} mean {err.mean():.6f} | z {np.abs(o['z'].cpu().numpy() - g['z_final']).max():.2e} z_pre {np.abs(o['z_pre'].cpu().numpy() - g['z_pre']).max():.2e} pose {np.abs(o['pose'].cpu().numpy() - g['pose']).max():.2e} loss {np.abs(o['loss'].cpu().numpy() - g['loss_hist'][:, -1]).max():.2e} iters {o['iters'].cpu().numpy()[:3]}", flush=True)
# ---- stage 4: timing
model = R.OracleModel()
for Bn in (1024, 4096, 8192):
    b = R.synth_inputs(model, Bn, trackers=6); d = to_device_batch(b, dev)
    outs = opt.optimize(**d, n_iter=50); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10): opt.optimize(**d, n_iter=50, out=outs)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    print(f"B={Bn}: {ms*1000:.1f} us/launch  {Bn/ms*1000/1e6:.2f} M frames/s  MFMA frac {Bn*50*35520/(ms*1e-3)/157.3e12:.3f}", flush=True)

#!/usr/bin/env python3
"""What shader clock does the headline launch run at, and does it depend on how long the GPU has been busy?  (round-4 review, item 1:
one 4096-frame launch took 0.1517 ms at 2.117 GHz, 16-round launches 0.129-0.131 ms per round -- the same cycles at 2.4 GHz.)

Runs on the GPU box.  Every launch writes dp_result.clock (shader cycles / 100 MHz ticks of workgroup 0's loop -> GHz); kernel time is taken
with HIP events around chunks of back-to-back launches.  Three experiments:
  A  a long back-to-back train of identical launches from an idle GPU: kernel time and clock against time since the first launch
  B  the same launch after idle gaps of 0 / 1 / 10 / 100 ms (does the clock state decay between launches?)
  C  batch sizes 256 ... 65536 frames through dp_w4 (rounds = frames / 4096): time per round and clock, steady state
Usage: python tools/clock_ramp.py [--train 4000] > gpurun_out/clock_ramp.txt
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_on_device  # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer  # noqa: E402

NAMES = ("z", "z_pre", "pose", "disp", "world_disp", "world_rot", "pos", "loss", "iters")


def ghz(c):
    c = c.cpu()
    return (c[..., 0].double() / c[..., 1].double().clamp(min=1) * 0.1)


def train(opt, batch, out, n, chunk, kernel="w4", iters=50):
    """n back-to-back launches; per chunk: (ms since start at chunk end, mean kernel ms, clock of the chunk's last launch)"""
    dev = opt.device
    nch = n // chunk
    clk = torch.zeros(nch, 2, dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nch + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for k in range(nch):
        for j in range(chunk):
            o = dict(out)
            if j == chunk - 1:
                o["clock"] = clk[k]
            opt.optimize(**batch, n_iter=iters, outputs=NAMES + (("clock",) if j == chunk - 1 else ()), out=o, kernel=kernel)
        ev[k + 1].record()
    torch.cuda.synchronize()
    g = ghz(clk)
    rows = []
    for k in range(nch):
        rows.append((ev[0].elapsed_time(ev[k + 1]), ev[k].elapsed_time(ev[k + 1]) / chunk, float(g[k])))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=4000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    opt = LatentOptimizer(device=dev)
    print(f"# {torch.cuda.get_device_name(0)}; dp_w4, 6 trackers, 50 iterations; clock = dp_result.clock (s_memtime / s_memrealtime of workgroup 0)")

    # ---- A
    B = 4096
    batch = synth_on_device(opt, B, 1234, dev)
    out = opt.optimize(**batch, n_iter=50, outputs=NAMES, kernel="w4")
    torch.cuda.synchronize()
    time.sleep(3.0)  # idle GPU
    print(f"\n## A: {args.train} back-to-back launches of {B} frames from an idle GPU (3 s of sleep before the first)")
    print("#  ms since start | kernel ms (mean of chunk) | sclk GHz | cycles per launch (kernel ms x GHz, k)")
    first = train(opt, batch, out, 40, 1)
    for t, ms, g in first[:10] + first[10::5]:
        print(f"  {t:10.3f}  {ms:8.4f}  {g:6.3f}  {ms * g * 1e3:8.1f}")
    rest = train(opt, batch, out, args.train, 100)
    t0 = first[-1][0]
    for t, ms, g in rest:
        print(f"  {t0 + t:10.3f}  {ms:8.4f}  {g:6.3f}  {ms * g * 1e3:8.1f}")

    # ---- B
    print("\n## B: one launch after an idle gap (host sleep after a synchronize), 8 samples each: kernel ms / sclk GHz (the launch itself timed by events)")
    for gap_ms in (0.0, 1.0, 10.0, 100.0, 1000.0):
        res = []
        for _ in range(8):
            torch.cuda.synchronize()
            time.sleep(gap_ms * 1e-3)
            clk = torch.zeros(2, dtype=torch.int64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = dict(out)
            o["clock"] = clk
            opt.optimize(**batch, n_iter=50, outputs=NAMES + ("clock",), out=o, kernel="w4")
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1), float(ghz(clk))))
        print(f"  gap {gap_ms:7.1f} ms: " + "  ".join(f"{a:.4f}/{b:.3f}" for a, b in res))

    # ---- C
    print("\n## C: steady state (300 ms of the same launch first), dp_w4 forced: frames | rounds | kernel ms | ms per round | sclk GHz | frac of the 157.3 TF fp32 roofline")
    for B in (256, 1024, 2048, 4096, 8192, 16384, 65536):
        batch = synth_on_device(opt, B, 1234, dev)
        out = opt.optimize(**batch, n_iter=50, outputs=NAMES, kernel="w4")
        torch.cuda.synchronize()
        per = 0.15 * max(1, B // 4096)
        n = max(20, int(300.0 / per))
        train(opt, batch, out, n, n)
        rows = train(opt, batch, out, max(20, n // 3), max(20, n // 3))
        ms, g = rows[-1][1], rows[-1][2]
        rounds = max(1, -(-B // 4096))
        print(f"  {B:6d}  {rounds:3d}  {ms:8.4f}  {ms / rounds:8.4f}  {g:6.3f}  {B * 50 * 35520 / (ms * 1e-3) / 157.3e12:6.3f}")


if __name__ == "__main__":
    main()

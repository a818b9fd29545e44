#!/usr/bin/env python3
"""Diagnostic: one dp_temporal_predict launch at few sequences -- one workgroup per sequence (variant 21) against TEAMS of 2 ... 16 workgroups
per sequence (variants 102 ... 116) and the library's own choice (0).  Device time per launch at the steady clock, back to back.
Usage: tools/team_latency.py [S ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diaglib import use_env_library

use_env_library()
from dragposer_amd.temporal import NativeTemporal, TemporalPredictor

torch.manual_seed(0)
dev = torch.device("cuda:0")
nat = NativeTemporal(TemporalPredictor().eval(), torch.zeros(24), torch.ones(24), device=dev)
Ss = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 64, 128]
for window in (0, 16):
    for S in Ss:
        lat, disp, hts = torch.randn(S, 60, 24, device=dev), torch.randn(S, 60, 3, device=dev), torch.randn(S, 60, 6, device=dev)
        out = torch.empty(S, window + 1, 24, device=dev)
        row = []
        for variant in (21, 102, 104, 108, 116, 0):
            nat._force_variant(variant)
            fn = lambda: nat.predict(lat, disp, hts, window, out=out)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.06:
                for _ in range(20):
                    fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            row.append(e0.elapsed_time(e1) / 20 * 1e3)
        assert nat._team_status() == 0
        print(f"window {window:2d} S = {S:4d}: one workgroup {row[0]:7.1f} us | teams of 2 / 4 / 8 / 16: {row[1]:7.1f} {row[2]:7.1f} {row[3]:7.1f} {row[4]:7.1f} us | "
              f"library's choice {row[5]:7.1f} us   (a forced team larger than the device holds falls back to one workgroup)", flush=True)

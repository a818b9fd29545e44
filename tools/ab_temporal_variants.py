#!/usr/bin/env python3
"""Diagnostic (GPU box): dp_temporal_predict at 1024 / 4096 sequences with forced kernel variants, steady clock; usage: tools/ab_temporal_variants.py 42,44"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from dragposer_amd.temporal import NativeTemporal, TemporalPredictor
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = TemporalPredictor().eval()
nat = NativeTemporal(model, torch.zeros(24), torch.ones(24), device=dev)
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [42, 44]
for window in (0, 16, 60):
    for S in (1024, 4096):
        lat, disp, hts = torch.randn(S, 60, 24, device=dev), torch.randn(S, 60, 3, device=dev), torch.randn(S, 60, 6, device=dev)
        out = torch.empty(S, window + 1, 24, device=dev)
        line, ref = [], None
        for v in variants:
            nat._force_variant(v)
            for _ in range(3): nat.predict(lat, disp, hts, window, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(3, int(60 / (0.3 * (1 + window / 8) * S / 1024)))
            for _ in range(n): nat.predict(lat, disp, hts, window, out=out)
            e0.record()
            for _ in range(10): nat.predict(lat, disp, hts, window, out=out)
            e1.record(); torch.cuda.synchronize()
            o = out.clone()
            if ref is None: ref = o
            line.append(f"v{v} {e0.elapsed_time(e1)/10:.4f} ms (max|d| {float((o-ref).abs().max()):.1e})")
        print(f"window {window:2d} S={S}: " + "  ".join(line), flush=True)

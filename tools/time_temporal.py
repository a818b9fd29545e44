#!/usr/bin/env python3
"""Diagnostic: time of one dp_temporal_predict launch (the whole temporal target block) at the reference's full size
(3 + 3 layers, feed-forward 2048) against the number of sequences and the window, beside the PyTorch-ROCm operator path
(nn.Transformer called window / 4 + 1 times).  Usage: tools/time_temporal.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.temporal import NativeTemporal, TemporalPredictor

torch.manual_seed(0)
dev = torch.device("cuda:0")
model = TemporalPredictor().eval()
nat = NativeTemporal(model, torch.zeros(24), torch.ones(24), device=dev)
gm = model.to(dev)


def torch_block(lat, disp, hts, window, step=4):
    idx = list(range(0, lat.shape[1], step))
    with torch.no_grad():
        enc_in = torch.cat((lat[:, idx[:-1]], torch.stack([disp[:, j:j + step].sum(dim=1) for j in idx[:-1]], dim=1), hts[:, idx[:-1]]), dim=-1)
        tgt = lat[:, idx[-1]].unsqueeze(1)
        buf = torch.zeros(lat.shape[0], window + 1, 24, device=lat.device)
        for i in range(0, window + 1, step):
            pred = gm(enc_in, tgt)
            tgt = torch.cat((tgt, pred[:, -1:]), dim=1)
            buf[:, i] = pred[:, -1]
    return buf


def block_flops(window, Te=14, d=48, F=2048, n_enc=3, n_dec=3, step=4, n_in=33):
    """algorithmic FLOPs of one temporal target block as the kernel computes it (the encoder ONCE -- the reference's nn.Transformer recomputes it
    in every autoregressive call --, then window / step + 1 decoder calls over 1, 2, ... target tokens): 2 FLOP per multiply-add of every linear
    layer and of the attention's two products; LayerNorm / softmax / bias not counted"""
    lin = lambda t, i, o: 2 * t * i * o
    att = lambda tq, tk: 2 * 2 * tq * tk * d
    enc = lin(Te, n_in, d) + n_enc * (lin(Te, d, 3 * d) + att(Te, Te) + lin(Te, d, d) + lin(Te, d, F) + lin(Te, F, d))
    dec = 0
    for T in range(1, window // step + 2):
        dec += lin(T, 24, d) + lin(1, d, 24)
        dec += n_dec * (lin(T, d, 3 * d) + att(T, T) + lin(T, d, d) + lin(T, d, d) + lin(Te, d, 2 * d) + att(T, Te) + lin(T, d, d) + lin(T, d, F) + lin(T, F, d))
    return enc + dec


WEIGHT_BYTES = 4 * sum(p.numel() for p in model.parameters())  # what one workgroup streams per layer pass (5.1 MB for the encoder + one decoder call)
NATIVE_ONLY = "--native-only" in sys.argv  # (for rocprofv3 --kernel-trace --stats: only dp_temporal_kernel launches)
for window in (0, 16, 60):
    for S in (1, 64, 256, 1024, 4096):
        lat, disp, hts = torch.randn(S, 60, 24, device=dev), torch.randn(S, 60, 3, device=dev), torch.randn(S, 60, 6, device=dev)
        out = torch.empty(S, window + 1, 24, device=dev)
        res = []
        for fn_i, fn in enumerate((lambda: nat.predict(lat, disp, hts, window, out=out), lambda: torch_block(lat, disp, hts, window))[:1 if NATIVE_ONLY else 2]):
            for _ in range(2):
                fn()
            if fn_i == 0:  # the steady shader clock (profiles/r05_clock_ramp.txt): 60 ms of the same launch first
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.06:
                    for _ in range(20):
                        fn()
            torch.cuda.synchronize()
            n = 5 if S >= 1024 else 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); e1.synchronize()
            res.append(e0.elapsed_time(e1) / n)
        fl = S * block_flops(window)
        print(f"window {window:2d} S={S:5d}: native {res[0]:9.3f} ms" + ("" if NATIVE_ONLY else f"   torch ops {res[1]:9.3f} ms")
              + f"   | {fl / 1e9:8.3f} GFLOP -> {fl / (res[0] * 1e-3) / 1e12:6.2f} TFLOP/s = {fl / (res[0] * 1e-3) / 157.3e12:.3f} of the fp32 MFMA roofline", flush=True)

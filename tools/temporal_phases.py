#!/usr/bin/env python3
"""Diagnostic: where the cycles of ONE dp_temporal_predict launch go (the -DDPT_STAMPS build, tools/temporal_phases.sh).
Usage: tools/temporal_phases.py [S=1] [window=0]   (prints per phase kind: count, shader cycles per call, total)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dragposer_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "_scratch", "lib_tstamps.so")
from dragposer_amd.temporal import NativeTemporal, TemporalPredictor

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
S = int(_pos[0]) if len(_pos) > 0 else 1
window = int(_pos[1]) if len(_pos) > 1 else 0
torch.manual_seed(0)
dev = torch.device("cuda:0")
nat = NativeTemporal(TemporalPredictor().eval(), torch.zeros(24), torch.ones(24), device=dev)
lib = nat._lib
lib.dp_temporal_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lat, disp, hts = torch.randn(S, 60, 24, device=dev), torch.randn(S, 60, 3, device=dev), torch.randn(S, 60, 6, device=dev)
out = torch.empty(S, window + 1, 24, device=dev)
buf = (C.c_ulonglong * 4096)()
for _ in range(300):  # (the steady clock)
    nat.predict(lat, disp, hts, window, out=out)
    if _ % 50 == 0:
        lib.dp_temporal_debug_read_stamps(buf, 4096)
lib.dp_temporal_debug_read_stamps(buf, 4096)
nat.predict(lat, disp, hts, window, out=out)
n = lib.dp_temporal_debug_read_stamps(buf, 4096)
NAMES = {0: "entry", 1: "lin_qkv", 2: "attention", 3: "out_proj", 4: "add_ln", 5: "ffn", 10: "tokens", 11: "in_proj_enc", 12: "enc_norm+mem", 13: "in_proj_dec",
         14: "final_ln+next_token", 15: "store", 20: "att: K in registers", 21: "att: turns done", 22: "qkv: wave 0 done", 23: "ffn: tiles done",
         24: "ffn: reduced + published", 25: "ln: wave 0 done"}
st = [(int(buf[i]) >> 48, int(buf[i]) & 0xFFFFFFFFFFFF) for i in range(n)]
tot = {}
for (i0, t0), (i1, t1) in zip(st[:-1], st[1:]):
    c, t = tot.get(i1, (0, 0))
    tot[i1] = (c + 1, t + (t1 - t0))
whole = st[-1][1] - st[0][1]
GHZ = 2.4  # (s_memtime counts shader cycles; the steady clock under this load is 2.38-2.40 GHz: profiles/r05_clock_ramp.txt)
print(f"S = {S}, window {window}: {n} stamps, {whole} shader cycles = {whole / GHZ / 1e3:.1f} us at {GHZ} GHz from entry to the last store "
      f"(a stamp costs about 200 cycles: the unstamped kernel is that much faster per stamp)")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {NAMES.get(k, k):>26}: {c:4d} x  {t / c:8.0f} cycles = {t / GHZ / 1e3:6.1f} us  ({100.0 * t / whole:4.1f} %)")
if "--timeline" in sys.argv:
    print("timeline (cycles since entry, delta, stamp):")
    for (i0, t0), (i1, t1) in zip(st[:-1], st[1:]):
        print(f"  {t1 - st[0][1]:8d} {t1 - t0:7d}  {NAMES.get(i1, i1)}")

#!/usr/bin/env python3
"""Soak of the temporal kernel's team exchange: launches of random sequence counts (1 ... 128: team sizes 8 / 4 / 2 / none in turn over ONE exchange area),
random windows and inputs, back to back; every launch compared with the one-workgroup-per-sequence kernel on the same inputs, the handle's status word
read at the end.  Usage: tools/team_soak.py [seconds=60]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dragposer_amd.temporal import NativeTemporal, TemporalPredictor

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
torch.manual_seed(7)
dev = torch.device("cuda:0")
model = TemporalPredictor().eval()
for p in model.parameters():
    if p.dim() == 1:
        p.data.add_(0.1 * torch.randn_like(p))
team = NativeTemporal(model, torch.zeros(24), torch.ones(24), device=dev)   # the library's choice: teams up to 128 sequences
solo = NativeTemporal(model, torch.zeros(24), torch.ones(24), device=dev)
solo._force_variant(21)
g = torch.Generator(device="cpu").manual_seed(3)
t0, n, worst, by_size = time.time(), 0, 0.0, {}
while time.time() - t0 < seconds:
    S = int(torch.randint(1, 129, (1,), generator=g))
    window = 4 * int(torch.randint(0, 16, (1,), generator=g))
    lat, disp, hts = torch.randn(S, 60, 24, generator=g).to(dev), (0.02 * torch.randn(S, 60, 3, generator=g)).to(dev), (1.0 + 0.3 * torch.randn(S, 60, 6, generator=g)).to(dev)
    reps = int(torch.randint(1, 6, (1,), generator=g))
    outs = [team.predict(lat, disp, hts, window) for _ in range(reps)]  # (back to back: tags count on within and across launches)
    want = solo.predict(lat, disp, hts, window)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), (S, window)
    err = float((outs[0] - want).abs().max())
    assert err <= 1e-5, (S, window, err)
    worst = max(worst, err)
    G = int(team._lib.dp_temporal_debug_team_size(256, S, 2048))  # (the library's own rule: all teams of a launch on at most half the CUs)
    by_size[G] = by_size.get(G, 0) + reps
    n += reps
assert team._team_status() == 0
print(f"{n} team launches in {time.time() - t0:.0f} s (by team size {dict(sorted(by_size.items()))}), sequences 1 ... 128, windows 0 ... 60: every launch equal to its repeats bit for bit, "
      f"max |team - one workgroup per sequence| = {worst:.2e}, status word 0")

#!/usr/bin/env python3
"""How far do the PRODUCT's own closed-loop runs of a recorded clip drift apart?  tests/golden/f1_*.npz hold the reference's run of a clip and
its twin (the same run from an initial latent moved by 1e-7); this runs the product's CLI on the same clip from the recorded initial latent
and from perturbed ones, and prints the sequence-level figures side by side.  GPU box: python tools/clip_twins.py f1_clip4 [f1_clip4_t ...]"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_torch as R  # noqa: E402  (model path only)


def run(g, meta, z0, tmp):
    from dragposer_amd import eval_drag as E

    cfg_path, z0_path = os.path.join(tmp, "cfg.json"), os.path.join(tmp, "z0.npy")
    with open(cfg_path, "w") as f:
        json.dump(meta["cfg"], f)
    np.save(z0_path, z0)
    bvh = os.path.join(ROOT, "tests", "data", meta["bvh"])
    argv = [R.DEFAULT_MODEL, bvh, "--config", cfg_path, "--initial-latent", z0_path, "--out-dir", os.path.join(tmp, "data")]
    if meta["temporal_on"]:
        sd = {k[len("temporal."):]: torch.tensor(g[k]) for k in g.files if k.startswith("temporal.")}
        ck = os.path.join(tmp, "temporal.pt")
        torch.save({"model_state_dict": sd, "means_latent": torch.tensor(g["means_latent"]), "stds_latent": torch.tensor(g["stds_latent"])}, ck)
        argv += ["--temporal-checkpoint", ck]
    return E.main(argv + ["--keep-frames"])[0]


for name in sys.argv[1:] or ["f1_clip4"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    print(f"## {name}: the reference: iterations/frame {g['iters'].mean():.2f}, MPJPE {float(g['mpjpe']) * 1000:.2f} mm, MPEEPE {float(g['mpeepe']) * 1000:.2f}; "
          f"its twin (1e-7): {g['twin_iters'].mean():.2f}, {float(g['twin_mpjpe']) * 1000:.2f}, {float(g['twin_mpeepe']) * 1000:.2f}; iteration counts equal on "
          f"{(g['twin_iters'] == g['iters']).mean():.3f} of the frames")
    for eps in (0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, 1e-4):
        with tempfile.TemporaryDirectory() as tmp:
            res = run(g, meta, g["initial_latent"] + np.float32(eps), tmp)
        same = res["iters"] == g["iters"]
        print(f"   product, initial latent + {eps:g}: iterations/frame {res['iters'].mean():.2f}, MPJPE {res['mpjpe'] * 1000:.2f} mm, MPEEPE {res['mpeepe'] * 1000:.2f}; "
              f"iteration counts equal to the reference's on {same.mean():.3f} of the frames (first difference at {int(np.argmin(same)) if not same.all() else -1}); "
              f"iterations by quarter {[round(float(res['iters'][a:a + 60].mean()), 2) for a in range(0, 240, 60)]} (reference {[round(float(g['iters'][a:a + 60].mean()), 2) for a in range(0, 240, 60)]})", flush=True)

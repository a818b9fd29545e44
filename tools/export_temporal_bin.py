#!/usr/bin/env python3
"""Temporal predictor -> temporal.bin, the flat file libDragPoserDLL.so's load_models() looks for beside dragposer_model.bin.
Source: the reference's `temporal.pt` (train_temporal.py:455-482: model_state_dict + means_latent + stds_latent), or a
sequence fixture (tests/golden/seq*.npz) that carries a reference Temporal state_dict under "temporal.*".
Format: DPM1 (tools/export_model_bin.py).  Usage: tools/export_temporal_bin.py SRC DST"""
import struct
import sys

import numpy as np


def tensors_from(src):
    if src.endswith(".npz"):
        g = np.load(src)
        out = {k[len("temporal."):]: g[k] for k in g.files if k.startswith("temporal.")}
        out["means_latent"], out["stds_latent"] = g["means_latent"], g["stds_latent"]
    else:
        import torch

        ck = torch.load(src, map_location="cpu")
        out = {k: v.numpy() for k, v in ck["model_state_dict"].items()}
        out["means_latent"], out["stds_latent"] = ck["means_latent"].numpy().reshape(-1), ck["stds_latent"].numpy().reshape(-1)
    out["sample_step"] = np.array([4.0], np.float32)  # train_temporal.py:15
    return out


def write(tensors, dst):
    with open(dst, "wb") as f:
        f.write(b"DPM1" + struct.pack("<I", len(tensors)))
        for k, v in tensors.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            name = k.encode()
            f.write(struct.pack("<I", len(name)) + name + struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())


if __name__ == "__main__":
    write(tensors_from(sys.argv[1]), sys.argv[2])
    print("wrote", sys.argv[2])

#!/usr/bin/env python3
"""model fixture (.npz) -> dragposer_model.bin, the flat file libDragPoserDLL.so's load_models() reads.
Format: "DPM1", u32 n; then n x { u32 name_len, name, u32 ndim, u32 dims[ndim], f32 data[...] } (int arrays as f32)."""
import struct
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "dragposer_amd/data/model_dancedb.npz"
dst = sys.argv[2] if len(sys.argv) > 2 else "dragposer_amd/data/dragposer_model.bin"
raw = np.load(src)
with open(dst, "wb") as f:
    f.write(b"DPM1" + struct.pack("<I", len(raw.files)))
    for k in raw.files:
        a = np.ascontiguousarray(raw[k], dtype=np.float32)
        name = k.encode()
        f.write(struct.pack("<I", len(name)) + name + struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape))
        f.write(a.tobytes())
print("wrote", dst)

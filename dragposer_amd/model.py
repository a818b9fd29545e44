"""Model fixture -> dp_model (host arrays handed to dp_create / dp_fold_decoder).

The fixture dragposer_amd/data/model_dancedb.npz holds the reference checkpoint's tensors
(state_dict keys of python/models/model_dancedb/generator.pt without the ``autoencoder.``
prefix, train.py:257-269) plus data.pt's means/stds and the skeleton of example.bvh.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

NJ = 22
DEFAULT_MODEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "model_dancedb.npz")


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class HostModel:
    """Keeps the numpy arrays alive and exposes them as a dp_model struct."""

    def __init__(self, path=DEFAULT_MODEL, weight_dtype="fp32", arrays=None):
        raw = arrays if arrays is not None else np.load(path)
        self.arrays = {
            "f_latent_w": _c32(raw["decoder.f_latent.weight"]),
            "f_latent_b": _c32(raw["decoder.f_latent.bias"]),
            "mean_q": _c32(np.asarray(raw["means.dqs"]).reshape(NJ, 8)[:, :4].reshape(88)),  # drag_pose.py:27-29
            "std_q": _c32(np.asarray(raw["stds.dqs"]).reshape(NJ, 8)[:, :4].reshape(88)),
            "mean_disp": _c32(raw["means.displacement"]),
            "std_disp": _c32(raw["stds.displacement"]),
            "offsets": _c32(raw["offsets"]),
        }
        self.parents = np.ascontiguousarray(raw["parents"], dtype=np.int32)
        for l in range(3):
            self.arrays[f"unpool_w{l}"] = _c32(raw[f"decoder.layers.{l}.0.weight"])
            self.arrays[f"conv_w{l}"] = _c32(np.asarray(raw[f"decoder.layers.{l}.1.weight"])[..., 0])
            self.arrays[f"conv_mask{l}"] = _c32(np.asarray(raw[f"decoder.layers.{l}.1.mask"])[..., 0])
            self.arrays[f"conv_b{l}"] = _c32(raw[f"decoder.layers.{l}.1.bias"])
        self.weight_dtype = {"fp32": _lib.DP_WEIGHTS_FP32, "bf16": _lib.DP_WEIGHTS_BF16}[weight_dtype]
        fp = lambda k: self.arrays[k].ctypes.data_as(C.POINTER(C.c_float))
        m = _lib.DpModel()
        m.f_latent_w, m.f_latent_b = fp("f_latent_w"), fp("f_latent_b")
        for l in range(3):
            m.unpool_w[l], m.conv_w[l] = fp(f"unpool_w{l}"), fp(f"conv_w{l}")
            m.conv_mask[l], m.conv_b[l] = fp(f"conv_mask{l}"), fp(f"conv_b{l}")
        m.mean_q, m.std_q, m.mean_disp, m.std_disp = fp("mean_q"), fp("std_q"), fp("mean_disp"), fp("std_disp")
        m.parents = self.parents.ctypes.data_as(C.POINTER(C.c_int))
        m.offsets = fp("offsets")
        m.weight_dtype = self.weight_dtype
        self.struct = m

    def fold(self):
        """dp_fold_decoder -> dict of the folded matrices (host only, no GPU needed)."""
        lib = _lib.load()
        out = _lib.DpFolded()
        rc = lib.dp_fold_decoder(C.byref(self.struct), C.byref(out))
        if rc != _lib.DP_OK:
            raise _lib.DragPoserError(rc, _lib.last_error())
        g = lambda name, shape: np.ctypeslib.as_array(getattr(out, name)).reshape(shape).copy()
        return dict(A0=g("A0", (40, 24)), c0=g("c0", (40,)), A1=g("A1", (60, 40)), b1=g("b1", (60,)),
                    A2=g("A2", (92, 60)), b2=g("b2", (92,))), out

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


def load_reference_model_folder(model_dir, skeleton_bvh=None):
    """The reference CLI's `model_path` (python/src/eval_drag.py:260-264): a FOLDER holding `generator.pt`
    ({"model_state_dict": state_dict of Generator_Model}, train.py:297-303) and `data.pt` ({"means": {"dqs",
    "displacement"}, "stds": {...}}, train.py:288-296), plain torch.load-able dicts of tensors -> the array dict HostModel /
    PoseEncoder read (the keys of data/model_dancedb.npz).  The skeleton is not part of those files: the reference takes
    parents and offsets from the BVH it evaluates (train.get_info_from_bvh, train.py:329-341), so does this
    (`skeleton_bvh`: a path or a loaded dragposer_amd.bvh.BVH)."""
    import torch

    sd = torch.load(os.path.join(model_dir, "generator.pt"), map_location="cpu", weights_only=True)["model_state_dict"]
    data = torch.load(os.path.join(model_dir, "data.pt"), map_location="cpu", weights_only=True)
    arrs = {k[len("autoencoder."):] if k.startswith("autoencoder.") else k: v.detach().cpu().numpy() for k, v in sd.items()}
    for grp in ("means", "stds"):
        arrs[f"{grp}.dqs"] = data[grp]["dqs"].detach().cpu().numpy().reshape(-1)
        arrs[f"{grp}.displacement"] = data[grp]["displacement"].detach().cpu().numpy().reshape(-1)
    if skeleton_bvh is not None:
        from .bvh import BVH

        bvh = skeleton_bvh if hasattr(skeleton_bvh, "get_data") else BVH().load(skeleton_bvh)
        _, _, parents, offsets, _ = bvh.get_data()
        arrs["parents"] = np.asarray(parents, dtype=np.int32).copy()
        arrs["parents"][0] = 0                                     # train.py:338
        arrs["offsets"] = np.asarray(offsets, dtype=np.float32).copy()
        arrs["offsets"][0] = 0.0                                   # train.py:340
    return arrs


def load_model_arrays(model_path, skeleton_bvh=None):
    """`model_path`: the reference's model folder (above) or this package's flat .npz fixture"""
    if os.path.isdir(model_path):
        return load_reference_model_folder(model_path, skeleton_bvh)
    return dict(np.load(model_path))

"""ORACLE (test infrastructure, not product): ctypes front-end of oracle/analytic.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_MODEL = os.path.join(os.path.dirname(HERE), "dragposer_amd", "data", "model_dancedb.npz")
NJ = 22
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_u8 = C.POINTER(C.c_ubyte)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def _lib(precision):
    path = os.path.join(HERE, "_build", f"liboracle_{precision}.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.ora_create.restype = C.c_void_p
    lib.ora_create.argtypes = [_f] * 18 + [_i, _f]
    lib.ora_destroy.argtypes = [C.c_void_p]
    lib.ora_get_folded.argtypes = [C.c_void_p] + [_f] * 6
    lib.ora_forward.argtypes = [C.c_void_p, C.c_int] + [_f] * 8
    lib.ora_grad.argtypes = [C.c_void_p, C.c_int] + [_f] * 6 + [_u8, C.c_float, C.c_float, _f, _f]
    lib.ora_optimize.argtypes = (
        [C.c_void_p, C.c_int] + [_f] * 6 + [_u8, C.c_int] + [C.c_double] * 4 + [C.c_float, C.c_float]
        + [C.c_int, C.c_float, C.c_float, C.c_float] + [_f] * 9 + [_i]
    )
    return lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def round_bf16(a):
    """fp32 -> nearest-even bf16 -> fp32 (numpy)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


class AnalyticOracle:
    def __init__(self, model_path=DEFAULT_MODEL, precision="f32", weight_rounding="none"):
        self.lib = _lib(precision)
        raw = np.load(model_path)
        rnd = round_bf16 if weight_rounding == "bf16" else (lambda x: x)
        args = [_c32(rnd(raw["decoder.f_latent.weight"])), _c32(raw["decoder.f_latent.bias"])]
        for l in range(3):
            args += [
                _c32(raw[f"decoder.layers.{l}.0.weight"]),
                _c32(rnd(raw[f"decoder.layers.{l}.1.weight"][..., 0])),
                _c32(raw[f"decoder.layers.{l}.1.mask"][..., 0]),
                _c32(raw[f"decoder.layers.{l}.1.bias"]),
            ]
        mu4 = _c32(raw["means.dqs"].reshape(NJ, 8)[:, :4].reshape(88))
        sd4 = _c32(raw["stds.dqs"].reshape(NJ, 8)[:, :4].reshape(88))
        args += [mu4, sd4, _c32(raw["means.displacement"]), _c32(raw["stds.displacement"])]
        parents = np.ascontiguousarray(raw["parents"], dtype=np.int32)
        offsets = _c32(raw["offsets"])
        self._keep = args + [parents, offsets]
        self.h = C.c_void_p(self.lib.ora_create(*[_fp(a) for a in args], parents.ctypes.data_as(_i), _fp(offsets)))

    def __del__(self):
        try:
            self.lib.ora_destroy(self.h)
        except Exception:
            pass

    def folded(self):
        out = [np.zeros(s, np.float32) for s in ((40, 24), (40,), (60, 40), (60,), (92, 60), (92,))]
        self.lib.ora_get_folded(self.h, *[_fp(a) for a in out])
        return dict(zip(("A0", "c0", "A1", "b1", "A2", "b2"), out))

    def forward(self, z, cur_rot):
        z, cur_rot = _c32(z), _c32(cur_rot)
        B = z.shape[0]
        o = dict(pose=np.zeros((B, 88), np.float32), disp_norm=np.zeros((B, 3), np.float32),
                 world_disp=np.zeros((B, 3), np.float32), world_rot=np.zeros((B, 4), np.float32),
                 pos=np.zeros((B, NJ, 3), np.float32), rot=np.zeros((B, NJ, 9), np.float32))
        self.lib.ora_forward(self.h, B, _fp(z), _fp(cur_rot), *[_fp(o[k]) for k in
                             ("pose", "disp_norm", "world_disp", "world_rot", "pos", "rot")])
        return o

    def grad(self, z, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, lam_rot=1.0, lam_tmp=0.02):
        z, z_tgt, cur_rot, tgt_pos, tgt_rot, w = map(_c32, (z, z_tgt, cur_rot, tgt_pos, tgt_rot, w))
        tracked = np.ascontiguousarray(tracked, dtype=np.uint8)
        B = z.shape[0]
        loss, grad = np.zeros((B, 3), np.float32), np.zeros((B, 24), np.float32)
        self.lib.ora_grad(self.h, B, _fp(z), _fp(z_tgt), _fp(cur_rot), _fp(tgt_pos), _fp(tgt_rot), _fp(w),
                          tracked.ctypes.data_as(_u8), lam_rot, lam_tmp, _fp(loss), _fp(grad))
        return loss, grad

    def optimize(self, z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter, lr=1e-2, betas=(0.9, 0.999),
                 eps=1e-8, lam_rot=1.0, lam_tmp=0.02, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=None):
        z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w = map(_c32, (z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w))
        tracked = np.ascontiguousarray(tracked, dtype=np.uint8)
        B = z0.shape[0]
        o = dict(z_final=np.zeros((B, 24), np.float32), z_pre=np.zeros((B, 24), np.float32),
                 pose=np.zeros((B, 88), np.float32), disp_norm=np.zeros((B, 3), np.float32),
                 world_disp=np.zeros((B, 3), np.float32), world_rot=np.zeros((B, 4), np.float32),
                 pos=np.zeros((B, NJ, 3), np.float32), rot=np.zeros((B, NJ, 9), np.float32),
                 loss=np.zeros((B, 3), np.float32))
        iters = np.zeros((B,), np.int32)
        early = min_loss_incr is not None or stop_eps_pos > 0 or stop_eps_rot > 0
        self.lib.ora_optimize(
            self.h, B, _fp(z0), _fp(z_tgt), _fp(cur_rot), _fp(tgt_pos), _fp(tgt_rot), _fp(w),
            tracked.ctypes.data_as(_u8), n_iter, lr, betas[0], betas[1], eps, lam_rot, lam_tmp,
            int(early), stop_eps_pos, stop_eps_rot, -np.inf if min_loss_incr is None else min_loss_incr,
            *[_fp(o[k]) for k in ("z_final", "z_pre", "pose", "disp_norm", "world_disp", "world_rot", "pos", "rot", "loss")],
            iters.ctypes.data_as(_i))
        o["iters"] = iters
        return o

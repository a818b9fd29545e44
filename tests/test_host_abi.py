"""CPU: the C ABI of include/dragposer.h driven END TO END without a GPU, through the HOST-ONLY build of it that SURVEY 8(b) asks for
(oracle/host_abi.c -> oracle/_build/libdragposer_hostonly.so: the same entry points, structs and status codes, implemented by the
plain-C restatement of the reference, "device" pointers read as host pointers).  Test infrastructure: the product never loads it
(dragposer_amd has no CPU fallback, tests/test_abi.py); this file loads it by explicit path.

What it buys: the structs of dragposer_amd/_lib.py (dp_model, dp_batch, dp_params, dp_result), the call sequence dp_create ->
dp_optimize / dp_forward -> dp_destroy, optional result pointers, argument validation and status codes are exercised here, on the
golden vectors the REAL reference produced -- the same calls tests/test_hip_parity.py makes on the GPU box."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dragposer_amd import _lib
from dragposer_amd.model import HostModel
from oracle import ref_torch as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_build", "libdragposer_hostonly.so")
KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w")
OUT = {"z": 24, "z_pre": 24, "pose": 88, "disp": 3, "world_disp": 3, "world_rot": 4, "pos": 66, "rot": 198, "loss": 3}


@pytest.fixture(scope="module")
def lib():
    lib = C.CDLL(PATH)
    lib.dp_last_error.restype = C.c_char_p
    lib.dp_last_error.argtypes = [C.c_void_p]
    lib.dp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_lib.DpModel), C.c_int]
    lib.dp_destroy.argtypes = [C.c_void_p]
    lib.dp_optimize.argtypes = [C.c_void_p, C.POINTER(_lib.DpBatch), C.POINTER(_lib.DpParams), C.POINTER(_lib.DpResult), C.c_void_p]
    lib.dp_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_lib.DpResult), C.c_void_p]
    return lib


def _ctx(lib, wd="fp32"):
    hm = HostModel(weight_dtype=wd)
    ctx = C.c_void_p()
    assert lib.dp_create(C.byref(ctx), C.byref(hm.struct), 0) == _lib.DP_OK, lib.dp_last_error(None)
    return ctx, hm


def _batch(g, n=None):
    b = _lib.DpBatch()
    arrs = {k: np.ascontiguousarray(g[k][:n], np.float32) for k in KEYS}
    arrs["tracked"] = np.ascontiguousarray(g["tracked"][:n], np.uint8)
    b.n_frames = len(arrs["z0"])
    for k, v in arrs.items():
        setattr(b, k, v.ctypes.data)
    return b, arrs


def _params(mt, **over):
    early = mt.get("early_stop", False)
    kw = dict(n_iter=mt["n_iter"], lr=mt["lr"], beta1=0.9, beta2=0.999, eps=1e-8, lambda_rot=1.0, lambda_tmp=mt["lambda_tmp"], early_stop=int(early),
              stop_eps_pos=mt["stop_eps_pos"], stop_eps_rot=mt["stop_eps_rot"], min_loss_incr=mt["min_loss_incr"] if early else float("-inf"),
              max_trackers=0, kernel=_lib.DP_KERNEL_AUTO)
    kw.update(over)
    return _lib.DpParams(**kw)


def _results(B, names=tuple(OUT) + ("iters",)):
    r, arrs = _lib.DpResult(), {}
    for n in names:
        arrs[n] = np.zeros((B,), np.int32) if n == "iters" else np.zeros((B, OUT[n]), np.float32)
        setattr(r, n, arrs[n].ctypes.data)
    return r, arrs


def test_exports_every_symbol_of_the_header(lib):
    hdr = open(os.path.join(ROOT, "include", "dragposer.h")).read()
    for sym in set(re.findall(r"^(?:int|const char\*)\s+(dp_\w+)\s*\(", hdr, flags=re.M)):
        assert hasattr(lib, sym), sym
    assert lib.dp_version() == 510
    assert "libdragposer_hostonly" not in open(os.path.join(ROOT, "dragposer_amd", "_lib.py")).read()  # the product does not know it


@pytest.mark.parametrize("name", ["s1", "s4", "es"])
def test_goldens_through_the_abi(lib, golden_dir, name):
    """the reference's recorded runs through dp_create / dp_optimize of the host-only build: every field of dp_result"""
    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = g["meta"]
    ctx, hm = _ctx(lib, "bf16" if mt["weight_rounding"] == "bf16" else "fp32")
    b, keep = _batch(g)
    r, o = _results(b.n_frames)
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(_params(mt)), C.byref(r), None) == _lib.DP_OK, lib.dp_last_error(ctx)
    assert (np.linalg.norm(o["pos"].reshape(-1, 22, 3) - g["pos"], axis=-1) * 1000).max() <= 0.05
    np.testing.assert_array_equal(o["iters"], g["iters"])
    np.testing.assert_allclose(o["z"], g["z_final"], atol=2e-5)
    np.testing.assert_allclose(o["z_pre"], g["z_pre"], atol=2e-5)
    np.testing.assert_allclose(o["world_rot"], g["world_rot"], atol=2e-6)
    np.testing.assert_allclose(o["world_disp"], g["world_disp"], atol=2e-7)
    np.testing.assert_allclose(o["rot"].reshape(-1, 22, 9), g["rot"], atol=2e-5)
    np.testing.assert_allclose(o["pose"], g["pose"], atol=2e-3)
    raw = np.load(R.DEFAULT_MODEL)
    np.testing.assert_allclose(o["disp"], g["disp_norm"] * raw["stds.displacement"] + raw["means.displacement"], atol=1e-7)  # metres, de-normalised
    last = g["loss_hist"][np.arange(len(g["iters"])), g["iters"] - 1]
    np.testing.assert_allclose(o["loss"], last, rtol=2e-3, atol=1e-8)
    assert lib.dp_destroy(ctx) == _lib.DP_OK


def test_optional_results_and_forward(lib, golden_dir):
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    ctx, hm = _ctx(lib)
    b, keep = _batch(g, 5)
    full_r, full = _results(5)
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(_params(g["meta"], n_iter=7)), C.byref(full_r), None) == _lib.DP_OK
    r, o = _results(5, ("z", "pos"))  # any result pointer may be NULL
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(_params(g["meta"], n_iter=7)), C.byref(r), None) == _lib.DP_OK
    np.testing.assert_array_equal(o["z"], full["z"])
    np.testing.assert_array_equal(o["pos"], full["pos"])
    fr, fo = _results(5, ("pose", "disp", "world_disp", "world_rot", "pos", "rot"))
    zs, cr = np.ascontiguousarray(g["z_src"][:5]), np.ascontiguousarray(g["cur_rot"][:5])
    assert lib.dp_forward(ctx, 5, zs.ctypes.data, cr.ctypes.data, C.byref(fr), None) == _lib.DP_OK
    trk = g["tracked"][:5].astype(bool)
    assert np.abs(fo["pos"].reshape(5, 22, 3)[trk] - g["tgt_pos"][:5][trk]).max() < 5e-6  # recipe S: the targets ARE this forward pass
    lib.dp_destroy(ctx)


def test_argument_validation_and_status_codes(lib, golden_dir):
    """the refusals of dp_host.cpp, one for one (tests/test_hip_parity.py::test_argument_validation makes the same calls on the GPU)"""
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    ctx, hm = _ctx(lib)
    b, keep = _batch(g, 4)
    r, o = _results(4)
    mt = g["meta"]
    for bad, msg in ((dict(n_iter=0), b"n_iter"), (dict(n_iter=_lib.DP_MAX_ITERS + 1), b"n_iter"), (dict(lr=0.0), b"Adam"), (dict(beta1=1.0), b"Adam"),
                     (dict(eps=0.0), b"eps must be > 0"), (dict(kernel=7), b"kernel selector")):
        assert lib.dp_optimize(ctx, C.byref(b), C.byref(_params(mt, **bad)), C.byref(r), None) == _lib.DP_ERR_INVALID, bad
        assert msg in lib.dp_last_error(ctx), (bad, lib.dp_last_error(ctx))
    b0, _ = _batch(g, 4)
    b0.n_frames = 0
    assert lib.dp_optimize(ctx, C.byref(b0), C.byref(_params(mt)), C.byref(r), None) == _lib.DP_ERR_INVALID
    b1, _ = _batch(g, 4)
    b1.w = None
    assert lib.dp_optimize(ctx, C.byref(b1), C.byref(_params(mt)), C.byref(r), None) == _lib.DP_ERR_INVALID and b"NULL input" in lib.dp_last_error(ctx)
    assert lib.dp_optimize(None, C.byref(b), C.byref(_params(mt)), C.byref(r), None) == _lib.DP_ERR_INVALID
    empty = _lib.DpModel()
    c2 = C.c_void_p()
    assert lib.dp_create(C.byref(c2), C.byref(empty), 0) == _lib.DP_ERR_INVALID and b"NULL" in lib.dp_last_error(None)
    lib.dp_optimize_sequence.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    lib.dp_optimize_sequence.restype = C.c_int
    assert lib.dp_optimize_sequence(ctx, 1, None, None, None, None, None, None, None) == _lib.DP_ERR_UNSUPPORTED
    lib.dp_destroy(ctx)


def test_a_struct_compiled_against_another_header_is_refused(lib, golden_dir):
    """include/dragposer.h 0.5.0: dp_params / dp_result start with struct_size.  A 0.4 caller's dp_params starts with n_iter (<= 256)
    and its dp_result with the pointer `z` -- both land in the size word (and reserved0) and are refused instead of read past."""
    import ctypes as C

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    ctx, hm = _ctx(lib)
    b, keep = _batch(g, 4)
    p = _params(g["meta"])
    r, arrs = _results(4)
    assert p.struct_size == C.sizeof(_lib.DpParams) and r.struct_size == C.sizeof(_lib.DpResult) and r.reserved0 == 0
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(p), C.byref(r), None) == _lib.DP_OK
    p.struct_size = 50  # a 0.4 dp_params: n_iter = 50 in the first word
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(p), C.byref(r), None) == _lib.DP_ERR_INVALID
    assert b"struct_size" in lib.dp_last_error(ctx)
    # a 0.4 dp_params that asks for 100 iterations (the sequence default): 13 words = 52 bytes, n_iter first.  100 passes for a size; the second
    # word is lr = 0.01, an iteration count of 1 008 981 770: refused on the two words every layout has, before anything is copied
    old = (C.c_uint32 * 13)()
    words = np.array([100, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.uint32)
    words[1:7] = np.array([1e-2, 0.9, 0.999, 1e-8, 1.0, 0.02], np.float32).view(np.uint32)
    old[:] = [int(w) for w in words]
    assert lib.dp_optimize(ctx, C.byref(b), C.cast(old, C.POINTER(_lib.DpParams)), C.byref(r), None) == _lib.DP_ERR_INVALID
    assert b"n_iter" in lib.dp_last_error(ctx) and b"pre-0.5" in lib.dp_last_error(ctx)
    p.struct_size = C.sizeof(_lib.DpParams)
    r.struct_size, r.reserved0 = 0x1a2b3c40, 0x7f12  # a 0.4 dp_result: the two halves of a device pointer
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(p), C.byref(r), None) == _lib.DP_ERR_INVALID
    r.struct_size, r.reserved0 = C.sizeof(_lib.DpResult), 0
    # the new result words: status (the restatement screens nothing: it reports what came out) and clock (zeros on a CPU)
    st, ck = np.full(4, -1, np.int32), np.full(2, 7, np.uint64)
    r.status, r.clock = st.ctypes.data, ck.ctypes.data
    assert lib.dp_optimize(ctx, C.byref(b), C.byref(p), C.byref(r), None) == _lib.DP_OK
    assert (st == 0).all() and (ck == 0).all()
    lib.dp_destroy(ctx)

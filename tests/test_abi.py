"""CPU: the C-ABI library loads, exports every symbol include/dragposer.h declares, and its
host-only logic (decoder folding, skeleton tables, argument validation) is right.  No compute
call is made here: without a GPU dp_create must fail loudly with DP_ERR_DEVICE."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from dragposer_amd import _lib
from dragposer_amd.model import HostModel
from oracle.analytic import AnalyticOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dragposer.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(dp_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.PUBLIC_SYMBOLS), declared ^ set(_lib.PUBLIC_SYMBOLS)
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.dp_version() == 510


def test_struct_layouts_match_header_sizes(tmp_path):
    """the ctypes mirrors of _lib.py against the C compiler's own view of include/dragposer.h: sizeof every struct, offsetof the
    fields the 0.5.0 size words guard (a binding that drifts from the header would otherwise only show up as garbage on the GPU)"""
    import shutil
    import subprocess

    ptr = C.sizeof(C.c_void_p)
    assert C.sizeof(_lib.DpModel) == 20 * ptr + 8  # 20 pointers + int (+pad)
    assert C.sizeof(_lib.DpBatch) == 8 + 7 * ptr
    assert C.sizeof(_lib.DpParams) == 14 * 4       # struct_size + 13 fields
    assert C.sizeof(_lib.DpResult) == 8 + 12 * ptr  # struct_size, reserved0, 12 pointers
    assert C.sizeof(_lib.DpSeqResults) == 8 + 7 * ptr
    assert C.sizeof(_lib.DpSeqState) == 5 * ptr + 10 * 4  # 5 pointers, history, n_heights, height_joints[8]
    assert C.sizeof(_lib.DpSeqStep) == 16 + 3 * ptr      # 2 ints + float (+pad), 3 pointers
    assert C.sizeof(_lib.DpFolded) == 4 * (40 * 24 + 40 + 60 * 40 + 60 + 92 * 60 + 92)
    if shutil.which("gcc") is None:
        pytest.skip("no gcc: sizes checked against the arithmetic above only")
    pairs = [("dp_model", _lib.DpModel), ("dp_batch", _lib.DpBatch), ("dp_params", _lib.DpParams), ("dp_result", _lib.DpResult),
             ("dp_seq_state", _lib.DpSeqState), ("dp_seq_step", _lib.DpSeqStep), ("dp_seq_frames", _lib.DpSeqFrames),
             ("dp_seq_results", _lib.DpSeqResults), ("dp_folded", _lib.DpFolded), ("dp_temporal_layer", _lib.DpTemporalLayer),
             ("dp_temporal_model", _lib.DpTemporalModel)]
    fields = [("dp_params", _lib.DpParams, "n_iter"), ("dp_params", _lib.DpParams, "kernel"), ("dp_result", _lib.DpResult, "z"),
              ("dp_result", _lib.DpResult, "iters"), ("dp_result", _lib.DpResult, "status"), ("dp_result", _lib.DpResult, "clock"),
              ("dp_seq_results", _lib.DpSeqResults, "pose_ret"), ("dp_seq_results", _lib.DpSeqResults, "status"),
              ("dp_seq_frames", _lib.DpSeqFrames, "z_tgt_seq"), ("dp_temporal_model", _lib.DpTemporalModel, "dec")]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "dragposer.h"\nint main(void) {\n'
                   + "".join(f'printf("%zu\\n", sizeof({c}));\n' for c, _ in pairs)
                   + "".join(f'printf("%zu\\n", offsetof({c}, {f}));\n' for c, _, f in fields)
                   + 'dp_params p = DP_PARAMS_INIT; dp_result r = DP_RESULT_INIT; dp_seq_results q = DP_SEQ_RESULTS_INIT;\n'
                     'printf("%u %u %u %d\\n", p.struct_size, r.struct_size, q.struct_size, p.n_iter + (r.z != 0) + (int)r.reserved0);\nreturn 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = subprocess.check_output([str(exe)]).decode().split("\n")
    want = [str(C.sizeof(t)) for _, t in pairs] + [str(getattr(t, f).offset) for _, t, f in fields]
    assert got[:len(want)] == want, list(zip([c for c, _ in pairs] + [c + "." + f for c, _, f in fields], got, want))
    assert got[len(want)] == f"{C.sizeof(_lib.DpParams)} {C.sizeof(_lib.DpResult)} {C.sizeof(_lib.DpSeqResults)} 0"


@pytest.mark.parametrize("wd", ["fp32", "bf16"])
def test_fold_decoder_matches_oracle_fold(wd):
    f, _ = HostModel(weight_dtype=wd).fold()
    fo = AnalyticOracle(weight_rounding="bf16" if wd == "bf16" else "none").folded()
    for k in f:
        np.testing.assert_array_equal(f[k], fo[k])
    assert np.count_nonzero(f["A2"]) < f["A2"].size  # joint-block structure survives folding


def test_fold_rejects_null_pointers():
    lib = _lib.load()
    m = _lib.DpModel()
    out = _lib.DpFolded()
    assert lib.dp_fold_decoder(C.byref(m), C.byref(out)) == _lib.DP_ERR_INVALID
    assert "NULL" in _lib.last_error()


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu():
    lib = _lib.load()
    hm = HostModel()
    ctx = C.c_void_p()
    rc = lib.dp_create(C.byref(ctx), C.byref(hm.struct), 0)
    assert rc == _lib.DP_ERR_DEVICE and not ctx.value
    assert "no CPU fallback" in _lib.last_error()


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_python_operator_has_no_cpu_fallback():
    from dragposer_amd.optimizer import LatentOptimizer

    with pytest.raises((RuntimeError, ValueError)):
        LatentOptimizer(device="cuda:0")
    with pytest.raises(ValueError):
        LatentOptimizer(device="cpu")


def test_item_table_and_skeleton_validation():
    import kernel_emu as KE

    hm = HostModel()
    _, _, items, smask = KE.host_tables(hm)
    assert 0 < bin(int(smask[3, KE.G_B2])).count("1") <= 26 and smask[7, KE.G_B1] == 0  # structural-zero step masks
    kinds = list(items["kind"])
    assert kinds[0] == 1 and kinds[1:22] == [0] * 21 and kinds[22] == 2
    assert kinds[23:25] == [3, 3] and list(items["src_quad"][23:25]) == [11, 11]  # joint 11's extra children
    assert sorted([int(items["ch_id"][11]), int(items["ch_id"][23]), int(items["ch_id"][24])]) == [12, 14, 18]
    # path of joint 17 (left wrist): 9,10,11,14,15,16,17
    p = [(int(items["path_lo"][17]) >> (5 * i)) & 31 for i in range(6)] + [int(items["path_hi"][17]) & 31]
    assert sorted(p) == [9, 10, 11, 14, 15, 16, 17]
    assert int(items["ch_sub"][22]) == (1 << 22) - 1
    # a non-topological parents array is rejected
    lib = _lib.load()
    bad = HostModel()
    bad.parents[5] = 7
    buf = np.zeros(32 * 32, np.float32)
    assert lib.dp_debug_items(C.byref(bad.struct), buf.ctypes.data_as(C.c_void_p)) == _lib.DP_ERR_INVALID


def test_skeleton_limits_are_reported():
    """The limits include/dragposer.h states: <= 3 root children, <= 3 extra child bones, chains of <= 7 bones."""
    lib = _lib.load()
    buf = np.zeros(32 * 32, np.float32)

    def rc_of(parents):
        hm = HostModel()
        hm.parents[:] = parents
        return lib.dp_debug_items(C.byref(hm.struct), buf.ctypes.data_as(C.c_void_p))

    ok = HostModel().parents.copy()
    assert rc_of(ok) == _lib.DP_OK
    star = np.zeros(22, np.int32)  # every joint a child of the root
    assert rc_of(star) == _lib.DP_ERR_UNSUPPORTED and "root" in _lib.last_error()
    chain = np.maximum(np.arange(22, dtype=np.int32) - 1, 0)  # one 21-bone chain
    assert rc_of(chain) == _lib.DP_ERR_UNSUPPORTED and "7 bones" in _lib.last_error()
    fan = np.array([0, 0] + [1] * 20, np.int32)  # joint 1 with 20 children
    assert rc_of(fan) == _lib.DP_ERR_UNSUPPORTED and "extra child" in _lib.last_error()


def test_temporal_predictor_argument_checks_and_no_cpu_fallback():
    """dp_temporal_create validates the model before it touches a device, and -- like dp_create -- has no CPU path."""
    import torch

    from dragposer_amd.temporal import NativeTemporal, TemporalPredictor

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.dp_temporal_create(C.byref(h), None, 0) == _lib.DP_ERR_INVALID and not h.value
    assert b"model is NULL" in lib.dp_temporal_last_error(None)
    m = _lib.DpTemporalModel()
    m.n_heights, m.dim_feedforward, m.n_encoder_layers, m.n_decoder_layers, m.max_len, m.sample_step = 6, 32, 9, 1, 30, 4
    assert lib.dp_temporal_create(C.byref(h), C.byref(m), 0) == _lib.DP_ERR_INVALID  # more than DP_TEMPORAL_MAX_LAYERS
    m.n_encoder_layers = 1
    assert lib.dp_temporal_create(C.byref(h), C.byref(m), 0) == _lib.DP_ERR_INVALID  # tensor pointers are NULL
    assert b"NULL" in lib.dp_temporal_last_error(None)
    assert lib.dp_temporal_destroy(None) == _lib.DP_ERR_INVALID
    torch.manual_seed(0)
    model = TemporalPredictor(n_encoder_layers=1, n_decoder_layers=1, dim_feedforward=32).eval()
    if not HAS_GPU:
        with pytest.raises(ValueError):
            NativeTemporal(model, torch.zeros(24), torch.ones(24), device="cpu")
        with pytest.raises(Exception) as e:  # a well-formed model, no device: refused, not emulated
            NativeTemporal(model, torch.zeros(24), torch.ones(24), device="cuda:0")
        assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_temporal_team_size_rule():
    """host arithmetic of dp_temporal_predict's kernel choice (dp_temporal.hip: teams of workgroups per sequence when there are few): every member on a
    CU of its own, at least one feed-forward tile per wave, 16 up to a quarter of the device"""
    lib = _lib.load()
    size = lambda n_seq, n_cu=256, ff=2048: lib.dp_temporal_debug_team_size(n_cu, n_seq, ff)
    # (the reference's 2048 hidden units are 64 tiles of 32: eight workgroups of eight waves have a tile each)
    # (round 6: all teams of a launch on at most HALF the CUs -- the slack that keeps a second handle's teams or another stream's kernel from
    #  starving a member)
    assert [size(s) for s in (1, 4, 5, 16, 17, 32, 33, 64, 65, 128, 129, 256, 1024)] == [8, 8, 8, 8, 4, 4, 2, 2, 1, 1, 1, 1, 1]
    for s in range(1, 300):
        g = size(s)
        assert g in (1, 2, 4, 8, 16) and (g == 1 or s * g <= 128)
    assert size(1, ff=200) == 1 and size(1, ff=512) == 2 and size(1, ff=1024) == 4 and size(1, ff=4096) == 16  # (7 / 16 / 32 / 128 tiles: a tile per wave of every member)
    assert size(1, n_cu=16) == 8 and size(3, n_cu=16) == 2 and size(5, n_cu=16) == 1 and size(0) == 1


def test_host_split_into_three_bf16_terms_is_exact():
    """the feed-forward weights of the temporal predictor reach the bf16 matrix pipe as three bf16 terms each (dp_temporal.hip: split precision).
    The host's split: every term the round-to-nearest-even bf16 of what is left, the three summing to the weight EXACTLY in fp32 arithmetic"""
    import numpy as np

    lib = _lib.load()
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(2000).astype(np.float32) * np.float32(0.2), np.float32([0.0, 1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9,
                        1.0 + 3 * 2.0 ** -9, 3.0e-20, -7.5e11, 0.1, 2.0 ** -126])])
    out = (C.c_ushort * 3)()

    def val(h):
        return np.array([int(h) << 16], dtype=np.uint32).view(np.float32)[0]

    def rne(x):  # numpy restatement of the rounding
        u = int(np.array([x], dtype=np.float32).view(np.uint32)[0])
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF

    for x in xs:
        lib.dp_temporal_debug_split3(float(x), out)
        h, m, l = (int(out[k]) for k in range(3))
        assert h == rne(x)
        r = np.float32(x) - val(h)
        assert m == rne(r)
        q = np.float32(r) - val(m)
        assert l == rne(q)
        assert np.float32(np.float32(val(h) + val(m)) + val(l)) == np.float32(x) or abs(x) < 2.0 ** -100  # (8 + 8 + 8 significant bits cover fp32's 24)


def test_rotation_target_validation():
    """LatentOptimizer.optimize(validate_targets=True) -> check_rotation_targets: the kernels evaluate |R - T|^2 in its quaternion
    form, equal to the reference's element-wise form only for rotation matrices (include/dragposer.h: dp_batch.tgt_rot)"""
    from dragposer_amd.optimizer import check_rotation_targets

    rs = np.random.RandomState(0)
    q = rs.randn(5, 22, 4)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=-1).astype(np.float32)
    trk = np.zeros((5, 22), np.uint8)
    trk[:, [0, 3, 7]] = 1
    check_rotation_targets(torch.tensor(R), torch.tensor(trk))  # rotations: passes
    bad = R.copy()
    bad[2, 5] *= 1.5  # an untracked joint may hold anything
    check_rotation_targets(torch.tensor(bad), torch.tensor(trk))
    bad[2, 3] *= 1.01  # a tracked one must be orthonormal ...
    with pytest.raises(ValueError):
        check_rotation_targets(torch.tensor(bad), torch.tensor(trk))
    refl = R.copy()
    refl[1, 7, 0:3] *= -1  # ... and proper (det +1)
    with pytest.raises(ValueError):
        check_rotation_targets(torch.tensor(refl), torch.tensor(trk))
    check_rotation_targets(torch.tensor(R), torch.zeros(5, 22, dtype=torch.uint8))  # nothing tracked: nothing to check

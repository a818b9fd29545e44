"""The native drop-in for the reference's Unity plugin ABI (DragPoserDLL/exportFunc.h:61-70): libDragPoserDLL.so.
CPU: the ten symbols exist and the host-only steps (BVH skeleton, model file, encoder) work / fail loudly.
GPU: drag_pose() frame by frame against the Python operator driven with the same targets, and -- with the temporal term on --
against what the reference itself returned for a Unity-shaped sequence (tests/golden/sequ.npz)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dragposer_amd", "lib", "libDragPoserDLL.so")
CLIP = os.path.join(ROOT, "tests", "data", "example_clip.bvh")
DATA = os.path.join(ROOT, "dragposer_amd", "data")
TEN = ("init_drag_poser", "set_reference_skeleton", "load_models", "set_mask_and_weights", "init_drag_model", "set_optim_params",
       "set_lambdas", "set_global_pos", "drag_pose", "destroy_drag_poser")


class F3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class F2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Qt(C.Structure):
    _fields_ = [("w", C.c_float), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


def _load():
    import dragposer_amd._lib as L

    L.load()  # pins torch's HIP runtime first
    lib = C.CDLL(LIB)
    lib.init_drag_poser.restype = C.c_void_p
    lib.drag_poser_last_error.restype = C.c_char_p
    lib.drag_poser_last_error.argtypes = [C.c_void_p]
    lib.set_reference_skeleton.argtypes = [C.c_void_p, C.c_char_p]
    lib.load_models.argtypes = [C.c_void_p, C.c_char_p]
    lib.set_mask_and_weights.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(F2)]
    lib.init_drag_model.argtypes = [C.c_void_p, F3, Qt]
    lib.set_optim_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_float]
    lib.set_lambdas.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    lib.set_global_pos.argtypes = [C.c_void_p, F3]
    lib.drag_pose.argtypes = [C.c_void_p, C.c_int, C.POINTER(F3), C.POINTER(Qt), C.POINTER(Qt), C.POINTER(F3)]
    lib.destroy_drag_poser.argtypes = [C.c_void_p]
    lib.drag_poser_last_iterations.argtypes = [C.c_void_p]
    lib.drag_poser_get_latent.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.drag_poser_set_latent.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    return lib


def test_exports_the_ten_reference_functions_and_fails_loudly():
    lib = _load()
    for sym in TEN:
        assert hasattr(lib, sym), sym
    h = lib.init_drag_poser()
    lib.load_models(h, DATA.encode())  # before the skeleton
    assert b"set_reference_skeleton" in lib.drag_poser_last_error(h)
    lib.set_reference_skeleton(h, b"/nonexistent.bvh")
    assert b"cannot open" in lib.drag_poser_last_error(h)
    lib.set_reference_skeleton(h, CLIP.encode())
    assert lib.drag_poser_last_error(h) == b""
    if not torch.cuda.is_available():
        lib.load_models(h, DATA.encode())
        assert b"no HIP device" in lib.drag_poser_last_error(h)  # no CPU fallback here either
    lib.destroy_drag_poser(h)


@pytest.mark.gpu
def test_drag_pose_matches_python_operator():
    from dragposer_amd import quat_np as Q
    from dragposer_amd.drag_pose import DragPose
    from dragposer_amd.optimizer import LatentOptimizer

    lib = _load()
    h = lib.init_drag_poser()
    lib.set_reference_skeleton(h, CLIP.encode())
    lib.load_models(h, DATA.encode())
    assert lib.drag_poser_last_error(h) == b"", lib.drag_poser_last_error(h)
    mask = np.zeros(22, np.float32)
    mask[R.TRACK6] = 1
    w = np.ones((22, 2), np.float32)
    for j, wj in R.W6.items():
        w[j] = wj
    lib.set_mask_and_weights(h, mask.ctypes.data_as(C.POINTER(C.c_float)), w.ctypes.data_as(C.POINTER(F2)))
    lib.set_optim_params(h, 1e-4, 1e-2, 10, 1e-2)  # Unity default budget: 10 iterations (Core/DragPoser.cs:34)
    lib.set_lambdas(h, 1.0, 0.02, 60)  # what the reference's own debug executable passes (DragPoserDLL/main.cpp)
    assert b"no temporal predictor" in lib.drag_poser_last_error(h)  # (no temporal.bin in that folder) accepted, the gap is reported, the pull term stays off
    lib.set_lambdas(h, 1.0, 0.0, 0)
    assert lib.drag_poser_last_error(h) == b""
    m = R.OracleModel()
    b = R.synth_inputs(m, 12, seed=5)  # 12 unrelated target frames, fed as a sequence
    cr0 = b["cur_rot"][0]
    lib.init_drag_model(h, F3(0.1, 0.2, 0.3), Qt(*[float(v) for v in cr0]))
    z0 = np.zeros(24, np.float32)
    lib.drag_poser_get_latent(h, z0.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.abs(z0).max() > 0 and np.isfinite(z0).all()

    opt = LatentOptimizer(device="cuda:0")
    dp = DragPose(opt, None, np.zeros(24), np.ones(24), n_sequences=1)
    dp.set_initial_state(z0, np.array([0.1, 0.2, 0.3], np.float32), cr0, np.zeros(6))
    idx = np.array(R.TRACK6)
    wj = np.array([R.W6[j] for j in R.TRACK6], np.float32)
    raw = np.load(R.DEFAULT_MODEL)
    for t in range(12):
        tp = b["tgt_pos"][t, idx]
        tR = b["tgt_rot"][t, idx].reshape(6, 3, 3)
        # rotation targets as quaternions (the plugin converts them back with to_matrix, run_drag.py:136)
        tq = np.stack([_mat_to_quat(M) for M in tR])
        res_pose = (Qt * 22)()
        res_pos = (F3 * 1)()
        lib.drag_pose(h, 6, (F3 * 6)(*[F3(*map(float, p)) for p in tp]), (Qt * 6)(*[Qt(*map(float, q)) for q in tq]), res_pose, res_pos)
        assert lib.drag_poser_last_error(h) == b""
        pose, gpos = dp.run(torch.tensor(tp), torch.tensor(Q.to_matrix(tq).astype(np.float32)), idx, wj, stop_eps_pos=1e-4, stop_eps_rot=1e-2,
                            max_iter=10, learning_rate=1e-2, lambda_rot=1.0, lambda_temporal=0.0, temporal_future_window=0)
        assert lib.drag_poser_last_iterations(h) == int(dp.last["iters"][0])
        np.testing.assert_allclose([res_pos[0].x, res_pos[0].y, res_pos[0].z], gpos.cpu().numpy(), atol=1e-6)
        qs = (pose.cpu().numpy().astype(np.float64) * opt.host_model.arrays["std_q"] + opt.host_model.arrays["mean_q"]).reshape(1, 22, 4)
        local = Q.from_root_space(qs, raw["parents"])[0]
        got = np.array([[q.w, q.x, q.y, q.z] for q in res_pose])
        np.testing.assert_allclose(got, local, atol=2e-6)
    lib.destroy_drag_poser(h)


def _mat_to_quat(M):
    w = np.sqrt(max(0.0, 1 + M[0, 0] + M[1, 1] + M[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + M[0, 0] - M[1, 1] - M[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - M[0, 0] + M[1, 1] - M[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - M[0, 0] - M[1, 1] + M[2, 2])) / 2
    return np.array([w, np.copysign(x, M[2, 1] - M[1, 2]), np.copysign(y, M[0, 2] - M[2, 0]), np.copysign(z, M[1, 0] - M[0, 1])])


def _drive_plugin(golden_dir, tmp_path, name):
    """the plug-in fed what the REFERENCE saw in tests/golden/<name>.npz, frame by frame, with the lambda the reference had at that
    frame (set_lambdas before every drag_pose, as Unity's SetLambdas may); returns per-frame errors against what the reference returned"""
    import shutil
    import sys

    from dragposer_amd import quat_np as Q

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_temporal_bin as X

    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    cfg, T = g["meta"]["cfg"], g["meta"]["T"]
    lam = [0.0 if t < cfg.get("lambda_switch_frame", 0) else float(cfg["lambda_temporal"]) for t in range(T)]
    shutil.copy(os.path.join(DATA, "dragposer_model.bin"), tmp_path / "dragposer_model.bin")
    X.write(X.tensors_from(os.path.join(golden_dir, f"{name}.npz")), str(tmp_path / "temporal.bin"))
    lib = _load()
    lib.drag_poser_has_temporal.argtypes = [C.c_void_p]
    h = lib.init_drag_poser()
    lib.set_reference_skeleton(h, CLIP.encode())
    lib.load_models(h, str(tmp_path).encode())
    assert lib.drag_poser_last_error(h) == b"", lib.drag_poser_last_error(h)
    assert lib.drag_poser_has_temporal(h) == 1
    mask, w = np.zeros(22, np.float32), np.ones((22, 2), np.float32)
    mask[g["mask_idx"]] = 1
    w[g["mask_idx"]] = g["weights"]
    lib.set_mask_and_weights(h, mask.ctypes.data_as(C.POINTER(C.c_float)), w.ctypes.data_as(C.POINTER(F2)))
    lib.set_optim_params(h, 0.01 * 0.01, 0.01, 100, 1e-2)
    lib.set_lambdas(h, 1.0, lam[0], int(cfg["temporal_future_window"]))
    assert lib.drag_poser_last_error(h) == b""  # a predictor is loaded: nothing to report
    lib.init_drag_model(h, F3(0.0, 0.0, 0.0), Qt(*[float(v) for v in g["init_rot"][0]]))
    z0 = np.ascontiguousarray(g["z0"][0], np.float32)
    lib.drag_poser_set_latent(h, z0.ctypes.data_as(C.POINTER(C.c_float)))  # the reference's start state
    raw = np.load(R.DEFAULT_MODEL)
    mean_q, std_q = raw["means.dqs"].reshape(22, 8)[:, :4].reshape(-1), raw["stds.dqs"].reshape(22, 8)[:, :4].reshape(-1)
    E = len(g["mask_idx"])
    gpos_mm, iters_equal, z_err, q_err = [], [], [], []
    for t in range(T):
        lib.set_lambdas(h, 1.0, lam[t], int(cfg["temporal_future_window"]))
        tp, tq = g["tgt_pos"][t, 0], np.stack([_mat_to_quat(M.astype(np.float64)) for M in g["tgt_rot"][t, 0]])
        res_pose, res_pos = (Qt * 22)(), (F3 * 1)()
        lib.drag_pose(h, E, (F3 * E)(*[F3(*map(float, p)) for p in tp]), (Qt * E)(*[Qt(*map(float, q)) for q in tq]), res_pose, res_pos)
        assert lib.drag_poser_last_error(h) == b"", lib.drag_poser_last_error(h)
        z = np.zeros(24, np.float32)
        lib.drag_poser_get_latent(h, z.ctypes.data_as(C.POINTER(C.c_float)))
        gpos_mm.append(np.abs(np.array([res_pos[0].x, res_pos[0].y, res_pos[0].z]) - g["gpos_ret"][t, 0]).max() * 1000.0)
        iters_equal.append(lib.drag_poser_last_iterations(h) == int(g["iters"][t, 0]))
        z_err.append(np.abs(z - g["latent"][t, 0]).max())
        want = Q.from_root_space((g["pose_ret"][t, 0].astype(np.float64) * std_q + mean_q).reshape(1, 22, 4), raw["parents"])[0]
        q_err.append(np.abs(np.array([[q.w, q.x, q.y, q.z] for q in res_pose]) - want).max())
    return lib, h, g, z0, E, lam, tuple(map(np.array, (gpos_mm, iters_equal, z_err, q_err)))


@pytest.mark.gpu
def test_plugin_reproduces_the_reference_sequence_with_the_temporal_term(golden_dir, tmp_path):
    """The plugin driven with what the REFERENCE saw and compared with what the reference returned (tests/golden/sequ.npz:
    the real DragPose.run over one sequence as the Unity path calls it -- run_drag.py:141-157: no joint adjustment, zero
    initial heights -- with lambda_temporal 0.02, a window of 8 frames and the reference's own Temporal class, whose
    state_dict becomes the plugin's temporal.bin).  Closed loop: strict over the first 16 frames, as test_hip_sequences."""
    lib, h, g, z0, E, lam, (gpos_mm, iters_equal, z_err, q_err) = _drive_plugin(golden_dir, tmp_path, "sequ")
    print(f"plugin vs reference (sequ): gpos {gpos_mm[:16].max():.4f} mm (all {gpos_mm.max():.4f}), latent {z_err[:16].max():.2e}, "
          f"local quaternions {q_err[:16].max():.2e}, same iteration count on {iters_equal.mean():.0%} of the frames")
    assert iters_equal[:16].mean() >= 0.95 and gpos_mm[:16].max() <= 0.02 and z_err[:16].max() <= 5e-4 and q_err[:16].max() <= 1e-4
    # and the pull term really is on: without it the same frames end elsewhere
    lib.set_lambdas(h, 1.0, 0.0, 0)
    lib.init_drag_model(h, F3(0.0, 0.0, 0.0), Qt(*[float(v) for v in g["init_rot"][0]]))
    lib.drag_poser_set_latent(h, z0.ctypes.data_as(C.POINTER(C.c_float)))
    for t in range(4):
        tp, tq = g["tgt_pos"][t, 0], np.stack([_mat_to_quat(M.astype(np.float64)) for M in g["tgt_rot"][t, 0]])
        res_pose, res_pos = (Qt * 22)(), (F3 * 1)()
        lib.drag_pose(h, E, (F3 * E)(*[F3(*map(float, p)) for p in tp]), (Qt * E)(*[Qt(*map(float, q)) for q in tq]), res_pose, res_pos)
    z = np.zeros(24, np.float32)
    lib.drag_poser_get_latent(h, z.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.abs(z - g["latent"][3, 0]).max() > 10 * z_err[3]
    lib.destroy_drag_poser(h)


@pytest.mark.gpu
def test_plugin_follows_the_references_predictor_schedule_when_the_term_is_switched_on_mid_window(golden_dir, tmp_path):
    """tests/golden/sequ_switch.npz: the reference's DragPose.run over the same kind of sequence with lambda_temporal = 0 for the first
    11 frames and 0.02 from frame 11 -- the fourth frame of the second window of 8.  The reference predicts at every window start
    whatever lambda is (drag_pose.py:247-291), so frames 11..15 pull towards the prediction made at frame 8, from the history as it
    stood then.  A plug-in that predicts only while the term is on (round 3: it restarted the window at switch-on and predicted from
    frame 11's history) gives another z_tgt for those frames; this one must give the reference's."""
    lib, h, g, z0, E, lam, (gpos_mm, iters_equal, z_err, q_err) = _drive_plugin(golden_dir, tmp_path, "sequ_switch")
    sw = g["meta"]["cfg"]["lambda_switch_frame"]
    assert lam[sw - 1] == 0.0 and lam[sw] > 0.0 and sw % g["meta"]["cfg"]["temporal_future_window"] != 0
    print(f"plugin vs reference (sequ_switch, term on from frame {sw}): gpos {gpos_mm[:16].max():.4f} mm (all {gpos_mm.max():.4f}), latent before the switch "
          f"{z_err[:sw].max():.2e}, frames {sw}..15 {z_err[sw:16].max():.2e}, local quaternions {q_err[:16].max():.2e}, same iteration count on "
          f"{iters_equal.mean():.0%} of the frames (reference: {g['iters'][:, 0].tolist()})")
    assert iters_equal[:16].mean() >= 0.9 and gpos_mm[:16].max() <= 0.05 and z_err[:16].max() <= 1e-3 and q_err[:16].max() <= 2e-4
    lib.destroy_drag_poser(h)

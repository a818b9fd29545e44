"""GPU (MI355X): `python -m dragposer_amd.eval_drag` against recordings of the REFERENCE'S OWN `eval_drag.main` on the same file
(tests/golden/f1_*.npz, tools/make_f1_goldens.py): BASELINE config 1's pipeline end to end -- BVH in, result BVH and the two
printed metrics out -- with the per-frame optimisation on the device.  Handed over from the recording: the initial latent (the
reference's normal draw depends on how much of torch's global generator its constructors consumed) and, for the runs with the
temporal term on, the predictor's weights (the reference's temporal.pt is not distributed; the recordings used a seeded, narrow one).

A sequence is a CLOSED LOOP (frame t starts from frame t - 1's latent, global pose and history): rounding differences between two
correct implementations are carried along and, wherever a frame's early-stop test or a LeakyReLU kink sits within rounding, they
become an iteration more or less and a different route.  So: the first frames must agree strictly (same iteration counts, returned
poses to fp32 rounding), after that the distance is printed frame range by frame range and bounded, and the sequence-level figures
the reference prints (MPJPE / MPEEPE of the written file, mean iterations per frame) must agree closely.
"""
import json
import os

import numpy as np
import pytest
import torch

from dragposer_amd import quat_np as Q
from dragposer_amd.bvh import BVH
from oracle import ref_torch as R
from test_f1_reference_pins import bvh_path, load

pytestmark = pytest.mark.gpu

STRICT = 12  # frames held to fp32 rounding


def _joint_positions(poses, raw):
    """returned poses [T, 88] (normalised root-space quaternions, root = world rotation) -> joint positions [T, 22, 3] with the root at
    the origin (what eval_pos_error measures)"""
    mu4 = raw["means.dqs"].reshape(22, 8)[:, :4].reshape(88)
    sd4 = raw["stds.dqs"].reshape(22, 8)[:, :4].reshape(88)
    q = (poses.astype(np.float64) * sd4 + mu4).reshape(-1, 22, 4)
    local = Q.from_root_space(q, list(raw["parents"]))
    return Q.fk(local, np.zeros((len(q), 3)), raw["offsets"].astype(np.float64), list(raw["parents"]))[0]


def _run_cli(g, tmp_path, per_frame=False):
    from dragposer_amd import eval_drag as E

    cfg_path, z0_path = str(tmp_path / "cfg.json"), str(tmp_path / "z0.npy")
    with open(cfg_path, "w") as f:
        json.dump(g["meta"]["cfg"], f)
    np.save(z0_path, g["initial_latent"])
    argv = [R.DEFAULT_MODEL, bvh_path(g), "--config", cfg_path, "--initial-latent", z0_path, "--out-dir", str(tmp_path / "data")]
    if g["meta"]["temporal_on"]:
        sd = {k[len("temporal."):]: torch.tensor(g[k]) for k in g if k.startswith("temporal.")}
        ck = str(tmp_path / "temporal.pt")
        torch.save({"model_state_dict": sd, "means_latent": torch.tensor(g["means_latent"]), "stds_latent": torch.tensor(g["stds_latent"])}, ck)
        argv += ["--temporal-checkpoint", ck]
    if per_frame:
        argv += ["--per-frame"]
    return E.main(argv + ["--keep-frames"])[0]


@pytest.mark.parametrize("name", ["f1_clip6", "f1_clip3", "f1_clip4", "f1_clip6_t", "f1_clip3_t", "f1_clip4_t"])
def test_cli_against_the_reference_run_of_the_clip(golden_dir, tmp_path, name):
    g = load(golden_dir, name)
    raw = np.load(R.DEFAULT_MODEL)
    res = _run_cli(g, tmp_path)
    T = int(g["n_frames"])
    poses, gpos, iters = res["poses"], res["pos"], res["iters"]
    assert poses.shape == (T, 88)
    d = np.linalg.norm(_joint_positions(poses, raw) - _joint_positions(g["pose_ret_all"], raw), axis=-1).max(axis=1) * 1000.0  # mm per frame
    dg = np.linalg.norm(gpos - g["gpos_ret"], axis=-1) * 1000.0
    same = iters == g["iters"]
    first = int(np.argmin(same)) if not same.all() else T
    rng = [(0, STRICT), (STRICT, 32), (32, 64), (64, 128), (128, T)]
    print(f"{name}: iteration counts equal on {same.mean():.3f} of {T} frames (first difference at frame {first}); mean iterations/frame {iters.mean():.2f} vs the "
          f"reference's {g['iters'].mean():.2f}; joint positions (root at origin) max mm per frame range "
          + ", ".join(f"[{a},{b}) {d[a:b].max():.4f}" for a, b in rng) + "; global position max mm " + ", ".join(f"[{a},{b}) {dg[a:b].max():.4f}" for a, b in rng)
          + f"; MPJPE {res['mpjpe'] * 1000:.3f} mm vs {float(g['mpjpe']) * 1000:.3f}, MPEEPE {res['mpeepe'] * 1000:.3f} vs {float(g['mpeepe']) * 1000:.3f}")
    # the reference's OWN closed loop, run twice by tools/make_f1_goldens.py: the second time from an initial latent moved by 1e-7 (one
    # fp32 ulp).  How far ITS two runs drift apart is the scale on which "the product follows the reference" can be read at all.
    dt = np.linalg.norm(_joint_positions(g["twin_pose_ret"], raw) - _joint_positions(g["pose_ret_all"], raw), axis=-1).max(axis=1) * 1000.0
    same_t = g["twin_iters"] == g["iters"]
    print(f"{name}: the reference against ITSELF from a latent 1e-7 away: iteration counts equal on {same_t.mean():.3f} of the frames; joint positions max mm "
          + ", ".join(f"[{a},{b}) {dt[a:b].max():.4f}" for a, b in rng) + f"; MPJPE {float(g['twin_mpjpe']) * 1000:.3f} mm vs {float(g['mpjpe']) * 1000:.3f}")
    six = len(np.nonzero(np.asarray(g["meta"]["cfg"]["mask"]))[0]) == 6
    # (without the feet's trackers the legs hang on the shared latent: fp32 rounding shows in THEIR joint positions first.  The 4-tracker clip with the
    #  pull term: the native predictor agrees with torch's to 1e-6 (tests/test_hip_temporal.py) and the reference's own twin run, 1e-7 away, is 0.01 mm
    #  apart in this window -- a 1e-6 difference in z_tgt is 0.1 mm in the legs after frame 4's 23 iterations: 0.056 mm with the predictor's first
    #  K-step order, 0.138 mm with the one of round 5's latency work, 0.018 mm with the feed-forward layers in split precision on the bf16 pipe --
    #  the same predictor to 1e-6 each time, iteration counts equal every time.  Fewer than six trackers: the strict window's bar)
    # (the looser bars are the 4-tracker clips' BY NAME: f1_clip3 / f1_clip3_t passed at 0.05 mm / 5e-3 / rtol 0.25 before those clips existed and still do)
    four = name.startswith("f1_clip4")
    assert same[:8].all() and d[:8].max() <= (0.2 if four else 0.05) and dg[:8].max() <= 0.05, (iters[:8], g["iters"][:8], d[:8].max())
    assert same[:STRICT].all() and d[:STRICT].max() <= 0.2, (iters[:STRICT], g["iters"][:STRICT], d[:STRICT].max())
    # after the strict window: no farther from the reference than three times what the reference's own twin run is (per frame range;
    # a floor of 10 mm where the twin happened to stay together), sequence-level figures within 5 % or three times the twins' spread
    spread = max(abs(float(g["twin_mpjpe"]) - float(g["mpjpe"])) / float(g["mpjpe"]), abs(float(g["twin_mpeepe"]) - float(g["mpeepe"])) / float(g["mpeepe"]))
    if six:
        for a, b in rng[1:]:
            assert d[a:b].max() <= 3.0 * max(dt[a:b].max(), dt[:b].max(), 10.0), (a, b, d[a:b].max(), dt[a:b].max())
        # (240 frames at 2-3 iterations each: nothing is converged, the latent carries the path; MPEEPE of the reference's twins is 4 % apart)
        np.testing.assert_allclose([res["mpjpe"], res["mpeepe"]], [float(g["mpjpe"]), float(g["mpeepe"])], rtol=max(0.08, 3.0 * spread))
    else:
        # head + two hands only: the legs hang on the shared latent alone and the loop amplifies whatever enters it -- the reference's
        # twin runs end 12 % apart in MPJPE without the temporal term (26 vs 30 cm of error: this configuration without a TRAINED predictor
        # is not a tracker at all), and stay together with it only as long as nothing perturbs them.  Held here: the first 32 frames
        # closely, the sequence-level figures loosely.
        assert d[:32].max() <= 1.0, d[:32].max()
        # (the 4-tracker clip without the pull term, profiles/r05_clip_twins.txt: the product's OWN runs from initial latents 1e-7 ... 1e-4 apart
        #  give MPEEPE 50 ... 71 mm at 1.7 ... 3.5 iterations per frame, the reference and its twin 80 / 73 mm at 4.1 / 3.3: a chaotic loop, one regime)
        np.testing.assert_allclose([res["mpjpe"], res["mpeepe"]], [float(g["mpjpe"]), float(g["mpeepe"])], rtol=max(0.35 if four else 0.25, 3.0 * spread))
    assert abs(iters.mean() - g["iters"].mean()) <= max(0.1, 3.0 * abs(g["twin_iters"].mean() - g["iters"].mean()) / g["iters"].mean()) * g["iters"].mean() + 1.5
    assert same.mean() >= 0.5 * same_t.mean()
    # the written file: on the strict window, the reference's MOTION block
    mine = BVH().load(res["out"]).motion
    dm = np.abs(mine[:STRICT] - g["result_motion_all"][:STRICT])
    dm[:, 3:] = np.minimum(dm[:, 3:], np.abs(dm[:, 3:] - 360.0))
    print(f"{name}: MOTION block against the reference's, strict window: first 8 frames {dm[:8].max():.2e}, all {dm.max():.2e} (degrees / metres)")
    assert dm[:8].max() <= (5e-2 if four else 5e-3) and dm.max() <= 5e-2, (dm[:8].max(), dm.max())  # degrees / metres as printed (6 decimals)


def test_cli_on_the_whole_example_file_against_the_reference_run(golden_dir, tmp_path):
    """BASELINE config 1 at full size (example.bvh, 5052 frames, 6 trackers, the reference's early-stop settings; lambda_temporal 0 on
    both sides: temporal.pt is not distributed) against the recording of the reference's own run of it"""
    g = load(golden_dir, "f1_example")
    raw = np.load(R.DEFAULT_MODEL)
    res = _run_cli(g, tmp_path)
    T, idx = int(g["n_frames"]), g["sample"]
    iters, gpos = res["iters"], res["pos"]
    d = np.linalg.norm(_joint_positions(res["poses"][idx], raw) - _joint_positions(g["pose_ret"], raw), axis=-1).max(axis=1) * 1000.0
    dg = np.linalg.norm(gpos - g["gpos_ret"], axis=-1) * 1000.0
    same = iters == g["iters"]
    edges = [0, STRICT, 64, 256, 1024, 2048, 4096, T]
    print(f"f1_example: {T} frames; iteration counts equal on {same.mean():.3f} (first difference at frame {int(np.argmin(same))}); mean iterations/frame {iters.mean():.2f} vs the "
          f"reference's {g['iters'].mean():.2f}; joint positions (sampled frames, root at origin) max mm by frame range "
          + ", ".join(f"[{a},{b}) {d[(idx >= a) & (idx < b)].max():.3f}" for a, b in zip(edges[:-1], edges[1:]))
          + "; global position max mm " + ", ".join(f"[{a},{b}) {dg[a:b].max():.3f}" for a, b in zip(edges[:-1], edges[1:]))
          + f"; MPJPE {res['mpjpe'] * 1000:.3f} mm vs {float(g['mpjpe']) * 1000:.3f}, MPEEPE {res['mpeepe'] * 1000:.3f} vs {float(g['mpeepe']) * 1000:.3f}; frame loop {res['time']:.3f} s")
    assert same[:STRICT].all() and d[idx < STRICT].max() <= 0.05 and dg[:STRICT].max() <= 0.05
    np.testing.assert_allclose([res["mpjpe"], res["mpeepe"]], [float(g["mpjpe"]), float(g["mpeepe"])], rtol=0.01)
    assert abs(iters.mean() - g["iters"].mean()) <= 0.05 * g["iters"].mean()
    # per frame the two runs are different routes through the same loop (7.8 iterations per frame: no frame is converged, the latent
    # carries the path): the distance between them is read against the distance between the reference's own twin runs
    dt = np.linalg.norm(_joint_positions(g["twin_pose_ret"], raw) - _joint_positions(g["pose_ret"], raw), axis=-1).max(axis=1) * 1000.0
    print(f"f1_example: per-frame joint distance to the reference: median {np.median(d):.2f} mm, p90 {np.percentile(d, 90):.2f}, p99 {np.percentile(d, 99):.2f}; the reference's "
          f"twin run (initial latent 1e-7 away) to the reference: median {np.median(dt):.2f} mm, p90 {np.percentile(dt, 90):.2f}, p99 {np.percentile(dt, 99):.2f}; its MPJPE "
          f"{float(g['twin_mpjpe']) * 1000:.3f} mm, iterations/frame {g['twin_iters'].mean():.2f}, iteration counts equal on {(g['twin_iters'] == g['iters']).mean():.3f} of the frames")
    for q in (50, 90, 99):
        assert np.percentile(d, q) <= 2.0 * np.percentile(dt, q) + 1.0, (q, np.percentile(d, q), np.percentile(dt, q))
    assert same.mean() >= 0.5 * (g["twin_iters"] == g["iters"]).mean()

"""CPU: the oracle (torch restatement + C restatement) against the golden vectors that the REAL
reference produced (tools/make_goldens.py) and the SURVEY 8.1 known-answer anchors."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle

torch.set_num_threads(4)


def _mm(a, b):
    return np.linalg.norm(a - b, axis=-1) * 1000.0


def _args(g):
    return [g[k] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]


@pytest.fixture(scope="module")
def anchors(golden_dir):
    return np.load(os.path.join(golden_dir, "anchors.npz"))


def test_anchor_a0_statistics():
    # SURVEY 8.1 A0 (data.pt)
    raw = np.load(R.DEFAULT_MODEL)
    np.testing.assert_allclose(raw["means.dqs"][:4], [0.9998175, -0.0000472, -0.0001902, 0.0008894], atol=5e-8)
    np.testing.assert_allclose(raw["stds.dqs"][:4], [0.0005840, 0.0070062, 0.0050380, 0.0163568], atol=5e-8)
    assert list(raw["parents"]) == [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 11, 18, 19, 20]


@pytest.mark.parametrize("impl", ["torch", "c_f32", "c_f64"])
def test_anchor_a1_forward(anchors, impl):
    z, c = np.zeros((1, 24), np.float32), np.array([[1.0, 0, 0, 0]], np.float32)
    if impl == "torch":
        m = R.OracleModel()
        mo, d = R.decoder_forward(m, torch.tensor(z))
        _, _, pos, rot, _ = R.pose_fk(m, mo, d, torch.tensor(c))
        pose, pos, rot = mo.numpy()[0], pos.numpy()[0], rot.numpy()[0].reshape(22, 9)
    else:
        f = AnalyticOracle(precision=impl[2:]).forward(z, c)
        pose, pos, rot = f["pose"][0], f["pos"][0], f["rot"][0]
    # SURVEY 8.1 A1 literal values, then the full vectors recorded from the reference
    np.testing.assert_allclose(pose[:4], [0.2888219, 0.2141083, 0.3866791, -0.3446664], atol=2e-4)
    np.testing.assert_allclose(pos[21], [-0.4412669, -0.0675104, 0.0905125], atol=2e-6)
    np.testing.assert_allclose(pose, anchors["a1_motion"], atol=2e-4)  # normalised space: /sigma amplifies
    np.testing.assert_allclose(pos, anchors["a1_pos"], atol=2e-6)
    np.testing.assert_allclose(rot, anchors["a1_rot"], atol=5e-6)


@pytest.mark.parametrize("impl", ["torch", "c_f32"])
def test_anchor_a2_loss_grad_and_first_adam_step(anchors, impl):
    w = np.zeros((1, 22, 2), np.float32); tr = np.zeros((1, 22), np.uint8)
    for j, wj in R.W6.items():
        w[0, j] = wj; tr[0, j] = 1
    tp = np.zeros((1, 22, 3), np.float32); tR = np.zeros((1, 22, 9), np.float32)
    tR[0, tr[0] > 0] = np.eye(3).reshape(9)
    z0, zt, c = np.zeros((1, 24), np.float32), np.full((1, 24), 0.1, np.float32), np.array([[1.0, 0, 0, 0]], np.float32)
    if impl == "torch":
        o = R.optimize(R.OracleModel(), z0, zt, c, tp, tR, w, tr, 1)
        loss, grad, z1 = o["loss_hist"][0, 0], o["grad"][0], o["z_final"][0]
    else:
        A = AnalyticOracle()
        l, g = A.grad(z0, zt, c, tp, tR, w, tr, 1.0, 0.02)
        loss, grad, z1 = l[0], g[0], A.optimize(z0, zt, c, tp, tR, w, tr, 1)["z_final"][0]
    np.testing.assert_allclose(loss, [0.698730052, 0.001586343, 0.000200000], rtol=2e-6)
    np.testing.assert_allclose(loss, anchors["a2_losses"], rtol=2e-6)
    np.testing.assert_allclose(grad, anchors["a2_grad"], atol=5e-7)
    np.testing.assert_allclose(z1, anchors["a2_z_after_one_step"], atol=1e-7)
    np.testing.assert_allclose(z1, -0.01 * np.sign(anchors["a2_grad"]), atol=1e-7)  # step 1 = -lr sign(g)


# tolerances (mm) per recipe: SURVEY 8(d).  S3 is chaotic under Adam's sign sensitivity; its max is
# reported in DESIGN.md (6.2 mm between the batch-1 reference and the batched torch restatement of the
# very same ops), the gate is on mean / p99.
TOL = {"s1": dict(max=0.05), "s4": dict(max=0.05), "es": dict(max=0.05), "s3": dict(mean=0.05, p99=1.0)}


@pytest.mark.parametrize("name", ["s1", "s3", "s4", "es"])
@pytest.mark.parametrize("impl", ["torch", "c_f32"])
def test_oracle_reproduces_reference_goldens(golden_dir, name, impl):
    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = g["meta"]
    kw = dict(lam_tmp=mt["lambda_tmp"], stop_eps_pos=mt["stop_eps_pos"], stop_eps_rot=mt["stop_eps_rot"],
              min_loss_incr=mt["min_loss_incr"])
    if impl == "torch":
        o = R.optimize(R.OracleModel(weight_rounding=mt["weight_rounding"]), *_args(g), mt["n_iter"], **kw)
        l3 = o["loss_hist"][:, :3]
        ok = ~np.isnan(g["loss_hist"][:, :3])
        np.testing.assert_allclose(l3[ok], g["loss_hist"][:, :3][ok], rtol=1e-5)  # first 3 iterations
    else:
        o = AnalyticOracle(weight_rounding=mt["weight_rounding"]).optimize(*_args(g), mt["n_iter"], **kw)
    err = _mm(o["pos"], g["pos"])
    tol = TOL[name]
    if "max" in tol:
        assert err.max() <= tol["max"], err.max()
        np.testing.assert_array_equal(o["iters"], g["iters"])
        np.testing.assert_allclose(o["z_final"], g["z_final"], atol=2e-5)
        np.testing.assert_allclose(o["world_rot"], g["world_rot"], atol=2e-6)
        np.testing.assert_allclose(o["world_disp"], g["world_disp"], atol=2e-7)
    else:
        assert err.mean() <= tol["mean"] and np.percentile(err, 99) <= tol["p99"], (err.mean(), np.percentile(err, 99))


@pytest.mark.parametrize("name,trackers,mixed,wd", [("full_s3_1024", 3, False, "none"), ("full_s4_1024", 6, True, "bf16")])
def test_c_oracle_against_the_reference_at_size(golden_dir, name, trackers, mixed, wd):
    """BASELINE configs 4 and 5 at 1024 frames: the C restatement against the REAL reference's fp32 run of the same inputs
    (tools/make_goldens.py --only full_s3_1024,full_s4_1024), in BASELINE's terms, next to the reference's own fp32-vs-fp64 pair."""
    ref = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = ref["meta"]
    b = R.synth_inputs(R.OracleModel(weight_rounding=wd), mt["B"], trackers=trackers, mixed=mixed, seed=mt["seed"])
    T6 = [0, 3, 7, 13, 17, 21]
    assert np.abs(b["tgt_pos"][:, T6] - ref["tgt_pos6"]).max() < 1e-5  # (the stored targets are the generating host's; its matmul bits may differ)
    b["tgt_pos"][:, T6], b["tgt_rot"][:, T6] = ref["tgt_pos6"], ref["tgt_rot6"]
    o = AnalyticOracle(weight_rounding=wd).optimize(*_args(b), mt["n_iter"], lam_tmp=mt["lambda_tmp"])
    e, pair = _mm(o["pos"], ref["pos"]), _mm(ref["pos"], ref["pos_f64"])
    flagged = pair.max(axis=1) > 0.02
    print(f"{name}: C oracle vs reference fp32: mean {e.mean():.5f} p99 {np.percentile(e, 99):.5f} max {e.max():.4f} mm; reference pair: "
          f"mean {pair.mean():.5f} p99 {np.percentile(pair, 99):.5f} max {pair.max():.4f}")
    assert e.mean() <= 0.05 and np.percentile(e, 99) <= (1.0 if trackers == 3 else 0.05)
    assert (e.max(axis=1) > 0.05).sum() <= max(2, 2 * int(flagged.sum())) and e.max() <= 10.0


def test_manual_adam_equals_torch_optim_adam(golden_dir):
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    m = R.OracleModel()
    a = [x[:8] for x in _args(g)]
    o1 = R.optimize(m, *a, 12)
    o2 = R.optimize(m, *a, 12, use_torch_adam=True)
    np.testing.assert_allclose(o1["z_final"], o2["z_final"], atol=2e-6)  # same formulas; fused-op rounding only


def test_reference_shaped_loop_equals_batched(golden_dir):
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    m = R.OracleModel()
    a = [x[:3] for x in _args(g)]
    o1 = R.optimize_reference_shaped(m, *a, 5)
    o2 = R.optimize(m, *a, 5)
    np.testing.assert_allclose(o1["z_final"], o2["z_final"], atol=2e-6)


def test_collapsed_fk_identity_n1():
    """the C restatement (collapsed FK) equals the matrix-chain torch restatement."""
    m = R.OracleModel()
    b = R.synth_inputs(m, 32)
    f = AnalyticOracle().forward(b["z_src"], b["cur_rot"])
    with torch.no_grad():
        mo, d = R.decoder_forward(m, torch.tensor(b["z_src"]))
        _, _, pos, rot, _ = R.pose_fk(m, mo, d, torch.tensor(b["cur_rot"]))
    np.testing.assert_allclose(f["pos"], pos.numpy(), atol=2e-6)
    np.testing.assert_allclose(f["rot"], rot.numpy().reshape(-1, 22, 9), atol=5e-6)

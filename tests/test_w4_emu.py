"""CPU: the host tables of the wave-private kernel (weight image, bias rows, per-quad constants) and the restated
kinematics (tests/w4_emu.py follows dp_w4.hip stage by stage) against the fp64 analytic oracle."""
import os

import numpy as np
import pytest

from dragposer_amd.model import HostModel
from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle

import w4_emu as W

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


@pytest.fixture(scope="module")
def tables():
    return W.host_tables(HostModel())


def test_weight_image_is_the_folded_decoder(tables):
    img, bias, pairs, items = tables
    f, _ = HostModel().fold()
    rs = np.random.RandomState(0)
    x = rs.randn(24)
    np.testing.assert_allclose(W.product(img, W.S_L0, x)[W.H0_ROW], f["A0"].astype(np.float64) @ x, rtol=1e-12, atol=1e-12)
    unused = np.setdiff1d(np.arange(64), W.H0_ROW)
    assert not W.product(img, W.S_L0, x)[unused].any() and not bias[0][unused].any()
    x = rs.randn(40)
    np.testing.assert_allclose(W.product(img, W.S_L1, x)[:60], f["A1"].astype(np.float64) @ x, rtol=1e-12, atol=1e-12)
    x = rs.randn(60)
    np.testing.assert_allclose(W.product(img, W.S_B1, x)[W.H0_ROW], f["A1"].astype(np.float64).T @ x, rtol=1e-12, atol=1e-12)
    x = rs.randn(40)  # bL0: lanes 0..31 hold K-steps 0..19, lanes 32..63 K-steps 20..39 of the same 24 rows
    lo, hi = W.product(img, W.S_B0, x[:20]), W.product(img, W.S_B0, x[20:])
    np.testing.assert_allclose((lo[:32] + hi[32:])[:24], f["A0"].astype(np.float64).T @ x, rtol=1e-12, atol=1e-12)
    # layer 2 carries the de-normalisation: rows of sigma * A2, bias sigma * b2 + mu, per item of a quad's two sides
    hm = HostModel()
    x = rs.randn(60)
    y = f["A2"].astype(np.float64) @ x + f["b2"]
    rq = W.layer2(img, bias, x)
    for s in (0, 1):
        r = rq[s]
        for b in range(16):
            item, kind = pairs["item"][b][s], pairs["kind"][b][s]
            if kind in (W.KIND_JOINT, W.KIND_ROOT, W.KIND_VIRT):
                j = item if item < 22 else int(items["src_quad"][item])
                want = y[4 * j:4 * j + 4] * hm.arrays["std_q"][4 * j:4 * j + 4] + hm.arrays["mean_q"][4 * j:4 * j + 4]
                np.testing.assert_allclose(r[b], want, rtol=1e-6, atol=1e-7)
            elif kind == W.KIND_DISP:
                np.testing.assert_allclose(r[b][:3], y[88:91] * hm.arrays["std_disp"] + hm.arrays["mean_disp"], rtol=1e-6, atol=1e-9)
                assert r[b][3] == 0
            else:
                np.testing.assert_array_equal(r[b], [1, 0, 0, 0])  # idle quads decode to the unit quaternion
    # every item appears exactly once; the root shares its quad with an idle side
    got = sorted(int(i) for i, k in zip(pairs["item"].reshape(-1), pairs["kind"].reshape(-1)) if i >= 0 and k != W.KIND_IDLE)
    nvirt = int((items["kind"] == W.KIND_VIRT).sum())
    assert got == list(range(23 + nvirt)) and pairs["item"][0][0] == 0 and pairs["item"][0][1] == -1


@pytest.mark.parametrize("name", ["s1", "s3", "s4"])
def test_emulated_iteration_matches_the_analytic_oracle(tables, golden_dir, name):
    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    if g["meta"]["weight_rounding"] != "none":
        pytest.skip("bf16-rounded weights are a different model (covered on the GPU)")
    lam = g["meta"]["lambda_tmp"]
    lo, gr = AnalyticOracle(precision="f64").grad(*[g[k] for k in KEYS], 1.0, lam)
    for f in range(0, len(g["z0"]), 5):
        loss, gz, _, _ = W.one_iteration(tables, *[g[k][f] for k in KEYS], lam_tmp=lam)
        np.testing.assert_allclose(loss, lo[f], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(gz, gr[f], rtol=2e-5, atol=1e-6)  # (sigma * A2 is rounded to fp32 once more than the oracle rounds)


def test_emulated_iteration_with_every_joint_tracked(tables):
    m = R.OracleModel()
    b = R.synth_inputs(m, 4)
    import torch

    with torch.no_grad():
        mo, dd = R.decoder_forward(m, torch.tensor(b["z_src"]))
        _, _, pos, rot, _ = R.pose_fk(m, mo, dd, torch.tensor(b["cur_rot"]))
    b["tracked"][:] = 1
    b["w"][:] = np.random.RandomState(1).uniform(0.1, 5.0, b["w"].shape).astype(np.float32)
    b["tgt_pos"] = pos.numpy().astype(np.float32)
    b["tgt_rot"] = rot.numpy().reshape(4, 22, 9).astype(np.float32)
    lo, gr = AnalyticOracle(precision="f64").grad(*[b[k] for k in KEYS], 1.0, 0.02)
    for f in range(4):
        loss, gz, _, _ = W.one_iteration(tables, *[b[k][f] for k in KEYS])
        np.testing.assert_allclose(loss, lo[f], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(gz, gr[f], rtol=2e-5, atol=1e-6)  # (sigma * A2 is rounded to fp32 once more than the oracle rounds)

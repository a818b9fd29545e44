"""GPU: skeletons other than the one the reference ships.  The reference's decoder and FK are topology-generic (skeleton.py:133-175, utils.py:109-149);
this repo's fp32 kernel (dp_w4) is laid out for 22 joints within three limits (include/dragposer.h: at most 3 children of the root, at most 3 extra
child bones, chains of at most 7 bones) and takes ANY tree inside them -- what depends on the tree is host-built tables (bone slots, path words,
subtree masks, the virtual items' duplicated rows of bL2), and this file runs two trees that exercise them differently from the Xsens hierarchy:
extra child bones on two different joints / three of them on one joint, arms at different spine levels, a 7-bone chain that is not the left arm.
Random offsets and perturbed decoder weights; targets = the C oracle's own FK of a source latent; the kernel against the C oracle (which follows the
reference term by term for any parents array).  dp_w16's slot map IS the Xsens tree: refused for these with DP_ERR_UNSUPPORTED, DP_KERNEL_AUTO takes dp_w4."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

TREES = {
    # two legs of four, a spine of two; neck + head (11-12-13) and the left arm (14..17) hang off joint 10, the right arm (18..21) off joint 9
    "arms_at_two_levels": [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 10, 14, 15, 16, 9, 18, 19, 20],
    # a trunk 1-2-3 whose end carries FOUR two-bone limbs (three extra child bones on ONE joint), a 7-bone tail 12..18, a short third root child
    "four_limbs_on_one_joint": [0, 0, 1, 2, 3, 4, 3, 6, 3, 8, 3, 10, 0, 12, 13, 14, 15, 16, 17, 0, 19, 20],
}


def _model_arrays(parents, seed):
    raw = dict(np.load(R.DEFAULT_MODEL))
    rng = np.random.default_rng(seed)
    raw["parents"] = np.asarray(parents, np.int32)
    off = rng.uniform(-0.25, 0.25, (22, 3)).astype(np.float32)
    off[0] = 0.0
    raw["offsets"] = off
    for k in list(raw):  # the decoder as a function changes too: every weight moved by up to 20 %
        if k.startswith("decoder.") and k.endswith("weight"):
            raw[k] = (raw[k] * (1.0 + 0.2 * rng.standard_normal(raw[k].shape))).astype(np.float32)
    return raw


@pytest.mark.parametrize("name", list(TREES))
def test_another_tree_through_dp_w4_against_the_c_oracle(name, tmp_path):
    from dragposer_amd import _lib
    from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
    from oracle.analytic import AnalyticOracle

    parents = TREES[name]
    raw = _model_arrays(parents, seed=len(name))
    path = str(tmp_path / "model.npz")
    np.savez(path, **raw)
    ora = AnalyticOracle(model_path=path, precision="f32")
    ora64 = AnalyticOracle(model_path=path, precision="f64")
    opt = LatentOptimizer(device="cuda:0", arrays=raw)
    B = 64
    g = torch.Generator().manual_seed(5)
    Zs, Z0 = (torch.randn(B, 24, generator=g) * 0.3).numpy(), (torch.randn(B, 24, generator=g) * 0.3).numpy()
    ZT = Z0 + 0.05 * torch.randn(B, 24, generator=g).numpy()
    CR = torch.randn(B, 4, generator=g).numpy()
    CR /= np.linalg.norm(CR, axis=-1, keepdims=True)
    fk = ora64.forward(Zs, CR)
    leaves = [j for j in range(1, 22) if j not in parents[1:]]
    tracked = np.zeros((B, 22), np.uint8)
    w = np.zeros((B, 22, 2), np.float32)
    for j in [0] + leaves:  # the root and every leaf: 6 / 8 trackers (more than six: stage G's general path)
        tracked[:, j] = 1
        w[:, j] = (10.0, 10.0) if j == 0 else (5.0, 0.01)
    batch = dict(z0=Z0, z_tgt=ZT, cur_rot=CR, tgt_pos=fk["pos"] * tracked[..., None], tgt_rot=fk["rot"] * tracked[..., None], w=w, tracked=tracked)
    d = to_device_batch(batch, opt.device)
    # the forward kernel first: decode + FK of the source latents against the oracle's
    f = opt.forward(d["z0"], d["cur_rot"], outputs=("pos", "rot"))
    want_f = ora.forward(Z0, CR)
    assert np.abs(f["pos"].cpu().numpy() - want_f["pos"]).max() * 1000.0 <= 0.01 and np.abs(f["rot"].cpu().numpy() - want_f["rot"]).max() <= 2e-5
    args = [batch[k] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")]
    for n_iter in (1, 20):
        want = ora.optimize(*args, n_iter)
        out = opt.optimize(**d, n_iter=n_iter, lambda_tmp=0.02, kernel="w4", outputs=("z", "pos", "loss", "status"))
        err = np.linalg.norm(out["pos"].cpu().numpy() - want["pos"], axis=-1).max(axis=1) * 1000.0
        print(f"{name}: {len(leaves) + 1} trackers, {n_iter} iterations: joint positions max {err.max():.5f} mm, |dz| {np.abs(out['z'].cpu().numpy() - want['z_final']).max():.2e}")
        assert (out["status"] == 0).all()
        if n_iter == 1:  # one Adam step = -lr sign(dL/dz): the whole backward through THIS tree, sign by sign (off the components of size 1e-7)
            assert (np.abs(out["z"].cpu().numpy() - want["z_final"]) > 1e-3).mean() <= 0.002
            np.testing.assert_allclose(out["loss"].cpu().numpy(), want["loss"], rtol=2e-5, atol=1e-7)
        assert np.sort(err)[-2] <= 0.05 and err.max() <= 5.0, err  # (one frame may sit on a LeakyReLU kink: BASELINE.md section 3)
    # the 16-frames-per-wave kernel is laid out for the reference's tree only
    with pytest.raises(_lib.DragPoserError) as e:
        opt.optimize(**d, n_iter=2, kernel="w16")
    assert e.value.code == _lib.DP_ERR_UNSUPPORTED
    assert opt.auto_kernel(1 << 20) == "w4"

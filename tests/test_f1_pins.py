"""CPU: the evaluation pipeline pinned as far as the reference's own data allows.

The reference's BVH I/O, Euler / dual-quaternion conventions and FK come from `upc-pymotion`, which is absent here, and the
reference holds no evaluation output: those conventions stay UNPINNED against the reference.  What its data
does pin: the three motion files it ships (python/data/example/{train,eval}/example{,_2,_3}.bvh -- the two folders are
byte-identical; staged by __graft_entry__.build() into tests/data/_local) must survive our reader / writer unchanged, must
look like the training data in the units of data.pt, must be reconstructed by the trained VAE, and the metric of
eval_metrics.py must behave as its definition says."""
import os

import numpy as np
import pytest
import torch

from dragposer_amd import motion as MO
from dragposer_amd import quat_np as Q
from dragposer_amd.bvh import BVH
from dragposer_amd.encoder import PoseEncoder
from dragposer_amd.eval_drag import eval_pos_error
from oracle import ref_torch as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"example.bvh": 5052, "example_2.bvh": 2920, "example_3.bvh": 3047}


def _path(name):
    p = os.path.join(ROOT, "tests", "data", "_local", name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not staged (python -c 'import __graft_entry__ as g; g.build()' where the reference is mounted)")
    return p


@pytest.fixture(scope="module")
def stats():
    raw = np.load(R.DEFAULT_MODEL)
    return ({"dqs": raw["means.dqs"], "displacement": raw["means.displacement"]},
            {"dqs": raw["stds.dqs"], "displacement": raw["stds.displacement"]}, raw)


@pytest.mark.parametrize("name", list(FILES))
def test_bvh_reader_writer_roundtrip(tmp_path, name, stats):
    src = _path(name)
    b = BVH().load(src)
    rot, pos, parents, offsets, order = b.get_data()
    assert rot.shape == (FILES[name], 22, 3) and b.motion.shape == (FILES[name], 69)  # 6 root channels + 21 x 3
    assert list(parents) == list(stats[2]["parents"]) and set(order) == {"xyz"}
    np.testing.assert_allclose(offsets, stats[2]["offsets"], atol=1e-6)
    b.set_data(rot, pos[:, 0])
    out = tmp_path / name
    b.save(out)
    a, c = open(src).read().splitlines(), open(out).read().splitlines()
    k = next(i for i, l in enumerate(a) if l.split() and l.split()[0] == "MOTION")
    assert a[:k] == c[:k]  # HIERARCHY text: identical
    assert a[k + 1].split() == c[k + 1].split() and abs(float(a[k + 2].split()[2]) - float(c[k + 2].split()[2])) < 1e-9
    m0 = np.array([[float(t) for t in l.split()] for l in a[k + 3:k + 3 + FILES[name]]])
    np.testing.assert_allclose(BVH().load(out).motion, m0, atol=1e-6)  # MOTION block: numerically equal


@pytest.mark.parametrize("name", list(FILES))
def test_preprocessing_in_the_units_of_the_trained_model(name, stats):
    means, stds, raw = stats
    m = MO.prepare_motion(BVH().load(_path(name)), means, stds)
    F = FILES[name]
    assert m["dqs"].shape == (F, 176)
    np.testing.assert_allclose(m["dqs_raw"].reshape(F, 22, 8)[0, 0, :4], [1, 0, 0, 0])  # frame 0: identity increment (motion_data.py:258)
    np.testing.assert_allclose(m["dqs_raw"].reshape(F, 22, 8)[0, 0, 4:], 0)              # ... and no displacement
    np.testing.assert_allclose(np.linalg.norm(m["root_quats"], axis=-1), 1.0, atol=1e-9)
    n = m["dqs"].reshape(F, 22, 8)
    real = np.ones((22, 8), bool)
    real[0, 7] = False  # the padding channel of the root's dual part (std 1, value 0)
    # dancedb statistics, another dancer: every channel within a few sigma of the training mean, spread of order one
    assert np.abs(n[:, real].mean(0)).max() < 2.0 and 0.1 < n[:, real].std(0).min() and n[:, real].std(0).max() < 2.5
    # the trained VAE reconstructs the poses (a wrong convention does not: tests/test_host_pipeline.py)
    enc, model = PoseEncoder(), R.OracleModel()
    sub = slice(0, F, 16)
    mu, _ = enc(torch.tensor(m["dqs"][sub]))
    mo, _ = R.decoder_forward(model, mu)
    q_rec = Q.normalize((mo * model.sd4 + model.mu4).reshape(-1, 22, 4).numpy().astype(np.float64))
    off, par = m["offsets"].astype(np.float64), m["parents"]
    err = np.linalg.norm(MO.root_space_translations(m["root_quats"][sub], off, par) - MO.root_space_translations(q_rec, off, par), axis=-1)
    assert err.mean() * 1000 < 30.0, err.mean() * 1000
    # heights fed to the temporal net: world y of joints [0, 4, 8, 13, 17, 21] (motion_data.py:261-270)
    q, pos, parents, offsets = MO.local_quats_from_bvh(BVH().load(_path(name)))
    gp, _ = Q.fk(q[:8], pos[:8, 0], offsets, parents)
    np.testing.assert_allclose(m["heights"][:8], gp[:, [0, 4, 8, 13, 17, 21], 1], atol=1e-5)


@pytest.mark.parametrize("name", list(FILES))
def test_position_error_metric_follows_its_definition(tmp_path, name):
    """eval_metrics.eval_pos_error (eval_metrics.py:6-32): FK of both files with the root at the origin; MPJPE = mean over all
    joints, MPEEPE = mean over sparse_joints[1:]."""
    src = _path(name)
    b = BVH().load(src)
    assert eval_pos_error(b, BVH().load(src)) == (0.0, 0.0)
    rot, pos, parents, offsets, order = b.get_data()
    moved = BVH().load(src)
    moved.set_data(rot, pos[:, 0] + np.array([1.0, -2.0, 3.0]))  # the root translation is not part of the metric
    mp, me = eval_pos_error(b, moved)
    assert mp < 1e-6 and me < 1e-6
    bent = BVH().load(src)
    r2 = rot.copy()
    r2[:, 16, 2] += 90.0  # left elbow (joint 16: child 17 = the wrist, an end effector) turned by 90 degrees about its last axis
    bent.set_data(r2, pos[:, 0])
    mp, me = eval_pos_error(b, bent)
    q, _, _, _ = MO.local_quats_from_bvh(b)
    qb, _, _, _ = MO.local_quats_from_bvh(bent)
    gp, _ = Q.fk(q, np.zeros((len(q), 3)), offsets, parents)
    gb, _ = Q.fk(qb, np.zeros((len(q), 3)), offsets, parents)
    d = np.linalg.norm(gp - gb, axis=-1)
    assert d[:, :17].max() < 1e-9 and d[:, 18:].max() < 1e-9  # only the wrist moves ...
    # ... on a circle about the forearm's last axis: never farther than sqrt(2) |forearm| from where it was
    assert 0 < d[:, 17].max() <= np.sqrt(2.0) * np.linalg.norm(offsets[17]) + 1e-9
    np.testing.assert_allclose(mp, d.mean(), rtol=1e-9)
    np.testing.assert_allclose(me, d[:, [4, 8, 13, 17, 21]].mean(), rtol=1e-9)


def test_reference_model_folder_contract(tmp_path):
    """The reference CLI takes a model FOLDER (eval_drag.py:260-264): generator.pt = {"model_state_dict": ...} with the
    Generator_Model's `autoencoder.` prefix (train.py:297-303), data.pt = {"means": {...}, "stds": {...}} (train.py:288-296).
    A folder written that way from the shipped tensors must load into exactly the arrays of the flat fixture, with the skeleton
    taken from the evaluated BVH as the reference does (train.py:329-341)."""
    from dragposer_amd.model import HostModel, load_model_arrays

    raw = np.load(R.DEFAULT_MODEL)
    sd = {"autoencoder." + k: torch.tensor(raw[k]) for k in raw.files if k.startswith(("encoder.", "decoder."))}
    torch.save({"model_state_dict": sd}, tmp_path / "generator.pt")
    torch.save({"means": {"dqs": torch.tensor(raw["means.dqs"]), "displacement": torch.tensor(raw["means.displacement"])},
                "stds": {"dqs": torch.tensor(raw["stds.dqs"]), "displacement": torch.tensor(raw["stds.displacement"])}}, tmp_path / "data.pt")
    arrs = load_model_arrays(str(tmp_path), skeleton_bvh=os.path.join(ROOT, "tests", "data", "example_clip.bvh"))
    for k in raw.files:
        if k == "offsets":
            np.testing.assert_allclose(arrs[k], raw[k], atol=1e-6)
        else:
            np.testing.assert_array_equal(np.asarray(arrs[k]).reshape(raw[k].shape), raw[k], err_msg=k)
    a, b = HostModel(arrays=arrs).fold()[0], HostModel().fold()[0]
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert load_model_arrays(R.DEFAULT_MODEL)["parents"].tolist() == raw["parents"].tolist()  # the flat fixture still loads

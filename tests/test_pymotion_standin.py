"""CPU: the harness-side stand-in for the absent `upc-pymotion` package (tools/pymotion_standin), through which the reference's own
evaluation plumbing was executed to produce tests/golden/f1_*.npz (tools/make_f1_goldens.py).  Its adapters must be self-consistent
-- Euler <-> quaternion and dual-quaternion round trips, the two FK forms against each other, the BVH adapter against the file --
or the fixtures would pin nonsense.  (What they cannot be checked against is pymotion itself: DESIGN.md section 2.)"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "pymotion_standin"))

from pymotion.io.bvh import BVH  # noqa: E402
from pymotion.ops.forward_kinematics import fk as fk_np  # noqa: E402
from pymotion.ops.forward_kinematics_torch import fk as fk_torch  # noqa: E402
from pymotion.ops.skeleton import from_root_dual_quat, to_root_dual_quat  # noqa: E402
from pymotion.rotations import dual_quat as dquat  # noqa: E402
from pymotion.rotations import quat, quat_torch  # noqa: E402

CLIP = os.path.join(ROOT, "tests", "data", "example_clip.bvh")


def _rand_quats(rs, *shape):
    q = rs.normal(size=shape + (4,))
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def test_euler_round_trip_in_bvh_channel_order():
    rs = np.random.RandomState(0)
    e = np.radians(rs.uniform(-80, 80, size=(50, 22, 3)))
    order = np.tile(np.array([list("xyz")] * 22), (50, 1, 1))
    q = quat.from_euler(e, order)
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, atol=1e-12)
    np.testing.assert_allclose(quat.to_euler(q, order), e, atol=1e-10)
    # R = Rx Ry Rz for channels X, Y, Z in file order: a rotation about z alone leaves z fixed
    qz = quat.from_euler(np.array([[[0.0, 0.0, 0.7]]]), np.array([[list("xyz")]]))
    np.testing.assert_allclose(quat.mul_vec(qz[0, 0], np.array([0.0, 0.0, 1.0])), [0, 0, 1], atol=1e-12)


def test_dual_quaternion_round_trip_and_unroll():
    rs = np.random.RandomState(1)
    r, t = _rand_quats(rs, 20, 22), rs.normal(size=(20, 22, 3))
    dq = dquat.from_rotation_translation(r, t)
    r2, t2 = dquat.to_rotation_translation(dq)
    np.testing.assert_allclose(r2, r, atol=1e-12)
    np.testing.assert_allclose(t2, t, atol=1e-12)
    # unroll: a smooth sequence with some frames negated comes back continuous (the same transforms, signs only)
    ang = np.linspace(0.1, 1.5, 20)[:, None]
    smooth = dquat.from_rotation_translation(np.concatenate([np.cos(ang / 2), np.sin(ang / 2), 0 * ang, 0 * ang], -1)[:, None], t[:, :1])
    flipped = smooth.copy()
    flipped[5:9] *= -1.0
    un = dquat.unroll(flipped, axis=0)
    np.testing.assert_allclose(un, smooth, atol=0)


def test_root_space_construction_and_both_fk_forms_agree_on_the_clip():
    b = BVH().load(CLIP)
    q, pos, parents, offsets, _, _ = b.get_data()
    parents = [0 if p is None else p for p in parents]
    offsets = offsets.copy()
    offsets[0] = 0.0
    dq = to_root_dual_quat(q, np.zeros((len(q), 3)), parents, offsets)
    local, root_t = from_root_dual_quat(dq, parents)
    np.testing.assert_allclose(np.abs(np.sum(local * q, axis=-1)), 1.0, atol=1e-9)  # the same rotations up to sign
    # positions: numpy FK = torch FK = root-space translations rotated by the root
    p_np, R_np = fk_np(q, pos[:, 0], offsets, parents)
    p_t, R_t = fk_torch(torch.tensor(q), torch.tensor(pos[:, 0]), torch.tensor(offsets), torch.tensor(parents))
    np.testing.assert_allclose(p_t.numpy(), p_np, atol=1e-9)
    np.testing.assert_allclose(R_t.numpy(), R_np, atol=1e-9)
    r, t = dquat.to_rotation_translation(dq)
    world = quat.mul_vec(r[:, 0:1], t) + pos[:, 0][:, None]
    np.testing.assert_allclose(world[:, 1:], p_np[:, 1:], atol=1e-9)


def test_bvh_adapter_exposes_what_the_reference_reads_and_writes_it_back(tmp_path):
    b = BVH().load(CLIP)
    d = b.data
    assert d["rotations"].shape == (240, 22, 3) and d["positions"].shape == (240, 22, 3) and d["parents"][0] is None
    assert d["rot_order"].shape == (22, 3) and set(d["rot_order"].ravel()) == {"x", "y", "z"}
    out = tmp_path / "copy.bvh"
    b.save(str(out))
    np.testing.assert_allclose(BVH().load(str(out)).data["rotations"], d["rotations"], atol=1e-6)


def test_torch_quaternion_helpers_match_the_numpy_ones():
    rs = np.random.RandomState(2)
    a, b, v = _rand_quats(rs, 30), _rand_quats(rs, 30), rs.normal(size=(30, 3))
    np.testing.assert_allclose(quat_torch.mul(torch.tensor(a), torch.tensor(b)).numpy(), quat.mul(a, b), atol=1e-12)
    np.testing.assert_allclose(quat_torch.mul_vec(torch.tensor(a), torch.tensor(v)).numpy(), quat.mul_vec(a, v), atol=1e-12)
    np.testing.assert_allclose(quat_torch.inverse(torch.tensor(a)).numpy(), quat.inverse(a), atol=0)

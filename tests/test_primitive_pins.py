"""CPU: the quaternion / Euler / dual-quaternion / BVH primitives of the host pipeline against HAND-COMPUTED values.

Why this file exists (advisor, round 4): the f1 fixtures come from running the reference through `tools/pymotion_standin`, whose Euler
composition, dual-quaternion layout, `unroll` and BVH reader are adapters over THIS repo's `dragposer_amd/quat_np.py` / `bvh.py` -- a
convention error there would sit on both sides of every f1 comparison and cancel.  The real `upc-pymotion` is absent and the reference
holds no vector for it, so the primitives cannot be pinned to pymotion itself; what CAN be done is to pin them to the conventions they claim
-- the BVH standard (channels in file order compose left to right: R = R_ch0 R_ch1 R_ch2, applied to column vectors), Hamilton (w, x, y, z)
quaternions, dual part 0.5 t (x) r -- with numbers worked out by hand below, independent of any code in the repo.  (The other, indirect,
evidence stays what it was: data prepared with these primitives has the dataset's own mean / std and the TRAINED decoder reconstructs
it to 19.9 mm MPJPE -- tests/test_host_pipeline.py, tests/test_f1_pins.py.)"""
import os

import numpy as np

from dragposer_amd import quat_np as Q
from dragposer_amd.bvh import BVH
from dragposer_amd.motion import dual_part

S = np.sqrt(0.5)


def test_hamilton_product_and_rotation_of_a_vector():
    i, j, k = np.array([0.0, 1, 0, 0]), np.array([0.0, 0, 1, 0]), np.array([0.0, 0, 0, 1])
    np.testing.assert_allclose(Q.mul(i, j), k)            # ij = k
    np.testing.assert_allclose(Q.mul(j, i), -k)           # ji = -k
    np.testing.assert_allclose(Q.mul(j, k), i)
    np.testing.assert_allclose(Q.mul(k, i), j)
    np.testing.assert_allclose(Q.mul(i, i), [-1, 0, 0, 0])
    qx90 = np.array([S, S, 0, 0])                          # 90 degrees about +x, active
    np.testing.assert_allclose(Q.mul_vec(qx90, np.array([0.0, 1, 0])), [0, 0, 1], atol=1e-12)   # y -> z
    np.testing.assert_allclose(Q.mul_vec(qx90, np.array([0.0, 0, 1])), [0, -1, 0], atol=1e-12)  # z -> -y
    np.testing.assert_allclose(Q.to_matrix(qx90), [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-12)
    np.testing.assert_allclose(Q.inverse(qx90), [S, -S, 0, 0])


def test_euler_channels_compose_left_to_right():
    """channels (X, Y, Z) = (90, 0, 90) degrees: R = Rx(90) Rz(90).  On the vector (1, 0, 0): Rz first takes it to (0, 1, 0), Rx then to
    (0, 0, 1).  The other composition (Rz Rx) would leave (0, 1, 0) -- the two orders differ on this input, so the assertion discriminates."""
    q = Q.from_euler(np.radians([90.0, 0.0, 90.0]), "xyz")
    np.testing.assert_allclose(Q.mul_vec(q, np.array([1.0, 0, 0])), [0, 0, 1], atol=1e-12)
    # as a quaternion: (cos45, sin45, 0, 0) (x) (cos45, 0, 0, sin45) = (1/2)(1, 1, -1, 1)   [i k = -j]
    np.testing.assert_allclose(q, [0.5, 0.5, -0.5, 0.5], atol=1e-12)
    # another order string: channels (Z, X, Y) = (90, 90, 0): R = Rz(90) Rx(90); on (0, 1, 0): Rx -> (0, 0, 1), Rz leaves z -> (0, 0, 1)
    q2 = Q.from_euler(np.radians([90.0, 90.0, 0.0]), "zxy")
    np.testing.assert_allclose(Q.mul_vec(q2, np.array([0.0, 1, 0])), [0, 0, 1], atol=1e-12)
    np.testing.assert_allclose(Q.mul_vec(q2, np.array([1.0, 0, 0])), [0, 1, 0], atol=1e-12)     # Rx leaves x, Rz takes x -> y
    for order in ("xyz", "zxy", "zyx", "yxz"):                                                   # to_euler inverts from_euler away from gimbal lock
        a = np.radians([[20.0, -35.0, 50.0], [-100.0, 40.0, 170.0]])
        np.testing.assert_allclose(Q.to_euler(Q.from_euler(a, order), order), a, atol=1e-10)


def test_dual_part_is_half_translation_times_rotation():
    ident = np.array([1.0, 0, 0, 0])
    np.testing.assert_allclose(dual_part(ident, np.array([2.0, 0, 0])), [0, 1, 0, 0])           # 0.5 (0, t) (x) 1
    qz90 = np.array([S, 0, 0, S])
    # 0.5 (0, 0, 2, 0) (x) (S, 0, 0, S) = (0, 1, 0, 0) (x) ... worked out: j (x) (S + S k) = S j + S i  ->  (0, S, S, 0)
    np.testing.assert_allclose(dual_part(qz90, np.array([0.0, 2, 0])), [0, S, S, 0], atol=1e-12)


def test_unroll_keeps_the_hemisphere_of_the_previous_sample():
    q = np.array([[1.0, 0, 0, 0], [-0.99, -0.1, 0, 0], [0.98, 0.2, 0, 0], [-0.97, -0.24, 0, 0]])
    u = Q.unroll(q)
    np.testing.assert_allclose(u, np.abs(q) * np.sign(q[0, 0]))   # every sample flipped into the first one's hemisphere
    assert (np.sum(u[1:] * u[:-1], axis=-1) > 0).all()


def test_root_space_and_back():
    """chain root -> a -> b with local rotations Rz(90), Rx(90), Ry(90): root-space rotation of b = local(a) local(b) (the root's own
    rotation excluded), and the inverse recovers the locals"""
    par = [0, 0, 1]
    loc = np.stack([Q.from_axis_angle("z", np.radians(np.array([90.0]))), Q.from_axis_angle("x", np.radians(np.array([90.0]))),
                    Q.from_axis_angle("y", np.radians(np.array([90.0])))], axis=1)  # [1, 3, 4]
    rs = Q.to_root_space(loc, par)
    np.testing.assert_allclose(rs[0, 1], loc[0, 1])
    np.testing.assert_allclose(rs[0, 2], Q.mul(loc[0, 1], loc[0, 2]))
    np.testing.assert_allclose(rs[0, 2], [0.5, 0.5, 0.5, 0.5], atol=1e-12)   # (S, S, 0, 0)(S, 0, S, 0) = 1/2 (1, 1, 1, 1)  [i j = k]
    np.testing.assert_allclose(Q.from_root_space(rs, par), loc, atol=1e-12)


BVH_TEXT = """HIERARCHY
ROOT Hips
{
  OFFSET 0.0 0.0 0.0
  CHANNELS 6 Xposition Yposition Zposition Xrotation Yrotation Zrotation
  JOINT Spine
  {
    OFFSET 0.0 1.0 0.0
    CHANNELS 3 Xrotation Yrotation Zrotation
    JOINT Head
    {
      OFFSET 0.0 0.5 0.0
      CHANNELS 3 Xrotation Yrotation Zrotation
      End Site
      {
        OFFSET 0.0 0.1 0.0
      }
    }
  }
}
MOTION
Frames: 2
Frame Time: 0.008333
1.0 2.0 3.0 0.0 0.0 0.0 0.0 0.0 0.0 0.0 0.0 0.0
0.0 0.0 0.0 90.0 0.0 0.0 0.0 0.0 90.0 0.0 0.0 0.0
"""


def test_bvh_file_with_joint_positions_worked_out_by_hand(tmp_path):
    """frame 0: no rotation, root at (1, 2, 3): spine (1, 3, 3), head (1, 3.5, 3).  frame 1: root Rx(90): its child's offset (0, 1, 0) goes to
    (0, 0, 1); the spine adds Rz(90) in its local frame: the head's offset (0, 0.5, 0) -> Rz: (-0.5, 0, 0) -> Rx: (-0.5, 0, 0); head at
    (-0.5, 0, 1)."""
    from dragposer_amd.motion import local_quats_from_bvh

    p = tmp_path / "three.bvh"
    p.write_text(BVH_TEXT)
    b = BVH().load(str(p))
    assert b.n_joints == 3 and b.rot_order() == ["xyz", "xyz", "xyz"]  # (End Sites are not joints)
    q, pos, parents, offsets = local_quats_from_bvh(b)
    assert list(parents) == [0, 0, 1]
    np.testing.assert_allclose(offsets, [[0, 0, 0], [0, 1, 0], [0, 0.5, 0]])
    gp, _ = Q.fk(q, pos[:, 0], offsets, list(parents))
    np.testing.assert_allclose(gp[0], [[1, 2, 3], [1, 3, 3], [1, 3.5, 3]], atol=1e-9)
    np.testing.assert_allclose(gp[1], [[0, 0, 0], [0, 0, 1], [-0.5, 0, 1]], atol=1e-9)
    # write and re-read: the MOTION block survives (6 decimals)
    out = tmp_path / "again.bvh"
    b.save(str(out))
    b2 = BVH().load(str(out))
    np.testing.assert_allclose(b2.motion, b.motion, atol=1e-6)

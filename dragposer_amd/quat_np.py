"""NumPy quaternion / kinematics helpers of the host pipeline (w, x, y, z; Hamilton; active).

These stand where the reference calls the un-vendored `upc-pymotion` package off the hot path
(`train.py:330-335,432-433,482-495`, `motion_data.py:52-68,254-274`): Euler <-> quaternion in BVH
channel order, unroll, root-space composition, FK.  Conventions are the BVH standard ones
(R = R_ch0 R_ch1 R_ch2 for channels in file order); they are cross-checked against the trained
model itself in tests/test_host_pipeline.py (dataset statistics and VAE reconstruction).
"""
import numpy as np


def mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), axis=-1)


def inverse(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def mul_vec(q, v):
    t = 2.0 * np.cross(q[..., 1:], v)
    return v + q[..., :1] * t + np.cross(q[..., 1:], t)


def normalize(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def unroll(q, axis=0):
    """flip signs so that consecutive quaternions along `axis` stay in the same hemisphere"""
    q = np.moveaxis(q.copy(), axis, 0)
    for i in range(1, q.shape[0]):
        flip = np.sum(q[i] * q[i - 1], axis=-1) < 0
        q[i][flip] = -q[i][flip]
    return np.moveaxis(q, 0, axis)


_AXIS = {"x": 0, "y": 1, "z": 2}


def from_axis_angle(axis, angle):
    out = np.zeros(angle.shape + (4,))
    out[..., 0] = np.cos(angle / 2)
    out[..., 1 + _AXIS[axis]] = np.sin(angle / 2)
    return out


def from_euler(angles, order):
    """angles [..., 3] in radians, in channel order `order` (e.g. 'xyz'): R = R_o0 R_o1 R_o2."""
    q = from_axis_angle(order[0], angles[..., 0])
    q = mul(q, from_axis_angle(order[1], angles[..., 1]))
    return mul(q, from_axis_angle(order[2], angles[..., 2]))


def to_matrix(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0] = 1 - 2 * (y * y + z * z); m[..., 0, 1] = 2 * (x * y - w * z); m[..., 0, 2] = 2 * (x * z + w * y)
    m[..., 1, 0] = 2 * (x * y + w * z); m[..., 1, 1] = 1 - 2 * (x * x + z * z); m[..., 1, 2] = 2 * (y * z - w * x)
    m[..., 2, 0] = 2 * (x * z - w * y); m[..., 2, 1] = 2 * (y * z + w * x); m[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return m


def to_euler(q, order):
    """inverse of from_euler for the six Tait-Bryan orders; returns radians [..., 3] in channel order."""
    m = to_matrix(normalize(q))
    i, j, k = (_AXIS[c] for c in order)
    sign = 1.0 if (j - i) % 3 == 1 else -1.0  # even / odd permutation
    # R = Ri(a) Rj(b) Rk(c):  R[i,k] = sign*sin(b)
    b = np.arcsin(np.clip(sign * m[..., i, k], -1.0, 1.0))
    a = np.arctan2(-sign * m[..., j, k], m[..., k, k])
    c = np.arctan2(-sign * m[..., i, j], m[..., i, i])
    return np.stack((a, b, c), axis=-1)


def to_root_space(local_rots, parents):
    """local rotations [F, J, 4] -> root-space rotations: the product of the local rotations from the
    root's child down to the joint, the root's own rotation excluded; entry 0 is untouched."""
    out = local_rots.copy()
    for j in range(1, len(parents)):
        p = parents[j]
        if p != 0:
            out[:, j] = mul(out[:, p], local_rots[:, j])
    return out


def from_root_space(root_rots, parents):
    """inverse of to_root_space (the reference's from_root_quat, utils.py:6-31 / train.py:409-434)."""
    out = root_rots.copy()
    for j in reversed(range(1, len(parents))):
        p = parents[j]
        if p != 0:
            out[:, j] = mul(inverse(out[:, p]), out[:, j])
    return out


def fk(local_rots, root_pos, offsets, parents):
    """local rotations [F, J, 4], root positions [F, 3] -> global positions [F, J, 3], rotations [F, J, 4]."""
    F, J = local_rots.shape[:2]
    gr = np.empty((F, J, 4))
    gp = np.empty((F, J, 3))
    gr[:, 0], gp[:, 0] = local_rots[:, 0], root_pos
    for j in range(1, J):
        p = parents[j]
        gr[:, j] = mul(gr[:, p], local_rots[:, j])
        gp[:, j] = gp[:, p] + mul_vec(gr[:, p], np.broadcast_to(offsets[j], (F, 3)))
    return gp, gr

"""pymotion.rotations.quat (numpy): adapters over dragposer_amd/quat_np.py.  Call sites in the reference: train.py:331-335,
432-433,482,487,495; motion_data.py:52-55,62,254-257,264; run_drag.py:136."""
import numpy as np

from dragposer_amd import quat_np as _Q

mul, mul_vec, inverse, normalize, unroll, to_matrix = _Q.mul, _Q.mul_vec, _Q.inverse, _Q.normalize, _Q.unroll, _Q.to_matrix


def _orders(order, n_joints):
    """`order`: array [..., J, 3] of 'x' / 'y' / 'z' (the reference tiles bvh.data["rot_order"] over the frames,
    train.py:330,486) -> one 3-letter string per joint (the order of a joint does not change over time)"""
    o = np.asarray(order)
    o = o.reshape(-1, n_joints, 3)[0]
    return ["".join(str(c) for c in o[j]) for j in range(n_joints)]


def from_euler(euler, order):
    """euler [..., J, 3] radians in channel order, order [..., J, 3] -> quaternions [..., J, 4]"""
    euler = np.asarray(euler, dtype=np.float64)
    J = euler.shape[-2]
    os_ = _orders(order, J)
    return np.stack([_Q.from_euler(euler[..., j, :], os_[j]) for j in range(J)], axis=-2)


def to_euler(q, order):
    q = np.asarray(q, dtype=np.float64)
    J = q.shape[-2]
    os_ = _orders(order, J)
    return np.stack([_Q.to_euler(q[..., j, :], os_[j]) for j in range(J)], axis=-2)

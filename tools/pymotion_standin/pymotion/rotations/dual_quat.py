"""pymotion.rotations.dual_quat (numpy).  Call sites in the reference: motion_data.py:61,68,263,272,359-360.
Convention (this repo's, dragposer_amd/motion.py): dq = (r, 0.5 (0, t) (x) r) -- a rotation r followed by a translation t."""
import numpy as np

from dragposer_amd import quat_np as _Q


def from_rotation_translation(r, t):
    tq = np.concatenate((np.zeros(t.shape[:-1] + (1,)), t), axis=-1)
    return np.concatenate((r, 0.5 * _Q.mul(tq, r)), axis=-1)


def to_rotation_translation(dq):
    r, d = dq[..., :4], dq[..., 4:]
    return r.copy(), 2.0 * _Q.mul(d, _Q.inverse(r))[..., 1:]


def unroll(dq, axis=0):
    """sign continuity along `axis`, decided on the rotation part, applied to all eight numbers"""
    dq = np.moveaxis(dq.copy(), axis, 0)
    for i in range(1, dq.shape[0]):
        flip = np.sum(dq[i][..., :4] * dq[i - 1][..., :4], axis=-1) < 0
        dq[i][flip] = -dq[i][flip]
    return np.moveaxis(dq, 0, axis)

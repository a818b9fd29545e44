"""pymotion.ops.skeleton (numpy).  Call sites in the reference: motion_data.py:58,260 (to_root_dual_quat)."""
import numpy as np

from dragposer_amd import quat_np as _Q

from ..rotations import dual_quat as _dq


def to_root_dual_quat(rotations, global_pos, parents, offsets):
    """local rotations [F, J, 4], root positions [F, 3], parents [J], offsets [J, 3] -> root-centred dual quaternions [F, J, 8]:
    joint 0 = (the root's rotation, its position); joint j > 0 = (rotation of j relative to the root frame, position of j in the
    root frame) -- SURVEY 8.2; the same construction as dragposer_amd/motion.py."""
    rotations = np.asarray(rotations, dtype=np.float64)
    F, J = rotations.shape[:2]
    parents = [int(p) if p is not None else 0 for p in parents]
    rs = _Q.to_root_space(rotations, parents)
    t = np.zeros((F, J, 3))
    for j in range(1, J):
        p = parents[j]
        o = np.broadcast_to(np.asarray(offsets, dtype=np.float64)[j], (F, 3))
        t[:, j] = o if p == 0 else t[:, p] + _Q.mul_vec(rs[:, p], o)
    t[:, 0] = global_pos
    return _dq.from_rotation_translation(rs, t)


def from_root_dual_quat(dq, parents):
    r, t = _dq.to_rotation_translation(dq)
    parents = [int(p) if p is not None else 0 for p in parents]
    return _Q.from_root_space(r, parents), t[:, 0]

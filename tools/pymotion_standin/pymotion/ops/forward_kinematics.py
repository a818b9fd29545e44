"""pymotion.ops.forward_kinematics (numpy): fk(rot [..., J, 4], global_pos [..., 3], offsets [J, 3], parents [J]) ->
positions [..., J, 3], rotation matrices [..., J, 3, 3]."""
import numpy as np

from dragposer_amd import quat_np as _Q


def fk(rot, global_pos, offsets, parents):
    rot = np.asarray(rot, dtype=np.float64)
    lead, J = rot.shape[:-2], rot.shape[-2]
    gp, gr = _Q.fk(rot.reshape(-1, J, 4), np.asarray(global_pos, dtype=np.float64).reshape(-1, 3), np.asarray(offsets, dtype=np.float64),
                   [int(p) for p in parents])
    return gp.reshape(lead + (J, 3)), _Q.to_matrix(gr).reshape(lead + (J, 3, 3))

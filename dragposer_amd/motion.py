"""Host-side motion preprocessing for the evaluation pipeline: BVH arrays -> the reference's
normalised root-space dual-quaternion representation (motion_data.py:225-324, train.py:329-341).

Representation: 22 joints x 8 channels.  Joint j>0: rotation of j relative to the root frame
(4) + dual part 0.5*(0,t_j)(x)r_j with t_j the joint position in the root frame (4).  Joint 0: the root's
incremental rotation inv(R[t-1]) R[t] (4) + the root displacement in the current root frame (3) + 0.
"""
import numpy as np

from . import quat_np as Q
from .model import NJ


def local_quats_from_bvh(bvh):
    """train.get_info_from_bvh (train.py:329-341): unrolled unit local quaternions, positions, parents, offsets."""
    rot_deg, pos, parents, offsets, order = bvh.get_data()
    q = np.stack([Q.from_euler(np.radians(rot_deg[:, j]), order[j]) for j in range(rot_deg.shape[1])], axis=1)
    q = Q.normalize(Q.unroll(q, axis=0))
    return q, pos, parents, offsets


def root_space_translations(root_rots, offsets, parents):
    """positions of the joints in the root frame (root at the origin, root rotation excluded)"""
    F = root_rots.shape[0]
    t = np.zeros((F, NJ, 3))
    for j in range(1, NJ):
        p = parents[j]
        o = np.broadcast_to(offsets[j], (F, 3))
        t[:, j] = o if p == 0 else t[:, p] + Q.mul_vec(root_rots[:, p], o)
    return t


def dual_part(r, t):
    tq = np.concatenate((np.zeros(t.shape[:-1] + (1,)), t), axis=-1)
    return 0.5 * Q.mul(tq, r)


def prepare_motion(bvh, means, stds, height_indices=(0, 4, 8, 13, 17, 21)):
    """-> dict with normalised dqs [F,176], global_pos [F,3], global_rot [F,4], heights [F,6], root-space unit
    quaternions [F,22,4] (entry 0 = incremental root rotation) and parents / offsets."""
    q, pos, parents, offsets = local_quats_from_bvh(bvh)
    F = q.shape[0]
    global_pos = pos[:, 0].copy()
    global_rot = q[:, 0].copy()
    disp = np.concatenate((np.zeros((1, 3)), global_pos[1:] - global_pos[:-1]), axis=0)
    disp = Q.mul_vec(Q.inverse(global_rot), disp)  # root-space displacement (motion_data.py:249-254)
    incr = global_rot.copy()
    incr[1:] = Q.mul(Q.inverse(global_rot[:-1]), global_rot[1:])
    incr[0] = [1.0, 0.0, 0.0, 0.0]
    rs = Q.to_root_space(q, parents)
    t = root_space_translations(rs, offsets, parents)
    world = Q.mul_vec(global_rot[:, None], t) + global_pos[:, None]
    heights = world[:, list(height_indices), 1]
    dq = np.zeros((F, NJ, 8))
    dq[:, :, :4] = rs
    dq[:, :, 4:] = dual_part(rs, t)
    dq[:, 0, :4] = incr
    # sign continuity of every joint's dual quaternion along time (dquat.unroll, motion_data.py:272)
    for j in range(NJ):
        for i in range(1, F):
            if np.dot(dq[i, j, :4], dq[i - 1, j, :4]) < 0:
                dq[i, j] = -dq[i, j]
    dq[:, 0, 4:7] = disp
    dq[:, 0, 7] = 0.0
    flat = dq.reshape(F, NJ * 8)
    norm = (flat - means["dqs"]) / stds["dqs"]
    root_quats = dq[:, :, :4].copy()
    return dict(dqs=norm.astype(np.float32), dqs_raw=flat, global_pos=global_pos.astype(np.float32),
                global_rot=global_rot.astype(np.float32), heights=heights.astype(np.float32), root_quats=root_quats,
                parents=parents, offsets=offsets.astype(np.float32), displacement=disp)

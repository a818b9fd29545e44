"""pymotion.ops.forward_kinematics_torch.  Call sites in the reference: eval_drag.py:190 (per-frame targets), eval_metrics.py:14,24.
fk(rot [..., J, 4], global_pos [..., 3], offsets [J, 3], parents [J]) -> positions [..., J, 3], rotation matrices [..., J, 3, 3],
in the tensors' own dtype (the reference calls it on fp32 tensors)."""
import torch

from ..rotations import quat_torch as _qt


def _to_matrix(q):
    w, x, y, z = q.unbind(-1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=-1).reshape(q.shape[:-1] + (3, 3))


def fk(rot, global_pos, offsets, parents):
    J = rot.shape[-2]
    par = [int(p) for p in parents]
    gr, gp = [rot[..., 0, :]], [global_pos.to(rot.dtype).reshape(rot.shape[:-2] + (3,))]
    for j in range(1, J):
        p = par[j]
        gr.append(_qt.mul(gr[p], rot[..., j, :]))
        gp.append(gp[p] + _qt.mul_vec(gr[p], offsets[j].to(rot.dtype).expand(rot.shape[:-2] + (3,))))
    gr, gp = torch.stack(gr, dim=-2), torch.stack(gp, dim=-2)
    return gp, _to_matrix(gr)

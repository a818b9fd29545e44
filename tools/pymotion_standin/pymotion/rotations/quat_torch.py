"""The four quaternion helpers of pymotion.rotations.quat_torch used on the hot path.

Call sites in the reference: mul -> drag_pose.py:88, utils.py:30; mul_vec -> drag_pose.py:102;
inverse -> utils.py:29,96; normalize -> autoencoder.py:248.  Quaternions are (w, x, y, z).
"""
import torch


def mul(q0, q1):
    w0, x0, y0, z0 = q0[..., 0:1], q0[..., 1:2], q0[..., 2:3], q0[..., 3:4]
    w1, x1, y1, z1 = q1[..., 0:1], q1[..., 1:2], q1[..., 2:3], q1[..., 3:4]
    w = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1
    x = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1
    y = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1
    z = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1
    return torch.cat((w, x, y, z), dim=-1)


def mul_vec(q, v):
    t = 2.0 * torch.cross(q[..., 1:], v, dim=-1)
    return v + q[..., 0:1] * t + torch.cross(q[..., 1:], t, dim=-1)


def inverse(q):
    return q * torch.tensor([1.0, -1.0, -1.0, -1.0], device=q.device, dtype=q.dtype)


def normalize(q):
    return q / torch.linalg.norm(q, dim=-1, keepdim=True)

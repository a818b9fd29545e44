"""NumPy model of the arithmetic the HIP kernel's kinematics phase (P3) performs, per frame.

This is the *kernel's* formulation -- everything after the root composition is done in the ROOT
frame (targets are rotated by R0^T instead of rotating every joint by R0), and the root
quaternion's gradient is assembled from per-tracker contributions -- written down here so that
the derivation can be checked on the CPU against the oracle before (and independently of) the
GPU.  Used by tests/test_kernel_model.py only.
"""
import numpy as np

NJ = 22


def qmat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def qmat_grad(q, X):
    w, x, y, z = q
    return 2 * np.array([
        -z * X[0, 1] + y * X[0, 2] + z * X[1, 0] - x * X[1, 2] - y * X[2, 0] + x * X[2, 1],
        y * X[0, 1] + z * X[0, 2] + y * X[1, 0] - 2 * x * X[1, 1] - w * X[1, 2] + z * X[2, 0] + w * X[2, 1] - 2 * x * X[2, 2],
        -2 * y * X[0, 0] + x * X[0, 1] + w * X[0, 2] + x * X[1, 0] + z * X[1, 2] - w * X[2, 0] + z * X[2, 1] - 2 * y * X[2, 2],
        -2 * z * X[0, 0] - w * X[0, 1] + x * X[0, 2] + w * X[1, 0] - 2 * z * X[1, 1] + y * X[1, 2] + x * X[2, 0] + y * X[2, 1]])


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def p3(y, c, parents, offsets, mu4, sd4, mu_d, sd_d, tp, tR, w, trk, lam_rot):
    """y[92] raw decoder output -> (loss_pos, loss_rot, gy[92], world outputs) the kernel's way."""
    y = np.asarray(y, np.float64)
    children = [[k for k in range(1, NJ) if parents[k] == j] for j in range(NJ)]
    path = [[] for _ in range(NJ)]
    for j in range(1, NJ):
        path[j] = path[parents[j]] + [j]
    sub = [[t for t in range(NJ) if j in ([0] + path[t] if j == 0 else path[t])] for j in range(NJ)]
    E = int(np.sum(trk))
    r = (y[:88] * sd4 + mu4).reshape(NJ, 4)
    inv = 1.0 / np.sqrt((r * r).sum(1))
    q = r * inv[:, None]
    d = y[88:91] * sd_d + mu_d
    qw = qmul(c, q[0])
    R0 = qmat(qw)
    M = [np.eye(3)] + [qmat(q[j]) for j in range(1, NJ)]
    bone = np.zeros((NJ, 3))
    for j in range(NJ):
        for k in children[j]:
            bone[k] = M[j] @ offsets[k]
    pr = np.array([d + sum((bone[k] for k in path[j]), np.zeros(3)) for j in range(NJ)])  # root frame
    lp = lr = 0.0
    gp = np.zeros((NJ, 3))
    gM = np.zeros((NJ, 3, 3))
    gqw = np.zeros(4)
    for t in range(NJ):
        if not trk[t]:
            continue
        cp, cr = w[t, 0] / (3 * E), lam_rot * w[t, 1] / (9 * E)
        tpr = R0.T @ tp[t]
        e = pr[t] - tpr
        lp += cp * (e @ e)
        gp[t] = 2 * cp * e
        tRr = R0.T @ tR[t].reshape(3, 3)
        eM = M[t] - tRr
        lr += cr * (eM * eM).sum()
        gM[t] = 2 * cr * eM
        C = -(np.outer(tp[t], gp[t]) + tR[t].reshape(3, 3) @ gM[t].T)
        gqw += qmat_grad(qw, C)
    gy = np.zeros(92)
    for j in range(1, NJ):
        X = gM[j].copy()
        for k in children[j]:
            S = sum((gp[t] for t in sub[k]), np.zeros(3))
            X += np.outer(S, offsets[k])
        gq = qmat_grad(q[j], X)
        gr = (gq - q[j] * (q[j] @ gq)) * inv[j]
        gy[4 * j:4 * j + 4] = sd4[4 * j:4 * j + 4] * gr
    gq0 = qmul(np.array([c[0], -c[1], -c[2], -c[3]]), gqw)
    gr0 = (gq0 - q[0] * (q[0] @ gq0)) * inv[0]
    gy[0:4] = sd4[0:4] * gr0
    gy[88:91] = sd_d * gp.sum(0)
    pos = (R0 @ pr.T).T
    rot = np.array([R0 @ M[j] for j in range(NJ)])
    return lp, lr, gy, dict(pos=pos, rot=rot, world_disp=R0 @ d, world_rot=qw, q=q)


def frame_grad(fold, y_fn_out, z, z_tgt, lam_tmp, gy, a0, a1):
    d1 = (fold["A2"].T @ gy) * np.where(a1 > 0, 1.0, 0.2)
    d0 = (fold["A1"].T @ d1) * np.where(a0 > 0, 1.0, 0.2)
    return fold["A0"].T @ d0 + 2 * lam_tmp * (z - z_tgt) / 24


def decode(fold, z):
    pre0 = fold["A0"] @ z + fold["c0"]
    a0 = np.maximum(pre0, 0.2 * pre0)
    pre1 = fold["A1"] @ a0 + fold["b1"]
    a1 = np.maximum(pre1, 0.2 * pre1)
    return fold["A2"] @ a1 + fold["b2"], a0, a1

"""NumPy (float64) emulation of ONE iteration of the wave-private kernel (dragposer_amd/csrc/dp_w4.hip), driven by the
very tables the host packs for it (dp_debug_pack_w4, dp_debug_pairs_w4, dp_debug_items): the product steps of the weight
image with the kernel's channel <-> (ABID, register) map, the three kinematics stages with the kernel's formulas (torque
form of the gradient, quaternion form of the rotation error), and the backward products.  It checks on the CPU what the
GPU tests check on the device: that the packing and the restated mathematics reproduce the oracle's loss and dL/dz."""
import ctypes as C

import numpy as np

from dragposer_amd import _lib

S_L0, S_L1, S_L2A, S_L2B, S_B2, S_B1, S_B0, N_STEPS = 0, 24, 64, 124, 184, 288, 348, 368
H0_ROW = np.array([c if c < 20 else 12 + c for c in range(40)])  # dp_w4.h: row (lane of layout D) of the first hidden layer's channel c
KIND_JOINT, KIND_ROOT, KIND_DISP, KIND_VIRT, KIND_IDLE = 0, 1, 2, 3, 4
PAIR = np.dtype([("sd", "<f4", (4, 2)), ("mu", "<f4", (4, 2)), ("off", "<f4", (3, 2)), ("sgn", "<f4", 2), ("rho", "<f4", 2),
                 ("item", "<i4", 2), ("kind", "<i4", 2), ("bone_slot", "<i4", 2), ("ch_sub", "<u4", 2), ("pad", "<i4", 2)])
ITEM = np.dtype([("sd", "<f4", 4), ("mu", "<f4", 4), ("ch_off", "<f4", 3), ("ch_id", "<i4"), ("ch_sub", "<u4"), ("path_lo", "<u4"),
                 ("path_hi", "<u4"), ("src_quad", "<i4"), ("dst_quad", "<i4"), ("kind", "<i4"), ("init_id", "<i4"),
                 ("init_off", "<f4", 3), ("pad", "<i4", 10)])
assert PAIR.itemsize == 144 and ITEM.itemsize == 128


def host_tables(hm):
    lib = _lib.load()
    _, folded = hm.fold()
    img = np.zeros((N_STEPS // 4, 64, 4), np.float32)
    bias = np.zeros((4, 64), np.float32)
    rc = lib.dp_debug_pack_w4(C.byref(folded), C.byref(hm.struct), img.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p))
    assert rc == 0, _lib.last_error()
    pairs = np.zeros(16, PAIR)
    assert lib.dp_debug_pairs_w4(C.byref(hm.struct), pairs.ctypes.data_as(C.c_void_p)) == 0
    items = np.zeros(32, ITEM)
    assert lib.dp_debug_items(C.byref(hm.struct), items.ctypes.data_as(C.c_void_p)) == 0
    return img, bias, pairs, items


def step_rows(img, s):
    """the 64 weight rows of step s (what lane l holds as the MFMA's B operand)"""
    return img[s >> 2, :, s & 3].astype(np.float64)


def product(img, s0, x):
    """sum over K-steps: step s0 + k multiplies channel k of x (X layout: quad k >> 2 = ABID, register k & 3)"""
    out = np.zeros(64)
    for k, xk in enumerate(x):
        out += step_rows(img, s0 + k) * xk
    return out


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def rot(q, a):  # R(q) a
    v = q[1:]
    t = np.cross(v, a)
    return a + 2.0 * (q[0] * t + np.cross(v, t))


def quat_from_rotmat(m):
    m = np.asarray(m, np.float64).reshape(3, 3)
    tr = np.trace(m)
    if tr > 0:
        s = np.sqrt(tr + 1) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = np.sqrt(1 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def lrelu(x):
    return np.maximum(x, 0.2 * x)


def layer2(img, bias, a1):
    """{side: [quad][channel]}: row 4b + r of block blk holds channel 2 blk + (r >> 1) of the side-(r & 1) item of quad b (dp_w4.h)"""
    blocks = [(product(img, s0, a1) + bias[2 + k]).reshape(16, 4) for k, s0 in enumerate((S_L2A, S_L2B))]
    return {s: np.stack([blocks[c >> 1][:, 2 * (c & 1) + s] for c in range(4)], axis=1) for s in (0, 1)}


def one_iteration(tables, z, z_tgt, cur, tgt_pos, tgt_rot, w, tracked, lam_rot=1.0, lam_tmp=0.02):
    """one frame: (loss_pos, loss_rot, loss_tmp), dL/dz[24], and the per-item unit quaternions / bones it decoded"""
    img, bias, pairs, items = tables
    z = np.asarray(z, np.float64)
    # ---- decoder forward (accumulators start from the bias rows; L2 leaves DE-NORMALISED channels)
    a0 = lrelu(product(img, S_L0, z) + bias[0])[H0_ROW]
    a1 = lrelu(product(img, S_L1, a0) + bias[1])[:60]
    rq = layer2(img, bias, a1)
    # ---- stage J
    QS, BN = np.zeros((32, 4)), np.zeros((32, 3))
    QS[30] = [1, 0, 0, 0]
    for k in range(3):  # constant bones of the root's children
        BN[items["init_id"][k]] = items["init_off"][k]
    J = {}
    for b in range(16):
        for s in range(2):
            kind, item = pairs["kind"][b][s], pairs["item"][b][s]
            r = rq[s][b]
            inv = 1.0 if kind == KIND_DISP else (0.0 if kind == KIND_IDLE else 1.0 / np.linalg.norm(r))
            q = r * inv
            u = rot(q, pairs["off"][b][:, s].astype(np.float64)) if kind != KIND_IDLE else np.zeros(3)
            J[b, s] = (q, u, inv)
            QS[item if item >= 0 else 31] = q
            BN[pairs["bone_slot"][b][s]] = u
    BN[23] = 0  # SLOT_ZERO is never written
    # ---- stage T
    joints = [j for j in range(22) if tracked[j]]
    E = len(joints)
    GP, RT, WT = np.zeros((24, 3)), np.zeros((24, 3)), np.zeros((32, 3))
    lp = lr = 0.0
    curq = np.asarray(cur, np.float64)
    for rank, j in enumerate(joints):
        tp = rot(conj(curq), np.asarray(tgt_pos[j], np.float64))
        qT = qmul(conj(curq), quat_from_rotmat(tgt_rot[j]))
        clp, clr = w[j][0] / (3.0 * E), lam_rot * w[j][1] / (9.0 * E)
        q0, qt, d = QS[0], (QS[30] if j == 0 else QS[j]), QS[22][:3]
        path = [(int(items["path_lo"][j]) >> (5 * k)) & 31 for k in range(6)] + [int(items["path_hi"][j]) & 31]
        p = d + sum(BN[s] for s in path)
        at = rot(conj(q0), tp)
        e = p - at
        gp = 2.0 * clp * e
        sq = qmul(qmul(conj(q0), qT), conj(qt))
        own = -8.0 * clr * sq[0] * sq[1:]
        GP[rank], RT[rank] = gp, np.cross(at, gp) + own
        WT[31 if j == 0 else j] = own
        lp += clp * (e @ e)
        lr += 8.0 * clr * (sq[1:] @ sq[1:])
    WT[30] = 0
    # ---- stage G
    gy = {0: np.zeros((16, 4)), 1: np.zeros((16, 4))}
    for b in range(16):
        tab = RT if pairs["kind"][b][0] == KIND_ROOT else GP
        for s in range(2):
            kind, item = pairs["kind"][b][s], pairs["item"][b][s]
            q, u, inv = J[b, s]
            S = sum((tab[rank] for rank, j in enumerate(joints) if (int(pairs["ch_sub"][b][s]) >> j) & 1), np.zeros(3))
            wt = WT[item] if kind == KIND_JOINT else WT[30]
            tau = np.cross(u, S) + pairs["rho"][b][s] * S + wt
            a = 2.0 * tau
            g = np.concatenate([[-(a @ q[1:])], q[0] * a + pairs["sgn"][b][s] * np.cross(a, q[1:])]) * inv
            if kind == KIND_DISP:
                g = np.array([S[0], S[1], S[2], 0.0])
            gy[s][b] = g
    # ---- decoder backward: K = channels of the 16 side-A items, then of side-B quads 1..10
    gyk = np.concatenate([gy[0].reshape(64), gy[1][1:11].reshape(40)])
    d1 = product(img, S_B2, gyk)[:60] * np.where(a1 > 0, 1.0, 0.2)
    d0 = product(img, S_B1, d1)[H0_ROW] * np.where(a0 > 0, 1.0, 0.2)
    # bL0 runs two K-steps per instruction: lanes 0..31 against channels 0..19, lanes 32..63 against channels 20..39
    half = np.zeros(64)
    for k in range(20):
        w = step_rows(img, S_B0 + k)
        half[:32] += w[:32] * d0[k]
        half[32:] += w[32:] * d0[20 + k]
    gz = (half[:32] + half[32:])[:24] + 2.0 * lam_tmp / 24.0 * (z - z_tgt)
    lt = lam_tmp * np.mean((z - z_tgt) ** 2)
    return (lp, lr, lt), gz, QS, BN

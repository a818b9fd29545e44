"""Lane-level NumPy emulation of one iteration of dp_optimize_kernel for ONE workgroup (16 frames).

It consumes the REAL host tables (dp_debug_pack / dp_debug_items from libdragposer_hip.so) and
re-enacts the kernel's data movement -- MFMA operand lanes, K-half planes, kcol column order,
per-item P3 roles incl. virtual items -- so that the host packing and the index algebra can be
verified on the CPU (tests/test_kernel_emu.py) against the oracle.  Test infrastructure only.
"""
import ctypes as C

import numpy as np

from dragposer_amd import _lib

NWAVE, W_REGS, NGEMM = 8, 84, 6
W_OFF = dict(L0=0, L1=6, L2=16, B2=32, B1=58, B0=74)
G_L0, G_L1, G_L2, G_B2, G_B1, G_B0 = range(6)
S_Y = 108


ITEM_DT = np.dtype([("sd", "f4", 4), ("mu", "f4", 4), ("ch_off", "f4", 3), ("ch_id", "i4"), ("ch_sub", "u4"),
                    ("path_lo", "u4"), ("path_hi", "u4"), ("src_quad", "i4"), ("dst_quad", "i4"), ("kind", "i4"),
                    ("init_id", "i4"), ("init_off", "f4", 3), ("pad", "i4", 10)])
assert ITEM_DT.itemsize == 128


def host_tables(host_model):
    lib = _lib.load()
    _, folded = host_model.fold()
    wfrag = np.zeros((NWAVE, W_REGS, 64), np.float32)
    bias = np.zeros((2, 64), np.float32)
    smask = np.zeros((NWAVE, NGEMM), np.uint32)
    rc = lib.dp_debug_pack(C.byref(folded), host_model.parents.ctypes.data_as(C.POINTER(C.c_int)),
                           wfrag.ctypes.data_as(C.POINTER(C.c_float)), bias.ctypes.data_as(C.POINTER(C.c_float)),
                           smask.ctypes.data_as(C.POINTER(C.c_uint)))
    assert rc == 0, _lib.last_error()
    items = np.zeros(32, ITEM_DT)
    rc = lib.dp_debug_items(C.byref(host_model.struct), items.ctypes.data_as(C.c_void_p))
    assert rc == 0, _lib.last_error()
    return wfrag, bias, items, smask


def mfma(a, b, acc):
    """v_mfma_f32_16x16x4_f32: a,b [64]; acc [64,4] (lane l: rows 4*(l>>4)+r, col l&15)."""
    A = a.reshape(4, 16).T.astype(np.float64)  # A[m][k] = a[m + 16k]
    Bm = b.reshape(4, 16).astype(np.float64)   # B[k][n] = b[n + 16k]
    D = A @ Bm                                 # [16 m][16 n]
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def load_b(buf_rows, step0, n):
    """B-operand floats of all 64 lanes for steps step0..step0+n-1: act[frame l&15][4*step + (l>>4)]."""
    out = np.zeros((n, 64))
    for l in range(64):
        for i in range(n):
            out[i, l] = buf_rows[l & 15, 4 * (step0 + i) + (l >> 4)]
    return out


def chain(wf, wave, off, b, mask, acc0=None):
    """two interleaved accumulators, masked steps (kernel: mfma_chain)."""
    acc = [np.zeros((64, 4)) if acc0 is None else acc0.copy(), np.zeros((64, 4))]
    for i in range(b.shape[0]):
        if (int(mask) >> i) & 1:
            acc[i & 1] = mfma(wf[wave, off + i], b[i], acc[i & 1])
        else:
            assert not np.any(wf[wave, off + i]), "masked step must have zero weights"
    return acc[0] + acc[1]


def store_tile(buf_rows, tile, acc):
    for l in range(64):
        buf_rows[l & 15, 16 * tile + 4 * (l >> 4): 16 * tile + 4 * (l >> 4) + 4] = acc[l]


def bias_frag(bias_row, tile):
    out = np.zeros((64, 4))
    for l in range(64):
        out[l] = bias_row[16 * tile + 4 * (l >> 4): 16 * tile + 4 * (l >> 4) + 4]
    return out


def lrelu(x):
    return np.maximum(x, 0.2 * x)


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mat_grad(q, X):
    w, x, y, z = q
    a, b, c = X[2, 1] - X[1, 2], X[0, 2] - X[2, 0], X[1, 0] - X[0, 1]
    s01, s02, s12 = X[0, 1] + X[1, 0], X[0, 2] + X[2, 0], X[1, 2] + X[2, 1]
    return 2 * np.array([x * a + y * b + z * c,
                         w * a + y * s01 + z * s02 - 2 * x * (X[1, 1] + X[2, 2]),
                         w * b + x * s01 + z * s12 - 2 * y * (X[0, 0] + X[2, 2]),
                         w * c + x * s02 + y * s12 - 2 * z * (X[0, 0] + X[1, 1])])


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def emulate_iteration(tables, z, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, lam_rot, lam_tmp):
    """One decode -> P3 -> backward for 16 frames.  Returns y[16,104], gy[16,104], gz[16,24], loss[16,2]."""
    wfrag, bias, items, smask = tables
    wf = wfrag.astype(np.float64)
    z = np.asarray(z, np.float64)
    zrows = np.zeros((16, 28)); zrows[:, :24] = z
    # ---- L0 (waves 0..2), L1 (waves 0..3): bias + LeakyReLU in the producer
    a0 = np.zeros((16, 52)); a0v = {}
    for wv in range(3):
        a0v[wv] = lrelu(chain(wf, wv, W_OFF["L0"], load_b(zrows, 0, 6), 0x3F, bias_frag(bias[0], wv)))
        store_tile(a0, wv, a0v[wv])
    a1 = np.zeros((16, 68)); a1v = {}
    for wv in range(4):
        a1v[wv] = lrelu(chain(wf, wv, W_OFF["L1"], load_b(a0, 0, 10), smask[wv, G_L1], bias_frag(bias[1], wv)))
        store_tile(a1, wv, a1v[wv])
    assert np.all(a1[:, 60] == 1.0)
    # ---- L2: tiles 0..3 on waves 0..3, tiles 4,5 in two K-halves on waves 4..7 (two planes)
    yp = np.zeros((2, 16, S_Y))
    for wv in range(4):
        store_tile(yp[0], wv, chain(wf, wv, W_OFF["L2"], load_b(a1, 0, 16), smask[wv, G_L2]))
    for wv in range(4, 8):
        half = ((wv >> 1) & 1) ^ (0 if wv & 1 else 1)  # dp_layout.h: l2_half
        store_tile(yp[half], 4 + (wv & 1), chain(wf, wv, W_OFF["L2"], load_b(a1, 8 * half, 8), smask[wv, G_L2]))
    ysum = yp[0] + yp[1]
    # ---- P3 per frame, per item
    gy = np.zeros((16, S_Y))
    loss = np.zeros((16, 2))
    for f in range(16):
        trk = np.asarray(tracked[f]) != 0
        tmask = sum(1 << j for j in range(22) if trk[j])
        E = int(trk.sum())
        ranks = {j: bin(tmask & ((1 << j) - 1)).count("1") for j in range(22)}
        bone = np.zeros((32, 4)); gpc = np.zeros((24, 4)); cq = np.zeros((24, 4)); qd = np.zeros(8)
        for k in range(3):
            bone[items[k]["init_id"], :3] = items[k]["init_off"]
        st = {}
        for it in range(32):
            ic = items[it]
            y4 = ysum[f, 4 * ic["src_quad"]: 4 * ic["src_quad"] + 4]
            r = y4 * ic["sd"] + ic["mu"]
            has_quat = ic["kind"] in (0, 1, 3)
            inv = 1.0 / np.sqrt(r @ r) if has_quat else 0.0
            q = r * inv
            M = quat_to_mat(q)
            if ic["kind"] == 1:
                qd[:4] = quat_mul(cur_rot[f], q)
                M = np.eye(3)
            if ic["kind"] == 2:
                qd[4:7] = r[:3]
            st[it] = (r, inv, q, M)
        for it in range(32):  # all bone writes precede the reads (wave_sync)
            ic = items[it]
            bone[ic["ch_id"], :3] = st[it][3] @ ic["ch_off"]
        assert np.all(bone[23] == 0)
        qw, d = qd[:4], qd[4:7]
        R0 = quat_to_mat(qw)
        gM = {it: np.zeros((3, 3)) for it in range(32)}
        for it in range(32):
            ic = items[it]
            p = d.copy()
            for i in range(7):
                k = (int(ic["path_lo"]) >> (5 * i)) & 31 if i < 6 else int(ic["path_hi"]) & 31
                p = p + bone[k, :3]
            if it < 22 and trk[it]:
                tp, tR = tgt_pos[f, it].astype(np.float64), tgt_rot[f, it].reshape(3, 3).astype(np.float64)
                clp, clr = w[f, it, 0] / (3 * E), lam_rot * w[f, it, 1] / (9 * E)
                e = p - R0.T @ tp
                gp = 2 * clp * e
                eM = st[it][3] - R0.T @ tR
                gM[it] = 2 * clr * eM
                Cm = -(np.outer(tp, gp) + tR @ gM[it].T)
                gpc[ranks[it], :3] = gp
                cq[ranks[it]] = quat_mat_grad(qw, Cm)
                loss[f, 0] += clp * (e @ e)
                loss[f, 1] += clr * (eM * eM).sum()
        order = [j for j in range(22) if trk[j]]
        for it in range(32):
            ic = items[it]
            r, inv, q, M = st[it]
            S = np.zeros(3)
            for e_, t in enumerate(order):
                if (int(ic["ch_sub"]) >> t) & 1:
                    S += gpc[e_, :3]
            if ic["kind"] == 1:
                gq = quat_mul(cur_rot[f] * np.array([1, -1, -1, -1]), cq[:E].sum(0))
            else:
                gq = quat_mat_grad(q, gM[it] + np.outer(S, ic["ch_off"]))
            gyv = ic["sd"] * (gq - q * (q @ gq)) * inv
            if ic["kind"] == 2:
                gyv = np.array([ic["sd"][0] * S[0], ic["sd"][1] * S[1], ic["sd"][2] * S[2], 0.0])
            if ic["dst_quad"] >= 0:
                gy[f, 4 * ic["dst_quad"]: 4 * ic["dst_quad"] + 4] = gyv
    # ---- backward products: mask by the producer's own activations
    d1 = np.zeros((16, 68))
    for wv in range(4):
        acc = chain(wf, wv, W_OFF["B2"], load_b(gy, 0, 26), smask[wv, G_B2])
        store_tile(d1, wv, np.where(a1v[wv] > 0, acc, 0.2 * acc))
    d0 = np.zeros((16, 52))
    for wv in range(3):
        acc = chain(wf, wv, W_OFF["B1"], load_b(d1, 0, 16), smask[wv, G_B1])
        store_tile(d0, wv, np.where(a0v[wv] > 0, acc, 0.2 * acc))
    gzr = np.zeros((16, 36))
    for wv in range(2):
        store_tile(gzr, wv, chain(wf, wv, W_OFF["B0"], load_b(d0, 0, 10), 0x3FF))
    gz = gzr[:, :24] + 2 * lam_tmp * (z - np.asarray(z_tgt, np.float64)) / 24
    return ysum[:, :104], gy[:, :104], gz, loss

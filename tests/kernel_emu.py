"""Lane-level NumPy emulation of one iteration of dp_optimize_kernel for ONE workgroup (16 frames).

It consumes the REAL host tables (dp_debug_pack / dp_debug_items from libdragposer_hip.so) and
re-enacts the kernel's data movement -- MFMA operand lanes, K-half planes, kcol column order,
per-item P3 roles incl. virtual items -- so that the host packing and the index algebra can be
verified on the CPU (tests/test_kernel_emu.py) against the oracle.  Test infrastructure only.
"""
import ctypes as C

import numpy as np

from dragposer_amd import _lib

NWAVE, W_REGS = 8, 51
W_OFF = dict(L0=0, L1=3, L2A=8, L2B=16, B2=24, B1=38, B0=46)
S_Y = 108


def kcol(K, i, h):
    nb = (K // 16) * 4
    return 16 * (i // 4) + 4 * h + (i % 4) if i < nb else (K // 16) * 16 + 2 * h + (i - nb)


ITEM_DT = np.dtype([("sd", "f4", 4), ("mu", "f4", 4), ("ch_off", "f4", 3), ("ch_id", "i4"), ("ch_sub", "u4"),
                    ("path_lo", "u4"), ("path_hi", "u4"), ("src_quad", "i4"), ("dst_quad", "i4"), ("kind", "i4"),
                    ("init_id", "i4"), ("init_off", "f4", 3), ("pad", "i4", 10)])
assert ITEM_DT.itemsize == 128


def host_tables(host_model):
    lib = _lib.load()
    _, folded = host_model.fold()
    wfrag = np.zeros((NWAVE, W_REGS, 64), np.float32)
    bias = np.zeros((2, 64), np.float32)
    rc = lib.dp_debug_pack(C.byref(folded), host_model.parents.ctypes.data_as(C.POINTER(C.c_int)),
                           wfrag.ctypes.data_as(C.POINTER(C.c_float)), bias.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, _lib.last_error()
    items = np.zeros(32, ITEM_DT)
    rc = lib.dp_debug_items(C.byref(host_model.struct), items.ctypes.data_as(C.c_void_p))
    assert rc == 0, _lib.last_error()
    return wfrag, bias, items


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


def load_row(buf_rows, K):
    """B-operand fragments of all 64 lanes: buf_rows [16][stride] -> [K/4][64]."""
    out = np.zeros((K // 4, 64))
    for l in range(64):
        f, h = l & 15, l >> 4
        for i in range(K // 4):
            out[i, l] = buf_rows[f, kcol(K, i, h)]
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
    wfrag, bias, items = tables
    wf = wfrag.astype(np.float64)
    z = np.asarray(z, np.float64)
    zrows = np.zeros((16, 28)); zrows[:, :24] = z
    zf = load_row(zrows, 24)
    # ---- L0
    a0p = np.zeros((2, 16, 52))
    for wv in range(6):
        t, hf = wv % 3, wv // 3
        acc = np.zeros((64, 4))
        if hf == 0:
            for l in range(64):
                acc[l] = bias[0, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4]
        for i in range(3):
            acc = mfma(wf[wv, W_OFF["L0"] + i], zf[3 * hf + i], acc)
        for l in range(64):
            a0p[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc[l]
    a0f = lrelu(load_row(a0p[0], 40) + load_row(a0p[1], 40))
    # ---- L1
    a1p = np.zeros((2, 16, 68))
    for wv in range(8):
        t, hf = wv & 3, wv >> 2
        acc = np.zeros((64, 4))
        if hf == 0:
            for l in range(64):
                acc[l] = bias[1, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4]
        for i in range(5):
            acc = mfma(wf[wv, W_OFF["L1"] + i], a0f[5 * hf + i], acc)
        for l in range(64):
            a1p[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc[l]
    a1f = lrelu(load_row(a1p[0], 64) + load_row(a1p[1], 64))
    a1f[12, 48:64] = 1.0  # lanes h == 3: constant-1 column 60
    # ---- L2
    yp = np.full((2, 16, S_Y), np.nan)
    for wv in range(8):
        t, hf = wv % 6, wv // 6
        acc0 = np.zeros((64, 4)); acc1 = np.zeros((64, 4))
        for i in range(8):
            acc0 = mfma(wf[wv, W_OFF["L2A"] + i], a1f[8 * hf + i], acc0)
            if wv < 4:
                acc1 = mfma(wf[wv, W_OFF["L2B"] + i], a1f[8 + i], acc1)
        for l in range(64):
            yp[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc0[l]
            if wv < 4:
                yp[1, l & 15, 16 * (wv + 2) + 4 * (l >> 4): 16 * (wv + 2) + 4 * (l >> 4) + 4] = acc1[l]
    ysum = yp[0] + yp[1]
    # ---- P3 per frame, per item
    gy = yp[0].copy()  # aliasing: gy overwrites plane 0
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
        pr = {}
        for it in range(32):
            ic = items[it]
            p = d.copy()
            for i in range(7):
                k = (int(ic["path_lo"]) >> (5 * i)) & 31 if i < 6 else int(ic["path_hi"]) & 31
                p = p + bone[k, :3]
            pr[it] = p
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
    assert not np.isnan(gy[:, :104]).any()
    # ---- bL2
    d1p = np.zeros((2, 16, 68))
    gyf = load_row(gy, 104)
    for wv in range(8):
        t, hf = wv & 3, wv >> 2
        acc = np.zeros((64, 4))
        steps = range(0, 12) if hf == 0 else range(12, 26)
        for n, i in enumerate(steps):
            acc = mfma(wf[wv, W_OFF["B2"] + n], gyf[i], acc)
        for l in range(64):
            d1p[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc[l]
    d1 = load_row(d1p[0], 64) + load_row(d1p[1], 64)
    d1 = np.where(a1f > 0, d1, 0.2 * d1)
    # ---- bL1
    d0p = np.zeros((2, 16, 52))
    for wv in range(6):
        t, hf = wv % 3, wv // 3
        acc = np.zeros((64, 4))
        for i in range(8):
            acc = mfma(wf[wv, W_OFF["B1"] + i], d1[8 * hf + i], acc)
        for l in range(64):
            d0p[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc[l]
    d0 = load_row(d0p[0], 40) + load_row(d0p[1], 40)
    d0 = np.where(a0f > 0, d0, 0.2 * d0)
    # ---- bL0
    gzp = np.zeros((2, 16, 36))
    for wv in range(4):
        t, hf = wv & 1, wv >> 1
        acc = np.zeros((64, 4))
        for i in range(5):
            acc = mfma(wf[wv, W_OFF["B0"] + i], d0[5 * hf + i], acc)
        for l in range(64):
            gzp[hf, l & 15, 16 * t + 4 * (l >> 4): 16 * t + 4 * (l >> 4) + 4] = acc[l]
    gzf = load_row(gzp[0], 24) + load_row(gzp[1], 24)
    gz = np.zeros((16, 24))
    for l in range(64):
        for i in range(6):
            gz[l & 15, kcol(24, i, l >> 4)] = gzf[i, l]
    gz += 2 * lam_tmp * (z - np.asarray(z_tgt, np.float64)) / 24
    return ysum[:, :104], gy[:, :104], gz, loss

"""CPU: the split-precision weight image of the 16-frames-per-wave kernel (dp_w16_host.cpp) against an independent statement
of its layout: every 16 x 32 block of every product, rebuilt from the image through the operand map of
v_mfma_f32_16x16x32_bf16 (lane l holds A[row l & 15][k = 8 (l >> 4) + j]) and the kernel's K order, must equal the folded
matrix entry EXACTLY as the sum of its three bf16 terms."""
import ctypes as C

import numpy as np
import pytest

from dragposer_amd import _lib
from dragposer_amd.model import HostModel

N_IN = [24, 40, 60, 96, 60, 40]
N_OUT = [40, 60, 96, 60, 40, 24]
NT_OUT = [3, 4, 6, 4, 3, 2]
NKB = [1, 2, 2, 3, 2, 2]
PAIR0 = [0, 3, 11, 23, 35, 41]
DISP = 22


def slot_item(t, g):  # dp_w16.h, restated
    if t < 4:
        return [1, 5, 14, 18][g] + t
    return [[0, DISP], [9, 10], [11, 12], [13, -1]][g][t - 4]


def dec_row(rho):
    item, r = slot_item(rho >> 4, (rho >> 2) & 3), rho & 3
    if item < 0:
        return -1
    if item == DISP:
        return 88 + r if r < 3 else -1
    return 4 * item + r


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("wd", ["fp32", "bf16"])
def test_image_reconstructs_the_folded_matrices_exactly(wd):
    hm = HostModel(weight_dtype=wd)
    f, folded_struct = hm.fold()
    lib = _lib.load()
    img = np.zeros(45 * 3 * 64 * 4, np.uint32)
    bias = np.zeros(13 * 16, np.float32)
    slots = np.zeros(24 * 12, np.float32)
    rc = lib.dp_debug_pack_w16(C.byref(folded_struct), C.byref(hm.struct), img.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p),
                               slots.ctypes.data_as(C.c_void_p))
    assert rc == _lib.DP_OK
    halves = img.view(np.uint16).reshape(45, 3, 64, 8)  # [pair][term][lane][j] (little endian: j even = low half)
    sd = np.zeros(96)
    mu = np.zeros(96)
    sdq, muq = hm.arrays["std_q"], hm.arrays["mean_q"]
    sdd, mud = hm.arrays["std_disp"], hm.arrays["mean_disp"]
    for rho in range(96):
        d = dec_row(rho)
        if d >= 88:
            sd[rho], mu[rho] = sdd[d - 88], mud[d - 88]
        elif d >= 0:
            sd[rho], mu[rho] = sdq[d], muq[d]
        elif slot_item(rho >> 4, (rho >> 2) & 3) < 0 and rho & 3 == 0:
            mu[rho] = 1.0
    A0, A1, A2 = f["A0"], f["A1"], f["A2"]

    def entry(layer, out, inn):
        if out >= N_OUT[layer] or inn >= N_IN[layer]:
            return np.float32(0)
        if layer == 0:
            return A0[out, inn]
        if layer == 1:
            return A1[out, inn]
        if layer == 2:
            d = dec_row(out)
            return np.float32(0) if d < 0 else np.float32(np.float64(sd[out]) * np.float64(A2[d, inn]))
        if layer == 3:
            d = dec_row(inn)
            return np.float32(0) if d < 0 else np.float32(np.float64(sd[inn]) * np.float64(A2[d, out]))
        if layer == 4:
            return A1[inn, out]
        return A0[inn, out]

    for layer in range(6):
        for n in range(NT_OUT[layer]):
            for kb in range(NKB[layer]):
                pair = PAIR0[layer] + n * NKB[layer] + kb
                terms = bf16_to_f32(halves[pair])  # [3][64][8]
                # a three-term sum of bf16 numbers, added small to large in fp32, is exact here (24 significant bits)
                total = (terms[2].astype(np.float64) + terms[1] + terms[0]).astype(np.float32)
                for lane in range(64):
                    for j in range(8):
                        inn = 16 * (2 * kb + (j >> 2)) + 4 * (lane >> 4) + (j & 3)
                        want = entry(layer, 16 * n + (lane & 15), inn)
                        assert total[lane, j] == want, (layer, n, kb, lane, j, total[lane, j], want)
    # bias rows: [tile][group][4]
    b = bias.reshape(13, 4, 4)
    for tile in range(13):
        for g in range(4):
            for r in range(4):
                if tile < 3:
                    c = 16 * tile + 4 * g + r
                    want = f["c0"][c] if c < 40 else 0
                elif tile < 7:
                    c = 16 * (tile - 3) + 4 * g + r
                    want = f["b1"][c] if c < 60 else 0
                else:
                    rho = 16 * (tile - 7) + 4 * g + r
                    d = dec_row(rho)
                    want = np.float32(np.float64(sd[rho]) * (np.float64(f["b2"][d]) if d >= 0 else 0.0) + mu[rho])
                assert b[tile, g, r] == np.float32(want), (tile, g, r)
    # slot constants: every joint exactly once, offsets of the skeleton
    sl = slots.reshape(24, 12)
    items = sl[:, 3].view(np.int32)
    assert sorted(int(i) for i in items) == [-1] + list(range(23))
    for k in range(24):
        if 0 < items[k] < 22:
            np.testing.assert_array_equal(sl[k, :3], hm.arrays["offsets"][items[k]])
        else:
            assert not sl[k, :3].any()


def _bf16_rne(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, on bit patterns (what v_cvt_pk_bf16_f32 and the host packer do)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _split3(x):
    h = _bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = _bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, _bf16_rne(r2)


def test_three_bf16_terms_are_exact_and_six_products_reach_fp32_accuracy():
    """The arithmetic dp_w16 rests on, emulated in numpy: (1) an fp32 number IS the sum of its three round-to-nearest bf16 terms,
    remainders taken in fp32; (2) a dot product of K = 96 such operands -- the six term products above 2^-24 of the leading one,
    each exact in fp32 (8 x 8 significant bits), accumulated in fp32 smallest first as the MFMA chain does -- is as close to the
    fp64 dot product as a plain fp32 dot product is (both within a few 2^-24 of the sum of magnitudes; same order of mean error)."""
    rs = np.random.RandomState(11)
    x = (rs.standard_normal(1 << 16) * np.exp(rs.uniform(-12, 6, 1 << 16))).astype(np.float32)
    h, m, l = _split3(x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))  # exactly
    K, N = 96, 4096
    w = (rs.standard_normal((N, K)) * np.exp(rs.uniform(-3, 1, (N, K)))).astype(np.float32)
    a = (rs.standard_normal((N, K)) * np.exp(rs.uniform(-6, 2, (N, K)))).astype(np.float32)
    wh, wm, wl = _split3(w)
    ah, am, al = _split3(a)
    acc = np.zeros(N, np.float32)
    for kb in range(K // 32):  # per K-block of 32: the six products, small ones first (dp_w16_impl.h: pair_mm)
        s = slice(32 * kb, 32 * kb + 32)
        for p, q in ((wl, ah), (wm, am), (wh, al), (wm, ah), (wh, am), (wh, ah)):
            prod = p[:, s].astype(np.float64) * q[:, s].astype(np.float64)  # 16 significant bits: exact in fp32 too
            assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)
            for k in range(32):  # (the MFMA's internal order is not specified; any fp32 order obeys the same bound)
                acc = (acc + prod[:, k].astype(np.float32)).astype(np.float32)
    exact = (w.astype(np.float64) * a.astype(np.float64)).sum(1)
    scale = (np.abs(w.astype(np.float64)) * np.abs(a.astype(np.float64))).sum(1)
    plain = np.zeros(N, np.float32)
    for k in range(K):
        plain = (plain + (w[:, k] * a[:, k]).astype(np.float32)).astype(np.float32)
    e_split, e_plain = np.abs(acc - exact) / scale, np.abs(plain - exact) / scale
    assert e_split.max() < 16 * 2.0 ** -24 and np.median(e_split) < 2.0 ** -24, (e_split.max(), np.median(e_split))
    # (measured: mean 1.2 vs 0.8 units of 2^-24 -- this emulation adds 576 terms one by one, the MFMA sums a block of 32 products at once)
    assert e_split.mean() < 2.0 * e_plain.mean(), (e_split.mean(), e_plain.mean())

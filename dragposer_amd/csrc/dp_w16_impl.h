#pragma once
// dp_w16_impl.h -- (included by FOUR translation units, one instantiation each -- dp_w16.hip / dp_w16_2w.hip: fixed iteration count,
// one / two waves per SIMD; dp_w16_es.hip / dp_w16_2w_es.hip: the same two with the reference's per-frame while-condition -- because
// the one-wave and the two-waves shapes want different instruction-scheduling strategies, __graft_entry__.py)
// dp_w16 -- the optimise loop (reference: DragPose.run's while loop, python/src/drag_pose.py:296-355) with SIXTEEN frames per
// wavefront, the decoder on v_mfma_f32_16x16x32_bf16 in split precision (dp_w16.h), for large batches.
//
//   * One wave owns 16 frames from the first decoder layer to the Adam step; lane (f = l & 15, g = l >> 4) holds four consecutive
//     channels of frame f of every 16-channel tile.  A product's result IS the next product's operand (dp_w16.h): between two
//     layers there is only element-wise work -- bias (the MFMA's C operand), LeakyReLU, the split of each fp32 value into three
//     bf16 terms -- and nothing moves between lanes.  Weights: a 135 KB LDS image shared by the workgroup, one ds_read_b128 per
//     (16 x 32 block, term), six MFMAs per block (the term pairs above 2^-24).
//   * Kinematics entirely in registers: after layer 2 a lane holds the quaternions of six joints of its frame -- a chain of four
//     and a chain of two of the skeleton (dp_w16.h: slot map) -- so bones, positions (prefix sums down a chain) and the gradient's
//     subtree sums (suffix sums up a chain) are lane-local; what crosses between chains (the root's rotation and displacement, the
//     spine joints the arms hang off) moves with ds_bpermute_b32 / v_permlane{16,32}_swap -- no LDS storage, no barrier.
//     The mathematics is dp_w4.hip's: the loop works in the frame of cur_rot, rotation error as a quaternion
//     product, gradients as torques.  Every joint slot may carry a tracker; a tile of slots no frame of the wave tracks is skipped.
//   * bf16 MFMAs leave the vector ALU free (profiles/r03_pair_probe.txt), fp32 MFMAs do not: here the element-wise and kinematics
//     work of one tile runs under the matrix work of the next.
// Fixed iteration count or, EARLY, the reference's per-frame while-condition; forward-only launches and whole-sequence launches stay
// with dp_w4.hip.
#include "dp_device.h"
#include "dp_w16.h"

using namespace dpw16;

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// LDS map (32-bit words)
constexpr int L16_IMG = 0;
constexpr int L16_BIAS = L16_IMG + IMG_U32;
constexpr int L16_TAB = L16_BIAS + BIAS_FLOATS; // per-iteration Adam scalars [MAX_ITERS][2]
constexpr int L16_FLAGS = L16_TAB + 2 * MAX_ITERS; // [8 waves][16 frames] DP_STATUS_BAD_* of the input screening, parked for the epilogue
constexpr int L16_CLK = L16_FLAGS + 8 * FPW;       // [4] the two start stamps of dp_result.clock (64 bits each)
constexpr int L16_ADX = L16_CLK + 4;               // [8 waves][2 doubles] Adam's running products beyond the argument table (LONG instantiations: adam_beyond)
constexpr int L16_END = L16_ADX + 8 * 4;

// ---------------------------------------------------------------- small vector helpers
DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
DEV V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 rot_q(Q4 q, V3 a)
{ // R(q) a = a + 2 (w t + v x t), t = v x a  -- what to_matrix_4 (utils.py:49-74) encodes for unit q
    const V3 v = {q.x, q.y, q.z};
    const V3 t = cross3(v, a), c = cross3(v, t);
    return {a.x + 2.f * (q.w * t.x + c.x), a.y + 2.f * (q.w * t.y + c.y), a.z + 2.f * (q.w * t.z + c.z)};
}
DEV V3 rot_qc(Q4 q, V3 a)
{ // R(conj q) a
    const V3 v = {q.x, q.y, q.z};
    const V3 t = cross3(v, a), c = cross3(v, t);
    return {a.x + 2.f * (c.x - q.w * t.x), a.y + 2.f * (c.y - q.w * t.y), a.z + 2.f * (c.z - q.w * t.z)};
}
DEV Q4 qconj(Q4 q) { return {q.w, -q.x, -q.y, -q.z}; }
// per-lane selects, component by component (a ?: on the structs themselves goes through the stack)
DEV V3 sel3(bool c, V3 a, V3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
DEV Q4 sel4(bool c, Q4 a, Q4 b) { return {c ? a.w : b.w, c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
DEV Q4 q_from_rotmat(const float* m)
{ // row-major 3x3 rotation -> unit quaternion (Shepperd's branches, one candidate selected, then normalised): dp_w4.hip's form
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float tr = m00 + m11 + m22;
    const float a = m21 - m12, b = m02 - m20, c = m10 - m01, d = m01 + m10, e = m02 + m20, f = m12 + m21;
    const bool s0 = tr > 0.f, s1 = m00 > m11 && m00 > m22, s2 = m11 > m22;
    Q4 q;
    q.w = s0 ? 1.f + tr : (s1 ? a : (s2 ? b : c));
    q.x = s0 ? a : (s1 ? 1.f + m00 - m11 - m22 : (s2 ? d : e));
    q.y = s0 ? b : (s1 ? d : (s2 ? 1.f + m11 - m00 - m22 : f));
    q.z = s0 ? c : (s1 ? e : (s2 ? f : 1.f + m22 - m00 - m11));
    const float n = 1.f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return {q.w * n, q.x * n, q.y * n, q.z * n};
}

// ---------------------------------------------------------------- across lane groups (lanes f, f + 16, f + 32, f + 48 = one frame)
// (W16_ABLATE_*: diagnostic builds that leave one part of the iteration out -- they compute nonsense, only their time is read: tools/ablate_w16.sh)
#ifdef W16_ABLATE_BPERM
DEV float bperm(int byte_addr, float v) { return v + __int_as_float(byte_addr & 1); }
#else
DEV float bperm(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }
#endif
DEV V3 bperm3(int a, V3 v) { return {bperm(a, v.x), bperm(a, v.y), bperm(a, v.z)}; }
DEV Q4 bperm4(int a, Q4 q) { return {bperm(a, q.w), bperm(a, q.x), bperm(a, q.y), bperm(a, q.z)}; }
DEV float sum_groups(float v)
{ // every lane of a frame gets the sum over its four lane groups (fixed order: deterministic)
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
DEV V3 sum_groups3(V3 v) { return {sum_groups(v.x), sum_groups(v.y), sum_groups(v.z)}; }

// ---------------------------------------------------------------- split precision
struct B3 { u4 t[3]; }; // the three bf16 terms of one K-block (two 16-channel tiles) of an activation
DEV unsigned cvt_pk(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{lo, hi}, bf2)); }
DEV void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l)
{ // x = h + m + l exactly (round-to-nearest bf16 at every stage; the remainders are exact fp32 differences)
    h = cvt_pk(x0, x1);
#ifdef W16_ABLATE_SPLIT
    m = h ^ 0x00010001u; l = h ^ 0x00020002u;
    return;
#endif
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk(s0, s1);
}
DEV B3 split_block(f4 t0, f4 t1)
{
    unsigned h[4], m[4], l[4];
    split_pair(t0.x, t0.y, h[0], m[0], l[0]);
    split_pair(t0.z, t0.w, h[1], m[1], l[1]);
    split_pair(t1.x, t1.y, h[2], m[2], l[2]);
    split_pair(t1.z, t1.w, h[3], m[3], l[3]);
    B3 b;
    b.t[0] = u4{h[0], h[1], h[2], h[3]};
    b.t[1] = u4{m[0], m[1], m[2], m[3]};
    b.t[2] = u4{l[0], l[1], l[2], l[3]};
    return b;
}
DEV f4 mm(u4 a, u4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0); }

// The weights of one (output tile, K-block) pair: three ds_read_b128 (the bf16 terms of the same 16 x 32 block).  The 45 pairs of an
// iteration are consumed in a fixed order, so every pair's reads are ISSUED ONE PAIR AHEAD of its six MFMAs (a lone wave otherwise
// sits out an LDS round trip per pair: 4.8 k of 16 k cycles per iteration waiting at s_waitcnt before this, 3.2 k after -- DESIGN.md 6.2;
// the counters of the kernel as it stands: profiles/r03_w16_pmc_16384.txt).
struct W3 { u4 h, m, l; };
// The image is 135 KB and a ds_read's offset field holds 16 bits: three lane bases 64 KB apart, each opaque to the compiler (it would
// fold them back into ONE base and pay a v_add_u32 per read beyond the first 64 KB: 71 per iteration).
struct Img { const u4* b[3]; };
template <int OFF> DEV u4 img_at(const Img& im) { return im.b[OFF >> 12][OFF & 4095]; } // (OFF in 16-byte units)
template <int PAIR> DEV W3 wload(const Img& im)
{
    constexpr int o = PAIR * (N_TERMS * 64);
    static_assert(((o + 128) >> 12) < 3, "image beyond the third 64 KB window");
    W3 r;
    r.h = img_at<o>(im); r.m = img_at<o + 64>(im); r.l = img_at<o + 128>(im);
    return r;
}
// pins the reads above this point (vector and matrix arithmetic may still move across; LDS operations may not)
DEV void pin_reads() { __builtin_amdgcn_sched_barrier(0x1 | 0x2 | 0x4 | 0x8 | 0x400); }
// one pair: acc += W x, the six term products above 2^-24, small ones first
DEV f4 pair_mm(f4 acc, const W3& w, const B3& b)
{
#ifdef W16_ABLATE_MM5
    return mm(w.h, b.t[0], acc);
#endif
    acc = mm(w.l, b.t[0], acc);
    acc = mm(w.m, b.t[1], acc);
    acc = mm(w.h, b.t[2], acc);
    acc = mm(w.m, b.t[0], acc);
    acc = mm(w.h, b.t[1], acc);
    acc = mm(w.h, b.t[0], acc);
    return acc;
}
// output tile N of product LAYER.  PREF (one wave per SIMD): from weights `w` of its first pair (already in flight); leaves the
// first pair of NEXT in `w`.  !PREF (two waves per SIMD: the partner covers the LDS latency, and registers are short): read at use.
template <int LAYER, int N, int NEXT, bool PREF> DEV f4 out_tile(const Img& img, W3& w, f4 acc, const B3* b)
{
    constexpr int P0 = PAIR0[LAYER] + N * NKB[LAYER];
#pragma unroll
    for (int kb = 0; kb < NKB[LAYER]; ++kb) {
        if (PREF) {
            W3 nx;
            if (kb + 1 < NKB[LAYER]) nx = kb == 0 ? wload<P0 + 1>(img) : wload<P0 + 2>(img);
            else nx = wload<NEXT>(img);
            pin_reads();
            acc = pair_mm(acc, w, b[kb]);
            w = nx;
        } else {
            const W3 wc = kb == 0 ? wload<P0>(img) : kb == 1 ? wload<P0 + 1>(img) : wload<P0 + 2>(img);
            acc = pair_mm(acc, wc, b[kb]);
        }
    }
    return acc;
}
constexpr int pair_of(int layer, int n) { return PAIR0[layer] + n * NKB[layer]; }

// component by component: an f4 * f4 becomes two v_pk_mul_f32, and a packed fp32 instruction cannot issue while a bf16 MFMA is in
// flight -- its own wave's or the partner wave's (profiles/r03_coissue_probe.txt: paired = serial for v_pk_*)
DEV f4 smul(f4 a, f4 b) { return f4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
DEV f4 smul(f4 a, float b) { return f4{a.x * b, a.y * b, a.z * b, a.w * b}; }
DEV f4 pull(f4 g, float c, f4 z, f4 zt) { return f4{g.x + c * (z.x - zt.x), g.y + c * (z.y - zt.y), g.z + c * (z.z - zt.z), g.w + c * (z.w - zt.w)}; } // temporal term
DEV f4 lrelu_factor16(f4 x)
{ // x > 0 ? 1 : 0.2 as med3(x * 2^127, 0.2, 1) (dp_w4.hip)
    const f4 t = smul(x, 0x1p127f);
    return f4{__builtin_amdgcn_fmed3f(t.x, 0.2f, 1.f), __builtin_amdgcn_fmed3f(t.y, 0.2f, 1.f), __builtin_amdgcn_fmed3f(t.z, 0.2f, 1.f),
              __builtin_amdgcn_fmed3f(t.w, 0.2f, 1.f)};
}
DEV f4 bias_row(const float* lds_bias, int tile, int g) { return *(const f4*)(lds_bias + (tile * 4 + g) * 4); }

// ------------------------------------------------------------------------------------------------
// NW waves per workgroup, WPS waves per SIMD (register budget 512 / WPS): one workgroup per CU either way (the LDS image).
// EARLY: the reference's per-frame while-condition (drag_pose.py:298-304, 351-355), as dp_w4.hip runs it: a frame whose condition fails
// takes this one last step (its final latent) and then keeps its pre-step latent, so that the forward passes it still takes part in
// reproduce its last one; a wave leaves the loop once all sixteen of its frames have stopped.  A frame is a lane column here, its
// state replicated over the four lane groups.
// LONG: n_iter beyond the kernel-argument table of Adam scalars (MAX_ITERS; the reference has no cap on max_iter): one uniform branch per iteration
// continues the two bias corrections in double on the device (adam_beyond), as dp_w4's LONG instantiations do.  Their own translation units
// (dp_w16_long*.hip): the ordinary launches' code is untouched.
template <int NW, int WPS, bool EARLY = false, bool LONG = false>
__global__ __launch_bounds__(NW * 64, WPS) void dp_w16_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[L16_END];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15, g = lane >> 4;
    const int nB = a.n_frames;
    const int f0 = (blockIdx.x * NW + wave) * FPW;
    const int gf = min(f0 + f, nB - 1); // (ragged tail: a copy of the last frame, not stored)
    const bool fvalid = f0 + f < nB;

    // ---- set-up: the weight image, bias rows and the Adam table into LDS (once per workgroup)
    for (int k = tid; k < IMG_U32 / 4; k += NW * 64) ((u4*)(lds + L16_IMG))[k] = ((const u4*)a.w16img)[k];
    for (int k = tid; k < BIAS_FLOATS; k += NW * 64) ((float*)lds)[L16_BIAS + k] = a.w16bias[k];
    for (int k = tid; k < min(a.n_iter, MAX_ITERS); k += NW * 64) *(f2*)((float*)lds + L16_TAB + 2 * k) = f2{a.tab.step[k], a.tab.bc2s[k]}; // (beyond the table: LONG)

    // latent and Adam state in the D layout: tile n, register r = latent dim 16 n + 4 g + r (dims 24..31: zero, stay zero)
    f4 z[2], zt[2], mA[2], vA[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 16 * n + 4 * g;
        const bool real = c < LAT;
        z[n] = real ? *(const f4*)(a.z0 + (size_t)gf * LAT + c) : f4{0.f, 0.f, 0.f, 0.f};
        zt[n] = real ? *(const f4*)(a.z_tgt + (size_t)gf * LAT + c) : f4{0.f, 0.f, 0.f, 0.f};
        mA[n] = vA[n] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const f4 cv = *(const f4*)(a.cur_rot + (size_t)gf * 4);
    const Q4 cur = {cv.x, cv.y, cv.z, cv.w};
    // input screening (include/dragposer.h: DP_STATUS_*; dp_w4.hip has the long story).  A frame is a lane column here and nothing crosses
    // columns, so a frame with a non-finite input needs no stand-in values: it is handed over AS non-finite -- the state (z) or the pull
    // target (z_tgt) set to NaN -- and the arithmetic then does what the reference's does (NaN loss on the first pass, the while-condition
    // fails, Adam writes NaN into the latent).  Inputs beyond DP_INPUT_LIMIT are treated the same way, as dp_w4 treats them.
    const auto oor = [](float x) { return !(fabsf(x) <= DP_INPUT_LIMIT); };
    float bad_s = (oor(cv.x) || oor(cv.y) || oor(cv.z) || oor(cv.w)) ? 1.f : 0.f, bad_t = 0.f, not_r = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { bad_s += oor(z[n][r]) ? 1.f : 0.f; bad_t += oor(zt[n][r]) ? 1.f : 0.f; }

    // my six slots: constants and tracker inputs (targets rotated into the frame of cur_rot once; rotation targets as unit
    // quaternions; loss coefficients with the per-frame mean denominators 3 E, 9 E folded in)
    V3 off[NTY];
    V3 tp[NTY];      // target position in the frame of cur_rot
    Q4 qT[NTY];      // target rotation in the frame of cur_rot
    float cgp[NTY];  // 2 w_pos / (3 E)   (0: untracked)
    float k8[NTY];   // -8 lam_rot w_rot / (9 E)
    float wp_raw[NTY], wr_raw[NTY];
    int n_trk = 0;
    unsigned slotmask = 0; // bit t: some frame of this wave tracks a slot of tile t
#pragma unroll
    for (int t = 0; t < NTY; ++t) {
        const SlotConst* sc = a.w16slots + t * 4 + g;
        off[t] = {sc->off[0], sc->off[1], sc->off[2]};
        const int item = sc->item;
        const bool joint = item >= 0 && item < NJ;
        const int j = joint ? item : 0;
        const bool trk = joint && a.tracked[(size_t)gf * NJ + j] != 0;
        const float* p = a.tgt_pos + ((size_t)gf * NJ + j) * 3;
        const float* rm = a.tgt_rot + ((size_t)gf * NJ + j) * 9;
        const float* wv = a.w + ((size_t)gf * NJ + j) * 2;
        float m9[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) m9[k] = rm[k];
        {
            bool bd = oor(p[0]) || oor(p[1]) || oor(p[2]) || oor(wv[0]) || oor(wv[1]);
#pragma unroll
            for (int k = 0; k < 9; ++k) bd = bd || oor(m9[k]);
            bad_t += trk && bd ? 1.f : 0.f;
            not_r += trk && !bd && not_rotation(m9) ? 1.f : 0.f; // (DP_STATUS_TARGET_NOT_ROTATION: reported, computed as given)
        }
        tp[t] = sel3(trk, rot_qc(cur, V3{p[0], p[1], p[2]}), V3{0.f, 0.f, 0.f});
        qT[t] = sel4(trk, quat_mul(qconj(cur), q_from_rotmat(m9)), Q4{1.f, 0.f, 0.f, 0.f});
        wp_raw[t] = trk ? wv[0] : 0.f;
        wr_raw[t] = trk ? wv[1] : 0.f;
        n_trk += trk ? 1 : 0;
        slotmask |= (__ballot(trk) != 0ull ? 1u : 0u) << t;
    }
    {
        const float E = sum_groups((float)n_trk);
        const float invE = E > 0.f ? 1.f / E : 0.f;
#pragma unroll
        for (int t = 0; t < NTY; ++t) {
            cgp[t] = 2.f * (wp_raw[t] * invE * (1.f / 3.f));
            k8[t] = -8.f * (a.lam_rot * wr_raw[t] * invE * (1.f / 9.f));
        }
    }
    // (the two flags go to LDS for the epilogue: held in registers across the loop they cost the one-wave unit two spills)
    {
        const bool state_bad = sum_groups(bad_s) > 0.f, tgt_bad = !state_bad && sum_groups(bad_t) > 0.f; // (the same in the frame's four lanes)
        const bool rot_bad = !state_bad && !tgt_bad && sum_groups(not_r) > 0.f;
        if (g == 0) lds[L16_FLAGS + wave * FPW + f] = (state_bad ? DP_STATUS_BAD_STATE : 0) + (tgt_bad ? DP_STATUS_BAD_TARGETS : 0) + (rot_bad ? DP_STATUS_TARGET_NOT_ROTATION : 0);
    if (__ballot(state_bad || tgt_bad) != 0ull) { // (uniform, rare)
        const float qnan = __builtin_nanf("");
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * n + 4 * g + r < LAT) { // (dims 24..31 stay zero)
                    z[n][r] = state_bad ? qnan : z[n][r];
                    zt[n][r] = tgt_bad ? qnan : zt[n][r];
                }
            }
    }
    }
#ifdef W16_ABLATE_T
    slotmask = 0;
#endif
    const bool is_root = g == 0, is_pad3 = g == 3; // slot 4 of group 0 = root, slot 5 of group 0 = displacement, slot 5 of group 3 = idle
    const int src_f0 = 4 * f, src_g2 = 4 * (f + 32), src_prev = 4 * (lane >= 16 ? lane - 16 : lane), src_next = 4 * (lane < 48 ? lane + 16 : lane);

    __syncthreads();
    if (f0 >= nB) return; // (uniform per wave; no barrier below)
    // dp_result.clock: shader cycles and 100 MHz ticks wave 0 of workgroup 0 spends from here to its last store
    // (the start stamps wait in LDS: four more scalar registers held across the loop cost this kernel spills)
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) { // (uniform)
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { *(unsigned long long*)(lds + L16_CLK) = c0; *(unsigned long long*)(lds + L16_CLK + 2) = r0; }
    }
#ifdef W16_STAGGER // (the waves of a workgroup started W16_STAGGER x 64 cycles apart: see dp_w4.hip "Stagger")
    for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(W16_STAGGER);
#endif
    // (Two waves per SIMD run the same program from the same barrier; a start delay for the second half of the workgroup -- so that one
    //  wave's matrix phases fall on the other's vector phases -- was measured and changes nothing: profiles/r03_w16_stagger.txt.)
    const float* lbias = (const float*)lds + L16_BIAS;
    Q4 q[NTY];       // unit quaternions of my slots (displacement slot: the de-normalised displacement in w, x, y)
    float inv[NTY];  // 1 / |raw quaternion|
    V3 u[NTY], P[NTY]; // bone parent -> joint and position relative to the root, both in the root frame
    Q4 q0 = {1.f, 0.f, 0.f, 0.f};
    V3 dsp = {0.f, 0.f, 0.f};
    f4 zpre[2] = {z[0], z[1]};
    float loss_p = 0.f, loss_r = 0.f;
    // EARLY: total loss of the frame's previous iteration, still iterating, iterations executed, losses of its last executed
    // iteration (pos, rot, tmp), latent after its last step
    float es_prev = 10000000.f, es_lp = 0.f, es_lr = 0.f, es_lt = 0.f;
    bool es_act = true;
    int es_iters = 0;
    f4 zfin[2] = {z[0], z[1]};

    if constexpr (LONG) { double* adx = (double*)(lds + L16_ADX) + 2 * wave; adx[0] = a.cont.b1t; adx[1] = a.cont.b2t; } // (every lane the same values: adam_beyond)
    for (int iter = 0; iter < a.n_iter; ++iter) {
        const bool last = iter == a.n_iter - 1;
        f2 adam_t = *(const f2*)((const float*)lds + L16_TAB + 2 * (LONG ? min(iter, MAX_ITERS - 1) : iter));
        if constexpr (LONG) {
            if (iter >= MAX_ITERS) { // (uniform: beyond the argument table)
                float st_ = 0.f, rb_ = 0.f;
                adam_beyond((double*)(lds + L16_ADX) + 2 * wave, a.cont, st_, rb_);
                adam_t = f2{st_, rb_};
            }
        }
        int o0 = lane, o1 = lane + 4096, o2 = lane + 8192;
        asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2)); // opaque per iteration: the weight reads stay inside the loop
        const Img img = {{(const u4*)(lds + L16_IMG) + o0, (const u4*)(lds + L16_IMG) + o1, (const u4*)(lds + L16_IMG) + o2}};
        if (!EARLY && last) { zpre[0] = z[0]; zpre[1] = z[1]; }

        // ================= forward: a0 = lrelu(A0 z + c0), a1 = lrelu(A1 a0 + b1), y = A2' a1 + b2'
        f4 fac0[3], fac1[4], y[NTY];
        B3 bz[1], b0[2], b1[2];
        constexpr bool PREF = WPS == 1;
        W3 wq;
        if (PREF) wq = wload<pair_of(L0, 0)>(img);
        bz[0] = split_block(z[0], z[1]);
        f4 a0[3], a1[4];
        a0[0] = out_tile<L0, 0, pair_of(L0, 1), PREF>(img, wq, bias_row(lbias, 0, g), bz);
        a0[1] = out_tile<L0, 1, pair_of(L0, 2), PREF>(img, wq, bias_row(lbias, 1, g), bz);
        a0[2] = out_tile<L0, 2, pair_of(L1, 0), PREF>(img, wq, bias_row(lbias, 2, g), bz);
#pragma unroll
        for (int n = 0; n < 3; ++n) { fac0[n] = lrelu_factor16(a0[n]); a0[n] = smul(a0[n], fac0[n]); }
        b0[0] = split_block(a0[0], a0[1]);
        b0[1] = split_block(a0[2], f4{0.f, 0.f, 0.f, 0.f});
        a1[0] = out_tile<L1, 0, pair_of(L1, 1), PREF>(img, wq, bias_row(lbias, 3, g), b0);
        a1[1] = out_tile<L1, 1, pair_of(L1, 2), PREF>(img, wq, bias_row(lbias, 4, g), b0);
        a1[2] = out_tile<L1, 2, pair_of(L1, 3), PREF>(img, wq, bias_row(lbias, 5, g), b0);
        a1[3] = out_tile<L1, 3, pair_of(L2, 4), PREF>(img, wq, bias_row(lbias, 6, g), b0);
#pragma unroll
        for (int n = 0; n < 4; ++n) { fac1[n] = lrelu_factor16(a1[n]); a1[n] = smul(a1[n], fac1[n]); }
        b1[0] = split_block(a1[0], a1[1]);
        b1[1] = split_block(a1[2], a1[3]);
        // the B chain's tiles first: the other chains wait for its quaternions
        y[4] = out_tile<L2, 4, pair_of(L2, 5), PREF>(img, wq, bias_row(lbias, 11, g), b1);
        y[5] = out_tile<L2, 5, pair_of(L2, 0), PREF>(img, wq, bias_row(lbias, 12, g), b1);
        y[0] = out_tile<L2, 0, pair_of(L2, 1), PREF>(img, wq, bias_row(lbias, 7, g), b1);
        y[1] = out_tile<L2, 1, pair_of(L2, 2), PREF>(img, wq, bias_row(lbias, 8, g), b1);
        y[2] = out_tile<L2, 2, pair_of(L2, 3), PREF>(img, wq, bias_row(lbias, 9, g), b1);
        y[3] = out_tile<L2, 3, pair_of(B2, 0), PREF>(img, wq, bias_row(lbias, 10, g), b1); // (bL2's first block rides through the kinematics)

#ifdef W16_ABLATE_KIN
        f4 gy[NTY];
        bool was_act = true, stop_now = false;
#pragma unroll
        for (int t = 0; t < NTY; ++t) { gy[t] = y[t]; q[t] = {y[t].x, y[t].y, y[t].z, y[t].w}; P[t] = {y[t].x, y[t].y, y[t].z}; }
#else
        // ================= stage J: normalise, bones, positions relative to the root
#pragma unroll
        for (int t = 0; t < NTY; ++t) {
            const float n2 = y[t].x * y[t].x + y[t].y * y[t].y + y[t].z * y[t].z + y[t].w * y[t].w;
            float iv = __builtin_amdgcn_rsqf(n2);
            if (t == 5) iv = is_root ? 1.f : iv; // the displacement passes through (the idle slot decodes to the unit quaternion)
            inv[t] = iv;
            q[t] = {y[t].x * iv, y[t].y * iv, y[t].z * iv, y[t].w * iv};
        }
        q0 = bperm4(src_f0, q[4]);                                   // the root's rotation (group 0, slot 4) to every lane of the frame
        dsp = bperm3(src_f0, V3{q[5].w, q[5].x, q[5].y});            // the displacement (group 0, slot 5)
        const Q4 ident = {1.f, 0.f, 0.f, 0.f};
        const Q4 q11 = bperm4(src_g2, q[4]), qpv = bperm4(src_prev, q[5]);
        const Q4 qextA = sel4(g >= 2, q11, ident); // the arms hang off joint 11 (group 2, slot 4); the legs off the root (identity in its own frame)
        const Q4 qextB = sel4(g >= 2, qpv, ident); // 11 hangs off 10 (group 1, slot 5), 13 off 12 (group 2, slot 5); 9 off the root
        u[0] = rot_q(qextA, off[0]);
        u[1] = rot_q(q[0], off[1]);
        u[2] = rot_q(q[1], off[2]);
        u[3] = rot_q(q[2], off[3]);
        u[4] = rot_q(qextB, off[4]);
        u[5] = rot_q(q[4], off[5]); // (group 0: root -> displacement, offset zero)
        const V3 zero3 = {0.f, 0.f, 0.f};
        {
            // B chains in dependency order: group 1 (9, 10) hangs off the root, group 2 (11, 12) off 10, group 3 (13) off 12
            V3 extB = zero3;
            P[4] = extB + u[4]; P[5] = P[4] + u[5];
            const V3 h1 = bperm3(src_prev, P[5]);
            extB = sel3(g == 2, h1, zero3);
            P[4] = extB + u[4]; P[5] = P[4] + u[5];
            const V3 h2 = bperm3(src_prev, P[5]);
            extB = sel3(g == 3, h2, extB);
            P[4] = extB + u[4]; P[5] = P[4] + u[5];
            const V3 p11 = bperm3(src_g2, P[4]);
            const V3 extA = sel3(g >= 2, p11, zero3);
            P[0] = extA + u[0]; P[1] = P[0] + u[1]; P[2] = P[1] + u[2]; P[3] = P[2] + u[3];
        }

        // ================= stage T: tracker terms of every tile some frame of the wave tracks (uniform branches)
        V3 gp[NTY], own[NTY]; // position gradient / own rotation torque of my slots (zero: untracked)
        V3 RT = zero3, GD = zero3; // my share of the torque on the root and of dL/d(displacement)
        float lp = 0.f, lr = 0.f;
#pragma unroll
        for (int t = 0; t < NTY; ++t) {
            gp[t] = zero3; own[t] = zero3;
            if ((slotmask >> t) & 1u) {
                const V3 pos = dsp + P[t];
                const V3 at = rot_qc(q0, tp[t]); // target position in the root frame
                const V3 e = pos - at;
                gp[t] = cgp[t] * e;
                // rotation error s = conj(q0) (x) qT (x) conj(qt): |M - T|_F^2 = 8 |vec s|^2; the root is the identity in its own frame
                const Q4 qt = t == 4 ? sel4(is_root, ident, q[t]) : q[t];
                const Q4 s = quat_mul(quat_mul(qconj(q0), qT[t]), qconj(qt));
                const float k = k8[t] * s.w;
                const V3 ow = {k * s.x, k * s.y, k * s.z};
                RT = RT + cross3(at, gp[t]) + ow;
                GD = GD + gp[t];
                own[t] = t == 4 ? sel3(is_root, zero3, ow) : ow; // (the root's own torque is part of the root sum only)
                if (EARLY || last) {
                    lp += 0.5f * cgp[t] * dot3(e, e);
                    lr -= k8[t] * (s.x * s.x + s.y * s.y + s.z * s.z);
                }
            }
        }
        if (!EARLY && last) { loss_p = lp; loss_r = lr; }
        bool was_act = true, stop_now = false;
        if (EARLY) { // the stop test of my frame, in all four of its lanes
            const float sp = sum_groups(lp), sr = sum_groups(lr);
            float l2 = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) { const f4 dz = z[n] - zt[n]; l2 += (dz.x * dz.x + dz.y * dz.y) + (dz.z * dz.z + dz.w * dz.w); }
            const float st = sum_groups(l2) * (a.lam_tmp * (1.f / 24.f));
            const float tot = (sp + sr) + st;
            const bool cont = (sp > a.stop_eps_pos || sr > a.stop_eps_rot) && (es_prev - tot > a.min_loss_incr) && !last;
            was_act = es_act;
            es_prev = es_act ? tot : es_prev;
            es_iters += es_act ? 1 : 0;
            es_lp = es_act ? sp : es_lp; es_lr = es_act ? sr : es_lr; es_lt = es_act ? st : es_lt;
            stop_now = es_act && !cont;
            es_act = es_act && cont;
        }

        // ================= stage G: subtree sums up the chains, torques, dL/dq, dL/dy
        RT = sum_groups3(RT);
        GD = sum_groups3(GD);
        V3 S[NTY], tau[NTY];
        S[3] = gp[3];
        S[2] = gp[2] + S[3];
        S[1] = gp[1] + S[2];
        S[0] = gp[0] + S[1];
        tau[3] = own[3];
        tau[2] = cross3(u[3], S[3]) + own[2];
        tau[1] = cross3(u[2], S[2]) + own[1];
        tau[0] = cross3(u[1], S[1]) + own[0];
        {
            // what the chain heads pass to their parents: subtree sum and the torque of the head's bone
            const V3 cA = cross3(u[0], S[0]);
            const V3 sA_n = bperm3(src_next, S[0]), cA_n = bperm3(src_next, cA); // group 3's arm -> joint 11 (group 2, slot 4)
            const V3 extS0 = sel3(g == 2, S[0] + sA_n, zero3), extC0 = sel3(g == 2, cA + cA_n, zero3);
            V3 extS1 = zero3, extC1 = zero3;
            // B chains in reverse dependency order: group 3 (13), then group 2 (11, 12) with 13's sums, then group 1 (9, 10) with 11's
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                S[5] = gp[5] + extS1;
                tau[5] = own[5] + extC1;
                S[4] = gp[4] + S[5] + extS0;
                tau[4] = cross3(u[5], S[5]) + own[4] + extC0;
                if (pass < 2) {
                    const V3 cB = cross3(u[4], S[4]);
                    const V3 sN = bperm3(src_next, S[4]), cN = bperm3(src_next, cB);
                    const bool take = pass == 0 ? g == 2 : g == 1;
                    extS1 = sel3(take, sN, extS1);
                    extC1 = sel3(take, cN, extC1);
                }
            }
        }
        f4 gy[NTY];
#pragma unroll
        for (int t = 0; t < NTY; ++t) {
            const bool root_slot = t == 4 && is_root;
            const V3 tq = t == 4 ? sel3(is_root, RT, tau[t]) : tau[t];
            const float sgn = root_slot ? -1.f : 1.f; // joints: dL/dq = (0, 2 tau) (x) q; the root rotates the targets: q (x) (0, 2 tau)
            const V3 av = {tq.x + tq.x, tq.y + tq.y, tq.z + tq.z};
            const Q4 qq = q[t];
            const V3 v = {qq.x, qq.y, qq.z};
            const V3 cx = cross3(av, v);
            const float g0 = -dot3(av, v);
            const float g1 = qq.w * av.x + sgn * cx.x, g2 = qq.w * av.y + sgn * cx.y, g3 = qq.w * av.z + sgn * cx.z;
            gy[t] = f4{g0 * inv[t], g1 * inv[t], g2 * inv[t], g3 * inv[t]}; // (already tangent: no projection; sigma is part of bL2's weights)
            if (t == 5) gy[t] = is_root ? f4{GD.x, GD.y, GD.z, 0.f} : (is_pad3 ? f4{0.f, 0.f, 0.f, 0.f} : gy[t]);
        }

#endif
        // ================= backward: d1 = (A2'^T gy) lrelu'(a1), d0 = (A1^T d1) lrelu'(a0), dL/dz = A0^T d0 + temporal term
        B3 by[3], bd1[2], bd0[2];
        by[0] = split_block(gy[0], gy[1]);
        by[1] = split_block(gy[2], gy[3]);
        by[2] = split_block(gy[4], gy[5]);
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f4 d1[4], d0[3], gz[2];
        d1[0] = smul(out_tile<B2, 0, pair_of(B2, 1), PREF>(img, wq, zero4, by), fac1[0]);
        d1[1] = smul(out_tile<B2, 1, pair_of(B2, 2), PREF>(img, wq, zero4, by), fac1[1]);
        d1[2] = smul(out_tile<B2, 2, pair_of(B2, 3), PREF>(img, wq, zero4, by), fac1[2]);
        d1[3] = smul(out_tile<B2, 3, pair_of(B1, 0), PREF>(img, wq, zero4, by), fac1[3]);
        bd1[0] = split_block(d1[0], d1[1]);
        bd1[1] = split_block(d1[2], d1[3]);
        d0[0] = smul(out_tile<B1, 0, pair_of(B1, 1), PREF>(img, wq, zero4, bd1), fac0[0]);
        d0[1] = smul(out_tile<B1, 1, pair_of(B1, 2), PREF>(img, wq, zero4, bd1), fac0[1]);
        d0[2] = smul(out_tile<B1, 2, pair_of(B0, 0), PREF>(img, wq, zero4, bd1), fac0[2]);
        bd0[0] = split_block(d0[0], d0[1]);
        bd0[1] = split_block(d0[2], zero4);
        gz[0] = pull(out_tile<B0, 0, pair_of(B0, 1), PREF>(img, wq, zero4, bd0), a.ctmp, z[0], zt[0]);
        gz[1] = pull(out_tile<B0, 1, pair_of(L0, 0), PREF>(img, wq, zero4, bd0), a.ctmp, z[1], zt[1]); // (the last read is a dummy: the next iteration re-reads it)
        if (a.dbg && iter == 0 && fvalid) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
                if (16 * n + 4 * g < LAT) *(f4*)(a.dbg + (size_t)gf * DBG_STRIDE + DBG_GZ + 16 * n + 4 * g) = gz[n];
        }
        // ================= Adam (torch.optim.Adam, single-tensor form; m, v start at 0, t = iter + 1)
        const float step = adam_t.x, rbc2s = adam_t.y;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { // (scalar on purpose: see smul)
                const float gr = gz[n][r];
                const float m1 = mA[n][r] + a.one_m_b1 * (gr - mA[n][r]);
                const float v1 = vA[n][r] * a.beta2 + a.one_m_b2 * (gr * gr);
                const float den = __builtin_amdgcn_sqrtf(v1) * rbc2s + a.eps;
                const float z1 = z[n][r] - step * (m1 * __builtin_amdgcn_rcpf(den));
                if (!EARLY) {
                    mA[n][r] = m1;
                    vA[n][r] = v1;
                    z[n][r] = z1;
                } else { // a frame that stops now: this step gives its final latent, z / m / v stay; a stopped frame: nothing moves
                    const bool go = was_act && !stop_now;
                    zpre[n][r] = stop_now ? z[n][r] : zpre[n][r]; // latent of the frame's LAST forward pass
                    zfin[n][r] = stop_now ? z1 : zfin[n][r];
                    mA[n][r] = go ? m1 : mA[n][r];
                    vA[n][r] = go ? v1 : vA[n][r];
                    z[n][r] = go ? z1 : z[n][r];
                }
            }
        if (EARLY && __ballot(es_act) == 0ull) break; // every frame of the wave has stopped
    }

    // ================= epilogue: outputs of the LAST forward pass (drag_pose.py:84-113 and what run() returns)
    float lsum_p, lsum_r, lt;
    if (EARLY) { // the losses of the frame's last executed iteration, as its stop test saw them
        lsum_p = es_lp; lsum_r = es_lr; lt = es_lt;
    } else {
        lsum_p = sum_groups(loss_p); lsum_r = sum_groups(loss_r);
        lt = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) { const f4 dz = zpre[n] - zt[n]; lt += dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w; }
        lt = sum_groups(lt) * a.lam_tmp * (1.f / 24.f);
    }
    const int bad_flags = (int)lds[L16_FLAGS + wave * FPW + f]; // (written by this wave's own lanes before the set-up's barrier)
    if ((bad_flags & (DP_STATUS_BAD_STATE | DP_STATUS_BAD_TARGETS)) != 0) lsum_p = lsum_r = lt = __builtin_nanf(""); // (the reference's total loss is NaN there; which of its terms are depends on the input)
    if (a.status) { // (uniform)
        float nf = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) nf += !(fabsf(EARLY ? zfin[n][r] : z[n][r]) <= 3.0e38f) ? 1.f : 0.f;
        nf = sum_groups(nf);
        if (fvalid && g == 0)
            a.status[gf] = (nf > 0.f ? DP_STATUS_NONFINITE_RESULT : 0) + bad_flags;
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) { // (uniform; the result stores that follow are a percent of the launch and change the ratio of the two counters by nothing)
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c1 - *(const unsigned long long*)(lds + L16_CLK); a.clk[1] = r1 - *(const unsigned long long*)(lds + L16_CLK + 2); }
    }
    if (!fvalid) return;
    const Q4 qw = quat_mul(cur, q0); // world rotation (drag_pose.py:88)
    const M3 R0 = quat_to_mat(qw);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 16 * n + 4 * g;
        if (c < LAT) {
            if (a.z) *(f4*)(a.z + (size_t)gf * LAT + c) = EARLY ? zfin[n] : z[n];
            if (a.z_pre) *(f4*)(a.z_pre + (size_t)gf * LAT + c) = zpre[n];
        }
    }
#pragma unroll
    for (int t = 0; t < NTY; ++t) {
        const SlotConst* sc = a.w16slots + t * 4 + g;
        const int item = sc->item;
        if (item == ITEM_DISP) {
            if (a.disp) { float* o = a.disp + (size_t)gf * 3; o[0] = dsp.x; o[1] = dsp.y; o[2] = dsp.z; }
            if (a.world_disp) { const V3 wd = mat_vec(R0, dsp); float* o = a.world_disp + (size_t)gf * 3; o[0] = wd.x; o[1] = wd.y; o[2] = wd.z; }
        } else if (item >= 0) {
            if (a.pose) *(f4*)(a.pose + (size_t)gf * 88 + 4 * item) =
                f4{(q[t].w - sc->mu[0]) / sc->sd[0], (q[t].x - sc->mu[1]) / sc->sd[1], (q[t].y - sc->mu[2]) / sc->sd[2], (q[t].z - sc->mu[3]) / sc->sd[3]};
            if (a.pos) { const V3 pw = mat_vec(R0, dsp + P[t]); float* o = a.pos + ((size_t)gf * NJ + item) * 3; o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; }
            if (a.rot) {
                M3 M = quat_to_mat(q[t]);
                if (item == 0) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
                const M3 G = mat_mat(R0, M);
                float* o = a.rot + ((size_t)gf * NJ + item) * 9;
                o[0] = G.m00; o[1] = G.m01; o[2] = G.m02; o[3] = G.m10; o[4] = G.m11; o[5] = G.m12; o[6] = G.m20; o[7] = G.m21; o[8] = G.m22;
            }
            if (item == 0) {
                if (a.world_rot) *(f4*)(a.world_rot + (size_t)gf * 4) = f4{qw.w, qw.x, qw.y, qw.z};
                if (a.loss) { float* o = a.loss + (size_t)gf * 3; o[0] = lsum_p; o[1] = lsum_r; o[2] = lt; }
                if (a.iters) a.iters[gf] = EARLY ? es_iters : a.n_iter;
            }
        }
    }
}


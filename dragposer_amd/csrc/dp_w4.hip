// dp_w4.hip -- wave-private variant of the fused latent-optimisation kernel for gfx950 (MI355X).
//
// Same mathematics as dp_kernel.hip (reference: DragPose.run's while loop, python/src/drag_pose.py:296-355), other
// decomposition: ONE wavefront owns FOUR frames from the first decoder layer to the Adam step, so an iteration has no
// workgroup barrier and no cross-wave traffic at all.
//   * Decoder forward / backward: v_mfma_f32_4x4x1_16b_f32 with the A-block broadcast (dp_w4.h): a step is a rank-1
//     update of a 64-channel x 4-frame tile; activations stay in registers between layers (a 4x4 register <-> lane
//     transpose inside lane quads turns a product's result into the next product's operand); weights are shared by
//     the waves of a workgroup and streamed from LDS, one ds_read_b128 per four steps.
//   * Kinematics (P3): lane 4b+i = item b of frame i (two rounds: items 0..15, 16..31); the cross-item traffic of a
//     frame (bones, tracker gradients) goes through that frame's own LDS block, written and read by this wave only.
//   * Adam: element-wise in the layout the last product leaves dL/dz in (lane = latent dim, register = frame).
// A workgroup is NW waves that share nothing but the weight image; grid = ceil(B / (4 NW)).
#include "dp_p3.h"
#include "dp_w4.h"
#include <utility>

using namespace dpw4;

template <class F, int... I> DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------------
// LDS map (floats)
constexpr int W4_R = 24;                        // tracker capacity per frame (>= NJ: every joint may carry one)
// one block per frame: bones | tracker gradients | d/d(qw) contributions | loss terms | qd row | tracker inputs | z rows
constexpr int FB_BONE = 0;                      // [32][4]
constexpr int FB_GPC = FB_BONE + 128;           // [R][4]
constexpr int FB_CQ = FB_GPC + 4 * W4_R;        // [R][4]
constexpr int FB_LP = FB_CQ + 4 * W4_R;         // [R][2]
constexpr int FB_QD = FB_LP + 2 * W4_R;         // [QD_S]  (dp_p3.h)
constexpr int FB_TRK = FB_QD + QD_S;            // [4][R][4]
constexpr int FB_ZPRE = FB_TRK + 16 * W4_R;     // [24] latent of the last forward pass (epilogue only)
constexpr int FB_ZT = FB_ZPRE + LAT;            // [24] z_tgt (epilogue only)
constexpr int FB_END = FB_ZT + LAT;
// the four frames of a wave sit in the four lanes of every quad: block stride = 16 banks (mod 64) apart, so that the
// quad's 16-byte accesses to the same row of four blocks never share a bank
constexpr int FB_STRIDE = ((FB_END - 16 + 63) / 64) * 64 + 16;
static_assert(FB_STRIDE >= FB_END && FB_STRIDE % 64 == 16 && FB_QD % 4 == 0 && FB_TRK % 4 == 0 && FB_ZPRE % 4 == 0, "frame block layout");

constexpr int L_IMG = 0;                        // weight image [N_GROUPS][64][4]
constexpr int L_ITEM = L_IMG + IMG_FLOATS;      // item constants, SoA: sd[32][4] | mu[32][4] | child offset[32][4]
constexpr int L_FR = L_ITEM + 3 * 32 * 4;       // frame blocks [NW * 4][FB_STRIDE]
template <int NW> constexpr int lds_total() { return L_FR + NW * FPW * FB_STRIDE; }

// ------------------------------------------------------------------------------------------------
template <int ABID> DEV f4 mfma_bc(float x, float w, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, c, 4, ABID, 0); }

// NG groups of four K-steps: step 4g+m multiplies channel 4(ABID0+g)+m of the X-layout operand `x` with the weight
// image w[g] (one ds_read_b128 per lane and group); two accumulators keep the 8-cycle issue rate
template <int NG, int ABID0> DEV void chain(f4& acc0, f4& acc1, const f4& x, const f4* w)
{
    static_for<NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        const f4 wv = w[g * 64];
        acc0 = mfma_bc<ABID0 + g>(x[0], wv[0], acc0);
        acc1 = mfma_bc<ABID0 + g>(x[1], wv[1], acc1);
        acc0 = mfma_bc<ABID0 + g>(x[2], wv[2], acc0);
        acc1 = mfma_bc<ABID0 + g>(x[3], wv[3], acc1);
    });
}

// D <-> X: transpose of (register index, lane-in-quad).  Two exchange stages (lane ^ 1, lane ^ 2), each a select whose
// first source comes through DPP quad_perm; hipcc does not fold the DPP into the select itself, hence the asm.
// (VALU write -> DPP read of the same VGPR needs 2 wait states: the leading s_mov + s_nop cover an input written just
// ahead of the statement; inside, every DPP source is at least two instructions old.)
DEV void quad_transpose(f4& r)
{
    const unsigned long long even1 = 0x5555555555555555ull, even2 = 0x3333333333333333ull;
    float n0, n1, n2, n3, m0, m1, m2, m3;
    asm volatile("s_mov_b64 vcc, %8\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32_dpp %0, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %2, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %1, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %3, %6, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "s"(even1)
                 : "vcc");
    asm volatile("s_mov_b64 vcc, %8\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
                 : "v"(n0), "v"(n1), "v"(n2), "v"(n3), "s"(even2)
                 : "vcc");
    r = f4{m0, m1, m2, m3};
}

DEV f4 splat(float v) { return f4{v, v, v, v}; }

// ------------------------------------------------------------------------------------------------
// Kinematics, lane = (item, frame).  The arithmetic of every item is that of dp_p3.h::p3_round, cut at its two
// wave-level exchanges into three stages, because the two rounds of a wave (items 0..15, 16..31) share the frames'
// LDS rows: both rounds finish a stage before either starts the next.
struct P3S { // what an item keeps in registers from one stage to the next
    Q4 q;
    float inv;
    M3 M, gM;
    f4 sd, cv, t0, t1, t2, t3;
};

template <int R>
DEV unsigned w4_setup(const KArgs& a, const ItemConst* icg, const ItemId& id, int item, unsigned tmask, int gf, bool optimise,
                      const FrameRows<R>& fr)
{
    const bool trk = optimise && id.is_joint && item < NJ && ((tmask >> item) & 1u);
    const int rank = __popc(tmask & ((1u << (item & 31)) - 1u));
    const int E = __popc(tmask);
    unsigned sel6 = 0, m = tmask;
    for (int u = 0; u < 6; ++u) {
        const int t = __builtin_ctz(m | 0x80000000u);
        m &= m - 1u;
        sel6 |= ((id.ch_sub >> t) & 1u) << u;
    }
    if (item == 0) {
        *(f4*)(fr.qd + 20) = *(const f4*)(a.cur_rot + (size_t)gf * 4);
        fr.qd[24] = __uint_as_float(tmask);
    }
    if (trk) {
        const float invE = 1.f / (float)E;
        const float* p = a.tgt_pos + (size_t)(gf * NJ + item) * 3;
        const float* q = a.tgt_rot + (size_t)(gf * NJ + item) * 9;
        const float wp = a.w[(gf * NJ + item) * 2 + 0], wr = a.w[(gf * NJ + item) * 2 + 1];
        const float clp = wp * invE * (1.f / 3.f);             // loss_pos coefficient  w_pos / (3E)
        const float clr = a.lam_rot * wr * invE * (1.f / 9.f); // loss_rot coefficient  lam w_rot / (9E)
        float* t = fr.trk + rank * 4;
        *(f4*)(t) = f4{p[0], p[1], p[2], 2.f * clp};
        *(f4*)(t + 4 * R) = f4{q[0], q[1], q[2], q[3]};
        *(f4*)(t + 8 * R) = f4{q[4], q[5], q[6], q[7]};
        *(f4*)(t + 12 * R) = f4{q[8], 2.f * clr, clp, clr};
    }
    // constant root-frame bones of the root's children
    if (item < MAX_ROOT_CH) *(f4*)(fr.bone + icg->init_id * 4) = f4{icg->init_off[0], icg->init_off[1], icg->init_off[2], 0.f};
    return (trk ? 1u : 0u) | ((unsigned)(rank & 31) << 1) | (sel6 << 8);
}

// stage 1: normalise, rotation matrix, child bone; root / displacement publish qw, R0, d
template <int R>
DEV void w4_s1(const KArgs& a, const ItemId& id, unsigned pk, const float* icl, const f4 y4, const FrameRows<R>& fr, P3S& s, int iter,
               int gf, bool fvalid)
{
    const bool trk = (pk & 1u) != 0u;
    const int rank = (int)((pk >> 1) & 31u);
    const float* tin = fr.trk + rank * 4;
    s.cv = *(const f4*)(fr.qd + 20); // cur_rot of my frame: used by the root lane only
    if (DBG_DUMP && a.dbg && iter == 0 && fvalid && id.dq == id.sq) *(f4*)(a.dbg + (size_t)gf * DBG_STRIDE + DBG_Y + 4 * id.sq) = y4;
    const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
    s.sd = sd;
    const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
    const float nn = rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z;
    const float inv = id.has_quat ? __builtin_amdgcn_rsqf(nn) : 0.f;
    s.inv = inv;
    const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
    s.q = q;
    // one quaternion -> matrix for every lane: the root lane's is the world rotation qw = cur (x) q_0 (its own M is
    // the identity in the root frame), every other lane's is its root-space joint rotation (1 (x) q = q exactly)
    const Q4 qs = quat_mul(id.is_root ? Q4{s.cv.x, s.cv.y, s.cv.z, s.cv.w} : Q4{1.f, 0.f, 0.f, 0.f}, q);
    M3 M = quat_to_mat(qs);
    if (id.is_root) {
        *(f4*)(fr.qd) = f4{qs.w, qs.x, qs.y, qs.z};
        *(f4*)(fr.qd + 8) = f4{M.m00, M.m01, M.m02, 0.f};
        *(f4*)(fr.qd + 12) = f4{M.m10, M.m11, M.m12, 0.f};
        *(f4*)(fr.qd + 16) = f4{M.m20, M.m21, M.m22, 0.f};
    }
    if (id.is_root) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (id.is_disp) *(f4*)(fr.qd + 4) = f4{rq.w, rq.x, rq.y, 0.f};
    s.M = M;
    {
        const f4 cho = *(const f4*)(icl + 256); // child offset (x,y,z)
        const V3 u = mat_vec(M, V3{cho.x, cho.y, cho.z});
        *(f4*)(fr.bone + id.ch_id * 4) = f4{u.x, u.y, u.z, 0.f};
    }
    if (trk) { // tracker inputs of my joint (loop-invariant LDS data)
        s.t0 = *(const f4*)(tin);          // tp, cgp
        s.t1 = *(const f4*)(tin + 4 * R);  // tR[0..3]
        s.t2 = *(const f4*)(tin + 8 * R);  // tR[4..7]
        s.t3 = *(const f4*)(tin + 12 * R); // tR[8], cgr, clp, clr
    }
}

// stage 2: root-frame position, tracker terms
template <int R> DEV void w4_s2(const ItemId& id, unsigned pk, const FrameRows<R>& fr, P3S& s)
{
    const bool trk = (pk & 1u) != 0u;
    const int rank = (int)((pk >> 1) & 31u);
    const f4 qwv = *(const f4*)(fr.qd);
    const f4 dv = *(const f4*)(fr.qd + 4);
    const f4 r0v = *(const f4*)(fr.qd + 8), r1v = *(const f4*)(fr.qd + 12), r2v = *(const f4*)(fr.qd + 16);
    const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
    const M3 R0 = {r0v.x, r0v.y, r0v.z, r1v.x, r1v.y, r1v.z, r2v.x, r2v.y, r2v.z};
    const M3& M = s.M;
    V3 pr = {dv.x, dv.y, dv.z}; // root-frame position: d + sum of the bones on the path
    {
        f4 bn[MAX_PATH];
#pragma unroll
        for (int i = 0; i < MAX_PATH; ++i) {
            const unsigned k = (i < 6) ? ((id.plo >> (5 * i)) & 31u) : (id.phi & 31u);
            bn[i] = *(const f4*)(fr.bone + k * 4);
        }
#pragma unroll
        for (int i = 0; i < MAX_PATH; ++i) { pr.x += bn[i].x; pr.y += bn[i].y; pr.z += bn[i].z; }
    }
    M3 gM = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (trk) { // tracker terms in the root frame
        const f4 t0 = s.t0, t1 = s.t1, t2 = s.t2, t3 = s.t3;
        const V3 tp = {t0.x, t0.y, t0.z};
        const M3 tR = {t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x};
        const float cgp = t0.w, cgr = t3.y;
        const V3 tpr = matT_vec(R0, tp);
        const V3 e = {pr.x - tpr.x, pr.y - tpr.y, pr.z - tpr.z};
        const V3 gp = {cgp * e.x, cgp * e.y, cgp * e.z};
        const M3 tRr = matT_mat(R0, tR);
        const M3 eM = {M.m00 - tRr.m00, M.m01 - tRr.m01, M.m02 - tRr.m02, M.m10 - tRr.m10, M.m11 - tRr.m11,
                       M.m12 - tRr.m12, M.m20 - tRr.m20, M.m21 - tRr.m21, M.m22 - tRr.m22};
        gM = {cgr * eM.m00, cgr * eM.m01, cgr * eM.m02, cgr * eM.m10, cgr * eM.m11, cgr * eM.m12,
              cgr * eM.m20, cgr * eM.m21, cgr * eM.m22};
        // dL/dR0 = -(tp gp^T + tR gM^T)  ->  contribution to dL/d(qw)
        M3 C = mat_matT(tR, gM);
        C.m00 = -(C.m00 + tp.x * gp.x); C.m01 = -(C.m01 + tp.x * gp.y); C.m02 = -(C.m02 + tp.x * gp.z);
        C.m10 = -(C.m10 + tp.y * gp.x); C.m11 = -(C.m11 + tp.y * gp.y); C.m12 = -(C.m12 + tp.y * gp.z);
        C.m20 = -(C.m20 + tp.z * gp.x); C.m21 = -(C.m21 + tp.z * gp.y); C.m22 = -(C.m22 + tp.z * gp.z);
        const Q4 gqw_t = quat_mat_grad(qw, C);
        *(f4*)(fr.gpc + rank * 4) = f4{gp.x, gp.y, gp.z, 0.f};
        *(f4*)(fr.cq + rank * 4) = f4{gqw_t.w, gqw_t.x, gqw_t.y, gqw_t.z};
        const float l_p = t3.z * (e.x * e.x + e.y * e.y + e.z * e.z);
        const float l_r = t3.w * (eM.m00 * eM.m00 + eM.m01 * eM.m01 + eM.m02 * eM.m02 + eM.m10 * eM.m10 + eM.m11 * eM.m11 +
                                  eM.m12 * eM.m12 + eM.m20 * eM.m20 + eM.m21 * eM.m21 + eM.m22 * eM.m22);
        *(f2*)(fr.lp + rank * 2) = f2{l_p, l_r}; // read by the epilogue after the last iteration
    }
    s.gM = gM;
}

// stage 3: subtree sums, dL/dq, projection -> the item's quad of dL/dy
template <int R>
DEV f4 w4_s3(const KArgs& a, const ItemId& id, unsigned pk, int Emax, const float* icl, const FrameRows<R>& fr, const P3S& s, int iter,
             int gf, bool fvalid)
{
    f4 S4 = {0.f, 0.f, 0.f, 0.f};
    {
        const float* tab = id.is_root ? fr.cq : fr.gpc;
        f4 g[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) g[u] = *(const f4*)(tab + u * 4);
        const unsigned sel6 = pk >> 8;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const float bb = (float)((sel6 >> u) & 1u);
            S4.x += bb * g[u].x; S4.y += bb * g[u].y; S4.z += bb * g[u].z; S4.w += bb * g[u].w;
        }
        if (Emax > 6) { // more than 6 trackers in a frame of this wave (uniform, rare): general path
            unsigned m = __float_as_uint(fr.qd[24]);
#pragma unroll
            for (int u = 0; u < 6; ++u) m &= m - 1u;
            for (int e0 = 6; e0 < Emax; ++e0) {
                const f4 ge = *(const f4*)(tab + e0 * 4);
                const int t = __builtin_ctz(m | 0x80000000u); // joint id of this rank (31 when exhausted)
                m &= m - 1u;
                const float bb = (float)((id.ch_sub >> t) & 1u);
                S4.x += bb * ge.x; S4.y += bb * ge.y; S4.z += bb * ge.z; S4.w += bb * ge.w;
            }
        }
    }
    const Q4 q = s.q;
    const f4 sd = s.sd, cv = s.cv;
    const float inv = s.inv;
    const Q4 gq_root = quat_mul(Q4{cv.x, -cv.y, -cv.z, -cv.w}, Q4{S4.x, S4.y, S4.z, S4.w});
    const f4 cho = *(const f4*)(icl + 256);
    M3 X = s.gM;
    X.m00 += S4.x * cho.x; X.m01 += S4.x * cho.y; X.m02 += S4.x * cho.z;
    X.m10 += S4.y * cho.x; X.m11 += S4.y * cho.y; X.m12 += S4.y * cho.z;
    X.m20 += S4.z * cho.x; X.m21 += S4.z * cho.y; X.m22 += S4.z * cho.z;
    const Q4 gq_joint = quat_mat_grad(q, X);
    const Q4 gq = id.is_root ? gq_root : gq_joint;
    const V3 S = {S4.x, S4.y, S4.z};
    const float dot = q.w * gq.w + q.x * gq.x + q.y * gq.y + q.z * gq.z;
    f4 gyv = {sd.x * (gq.w - q.w * dot) * inv, sd.y * (gq.x - q.x * dot) * inv,
              sd.z * (gq.y - q.y * dot) * inv, sd.w * (gq.z - q.z * dot) * inv};
    if (id.is_disp) gyv = f4{sd.x * S.x, sd.y * S.y, sd.z * S.z, 0.f}; // ch_sub = every joint
    if (id.dq < 0) gyv = f4{0.f, 0.f, 0.f, 0.f};
    if (DBG_DUMP && a.dbg && iter == 0 && fvalid && id.dq >= 0) *(f4*)(a.dbg + (size_t)gf * DBG_STRIDE + DBG_GY + 4 * id.dq) = gyv;
    return gyv;
}

// outputs of the LAST forward pass of (item, frame gf): the same as dp_p3.h::p3_outputs with the decoder quad in registers
template <int R>
DEV void w4_outputs(const KArgs& a, const ItemId& id, const float* icl, const f4 y4, const FrameRows<R>& fr, int item, int gf, bool optimise,
                    const float* zpre_row, const float* zt_row)
{
    const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
    const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
    const float inv = id.has_quat ? __builtin_amdgcn_rsqf(rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z) : 0.f;
    const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
    M3 M = quat_to_mat(q);
    if (id.is_root) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    const f4 qwv = *(const f4*)(fr.qd);
    const f4 dv = *(const f4*)(fr.qd + 4);
    const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
    const M3 R0 = quat_to_mat(qw);
    if (id.is_joint) {
        if (a.pose) {
            float* o = a.pose + (size_t)gf * 88 + 4 * item;
            o[0] = (q.w - mu.x) / sd.x; o[1] = (q.x - mu.y) / sd.y;
            o[2] = (q.y - mu.z) / sd.z; o[3] = (q.z - mu.w) / sd.w;
        }
        if (a.pos) {
            V3 pr = {dv.x, dv.y, dv.z};
            for (int i = 0; i < MAX_PATH; ++i) {
                const unsigned k = (i < 6) ? ((id.plo >> (5 * i)) & 31u) : (id.phi & 31u);
                const f4 bn = *(const f4*)(fr.bone + k * 4);
                pr.x += bn.x; pr.y += bn.y; pr.z += bn.z;
            }
            const V3 pw = mat_vec(R0, pr);
            float* o = a.pos + ((size_t)gf * NJ + item) * 3;
            o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
        }
        if (a.rot) {
            const M3 G = mat_mat(R0, M);
            float* o = a.rot + ((size_t)gf * NJ + item) * 9;
            o[0] = G.m00; o[1] = G.m01; o[2] = G.m02; o[3] = G.m10; o[4] = G.m11; o[5] = G.m12; o[6] = G.m20; o[7] = G.m21; o[8] = G.m22;
        }
    }
    if (id.is_root) {
        if (a.world_rot) { float* o = a.world_rot + (size_t)gf * 4; o[0] = qw.w; o[1] = qw.x; o[2] = qw.y; o[3] = qw.z; }
        if (optimise && a.loss) {
            float lsum_p = 0.f, lsum_r = 0.f, lt = 0.f;
            const int E = __popc(__float_as_uint(fr.qd[24]));
            for (int e0 = 0; e0 < E; ++e0) { const f2 l = *(const f2*)(fr.lp + e0 * 2); lsum_p += l.x; lsum_r += l.y; }
            for (int k = 0; k < LAT; k += 4) {
                const f4 dz = *(const f4*)(zpre_row + k) - *(const f4*)(zt_row + k);
                lt += dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
            }
            a.loss[(size_t)gf * 3 + 0] = lsum_p;
            a.loss[(size_t)gf * 3 + 1] = lsum_r;
            a.loss[(size_t)gf * 3 + 2] = lt * a.lam_tmp * (1.f / 24.f);
        }
    }
    if (id.is_disp) {
        if (a.disp) { float* o = a.disp + (size_t)gf * 3; o[0] = rq.w; o[1] = rq.x; o[2] = rq.y; }
        if (a.world_disp) {
            const V3 wd = mat_vec(R0, V3{rq.w, rq.x, rq.y});
            float* o = a.world_disp + (size_t)gf * 3; o[0] = wd.x; o[1] = wd.y; o[2] = wd.z;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void dp_w4_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[lds_total<NW>()];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, i = lane & 3; // quad = item (block of the products), lane in quad = frame
    const int nB = a.n_frames;
    const int f0 = (blockIdx.x * NW + wave) * FPW;
    const bool optimise = (a.mode == 0);

    // ---- weight image and item tables into LDS (the only data the waves of a workgroup share)
    for (int k = tid; k < IMG_FLOATS / 4; k += NW * 64) ((f4*)(lds + L_IMG))[k] = ((const f4*)a.w4img)[k];
    if (tid < 32 * 3) {
        const int it = tid & 31, k = tid >> 5;
        const float* src = (const float*)(a.items + it) + (k == 0 ? 0 : k == 1 ? 4 : 8);
        *(f4*)(lds + L_ITEM + k * 128 + 4 * it) = f4{src[0], src[1], src[2], k == 2 ? 0.f : src[3]};
    }
    float* fb0 = lds + L_FR + wave * FPW * FB_STRIDE; // this wave's four frame blocks
    for (int k = lane; k < FPW * FB_STRIDE; k += 64) fb0[k] = 0.f;
    __syncthreads();
    if (f0 >= nB) return; // (uniform per wave) no barrier below this line

    const int gfi = min(f0 + i, nB - 1); // my frame as the lane of a quad (clamped: ragged tails compute a copy)
    const bool fvalid = f0 + i < nB;
    float* fb = fb0 + i * FB_STRIDE;
    const FrameRows<W4_R> fr = {fb + FB_BONE, fb + FB_GPC, fb + FB_CQ, fb + FB_LP, fb + FB_QD, fb + FB_TRK};

    // ---- per-lane accumulator seeds (bias rows of L0, L1, L2A, L2B)
    const float bias0 = a.w4bias[lane], bias1 = a.w4bias[64 + lane], bias2a = a.w4bias[128 + lane], bias2b = a.w4bias[192 + lane];

    // ---- latent and Adam state in the D layout of the last product: lane = latent dim, register r = frame f0 + r
    f4 zD = {0.f, 0.f, 0.f, 0.f}, ztD = zD, mD = zD, vD = zD, zpreD = zD;
    if (lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            const int gf = min(f0 + r, nB - 1);
            zD[r] = a.z0[(size_t)gf * LAT + lane];
            if (optimise) ztD[r] = a.z_tgt[(size_t)gf * LAT + lane];
        }
    }

    // ---- kinematics identity of my two items
    const int itA = b, itB = ITEMS_A + b;
    const ItemId idA = load_item(a.items + itA), idB = load_item(a.items + itB);
    const float* iclA = lds + L_ITEM + 4 * itA;
    const float* iclB = lds + L_ITEM + 4 * itB;
    unsigned tmask = 0;
    if (optimise)
        for (int j = 0; j < NJ; ++j) tmask |= (a.tracked[(size_t)gfi * NJ + j] != 0 ? 1u : 0u) << j;
    const unsigned pkA = w4_setup<W4_R>(a, a.items + itA, idA, itA, tmask, gfi, optimise, fr);
    const unsigned pkB = w4_setup<W4_R>(a, a.items + itB, idB, itB, tmask, gfi, optimise, fr);
    const int E = __popc(tmask);
    const int Emax = max(max(__builtin_amdgcn_readlane(E, 0), __builtin_amdgcn_readlane(E, 1)),
                         max(__builtin_amdgcn_readlane(E, 2), __builtin_amdgcn_readlane(E, 3)));
    wave_sync();

    f4 yA = {0.f, 0.f, 0.f, 0.f}, yB = yA;
    P3S sA, sB;
    for (int iter = 0; iter < a.n_iter; ++iter) {
        const bool last = (iter == a.n_iter - 1);
        const float step = a.tab.step[iter], rbc2s = a.tab.bc2s[iter];
        int o = lane;
        asm volatile("" : "+v"(o)); // opaque per iteration: keeps the weight reads inside the loop (nothing to hoist and spill)
        const f4* w = (const f4*)(lds + L_IMG) + o;

        // ================= L0: a0 = lrelu(A0 z + c0)
        f4 x = zD;
        quad_transpose(x);
        f4 acc0 = splat(bias0), acc1 = splat(0.f);
        chain<6, 0>(acc0, acc1, x, w + (S_L0 / 4) * 64);
        const f4 a0D = lrelu4(acc0 + acc1);
        // ================= L1: a1 = lrelu(A1 a0 + b1)
        x = a0D;
        quad_transpose(x);
        acc0 = splat(bias1); acc1 = splat(0.f);
        chain<10, 0>(acc0, acc1, x, w + (S_L1 / 4) * 64);
        const f4 a1D = lrelu4(acc0 + acc1);
        // ================= L2: y = A2 a1 + b2, two 64-row blocks (items 0..15 | 16..31)
        x = a1D;
        quad_transpose(x);
        {
            f4 pa0 = splat(bias2a), pa1 = splat(0.f), pb0 = splat(bias2b), pb1 = splat(0.f);
            chain<15, 0>(pa0, pa1, x, w + (S_L2A / 4) * 64);
            chain<15, 0>(pb0, pb1, x, w + (S_L2B / 4) * 64);
            yA = pa0 + pa1;
            yB = pb0 + pb1;
        }
        quad_transpose(yA); // lane (b, i): the decoder quad of item b / 16 + b of frame i
        quad_transpose(yB);

        // ================= P3: normalise, FK, loss, backward to dL/dy
        w4_s1<W4_R>(a, idA, pkA, iclA, yA, fr, sA, iter, gfi, fvalid);
        w4_s1<W4_R>(a, idB, pkB, iclB, yB, fr, sB, iter, gfi, fvalid);
        wave_sync();
        if (!optimise) break; // forward-only launch (uniform)
        w4_s2<W4_R>(idA, pkA, fr, sA);
        w4_s2<W4_R>(idB, pkB, fr, sB);
        wave_sync();
        const f4 gyA = w4_s3<W4_R>(a, idA, pkA, Emax, iclA, fr, sA, iter, gfi, fvalid);
        const f4 gyB = w4_s3<W4_R>(a, idB, pkB, Emax, iclB, fr, sB, iter, gfi, fvalid);
        wave_sync(); // the next iteration's stage 1 overwrites rows stage 3 has read

        // ================= bL2: d1 = (A2^T gy) * lrelu'(a1): K = 4 channels of items 0..15 (gyA), then items 16..25 (gyB)
        acc0 = splat(0.f); acc1 = splat(0.f);
        chain<16, 0>(acc0, acc1, gyA, w + (S_B2 / 4) * 64);
        chain<10, 0>(acc0, acc1, gyB, w + (S_B2 / 4 + 16) * 64);
        x = dlrelu4(a1D, acc0 + acc1);
        quad_transpose(x);
        // ================= bL1: d0 = (A1^T d1) * lrelu'(a0)
        acc0 = splat(0.f); acc1 = splat(0.f);
        chain<15, 0>(acc0, acc1, x, w + (S_B1 / 4) * 64);
        x = dlrelu4(a0D, acc0 + acc1);
        quad_transpose(x);
        // ================= bL0 + Adam (torch.optim.Adam, single-tensor form; m, v start at 0, t = iter + 1)
        acc0 = splat(0.f); acc1 = splat(0.f);
        chain<10, 0>(acc0, acc1, x, w + (S_B0 / 4) * 64);
        const f4 g = (acc0 + acc1) + a.ctmp * (zD - ztD);
        if (DBG_DUMP && a.dbg && iter == 0 && lane < LAT) {
#pragma unroll
            for (int r = 0; r < FPW; ++r)
                if (f0 + r < nB) a.dbg[(size_t)(f0 + r) * DBG_STRIDE + DBG_GZ + lane] = g[r];
        }
        if (last) zpreD = zD; // latent of this (the last) forward pass
        mD = mD + a.one_m_b1 * (g - mD);
        vD = vD * a.beta2 + a.one_m_b2 * (g * g);
        const f4 den = f4{__builtin_amdgcn_sqrtf(vD.x), __builtin_amdgcn_sqrtf(vD.y), __builtin_amdgcn_sqrtf(vD.z),
                          __builtin_amdgcn_sqrtf(vD.w)} * rbc2s + a.eps;
        zD = zD - step * (mD * f4{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                  __builtin_amdgcn_rcpf(den.w)});
    }

    // ================= epilogue: outputs of the LAST forward pass (decoder quads still in registers, bones / qw / d and
    // the tracker loss terms in the frame blocks)
    if (!optimise) zpreD = zD;
    if (lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            fb0[r * FB_STRIDE + FB_ZPRE + lane] = zpreD[r];
            fb0[r * FB_STRIDE + FB_ZT + lane] = ztD[r];
        }
    }
    wave_sync();
    if (fvalid) {
        w4_outputs<W4_R>(a, idA, iclA, yA, fr, itA, gfi, optimise, fb + FB_ZPRE, fb + FB_ZT);
        w4_outputs<W4_R>(a, idB, iclB, yB, fr, itB, gfi, optimise, fb + FB_ZPRE, fb + FB_ZT);
    }
    if (optimise && lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            if (f0 + r < nB) {
                if (a.z) a.z[(size_t)(f0 + r) * LAT + lane] = zD[r];
                if (a.z_pre) a.z_pre[(size_t)(f0 + r) * LAT + lane] = zpreD[r];
            }
        }
    }
    if (optimise && a.iters && lane < FPW && f0 + lane < nB) a.iters[f0 + lane] = a.n_iter;
}

extern "C" hipError_t dp_launch_w4(const KArgs* args, hipStream_t stream)
{
    constexpr int NW = 4;
    const int grid = (args->n_frames + NW * FPW - 1) / (NW * FPW);
    hipLaunchKernelGGL(dp_w4_kernel<NW>, dim3(grid), dim3(NW * 64), 0, stream, *args);
    return hipGetLastError();
}

extern "C" int dp_w4_lds_bytes(void) { return lds_total<4>() * 4; }
extern "C" int dp_w4_frames_per_block(void) { return 4 * FPW; }

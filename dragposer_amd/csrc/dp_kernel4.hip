// dp_kernel4.hip -- co-resident variant of the fused latent-optimisation kernel (see dp_kernel.hip for the algorithm).
//
// Same mathematics, same instruction order inside every product and every kinematics item -- results are bit-identical
// to dp_kernel.hip -- but a workgroup is 4 waves (256 threads) with <= 76 KB of LDS, so TWO workgroups share a CU:
// while one is in its VALU-bound kinematics phase the other runs its latency-bound matrix phases, which the single
// 8-wave workgroup of dp_kernel.hip can not overlap.
//   ROUNDS = 2: 16 frames per workgroup, the kinematics phase runs twice per iteration (wave w: frames 4w+2r, 4w+2r+1).
//               For batches that give every CU at least two 16-frame workgroups.
//   ROUNDS = 1:  8 frames per workgroup (MFMA columns 8..15 idle), one kinematics round (wave w: frames 2w, 2w+1).
//               For smaller batches: twice the workgroups, so twice the CUs (or two groups per CU) are busy.
// Differences in data placement:
//   * layer 2's six tiles: wave w computes tile w (16 steps) AND one K-half of tile 4 / 5 (the half that wave w+4 of
//     the 8-wave kernel computes, same registers of the weight image);
//   * backward weights are not parked in LDS (that is what makes two groups fit): each matrix wave streams them from
//     the global weight image (L2-resident, 256 B coalesced per register) one phase ahead of their use;
//   * tracker inputs / gradients are stored by tracker RANK (capacity K4_R per frame) instead of by joint.
// Fixed iteration count only (no early stop); dp_host.cpp dispatches here when the caller states max_trackers <= K4_R.
#include "dp_device.h"

constexpr int K4_R = 16; // tracker capacity per frame
constexpr int K4_NT = 256;

// ------------------------------------------------------------------------------------------------
// LDS map (floats); rows are always 16 (MFMA columns), also when only 8 carry frames
constexpr int K_Z = 0;                          // zs [16][S_Z]
constexpr int K_A0 = K_Z + FPB * S_Z;           // a0 / d0
constexpr int K_A1 = K_A0 + FPB * S_A0;         // a1 / d1
constexpr int K_GY = K_A1 + FPB * S_A1;         // dL/dy quads
constexpr int K_Y = K_GY + FPB * S_Y;           // y [2 planes][16][S_Y]
constexpr int K_BONE = K_Y + 2 * FPB * S_Y;     // bone[16][32][4]
constexpr int K_GPC = K_BONE + FPB * 32 * 4;    // gpc [16][R][4]
constexpr int K_CQ = K_GPC + FPB * K4_R * 4;    // cq  [16][R][4]
constexpr int K_LP = K_CQ + FPB * K4_R * 4;     // lp  [16][R][2]
constexpr int QD_S = 28;                        // qd row: qw[4] | d[3],0 | R0 rows (3 x [3],0) | cur[4] | tracked-joint mask, -, -, -
constexpr int K_QD = K_LP + FPB * K4_R * 2;     // qd  [16][QD_S]
constexpr int K_TRK = K_QD + FPB * QD_S;         // tracker inputs [16 frames][4 quads][R ranks][4]
constexpr int K_ITEM = K_TRK + FPB * K4_R * 16; // item constants, SoA
constexpr int K_BIAS = K_ITEM + 3 * 32 * 4;
constexpr int K_ZT = K_BIAS + 128;
constexpr int K_ZPRE = K_ZT + FPB * S_Z;
constexpr int K_ADM = K_ZPRE + FPB * S_Z;
constexpr int K_ADV = K_ADM + FPB * S_Z;
constexpr int K_TOTAL = K_ADV + FPB * S_Z;
static_assert(K_TOTAL * 4 * 2 <= 160 * 1024, "two workgroups per CU");

// loop-invariant global loads would be hoisted out of the iteration loop (and spilled); an opaque copy of the
// pointer per iteration keeps every stream of weights where it is issued
DEV const float* opaque(const float* p)
{
    asm volatile("" : "+s"(p));
    return p;
}

template <int ROUNDS>
__global__ __launch_bounds__(K4_NT, 2) void dp_optimize_kernel4(const KArgs a)
{
    constexpr int FR = 8 * ROUNDS; // frames per workgroup
    __shared__ __attribute__((aligned(16))) float lds[K_TOTAL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f16 = lane & 15, h = lane >> 4;
    const bool fhi = f16 >= 8;
    const int hs = h ^ (fhi ? 2 : 0);
    const int pf0 = (ROUNDS == 2 ? 4 : 2) * wave + (lane >> 5); // P3 frame of round 0 (round r: + 2r)
    const int it_id = lane & 31;
    const int blk0 = blockIdx.x * FR;
    const int nB = a.n_frames;
    const bool optimise = (a.mode == 0);

    float* zs = lds + K_Z + f16 * S_Z;
    float* a0r = lds + K_A0 + f16 * S_A0;
    float* a1r = lds + K_A1 + f16 * S_A1;
    float* yr = lds + K_Y + f16 * S_Y;
    const float* gyr = lds + K_GY + f16 * S_Y;
    float* yp = lds + K_Y;
    const ItemConst* icg = a.items + it_id;
    const float* icl = lds + K_ITEM + 4 * it_id;

    for (int i = tid; i < K_TOTAL; i += K4_NT) lds[i] = 0.f;
    __syncthreads();
    if (tid < 32 * 3) {
        const int it = tid & 31, k = tid >> 5;
        const float* src = (const float*)(a.items + it) + (k == 0 ? 0 : k == 1 ? 4 : 8);
        *(f4*)(lds + K_ITEM + k * 128 + 4 * it) = f4{src[0], src[1], src[2], k == 2 ? 0.f : src[3]};
    }
    if (tid < 128) lds[K_BIAS + tid] = a.bias[tid];

    // ---- forward weights: VGPR-resident; [L0:6][L1:10][L2:16] of wave w, then the first 8 L2 registers of wave w+4
    constexpr int WF = W_FWD + L2_HALF_STEPS;
    float W[WF];
#pragma unroll
    for (int i = 0; i < W_FWD; ++i) W[i] = a.wfrag[(wave * W_REGS + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < L2_HALF_STEPS; ++i) W[W_FWD + i] = a.wfrag[((wave + 4) * W_REGS + W_OFF_L2 + i) * 64 + lane];
    const unsigned mk_l1 = a.smask[wave][G_L1], mk_l2 = a.smask[wave][G_L2], mk_l2h = a.smask[wave + 4][G_L2],
                   mk_b2 = a.smask[wave][G_B2], mk_b1 = a.smask[wave][G_B1];
    const int half = l2_half(wave + 4);
    const float* wbw = a.wfrag + (size_t)(wave * W_REGS) * 64; // this wave's weight image (uniform): [register][lane]

    const int zd = 16 * (wave & 1) + 4 * h;
    const bool zvalid = wave < 2 && zd < LAT && f16 < FR;
    if (zvalid) {
        const int gf = min(blk0 + f16, nB - 1);
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        *(f4*)(zs + zd) = swz4(*(const f4*)(a.z0 + (size_t)gf * LAT + zd), fhi);
        *(f4*)(lds + K_ZT + f16 * S_Z + zd) = optimise ? *(const f4*)(a.z_tgt + (size_t)gf * LAT + zd) : zero4;
    }

    // ---- P3 per-lane identity
    const int sq = icg->src_quad, dq = icg->dst_quad;
    const int ch_id = icg->ch_id;
    const unsigned ch_sub = icg->ch_sub, plo = icg->path_lo, phi = icg->path_hi;
    const int kind = icg->kind;
    const bool is_joint = kind == KIND_JOINT || kind == KIND_ROOT;
    const bool has_quat = kind != KIND_DISP && kind != KIND_IDLE;
    const bool is_root = kind == KIND_ROOT;
    const bool is_disp = kind == KIND_DISP;
    // per-round tracker bookkeeping: one packed word per lane (bit 0 tracked, bits 1..5 rank, bits 8..13 sel6: the tracked
    // joint of rank u lies below my child bone); what is per frame (cur_rot, the tracked-joint mask) sits in the qd row
    unsigned pk_[ROUNDS];
    int Emax_[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int pf = pf0 + 2 * r;
        const int gfc = min(blk0 + pf, nB - 1);
        bool trk = false;
        if (optimise && is_joint) trk = a.tracked[gfc * NJ + it_id] != 0;
        unsigned long long bal = __ballot(trk);
        unsigned tmask = (lane >> 5) ? (unsigned)(bal >> 32) : (unsigned)bal;
        const int rank = __popc(tmask & ((1u << it_id) - 1u));
        if (rank >= K4_R) trk = false; // beyond the stated capacity: ignored (memory-safe; the host never dispatches such batches here)
        bal = __ballot(trk);
        tmask = (lane >> 5) ? (unsigned)(bal >> 32) : (unsigned)bal;
        const int E = __popc(tmask);
        unsigned sel6 = 0, m = tmask;
        for (int u = 0; u < 6; ++u) {
            const int t = __builtin_ctz(m | 0x80000000u);
            m &= m - 1u;
            sel6 |= ((ch_sub >> t) & 1u) << u;
        }
        pk_[r] = (trk ? 1u : 0u) | ((unsigned)(rank & 31) << 1) | (sel6 << 8);
        Emax_[r] = max(__builtin_amdgcn_readlane(E, 0), __builtin_amdgcn_readlane(E, 32));
        if (it_id == 0) {
            float* qd = lds + K_QD + pf * QD_S;
            *(f4*)(qd + 20) = *(const f4*)(a.cur_rot + (size_t)gfc * 4);
            qd[24] = __uint_as_float(tmask);
        }
        if (trk) {
            const float invE = 1.f / (float)E;
            const float* p = a.tgt_pos + (size_t)(gfc * NJ + it_id) * 3;
            const float* q = a.tgt_rot + (size_t)(gfc * NJ + it_id) * 9;
            const float wp = a.w[(gfc * NJ + it_id) * 2 + 0], wr = a.w[(gfc * NJ + it_id) * 2 + 1];
            const float clp = wp * invE * (1.f / 3.f);
            const float clr = a.lam_rot * wr * invE * (1.f / 9.f);
            float* t = lds + K_TRK + (pf * 4 * K4_R + rank) * 4;
            *(f4*)(t) = f4{p[0], p[1], p[2], 2.f * clp};
            *(f4*)(t + 4 * K4_R) = f4{q[0], q[1], q[2], q[3]};
            *(f4*)(t + 8 * K4_R) = f4{q[4], q[5], q[6], q[7]};
            *(f4*)(t + 12 * K4_R) = f4{q[8], 2.f * clr, clp, clr};
        }
        if (it_id < MAX_ROOT_CH)
            *(f4*)(lds + K_BONE + pf * 128 + icg->init_id * 4) = f4{icg->init_off[0], icg->init_off[1], icg->init_off[2], 0.f};
    }

    f4 a0v = {0.f, 0.f, 0.f, 0.f}, a1v = a0v;
    __syncthreads();
#ifdef DP_PROFILE
    unsigned long long prof[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif

    for (int iter = 0; iter < a.n_iter; ++iter) {
        const bool last = (iter == a.n_iter - 1);
        const float step = a.tab.step[iter], rbc2s = a.tab.bc2s[iter];
        const float* wb = opaque(wbw);

        // ================= L0
        if (wave < 3) {
            float b[6];
            load_b<6>(zs, hs, 0, b);
            const f4 bias = *(const f4*)(lds + K_BIAS + 16 * wave + 4 * h);
            float w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = W[W_OFF_L0 + i];
            a0v = lrelu4(mfma_chain_dense<6>(w, b, bias));
            *(f4*)(a0r + 16 * wave + 4 * h) = swz4(a0v, fhi);
        }
        STAMP(0);
        __syncthreads();
        STAMP(1);

        // ================= L1
        {
            float b[10];
            load_b<10>(a0r, hs, 0, b);
            const f4 bias = *(const f4*)(lds + K_BIAS + 64 + 16 * wave + 4 * h);
            float w[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) w[i] = W[W_OFF_L1 + i];
            a1v = lrelu4(mfma_chain<10>(w, b, mk_l1, bias));
            *(f4*)(a1r + 16 * wave + 4 * h) = swz4(a1v, fhi);
        }
        STAMP(2);
        __syncthreads();
        STAMP(3);

        // ================= L2: tile `wave` over all 16 steps, plus one K-half of tile 4 / 5
        {
            float b[16], w[16], bh[8], wh[8];
            load_b<16>(a1r, hs, 0, b);
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = W[W_OFF_L2 + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) wh[i] = W[W_FWD + i];
            const f4 acc = mfma_chain<16>(w, b, mk_l2, f4{0.f, 0.f, 0.f, 0.f});
            f4 acch;
            if (half) { // (uniform; a select between b[i] and b[8 + i] would be lowered to a scratch array)
#pragma unroll
                for (int i = 0; i < 8; ++i) bh[i] = b[8 + i];
                acch = mfma_chain<8>(wh, bh, mk_l2h, f4{0.f, 0.f, 0.f, 0.f});
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) bh[i] = b[i];
                acch = mfma_chain<8>(wh, bh, mk_l2h, f4{0.f, 0.f, 0.f, 0.f});
            }
            *(f4*)(yr + 16 * wave + 4 * h) = acc;
            *(f4*)(yr + half * FPB * S_Y + 16 * (L2_SPLIT_TILE0 + (wave & 1)) + 4 * h) = acch;
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);

        // ================= P3 (wave-private rows), ROUNDS x 2 frames per wave
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int pf = pf0 + 2 * r;
            const int gfp = blk0 + pf;
            const bool fvalid = gfp < nB && pf < FR;
            float* bone = lds + K_BONE + pf * 128;
            float* gpc = lds + K_GPC + pf * (4 * K4_R);
            float* cqb = lds + K_CQ + pf * (4 * K4_R);
            float* lpb = lds + K_LP + pf * (2 * K4_R);
            float* qdb = lds + K_QD + pf * QD_S;
            const bool trk = (pk_[r] & 1u) != 0u;
            const int rank = (int)((pk_[r] >> 1) & 31u);
            const float* tin = lds + K_TRK + (pf * 4 * K4_R + rank) * 4;

            const f4 y4 = *(const f4*)(yp + pf * S_Y + 4 * sq) + *(const f4*)(yp + (FPB + pf) * S_Y + 4 * sq);
            if (DBG_DUMP && a.dbg && iter == 0 && fvalid && dq == sq) *(f4*)(a.dbg + (size_t)gfp * DBG_STRIDE + DBG_Y + 4 * sq) = y4;
            const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
            const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
            const float nn = rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z;
            const float inv = has_quat ? __builtin_amdgcn_rsqf(nn) : 0.f;
            const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
            M3 M = quat_to_mat(q);
            if (is_root) {
                const f4 cv = *(const f4*)(qdb + 20);
                const Q4 qw0 = quat_mul(Q4{cv.x, cv.y, cv.z, cv.w}, q);
                const M3 R = quat_to_mat(qw0);
                *(f4*)(qdb) = f4{qw0.w, qw0.x, qw0.y, qw0.z};
                *(f4*)(qdb + 8) = f4{R.m00, R.m01, R.m02, 0.f};
                *(f4*)(qdb + 12) = f4{R.m10, R.m11, R.m12, 0.f};
                *(f4*)(qdb + 16) = f4{R.m20, R.m21, R.m22, 0.f};
                M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
            }
            if (is_disp) *(f4*)(qdb + 4) = f4{rq.w, rq.x, rq.y, 0.f};
            {
                const f4 cho = *(const f4*)(icl + 256);
                const V3 u = mat_vec(M, V3{cho.x, cho.y, cho.z});
                *(f4*)(bone + ch_id * 4) = f4{u.x, u.y, u.z, 0.f};
            }
            wave_sync();
            STAMP(6);

            const f4 qwv = *(const f4*)(qdb);
            const f4 dv = *(const f4*)(qdb + 4);
            const f4 r0v = *(const f4*)(qdb + 8), r1v = *(const f4*)(qdb + 12), r2v = *(const f4*)(qdb + 16);
            f4 t0, t1, t2, t3;
            if (trk) {
                t0 = *(const f4*)(tin);
                t1 = *(const f4*)(tin + 4 * K4_R);
                t2 = *(const f4*)(tin + 8 * K4_R);
                t3 = *(const f4*)(tin + 12 * K4_R);
            }
            const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
            const M3 R0 = {r0v.x, r0v.y, r0v.z, r1v.x, r1v.y, r1v.z, r2v.x, r2v.y, r2v.z};
            V3 pr = {dv.x, dv.y, dv.z};
            {
                f4 b[MAX_PATH];
#pragma unroll
                for (int i = 0; i < MAX_PATH; ++i) {
                    const unsigned k = (i < 6) ? ((plo >> (5 * i)) & 31u) : (phi & 31u);
                    b[i] = *(const f4*)(bone + k * 4);
                }
#pragma unroll
                for (int i = 0; i < MAX_PATH; ++i) { pr.x += b[i].x; pr.y += b[i].y; pr.z += b[i].z; }
            }
            M3 gM = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (trk) {
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
                M3 C = mat_matT(tR, gM);
                C.m00 = -(C.m00 + tp.x * gp.x); C.m01 = -(C.m01 + tp.x * gp.y); C.m02 = -(C.m02 + tp.x * gp.z);
                C.m10 = -(C.m10 + tp.y * gp.x); C.m11 = -(C.m11 + tp.y * gp.y); C.m12 = -(C.m12 + tp.y * gp.z);
                C.m20 = -(C.m20 + tp.z * gp.x); C.m21 = -(C.m21 + tp.z * gp.y); C.m22 = -(C.m22 + tp.z * gp.z);
                const Q4 gqw_t = quat_mat_grad(qw, C);
                *(f4*)(gpc + rank * 4) = f4{gp.x, gp.y, gp.z, 0.f};
                *(f4*)(cqb + rank * 4) = f4{gqw_t.w, gqw_t.x, gqw_t.y, gqw_t.z};
                const float l_p = t3.z * (e.x * e.x + e.y * e.y + e.z * e.z);
                const float l_r = t3.w * (eM.m00 * eM.m00 + eM.m01 * eM.m01 + eM.m02 * eM.m02 + eM.m10 * eM.m10 + eM.m11 * eM.m11 +
                                          eM.m12 * eM.m12 + eM.m20 * eM.m20 + eM.m21 * eM.m21 + eM.m22 * eM.m22);
                *(f2*)(lpb + rank * 2) = f2{l_p, l_r};
            }
            wave_sync();
            STAMP(7);

            V3 S = {0.f, 0.f, 0.f};
            Q4 gqw = {0.f, 0.f, 0.f, 0.f};
            {
                f4 g[6], c[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) g[u] = *(const f4*)(gpc + u * 4);
                if (is_root) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) c[u] = *(const f4*)(cqb + u * 4);
#pragma unroll
                    for (int u = 0; u < 6; ++u) { gqw.w += c[u].x; gqw.x += c[u].y; gqw.y += c[u].z; gqw.z += c[u].w; }
                }
                const unsigned sel6 = pk_[r] >> 8;
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const float b = (float)((sel6 >> u) & 1u);
                    S.x += b * g[u].x; S.y += b * g[u].y; S.z += b * g[u].z;
                }
                if (Emax_[r] > 6) {
                    unsigned m = __float_as_uint(qdb[24]);
#pragma unroll
                    for (int u = 0; u < 6; ++u) m &= m - 1u;
                    for (int e0 = 6; e0 < Emax_[r]; ++e0) {
                        const f4 ge = *(const f4*)(gpc + e0 * 4);
                        const int t = __builtin_ctz(m | 0x80000000u);
                        m &= m - 1u;
                        const float b = (float)((ch_sub >> t) & 1u);
                        S.x += b * ge.x; S.y += b * ge.y; S.z += b * ge.z;
                        if (is_root) { const f4 ce = *(const f4*)(cqb + e0 * 4); gqw.w += ce.x; gqw.x += ce.y; gqw.y += ce.z; gqw.z += ce.w; }
                    }
                }
            }
            Q4 gq;
            if (is_root) {
                const f4 cv = *(const f4*)(qdb + 20);
                gq = quat_mul(Q4{cv.x, -cv.y, -cv.z, -cv.w}, gqw);
            } else {
                const f4 cho = *(const f4*)(icl + 256);
                M3 X = gM;
                X.m00 += S.x * cho.x; X.m01 += S.x * cho.y; X.m02 += S.x * cho.z;
                X.m10 += S.y * cho.x; X.m11 += S.y * cho.y; X.m12 += S.y * cho.z;
                X.m20 += S.z * cho.x; X.m21 += S.z * cho.y; X.m22 += S.z * cho.z;
                gq = quat_mat_grad(q, X);
            }
            const float dot = q.w * gq.w + q.x * gq.x + q.y * gq.y + q.z * gq.z;
            f4 gyv = {sd.x * (gq.w - q.w * dot) * inv, sd.y * (gq.x - q.x * dot) * inv,
                      sd.z * (gq.y - q.y * dot) * inv, sd.w * (gq.z - q.z * dot) * inv};
            if (is_disp) gyv = f4{sd.x * S.x, sd.y * S.y, sd.z * S.z, 0.f};
            if (dq >= 0) {
                *(f4*)(lds + K_GY + pf * S_Y + 4 * dq) = swz4(gyv, pf >= 8);
                if (DBG_DUMP && a.dbg && iter == 0 && fvalid) *(f4*)(a.dbg + (size_t)gfp * DBG_STRIDE + DBG_GY + 4 * dq) = gyv;
            }
            if (ROUNDS > 1) __builtin_amdgcn_sched_barrier(0); // keep the rounds apart: interleaving them doubles the live temporaries
        }
        if (!optimise) break;
        float WB2[26]; // bL2 weights: streamed from the global image, in flight across the barrier
#pragma unroll
        for (int i = 0; i < 26; ++i) WB2[i] = wb[(W_OFF_B2 + i) * 64 + lane];
        STAMP(8);
        __syncthreads();
        STAMP(9);

        // ================= bL2
        float WB1[16];
        {
            float b[26];
            load_b<26>(gyr, hs, 0, b);
            if (wave < 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) WB1[i] = wb[(W_OFF_B1 + i) * 64 + lane];
            }
            const f4 acc = mfma_chain<26>(WB2, b, mk_b2, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(a1r + 16 * wave + 4 * h) = swz4(dlrelu4(a1v, acc), fhi);
        }
        STAMP(10);
        __syncthreads();
        STAMP(11);

        // ================= bL1
        float WB0[10];
        if (wave < 3) {
            float b[16];
            load_b<16>(a1r, hs, 0, b);
            if (wave < 2) {
#pragma unroll
                for (int i = 0; i < 10; ++i) WB0[i] = wb[(W_OFF_B0 + i) * 64 + lane];
            }
            const f4 acc = mfma_chain<16>(WB1, b, mk_b1, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(a0r + 16 * wave + 4 * h) = swz4(dlrelu4(a0v, acc), fhi);
        }
        STAMP(12);
        __syncthreads();
        STAMP(13);

        // ================= bL0 + Adam
        if (wave < 2) {
            float b[10];
            load_b<10>(a0r, hs, 0, b);
            f4 z4 = {0.f, 0.f, 0.f, 0.f}, zt4 = z4, m4 = z4, v4 = z4;
            if (zvalid) {
                z4 = swz4(*(const f4*)(zs + zd), fhi);
                zt4 = *(const f4*)(lds + K_ZT + f16 * S_Z + zd);
                m4 = *(const f4*)(lds + K_ADM + f16 * S_Z + zd);
                v4 = *(const f4*)(lds + K_ADV + f16 * S_Z + zd);
            }
            const f4 gz = mfma_chain_dense<10>(WB0, b, f4{0.f, 0.f, 0.f, 0.f});
            const f4 g = gz + a.ctmp * (z4 - zt4);
            if (DBG_DUMP && a.dbg && iter == 0 && zvalid && blk0 + f16 < nB) *(f4*)(a.dbg + (size_t)(blk0 + f16) * DBG_STRIDE + DBG_GZ + zd) = g;
            if (zvalid && last) *(f4*)(lds + K_ZPRE + f16 * S_Z + zd) = z4;
            m4 = m4 + a.one_m_b1 * (g - m4);
            v4 = v4 * a.beta2 + a.one_m_b2 * (g * g);
            const f4 den = f4{__builtin_amdgcn_sqrtf(v4.x), __builtin_amdgcn_sqrtf(v4.y), __builtin_amdgcn_sqrtf(v4.z),
                              __builtin_amdgcn_sqrtf(v4.w)} * rbc2s + a.eps;
            z4 = z4 - step * (m4 * f4{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                      __builtin_amdgcn_rcpf(den.w)});
            if (zvalid) {
                *(f4*)(zs + zd) = swz4(z4, fhi);
                *(f4*)(lds + K_ADM + f16 * S_Z + zd) = m4;
                *(f4*)(lds + K_ADV + f16 * S_Z + zd) = v4;
            }
        }
        STAMP(14);
        __syncthreads();
        STAMP(15);
    }
#ifdef DP_PROFILE
    if (a.dbg && tid == 0) {
        unsigned long long* o = (unsigned long long*)a.dbg + (size_t)blockIdx.x * 20;
        for (int i = 0; i < 20; ++i) o[i] = prof[i];
    }
#endif

    // ================= epilogue: outputs of the LAST forward pass, rebuilt from what it left in LDS
    if (!optimise) {
        if (zvalid) *(f4*)(lds + K_ZPRE + f16 * S_Z + zd) = swz4(*(const f4*)(zs + zd), fhi);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int pf = pf0 + 2 * r;
        const int gfp = blk0 + pf;
        if (!(gfp < nB && pf < FR)) continue;
        const float* bone = lds + K_BONE + pf * 128;
        const float* lpb = lds + K_LP + pf * (2 * K4_R);
        const float* qdb = lds + K_QD + pf * QD_S;
        const f4 y4 = *(const f4*)(yp + pf * S_Y + 4 * sq) + *(const f4*)(yp + (FPB + pf) * S_Y + 4 * sq);
        const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
        const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
        const float inv = has_quat ? __builtin_amdgcn_rsqf(rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z) : 0.f;
        const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
        M3 M = quat_to_mat(q);
        if (is_root) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        const f4 qwv = *(const f4*)(qdb);
        const f4 dv = *(const f4*)(qdb + 4);
        const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
        const M3 R0 = quat_to_mat(qw);
        if (is_joint) {
            if (a.pose) {
                float* o = a.pose + (size_t)gfp * 88 + 4 * it_id;
                o[0] = (q.w - mu.x) / sd.x; o[1] = (q.x - mu.y) / sd.y;
                o[2] = (q.y - mu.z) / sd.z; o[3] = (q.z - mu.w) / sd.w;
            }
            if (a.pos) {
                V3 pr = {dv.x, dv.y, dv.z};
                for (int i = 0; i < MAX_PATH; ++i) {
                    const unsigned k = (i < 6) ? ((plo >> (5 * i)) & 31u) : (phi & 31u);
                    const f4 b = *(const f4*)(bone + k * 4);
                    pr.x += b.x; pr.y += b.y; pr.z += b.z;
                }
                const V3 pw = mat_vec(R0, pr);
                float* o = a.pos + ((size_t)gfp * NJ + it_id) * 3;
                o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
            }
            if (a.rot) {
                const M3 G = mat_mat(R0, M);
                float* o = a.rot + ((size_t)gfp * NJ + it_id) * 9;
                o[0] = G.m00; o[1] = G.m01; o[2] = G.m02; o[3] = G.m10; o[4] = G.m11; o[5] = G.m12; o[6] = G.m20; o[7] = G.m21; o[8] = G.m22;
            }
        }
        if (is_root) {
            if (a.world_rot) { float* o = a.world_rot + (size_t)gfp * 4; o[0] = qw.w; o[1] = qw.x; o[2] = qw.y; o[3] = qw.z; }
            if (optimise) {
                float lsum_p = 0.f, lsum_r = 0.f, lt = 0.f;
                const int E = __popc(__float_as_uint(qdb[24]));
                for (int e0 = 0; e0 < E; ++e0) { const f2 l = *(const f2*)(lpb + e0 * 2); lsum_p += l.x; lsum_r += l.y; }
                const float* zrow = lds + K_ZPRE + pf * S_Z;
                const float* ztrow = lds + K_ZT + pf * S_Z;
                for (int k = 0; k < LAT; k += 4) {
                    const f4 zz = *(const f4*)(zrow + k), zt = *(const f4*)(ztrow + k);
                    if (a.z_pre) *(f4*)(a.z_pre + (size_t)gfp * LAT + k) = zz;
                    const f4 dz = zz - zt;
                    lt += dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
                }
                if (a.loss) {
                    a.loss[(size_t)gfp * 3 + 0] = lsum_p;
                    a.loss[(size_t)gfp * 3 + 1] = lsum_r;
                    a.loss[(size_t)gfp * 3 + 2] = lt * a.lam_tmp * (1.f / 24.f);
                }
            }
        }
        if (is_disp) {
            if (a.disp) { float* o = a.disp + (size_t)gfp * 3; o[0] = rq.w; o[1] = rq.x; o[2] = rq.y; }
            if (a.world_disp) {
                const V3 wd = mat_vec(R0, V3{rq.w, rq.x, rq.y});
                float* o = a.world_disp + (size_t)gfp * 3; o[0] = wd.x; o[1] = wd.y; o[2] = wd.z;
            }
        }
    }

    if (optimise && zvalid && blk0 + f16 < nB) {
        if (a.z) *(f4*)(a.z + (size_t)(blk0 + f16) * LAT + zd) = swz4(*(const f4*)(zs + zd), fhi);
        if (a.iters && wave == 0 && h == 0) a.iters[blk0 + f16] = a.n_iter;
    }
}

extern "C" hipError_t dp_launch_optimize4(const KArgs* args, int rounds, hipStream_t stream)
{
    const int fr = 8 * rounds;
    const int grid = (args->n_frames + fr - 1) / fr;
    if (rounds == 2)
        hipLaunchKernelGGL(dp_optimize_kernel4<2>, dim3(grid), dim3(K4_NT), 0, stream, *args);
    else
        hipLaunchKernelGGL(dp_optimize_kernel4<1>, dim3(grid), dim3(K4_NT), 0, stream, *args);
    return hipGetLastError();
}

extern "C" int dp_kernel4_lds_bytes(void) { return K_TOTAL * 4; }
extern "C" int dp_kernel4_max_trackers(void) { return K4_R; }

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
#include "dp_p3.h"

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

DEV FrameRows<K4_R> rows(float* lds, int pf)
{
    return {lds + K_BONE + pf * 128, lds + K_GPC + pf * (4 * K4_R), lds + K_CQ + pf * (4 * K4_R),
            lds + K_LP + pf * (2 * K4_R), lds + K_QD + pf * QD_S, lds + K_TRK + pf * (16 * K4_R)};
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
    stage_item_tables(a, lds + K_ITEM, lds + K_BIAS, tid);

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

    // ---- P3 per-lane identity; per round one packed tracker word (dp_p3.h)
    const ItemId id = load_item(icg);
    unsigned pk_[ROUNDS];
    int Emax_[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        pk_[r] = p3_setup<K4_R>(a, icg, id, lane, it_id, min(blk0 + pf0 + 2 * r, nB - 1), optimise, rows(lds, pf0 + 2 * r), Emax_[r]);

    f4 a0v = {0.f, 0.f, 0.f, 0.f}, a1v = a0v;
    __syncthreads();
    Prof prof;
    prof.start();

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

        // ================= P3 (wave-private rows; dp_p3.h), ROUNDS x 2 frames per wave
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int pf = pf0 + 2 * r;
            if (blk0 + (ROUNDS == 2 ? 4 : 2) * wave + 2 * r < nB) // (uniform) at least one of the round's two frames exists
                p3_round<K4_R>(a, id, pk_[r], Emax_[r], icl, yp + pf * S_Y, lds + K_GY + pf * S_Y, pf >= 8, rows(lds, pf), iter,
                           blk0 + pf, blk0 + pf < nB && pf < FR, prof, []() {});
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
    prof.store(a.dbg, tid, blockIdx.x);

    // ================= epilogue: outputs of the LAST forward pass, rebuilt from what it left in LDS
    if (!optimise) {
        if (zvalid) *(f4*)(lds + K_ZPRE + f16 * S_Z + zd) = swz4(*(const f4*)(zs + zd), fhi);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int pf = pf0 + 2 * r;
        if (blk0 + pf < nB && pf < FR)
            p3_outputs<K4_R>(a, id, icl, yp + pf * S_Y, rows(lds, pf), it_id, blk0 + pf, optimise, nullptr, lds + K_ZPRE + pf * S_Z,
                             lds + K_ZT + pf * S_Z);
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

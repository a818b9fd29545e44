// dp_kernel.hip -- the fused latent-optimisation kernel for gfx950 (MI355X).
//
// One launch runs ALL iterations of decode -> FK -> loss -> backward -> Adam for every frame of
// the batch (reference: DragPose.run's while loop, python/src/drag_pose.py:296-355).
//
// Geometry: workgroup = 512 threads (8 waves) = 16 frames; grid = ceil(B/16).
//   * Decoder forward/backward: six products per iteration on v_mfma_f32_16x16x4_f32.  Output tile
//     t (16 channels x 16 frames) belongs to wave t -- one matrix wave per SIMD -- which runs two
//     interleaved accumulator chains (even / odd K steps), adds bias + LeakyReLU (forward) or
//     multiplies by the LeakyReLU derivative it still holds in registers (backward) and writes ONE
//     plane to LDS; the next product reads its B operand from there, one float per MFMA step.
//     Forward weights live in VGPRs for the whole kernel, backward weights in LDS (fetched ahead
//     of the barrier that precedes their use).  Structurally zero (tile, step) blocks of the
//     folded matrices are skipped through per-wave step masks.
//   * Kinematics phase (P3): wave w owns frames 2w,2w+1, 32 lanes per frame, one lane per joint
//     (+1 for the root displacement, +1 per extra child bone); cross-joint traffic (bones, tracker
//     gradients) goes through wave-private LDS rows, so P3 needs no workgroup barrier inside.
//   * bL0 and Adam are fused: the two waves that produce dL/dz own z, m, v in registers.
// 7 workgroup barriers per iteration, no global memory traffic inside the loop.
#include "dp_p3.h"

// ------------------------------------------------------------------------------------------------
// LDS map (floats)
constexpr int L_Z = 0;                             // zs [16][S_Z]      latent, B operand of L0
constexpr int L_A0 = L_Z + FPB * S_Z;              // a0 [16][S_A0]     (aliased by d0)
constexpr int L_A1 = L_A0 + FPB * S_A0;            // a1 [16][S_A1]     (aliased by d1)
constexpr int L_GY = L_A1 + FPB * S_A1;            // gy [16][S_Y]      dL/dy quads (26 used)
constexpr int L_Y = L_GY + FPB * S_Y;              // y  [2][16][S_Y]   plane 1: 2nd K-half of tiles 4,5 (kept for the epilogue)
constexpr int L_ZERO0 = L_Y + FPB * S_Y;           // ---- everything from here on is zeroed at start (incl. y plane 1)
constexpr int L_BONE = L_ZERO0 + FPB * S_Y;        // bone[16][32][4]
constexpr int R8 = 24;                             // tracker capacity per frame (>= NJ: every joint may carry one)
constexpr int L_GPC = L_BONE + FPB * 32 * 4;       // gpc [16][R8][4]  tracker position gradients by rank
constexpr int L_CQ = L_GPC + FPB * R8 * 4;         // cq  [16][R8][4]  tracker contributions to d/d(qw)
constexpr int L_LP = L_CQ + FPB * R8 * 4;          // lp  [16][R8][2]  tracker loss terms
constexpr int L_QD = L_LP + FPB * R8 * 2;          // qd  [16][QD_S]   (dp_p3.h)
constexpr int L_TRK = L_QD + FPB * QD_S;           // tracker inputs by rank: [16 frames][4 quads][R8][4]
constexpr int L_ITEM = L_TRK + FPB * R8 * 16;      // item constants, SoA: sd[32][4] | mu[32][4] | child offset[32][4]
constexpr int L_BIAS = L_ITEM + 3 * 32 * 4;        // bias rows of L0 (48) and L1 (64), padded to 64 each
constexpr int L_ZT = L_BIAS + 128;                 // z_tgt [16][S_Z]
constexpr int WB_STRIDE = 60;                      // per lane: bL2[26] pad2 | bL1[16] | bL0[10] | pad6
constexpr int WB_B2 = 0, WB_B1 = 28, WB_B0 = 44;
constexpr int L_ZPRE = L_ZT + FPB * S_Z;           // latent of the last forward pass [16][S_Z]
constexpr int L_ZFIN = L_ZPRE + FPB * S_Z;          // early stop: latent after a frame's last step [16][S_Z]
constexpr int L_ES = L_ZFIN + FPB * S_Z;           // early stop: per frame {active, stop_now, iters, next_active, loss_pos, loss_rot, loss_tmp, -}
constexpr int L_LT = L_ES + FPB * 8;               // early stop: partial sums of |z - z_tgt|^2 from the latent lanes [16][8]
constexpr int L_ADM = L_LT + FPB * 8;            // Adam m [16][S_Z]   (zeroed at start: lies below L_ITEM? no -> zeroed explicitly)
constexpr int L_ADV = L_ADM + FPB * S_Z;           // Adam v [16][S_Z]
constexpr int L_WB = L_ADV + FPB * S_Z;            // wb[4 matrix waves][64 lanes][WB_STRIDE] backward weights
constexpr int L_TOTAL = L_WB + 4 * 64 * WB_STRIDE;
static_assert(L_TOTAL * 4 <= 160 * 1024, "LDS budget");
static_assert((WB_STRIDE % 4) == 0 && ((WB_STRIDE / 4) & 1) == 1, "16-byte rows, odd stride/4 (conflict-free b128)");

// ------------------------------------------------------------------------------------------------
template <bool EARLY>
__global__ __launch_bounds__(NTHREADS, 2) void dp_optimize_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[L_TOTAL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f16 = lane & 15, h = lane >> 4;   // MFMA roles: frame column, K / row group
    const bool fhi = f16 >= 8;                  // swizzled row (see swz4)
    const int hs = h ^ (fhi ? 2 : 0);           // K-group as laid out in my frame's row
    const int pf = 2 * wave + (lane >> 5);      // P3 roles: frame (row of every LDS buffer)
    const int it_id = lane & 31;                //           item id
    const int blk0 = blockIdx.x * FPB;
    const int nB = a.n_frames;
    const bool optimise = (a.mode == 0);

    float* zs = lds + L_Z + f16 * S_Z;
    float* a0r = lds + L_A0 + f16 * S_A0;       // my frame's row of a0 / d0
    float* a1r = lds + L_A1 + f16 * S_A1;       //                   a1 / d1
    float* yr = lds + L_Y + f16 * S_Y;          //                   y plane 0
    const float* gyr = lds + L_GY + f16 * S_Y;  //                   gy
    float* yp = lds + L_Y;
    const FrameRows<R8> fr = {lds + L_BONE + pf * 128, lds + L_GPC + pf * (4 * R8), lds + L_CQ + pf * (4 * R8),
                              lds + L_LP + pf * (2 * R8), lds + L_QD + pf * QD_S, lds + L_TRK + pf * (16 * R8)};
    const ItemConst* icg = a.items + it_id;                       // global copy: read once, before the loop
    const float* icl = lds + L_ITEM + 4 * it_id;                  // LDS copy (SoA): sd, +128: mu, +256: child offset
    const float* wbl = lds + L_WB + ((wave & 3) * 64 + lane) * WB_STRIDE;

    // ---- zero the scratch part of the LDS (incl. plane 1 of y), copy the small tables
    for (int i = tid; i < L_ITEM - L_ZERO0; i += NTHREADS) lds[L_ZERO0 + i] = 0.f;
    stage_item_tables(a, lds + L_ITEM, lds + L_BIAS, tid);
    __syncthreads();

    // ---- loop-invariant MFMA A operands: forward weights stay in VGPRs for the whole kernel, the
    //      backward weights of the four matrix waves are parked in LDS (lane-major)
    float W[W_FWD];
#pragma unroll
    for (int i = 0; i < W_FWD; ++i) W[i] = a.wfrag[(wave * W_REGS + i) * 64 + lane];
    if (wave < 4) {
        float* wo = lds + L_WB + (wave * 64 + lane) * WB_STRIDE;
        for (int i = 0; i < 26; ++i) wo[WB_B2 + i] = a.wfrag[(wave * W_REGS + W_OFF_B2 + i) * 64 + lane];
        for (int i = 0; i < 16; ++i) wo[WB_B1 + i] = a.wfrag[(wave * W_REGS + W_OFF_B1 + i) * 64 + lane];
        for (int i = 0; i < 10; ++i) wo[WB_B0 + i] = a.wfrag[(wave * W_REGS + W_OFF_B0 + i) * 64 + lane];
    }
    const unsigned mk_l1 = a.smask[wave][G_L1], mk_l2 = a.smask[wave][G_L2], mk_b2 = a.smask[wave][G_B2],
                   mk_b1 = a.smask[wave][G_B1];

    // ---- latent, Adam state: waves 0,1 = the two output tiles of bL0; lane (f16,h) owns dims zd..zd+3
    const int zd = 16 * (wave & 1) + 4 * h;
    const bool zvalid = wave < 2 && zd < LAT;
    if (zvalid) { // z, z_tgt, m, v live in LDS rows (z doubles as the B operand of L0)
        const int gf = min(blk0 + f16, nB - 1);
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        *(f4*)(zs + zd) = swz4(*(const f4*)(a.z0 + (size_t)gf * LAT + zd), fhi);
        *(f4*)(lds + L_ZT + f16 * S_Z + zd) = optimise ? *(const f4*)(a.z_tgt + (size_t)gf * LAT + zd) : zero4;
        *(f4*)(lds + L_ADM + f16 * S_Z + zd) = zero4;
        *(f4*)(lds + L_ADV + f16 * S_Z + zd) = zero4;
        if (EARLY) {
            const f4 dz = *(const f4*)(a.z0 + (size_t)gf * LAT + zd) - *(const f4*)(a.z_tgt + (size_t)gf * LAT + zd);
            lds[L_LT + f16 * 8 + 4 * (wave & 1) + h] = dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
        }
    }
    if (EARLY && tid < FPB) { // per-frame early-stop record; frames of waves that sit P3 out count as stopped
        const float live = (blk0 + (tid & ~1) < nB) ? 1.f : 0.f;
        *(f4*)(lds + L_ES + tid * 8) = f4{live, 0.f, 0.f, live};
    }

    // ---- P3 per-lane identity (a few integers stay in registers, the float constants are re-read from
    //      LDS every iteration to keep the register budget for the kinematics temporaries)
    const ItemId id = load_item(icg);
    const int gfp = blk0 + pf;
    const bool fvalid = gfp < nB;
    const bool wave_live = blk0 + 2 * wave < nB; // (uniform) at least one of this wave's two frames exists
    int Emax;
    const unsigned pk = p3_setup<R8>(a, icg, id, lane, it_id, min(gfp, nB - 1), optimise, fr, Emax);

    f4 a0v = {0.f, 0.f, 0.f, 0.f}, a1v = a0v; // my tile of a0 / a1 (kept for the LeakyReLU derivative)
    float es_prev = 10000000.f; // early stop, root lane: previous total loss (drag_pose.py:297), active flag, count
    bool es_act = true;
    int es_iters = 0;
    __syncthreads();
    Prof prof;
    prof.start();

    for (int iter = 0; iter < a.n_iter; ++iter) {
        const bool last = (iter == a.n_iter - 1);
        const float step = a.tab.step[iter], rbc2s = a.tab.bc2s[iter]; // scalar loads, issued a whole iteration ahead of their use

        // ================= L0: a0 = lrelu(A0 z + c0)   (24 -> 40), tiles on waves 0..2
        if (wave < 3) {
            float b[6];
            load_b<6>(zs, hs, 0, b);
            const f4 bias = *(const f4*)(lds + L_BIAS + 16 * wave + 4 * h);
            float w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = W[W_OFF_L0 + i];
            a0v = lrelu4(mfma_chain_dense<6>(w, b, bias));
            *(f4*)(a0r + 16 * wave + 4 * h) = swz4(a0v, fhi);
        }
        STAMP(0);
        __syncthreads();
        STAMP(1);

        // ================= L1: a1 = lrelu(A1 a0 + b1)  (40 -> 60), tiles on waves 0..3
        if (wave < 4) {
            float b[10];
            load_b<10>(a0r, hs, 0, b);
            const f4 bias = *(const f4*)(lds + L_BIAS + 64 + 16 * wave + 4 * h);
            float w[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) w[i] = W[W_OFF_L1 + i];
            a1v = lrelu4(mfma_chain<10>(w, b, mk_l1, bias));
            *(f4*)(a1r + 16 * wave + 4 * h) = swz4(a1v, fhi);
        }
        STAMP(2);
        __syncthreads();
        STAMP(3);

        // ================= L2: y = A2 a1 + b2  (60 -> 92; b2 rides on the constant-1 column 60)
        // tiles 0..3 on waves 0..3 (16 steps), tiles 4,5 in two K-halves on waves 4..7 (8 steps each)
        if (wave < 4) {
            float b[16], w[16];
            load_b<16>(a1r, hs, 0, b);
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = W[W_OFF_L2 + i];
            const f4 acc = mfma_chain<16>(w, b, mk_l2, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(yr + 16 * wave + 4 * h) = acc;
        } else {
            const int half = l2_half(wave);
            float b[8], w[8];
            load_b<8>(a1r, hs, L2_HALF_STEPS * half, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = W[W_OFF_L2 + i];
            const f4 acc = mfma_chain<8>(w, b, mk_l2, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(yr + half * FPB * S_Y + 16 * (L2_SPLIT_TILE0 + (wave & 1)) + 4 * h) = acc;
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);

        // ================= P3: normalise, FK, loss, backward to gy   (wave-private rows; dp_p3.h)
        // (waves whose two frames both lie beyond the batch sit the phase out: their rows of dL/dy stay zero, and a
        //  short batch -- one sequence -- does not pay for the SIMD partner's instruction stream)
        if (wave_live) p3_round<R8>(a, id, pk, Emax, icl, yp + pf * S_Y, lds + L_GY + pf * S_Y, pf >= 8, fr, iter, gfp, fvalid, prof, [&]() {
            if (EARLY) { // root lane: per-frame stop test of the reference's while loop (drag_pose.py:300-304,351-355)
                const int E = __popc(__float_as_uint(fr.qd[24]));
                float lp = 0.f, lr = 0.f;
                for (int e0 = 0; e0 < E; ++e0) { const f2 l = *(const f2*)(fr.lp + e0 * 2); lp += l.x; lr += l.y; }
                const f4 p0 = *(const f4*)(lds + L_LT + pf * 8);
                const f2 p1 = *(const f2*)(lds + L_LT + pf * 8 + 4);
                const float lt = (((p0.x + p0.y) + (p0.z + p0.w)) + (p1.x + p1.y)) * a.lam_tmp * (1.f / 24.f);
                const float tot = (lp + lr) + lt;
                const bool cont = (lp > a.stop_eps_pos || lr > a.stop_eps_rot) && (es_prev - tot > a.min_loss_incr) && !last;
                float* es = lds + L_ES + pf * 8;
                if (es_act) {
                    es_prev = tot;
                    ++es_iters;
                    *(f4*)(es + 4) = f4{lp, lr, lt, 0.f}; // losses of this frame's last executed iteration
                }
                *(f4*)(es) = f4{es_act ? 1.f : 0.f, (es_act && !cont) ? 1.f : 0.f, (float)es_iters, (es_act && cont) ? 1.f : 0.f};
                es_act = es_act && cont;
            }
        });
        if (!optimise) break; // forward-only launch (uniform)
        float WB2[26]; // bL2 weights: issued before the barrier, landed by the time it opens
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f4 wv = *(const f4*)(wbl + WB_B2 + 4 * i);
                WB2[4 * i] = wv.x; WB2[4 * i + 1] = wv.y; WB2[4 * i + 2] = wv.z; WB2[4 * i + 3] = wv.w;
            }
            const f2 wv = *(const f2*)(wbl + WB_B2 + 24);
            WB2[24] = wv.x; WB2[25] = wv.y;
        }
        STAMP(8);
        __syncthreads();
        STAMP(9);

        // ================= bL2: d1 = (A2^T gy) * lrelu'(a1)  (92 + virtual quads -> 60), tiles on waves 0..3
        float WB1[16];
        if (wave < 4) {
            float b[26];
            load_b<26>(gyr, hs, 0, b);
            const f4 acc = mfma_chain<26>(WB2, b, mk_b2, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(a1r + 16 * wave + 4 * h) = swz4(dlrelu4(a1v, acc), fhi); // d1 aliases a1
            if (wave < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f4 wv = *(const f4*)(wbl + WB_B1 + 4 * i);
                    WB1[4 * i] = wv.x; WB1[4 * i + 1] = wv.y; WB1[4 * i + 2] = wv.z; WB1[4 * i + 3] = wv.w;
                }
            }
        }
        STAMP(10);
        __syncthreads();
        STAMP(11);

        // ================= bL1: d0 = (A1^T d1) * lrelu'(a0)  (60 -> 40), tiles on waves 0..2
        float WB0[10];
        if (wave < 3) {
            float b[16];
            load_b<16>(a1r, hs, 0, b);
            const f4 acc = mfma_chain<16>(WB1, b, mk_b1, f4{0.f, 0.f, 0.f, 0.f});
            *(f4*)(a0r + 16 * wave + 4 * h) = swz4(dlrelu4(a0v, acc), fhi); // d0 aliases a0
            if (wave < 2) {
                const f4 w0 = *(const f4*)(wbl + WB_B0), w1 = *(const f4*)(wbl + WB_B0 + 4);
                const f2 w2 = *(const f2*)(wbl + WB_B0 + 8);
                WB0[0] = w0.x; WB0[1] = w0.y; WB0[2] = w0.z; WB0[3] = w0.w; WB0[4] = w1.x; WB0[5] = w1.y; WB0[6] = w1.z; WB0[7] = w1.w;
                WB0[8] = w2.x; WB0[9] = w2.y;
            }
        }
        STAMP(12);
        __syncthreads();
        STAMP(13);

        // ================= bL0 + Adam: gz = A0^T d0 (40 -> 24) on waves 0,1, which own z, m, v of their
        // 16 / 8 latent dims (torch.optim.Adam, single-tensor form; m, v start at 0, t = iter+1)
        if (wave < 2) {
            float b[10];
            load_b<10>(a0r, hs, 0, b);
            // Adam state of my 4 latent dims: independent of the product, fetched while it runs
            f4 z4 = {0.f, 0.f, 0.f, 0.f}, zt4 = z4, m4 = z4, v4 = z4;
            if (zvalid) {
                z4 = swz4(*(const f4*)(zs + zd), fhi);
                zt4 = *(const f4*)(lds + L_ZT + f16 * S_Z + zd);
                m4 = *(const f4*)(lds + L_ADM + f16 * S_Z + zd);
                v4 = *(const f4*)(lds + L_ADV + f16 * S_Z + zd);
            }
            bool f_act = true, f_stop = false;
            if (EARLY) {
                const f2 fl = *(const f2*)(lds + L_ES + f16 * 8);
                f_act = fl.x != 0.f;
                f_stop = fl.y != 0.f;
            }
            const f4 gz = mfma_chain_dense<10>(WB0, b, f4{0.f, 0.f, 0.f, 0.f});
            const f4 g = gz + a.ctmp * (z4 - zt4);
            if (DBG_DUMP && a.dbg && iter == 0 && zvalid && blk0 + f16 < nB) *(f4*)(a.dbg + (size_t)(blk0 + f16) * DBG_STRIDE + DBG_GZ + zd) = g;
            if (zvalid && f_act && (EARLY || last)) *(f4*)(lds + L_ZPRE + f16 * S_Z + zd) = z4; // latent of this forward pass
            m4 = m4 + a.one_m_b1 * (g - m4);
            v4 = v4 * a.beta2 + a.one_m_b2 * (g * g);
            const f4 den = f4{__builtin_amdgcn_sqrtf(v4.x), __builtin_amdgcn_sqrtf(v4.y), __builtin_amdgcn_sqrtf(v4.z),
                              __builtin_amdgcn_sqrtf(v4.w)} * rbc2s + a.eps;
            z4 = z4 - step * (m4 * f4{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                      __builtin_amdgcn_rcpf(den.w)});
            if (zvalid && f_act) {
                if (EARLY && f_stop) {
                    // this frame's loop ends here: keep the stepped latent aside and leave zs at the pre-step latent, so
                    // that the (wasted) forward passes of the remaining iterations reproduce this one bit for bit
                    *(f4*)(lds + L_ZFIN + f16 * S_Z + zd) = z4;
                } else {
                    *(f4*)(zs + zd) = swz4(z4, fhi);
                    *(f4*)(lds + L_ADM + f16 * S_Z + zd) = m4;
                    *(f4*)(lds + L_ADV + f16 * S_Z + zd) = v4;
                    if (EARLY) {
                        const f4 dz = z4 - zt4;
                        lds[L_LT + f16 * 8 + 4 * (wave & 1) + h] = dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
                    }
                }
            }
        }
        STAMP(14);
        __syncthreads();
        STAMP(15);
        if (EARLY) { // every frame of the workgroup has stopped (same LDS words for all waves: uniform)
            if (__ballot(lds[L_ES + (lane & 15) * 8 + 3] != 0.f) == 0ull) break;
        }
    }
    prof.store(a.dbg, tid, blockIdx.x);

    // ================= epilogue: outputs of the LAST forward pass, rebuilt from what it left in LDS
    // (y planes, qw / d, bones, tracker loss terms, the pre-step latent) -- kept out of the hot loop
    if (!optimise) { // forward-only launch: the one forward pass ran on z itself
        if (zvalid) *(f4*)(lds + L_ZPRE + f16 * S_Z + zd) = swz4(*(const f4*)(zs + zd), fhi);
    }
    __syncthreads();
    if (fvalid)
        p3_outputs<R8>(a, id, icl, yp + pf * S_Y, fr, it_id, gfp, optimise, EARLY ? lds + L_ES + pf * 8 : nullptr,
                       lds + L_ZPRE + pf * S_Z, lds + L_ZT + pf * S_Z);

    if (optimise && zvalid && blk0 + f16 < nB) { // own writes, same lane
        if (a.z) *(f4*)(a.z + (size_t)(blk0 + f16) * LAT + zd) = EARLY ? *(const f4*)(lds + L_ZFIN + f16 * S_Z + zd) : swz4(*(const f4*)(zs + zd), fhi);
        if (!EARLY && a.iters && wave == 0 && h == 0) a.iters[blk0 + f16] = a.n_iter;
    }
}

extern "C" hipError_t dp_launch_optimize(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + FPB - 1) / FPB;
    if (args->early_stop)
        hipLaunchKernelGGL(dp_optimize_kernel<true>, dim3(grid), dim3(NTHREADS), 0, stream, *args);
    else
        hipLaunchKernelGGL(dp_optimize_kernel<false>, dim3(grid), dim3(NTHREADS), 0, stream, *args);
    return hipGetLastError();
}

extern "C" int dp_kernel_lds_bytes(void) { return L_TOTAL * 4; }

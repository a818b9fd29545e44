// dp_temporal.hip -- the temporal predictor on the device (reference: Temporal, python/src/temporal_transformer.py:7-77,
// positional_encoding.py:6-32; the temporal target block of DragPose.run, drag_pose.py:234-294).
//
// A torch.nn.Transformer (post-norm layers, ReLU, final encoder / decoder LayerNorm, no masks, dropout off) of width 48
// over at most 32 tokens: far too small for one GPU to be busy with one sequence, so the unit of work is ONE WORKGROUP PER
// SEQUENCE (or per two when there are many; a TEAM of up to eight per sequence when there are few: below) that runs the whole block -- token
// assembly from the history buffers, the encoder once, window / step + 1
// autoregressive decoder calls (the reference passes no target mask, so every call recomputes all target positions),
// de-normalisation and the step-hold "lerp" -- with every activation in LDS and the weights (7.7 MB at the reference's
// size) streamed from L2.
//   * the small linear layers (in_proj, out_proj: 6 % of the FLOPs) on v_mfma_f32_16x16x4_f32, fp32: M = token, N = output channel, K = input
//     channel; the weights keep torch's [out][in] layout (rows padded to a multiple of four floats), a lane reads a quarter of a row and of its
//     token's activations in 16-byte words (lin);
//   * the feed-forward block (48 -> F -> 48, 94 % of the FLOPs and of the weight bytes) on v_mfma_f32_16x16x32_bf16 in SPLIT PRECISION -- every
//     fp32 operand the exact sum of three bf16 terms, six term products per K-block, fp32 accumulation: the arithmetic of an fp32 product to its
//     last bit or two (dp_w16.h) -- per tile of 32 hidden units: H^T = W1 X^T, bias + ReLU in registers, then OUT += H W2^T with the first
//     product's result AS the second's A operand (a lane's eight hidden units are its eight K-slots by the host's packing of W2: no LDS, no
//     transposition).  The tiles are dealt to the 8 waves, a tile's image (weights already split: 21 + 2 sixteen-byte words per lane) passes
//     through the registers in three parts, the waves' partial outputs are summed through LDS (see the comment above ffn_p1).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/dragposer.h"

namespace {

constexpr int D = DP_TEMPORAL_D_MODEL, NHD = DP_TEMPORAL_HEADS, HD = D / NHD, LAT = 24;
constexpr int MAXT = DP_TEMPORAL_MAX_TOKENS, MAXL = DP_TEMPORAL_MAX_LAYERS;
constexpr int NT = 512, NWV = NT / 64;              // threads / waves per workgroup (two waves per SIMD)
constexpr int LN_MAX = (MAXL * 4 + MAXL * 6 + 4) * D;  // floats of the LayerNorm block at the largest architecture
constexpr int MAX_IN = 36;                          // 24 + 3 + 8 heights, padded to a multiple of 4 (K-steps)
constexpr int FT = 32;                              // hidden units per feed-forward tile (the K of one v_mfma_f32_16x16x32_bf16)
constexpr int FFN_IMG_V = 12 + 9 + 2;               // 16-byte words per lane of a tile's image: W1 [2 M-tiles][2 K-blocks][3 terms], W2 [3 column tiles][3 terms], bias1 [2]
constexpr int FFN_TILE_FLOATS = FFN_IMG_V * 64 * 4; // ... in 32-bit words
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

struct TLayer { // offsets (in floats) into the device weight buffer
    int sa_in_wT, sa_in_b, sa_out_wT, sa_out_b, ca_in_wT, ca_in_b, ca_out_wT, ca_out_b;
    int ffn_pack, lin2_b, n1w, n1b, n2w, n2b, n3w, n3b; // ffn_pack: [ceil(F / 16)][7][64 lanes][4] (dp_temporal_create)
};
struct TArgs {
    const float* w;
    int enc_tab, dec_tab; // offsets of the TLayer tables inside the weight buffer (a kernel-argument array indexed by the
                          // layer loop would be copied into registers: 288 SGPRs)
    int n_enc, n_dec, ff, n_in, nh, max_len, step;
    int ipe_wT, ipe_b, ipd_wT, ipd_b, op_wT, op_b, pe, encn_w, encn_b, decn_w, decn_b, mean, stdv;
    int ln0, ln_len; // all LayerNorm rows (the layers' and the two final ones) are one block: the TEAM kernel keeps it in LDS
    // per call
    const float *latent_buf, *disp_buf, *heights_buf;
    float* target;
    int H, n_seq, window;
    // a TEAM of G workgroups per sequence (few sequences: latency; below): the exchange area of the handle, the tag base of this launch
    float* xch;
    unsigned* epochs; // one word per team: the tag of the team's last exchange (device-resident, so that a captured launch can be replayed)
    int* tstatus;     // the handle's status word in DEVICE memory: what every team launch checks at entry and while it waits
    int* hstatus;     // ... and its mirror in page-locked HOST memory: what dp_temporal_status / the next dp_temporal_predict read without a synchronise
    int G;
    int poll_limit;   // re-reads of a granule set before a member gives up (XCH_POLL_LIMIT; the debug hook shortens it)
    int dbg_skip_team, dbg_skip_member; // private test hook: that member of that team never publishes (-1: nobody)
};

#define DEV __device__ __forceinline__
// PAIR (round 6, many sequences): ONE workgroup of 2 NT threads per CU instead of two of NT -- two HALVES, each the NS = 2 workgroup it was (its own
// LDS arrays, its own two sequences, every phase unchanged) -- so that the feed-forward layers of the calls over at most 8 tokens can share their
// weight fetches across all four sequences (ffn: COOP).  Every phase indexes by the thread / wave WITHIN ITS HALF:
DEV int ltid() { return (int)threadIdx.x & (512 - 1); }
DEV int lwave() { return ((int)threadIdx.x >> 6) & (512 / 64 - 1); }

// Diagnostic build (-DDPT_STAMPS, tools/temporal_phases.sh): thread 0 of workgroup 0 records (phase id, s_memtime) at every phase boundary
#ifdef DPT_STAMPS
__device__ unsigned long long g_stamps[4096];
__device__ int g_nstamps;
#define STAMP(id)                                                                                                          \
    do {                                                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_nstamps < 4096) {                                                     \
            g_stamps[g_nstamps++] = ((unsigned long long)(id) << 48) | (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFull); \
        }                                                                                                                  \
    } while (0)
#else
#define STAMP(id) do { } while (0)
#endif

DEV f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Token layout.  The activations of a workgroup are rows of [MAXT][48] arrays, handled in tiles of 16 rows (the M or N
// dimension of an MFMA).  NS = 1: one sequence, its T <= 32 tokens in rows 0 .. T-1 (the second tile only when T > 16).
// NS = 2: two sequences, sequence s in tile s, T <= 16 tokens each -- every token-wise product then serves both with one
// fetch of its weights.
// Round 5: the rows a sequence gets are a PARAMETER of every phase (R, a power of two; NS >= 2 only): R = 16 is the tile per sequence above; R = 8
// puts two sequences into ONE tile (sequence s in rows 8 s .. 8 s + T - 1) for the decoder calls over at most 8 target tokens -- the first call of
// every block has ONE token per sequence, and a tile of 16 rows per sequence spends 15/16 of its matrix work on nothing.
template <int NS> DEV int rows_per_seq(int T) { return (NS == 2 && T <= 8) ? 8 : 16; } // (derived from the token count wherever it is needed: nothing to keep alive)
template <int NS> DEV int n_ttiles(int T, int R) { return NS == 1 ? (T + 15) >> 4 : (NS * R + 15) >> 4; }
template <int NS> DEV bool row_valid(int g, int T, int R) { return NS == 1 ? g < T : (g & (R - 1)) < T; } // row g = 16 tile + row in tile
template <int NS> DEV int row_pos(int g, int R) { return NS == 1 ? g : g & (R - 1); }                      // its position in its sequence
template <int NS> DEV int row_of(int g, int R, int R2)
{ // the same token in a layout of R2 rows per sequence (R, R2 in {8, 16}: no integer division in the kernel)
    if (NS == 1 || R == R2) return g;
    return R == 8 ? ((g >> 3) << 4) + (g & 7) : ((g >> 4) << 3) + (g & 15);
}

// KS consecutive floats at p (16-byte aligned when KS is a multiple of 4, 8-byte when even): the fewest loads -- the token-wise products of this
// file are bound by the NUMBER of instructions a lone wave issues (profiles/r05_temporal_phases_team.txt), and a lane's K values are consecutive
// in memory by the choice of which K-steps a lane serves (below)
template <int KS> DEV void read_k(float (&v)[KS], const float* p)
{
    if constexpr (KS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f4 t = *(const f4*)(p + 4 * i);
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
        }
    } else if constexpr (KS % 2 == 0) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < KS / 2; ++i) {
            const f2 t = *(const f2*)(p + 2 * i);
            v[2 * i] = t[0]; v[2 * i + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) v[i] = p[i];
    }
}

// out[t][n] = b[n] + sum_k in[t][k] * W[n][k] (+ pe[pos(t)][n])   (n < N, k < K <= 4 KS; W = Linear.weight, rows padded to 4 KS floats)
// One 16 x 16 output tile per wave and turn.  An MFMA K-step takes four k values, one per lane group q; WHICH four is free as long as both
// operands agree: lane group q serves k = KS q + ks in step ks, so its KS values of a row are consecutive -- of the activations in LDS and of
// the weight row in memory: 3 + 3 16-byte loads per tile where 4 ks + q needed 12 + 12 single words, each with its own address arithmetic.
template <int KS, int NS>
DEV void lin(float* out, int ldo, const float* in, int ldi, int T, const float* W, const float* b, int N, int K, const float* pe = nullptr, int Rin = 0)
{ // rows per sequence of `out`: rows_per_seq(T); of `in`: Rin (0: the same)
    const int R = rows_per_seq<NS>(T);
    if (Rin == 0) Rin = R;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // opaque per call: the per-lane weight addresses are recomputed (a few VALU operations) instead
                                   // of being hoisted out of the layer loops into registers the kernel does not have
    const int wave = lwave(), l16 = lane & 15, q = lane >> 4;
    const int ntiles = (N + 15) >> 4, jobs = ntiles * n_ttiles<NS>(T, R);
#pragma unroll 1
    for (int job = wave; job < jobs; job += NWV) {
        const int nt = job % ntiles, tt = job / ntiles, n = 16 * nt + l16;
        const bool rv = row_valid<NS>(16 * tt + l16, T, R);
        const int rin = row_of<NS>(16 * tt + l16, R, Rin);
        float bw[KS], av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[ks] = av[ks] = 0.f;
        if (n < N) read_k<KS>(bw, W + (size_t)n * (4 * KS) + KS * q); // (columns k >= K of a weight row are zero: the packer pads)
        if (rv) read_k<KS>(av, in + rin * ldi + KS * q);
        if (4 * KS > K) { // (the activation row may hold anything beyond K: keep it out of the products -- 0 * NaN)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) if (KS * q + ks >= K) av[ks] = 0.f;
        }
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            acc0 = mfma(av[ks], bw[ks], acc0);
            if (ks + 1 < KS) acc1 = mfma(av[ks + 1], bw[ks + 1], acc1);
        }
        if (n < N) {
            const float bias = b[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = 16 * tt + 4 * q + r;
                if (row_valid<NS>(g, T, R)) out[g * ldo + n] = acc0[r] + acc1[r] + bias + (pe ? pe[row_pos<NS>(g, R) * D + n] : 0.f);
            }
        }
    }
    __syncthreads();
}

// the next target token of every sequence: out_proj of its LAST position (row T - 1 of its tile) -> 24 channels; the
// sequences are the rows of one MFMA tile
template <int NS>
DEV void next_token(float* tok, float* preds, const float* x, int T, int it, bool feed, const float* W, const float* b)
{ // (x in the layout of rows_per_seq(T) rows per sequence; tok keeps 16 rows per sequence whatever the calls' layouts are)
    const int R = rows_per_seq<NS>(T);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = lwave(), l16 = lane & 15, q = lane >> 4;
    if (wave < 2) { // two 16-column tiles of the 24 outputs
        const int n = 16 * wave + l16;
        float bw[D / 4], av[D / 4];
#pragma unroll
        for (int ks = 0; ks < D / 4; ++ks) bw[ks] = av[ks] = 0.f;
        if (n < LAT) read_k<D / 4>(bw, W + n * D + (D / 4) * q); // (lane group q serves k = 12 q + ks: lin)
        if (l16 < NS) read_k<D / 4>(av, x + ((NS == 1 ? 0 : R * l16) + T - 1) * D + (D / 4) * q);
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int ks = 0; ks < D / 4; ks += 2) {
            acc0 = mfma(av[ks], bw[ks], acc0);
            acc1 = mfma(av[ks + 1], bw[ks + 1], acc1);
        }
        if (n < LAT && q == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < NS) { // row r = sequence r
                    const float v = acc0[r] + acc1[r] + b[n];
                    preds[(r * (MAXT + 1) + it) * LAT + n] = v;
                    if (feed) tok[((NS == 1 ? 0 : 16 * r) + T) * LAT + n] = v; // (read by the next call as target position T)
                }
            }
        }
    }
    __syncthreads();
}

// Reductions over the 16 lanes of a DPP row (xor 1, xor 2, half mirror, mirror: every lane ends with the row's result) --
// register-to-register, where __shfl_xor goes through the LDS crossbar (ds_bpermute: ~100 cycles a step)
template <int CTRL> DEV float dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
DEV float row_sum(float v)
{
    v += dpp<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp<0x141>(v); // row_half_mirror
    v += dpp<0x140>(v); // row_mirror
    return v;
}
DEV float row_max(float v)
{
    v = fmaxf(v, dpp<0xB1>(v));
    v = fmaxf(v, dpp<0x4E>(v));
    v = fmaxf(v, dpp<0x141>(v));
    v = fmaxf(v, dpp<0x140>(v));
    return v;
}
DEV float wave_sum(float v)
{ // the four row sums through scalar reads
    v = row_sum(v);
    const int b = __float_as_int(v); // (the builtin reads an int: pass the bits, not the value)
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// x[t] = LayerNorm(x[t] + o[t]) (o may be null), eps 1e-5, biased variance (torch.nn.LayerNorm).  Round 6: FOUR tokens per wave, one per DPP row of
// 16 lanes, three channels per lane (c = l16 + 16 j: a row's lanes read consecutive words) -- both sums are three adds and a register-to-register
// row reduction, no scalar round trip (the one-token-per-wave form went through v_readlane twice per token and took two turns for a 14-token
// encoder pass: 17 LayerNorms stand in every block's latency chain).
template <int NS>
DEV void add_ln(float* x, const float* o, int T, const float* g, const float* b)
{
    const int R = rows_per_seq<NS>(T);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in lin: the per-lane addresses of the two parameter rows are not to be hoisted out of the call loop and spilled)
    const int wave = lwave(), l16 = lane & 15, r4 = lane >> 4;
    float gc[3], bc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gc[k] = g[l16 + 16 * k]; bc[k] = b[l16 + 16 * k]; }
    const int rows = 16 * n_ttiles<NS>(T, R);
#pragma unroll 1
    for (int t0 = 4 * wave; t0 < rows; t0 += 4 * NWV) {
        const int t = t0 + r4;
        if (t < rows && row_valid<NS>(t, T, R)) { // (per DPP row: the reductions below stay inside a row)
            float v[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = x[t * D + l16 + 16 * k] + (o ? o[t * D + l16 + 16 * k] : 0.f);
            const float mean = row_sum((v[0] + v[1]) + v[2]) * (1.f / D);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean;
            const float r = 1.f / sqrtf(row_sum((d0 * d0 + d1 * d1) + d2 * d2) * (1.f / D) + 1e-5f);
            x[t * D + l16] = d0 * r * gc[0] + bc[0];
            x[t * D + l16 + 16] = d1 * r * gc[1] + bc[1];
            x[t * D + l16 + 32] = d2 * r * gc[2] + bc[2];
        }
    }
    STAMP(25);
    __syncthreads();
}

// q | k | v = in_proj(xq | xkv | xkv) in ONE phase: the nine 16-row tiles of the packed in_proj weight [144][48] (x token
// tiles) are dealt to the waves; output channels 0..47 take the query tokens, the rest the key / value tokens.
template <int NS>
DEV void lin_qkv(float* qkv, const float* xq, int Tq, const float* xkv, int Tk, const float* W, const float* b, bool v_only = false)
{ // (q rows in the layout of rows_per_seq(Tq) rows per sequence, k / v rows in that of rows_per_seq(Tk); v_only (uniform): the three value tiles alone -- mha)
    const int Rq = rows_per_seq<NS>(Tq), Rk = rows_per_seq<NS>(Tk);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in lin)
    const int wave = lwave(), l16 = lane & 15, qd = lane >> 4;
    const int ttq = v_only ? 0 : n_ttiles<NS>(Tq, Rq), ttk = n_ttiles<NS>(Tk, Rk), jobs = 3 * ttq + (v_only ? 3 : 6) * ttk;
    // a wave takes its jobs two at a time -- job and job + 8 (with nine jobs only wave 0 has a second one): both jobs' operands are requested
    // before the first product (a weight row comes from cold memory: the second round trip is what the other seven waves would wait for)
#pragma unroll 1
    for (int job0 = wave; job0 < jobs; job0 += 2 * NWV) {
        float bw[2][D / 4], av[2][D / 4];
        int nn[2], tts[2];
        bool isqs[2];
        const bool two = job0 + NWV < jobs; // (uniform)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int job = job0 + u * NWV;
            const bool isq = job < 3 * ttq;
            const int j2 = isq ? job : job - 3 * ttq, nt = isq ? j2 % 3 : v_only ? 6 + j2 % 3 : 3 + j2 % 6, tt = isq ? j2 / 3 : v_only ? j2 / 3 : j2 / 6;
            const float* in = isq ? xq : xkv;
            const int Tj = isq ? Tq : Tk, Rj = isq ? Rq : Rk, n = 16 * nt + l16;
            nn[u] = n; tts[u] = tt; isqs[u] = isq;
            const bool on = u == 0 || two;
#pragma unroll
            for (int ks = 0; ks < D / 4; ++ks) bw[u][ks] = av[u][ks] = 0.f;
            if (on) read_k<D / 4>(bw[u], W + n * D + (D / 4) * qd); // (lane group qd serves k = 12 qd + ks: lin)
            if (on && row_valid<NS>(16 * tt + l16, Tj, Rj)) read_k<D / 4>(av[u], in + (16 * tt + l16) * D + (D / 4) * qd);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int ks = 0; ks < D / 4; ks += 2) {
                acc0 = mfma(av[u][ks], bw[u][ks], acc0);
                acc1 = mfma(av[u][ks + 1], bw[u][ks + 1], acc1);
            }
            const int n = nn[u], Tj = isqs[u] ? Tq : Tk, Rj = isqs[u] ? Rq : Rk;
            const float bias = b[n];
            float* out = qkv + (n / D) * (MAXT * D) + (n % D); // q, k, v are consecutive [MAXT][D] arrays
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = 16 * tts[u] + 4 * qd + r;
                if (row_valid<NS>(g, Tj, Rj)) out[g * D] = acc0[r] + acc1[r] + bias;
            }
        }
    }
    STAMP(22);
    __syncthreads();
}

// multi-head attention of Tq queries over Tk <= 32 keys (no mask), one phase:
//   ao[i][h*HD + c] = sum_j softmax_j(q_i . k_j / sqrt(HD)) v[j][h*HD + c]
// NS = 2: one wave per (sequence, head).  NS = 1: TWO waves per head, each with every other turn of four queries.
// Lane (row r, j) -- r = lane / 16, j = lane % 16 -- holds keys j and j + 16; a turn of the loop takes four queries, one per
// DPP row: 12-term dot products, the maximum and the sum over the keys by row reductions in registers, the probabilities
// through the wave's score rows in LDS (same wave: no barrier); then lane (query of the turn, channel) accumulates the output --
// the keys in their order, eight probabilities and values requested at a time (one LDS round trip per eight keys, not per key).
template <int NS>
DEV void attention(float* ao, const float* q, const float* k, const float* v, float* sc, int Tq, int Tk)
{
    const int Rq = rows_per_seq<NS>(Tq), Rk = rows_per_seq<NS>(Tk);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in lin)
    const int wave = lwave();
    const int h = wave & (NHD - 1), sq = NS == 1 ? 0 : wave / NHD; // NS = 1: waves h and h + 4 share head h; NS = 2: wave = 4 * sequence + head
    constexpr int SHARE = NS == 1 ? NWV / NHD : 1;                  // waves per (sequence, head)
    const int part = NS == 1 ? wave / NHD : 0;
    if (sq < NS) {
        q += Rq * sq * D; k += Rk * sq * D; v += Rk * sq * D; ao += Rq * sq * D; // (the sequence's rows)
        const float scale = 1.f / sqrtf((float)HD);
        const int j = lane & 15, r = lane >> 4;
        const bool live0 = j < Tk, live1 = j + 16 < Tk;
        float k0[HD], k1[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) k0[c] = k1[c] = 0.f;
        if (live0) read_k<HD>(k0, k + j * D + h * HD); // (a head's 12 channels: three 16-byte words)
        if (live1) read_k<HD>(k1, k + (j + 16) * D + h * HD);
        STAMP(20);
        float* sch = sc + (NS == 1 ? h * MAXT * MAXT : wave * 16 * MAXT);
        const int iq = lane / HD, cq = lane - iq * HD; // second half of a turn: lane (query iq of the turn, channel cq), lanes 0 .. 47
#pragma unroll 1
        for (int i0 = 4 * part; i0 < Tq; i0 += 4 * SHARE) {
            const int i = min(i0 + r, Tq - 1); // (a row beyond the last query recomputes it)
            float s0 = 0.f, s1 = 0.f, qv[HD];
            read_k<HD>(qv, q + i * D + h * HD);
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                s0 = fmaf(qv[c], k0[c], s0);
                s1 = fmaf(qv[c], k1[c], s1);
            }
            s0 = live0 ? s0 * scale : -3.0e38f;
            s1 = live1 ? s1 * scale : -3.0e38f;
            const float m = row_max(fmaxf(s0, s1));
            const float e0 = live0 ? expf(s0 - m) : 0.f, e1 = live1 ? expf(s1 - m) : 0.f;
            const float inv = 1.f / row_sum(e0 + e1);
            sch[i * MAXT + j] = e0 * inv; // (a key beyond the last: probability 0, so that the rows below can be read eight at a time)
            sch[i * MAXT + j + 16] = e1 * inv;
            // (LDS operations of a wave execute in order: the probabilities written above are visible below)
            const int io = i0 + iq;
            if (iq < 4 && io < Tq) {
                const float* p = sch + io * MAXT;
                const float* vc = v + h * HD + cq;
                float acc = 0.f;
#pragma unroll 1
                for (int j0 = 0; j0 < Tk; j0 += 8) {
                    float pj[8], vj[8];
                    read_k<8>(pj, p + j0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) vj[u] = vc[min(j0 + u, Tk - 1) * D]; // (beyond the last key: its probability is 0, the value any finite one)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fmaf(pj[u], vj[u], acc);
                }
                ao[io * D + h * HD + cq] = acc;
            }
        }
        STAMP(21);
    }
    __syncthreads();
}

// o = out_proj(attention(in_proj_q(xq), in_proj_k(xkv), in_proj_v(xkv)))
template <int NS>
DEV void mha(float* o, const float* xq, int Tq, const float* xkv, int Tk, const float* w, int in_wT, int in_b, int out_wT, int out_b,
             float* q, float* k, float* v, float* ao, float* sc)
{
    if (xq == xkv && Tq == 1) { // (uniform) ONE token attending to itself -- the first decoder call of every block: the softmax over one key is exactly 1
        // (exp(0) / 1), the head outputs ARE the value rows (fma(1, v, 0)), bit for bit: no query, no key, no attention phase
        lin_qkv<NS>(q, xq, Tq, xkv, Tk, w + in_wT, w + in_b, true);
        STAMP(1);
        lin<D / 4, NS>(o, D, v, D, Tq, w + out_wT, w + out_b, D, D);
        STAMP(3);
        return;
    }
    lin_qkv<NS>(q, xq, Tq, xkv, Tk, w + in_wT, w + in_b);
    STAMP(1);
    attention<NS>(ao, q, k, v, sc, Tq, Tk);
    STAMP(2);
    lin<D / 4, NS>(o, D, ao, D, Tq, w + out_wT, w + out_b, D, D);
    STAMP(3);
}

// ---- the TEAM exchange (few sequences: G workgroups run one sequence's block together, below) -------------------------------------------
// Unit of exchange: a 16-byte GRANULE = three partial sums + a tag, written with ONE write-through store (`sc1`: the bytes leave the
// writer's XCD) and read with `sc1` loads (never served by the reader's L1): the tag arrives with its data, so there is no flag, no
// drain and no fence -- a reader re-reads the G granules of its three outputs until all carry the tag of this exchange.  Granule i of the
// G workgroups is contiguous ([granule][workgroup]): one address register, immediate offsets.
// Tags: every team counts its exchanges in a word of device memory that lives as long as the handle (read by all its workgroups at entry, written
// back by the first at exit -- which it reaches only after every member has entered); a slot (team, granule, member) is only ever written by that
// member of that team, whatever the team size of the launch (the layout is that of the largest team), so what a slot holds is always an OLDER tag of
// the same counter: never the awaited one.  Nothing of this lives in kernel arguments: a launch captured into a graph can be replayed.
constexpr int XCH_GRANULES = 2 * 16 * D / 3;          // two token tiles of 16 x 48 partial sums, three per granule
constexpr int XCH_GMAX = 16;                          // the largest team; slots per granule in the layout
constexpr int XCH_POLL_LIMIT = 1 << 19;               // (~1 s: then the member gives up -- see "time-out" below)
// Time-out.  A team waits for its members, so all of them must be resident.  The host sizes teams so that they are (one workgroup per CU by the
// occupancy query, at most HALF the CUs the stream may use: dp_temporal_predict) -- against its own launch; another stream's long kernel, a CU mask
// set behind the library's back or a second process can still keep a member off the device.  What happens then is never silent:
//   * the member that has re-read its granules poll_limit times (~1 s) latches the handle's status word -- in device memory AND in its page-locked
//     host mirror --, stops waiting for good (`dead`) and, at the end of the block, writes NaN into EVERY target row of its sequence: the optimise
//     kernel's input screening then reports DP_STATUS_BAD_TARGETS for that sequence (include/dragposer.h) instead of being pulled towards garbage;
//   * members still waiting see the device word within 256 re-reads and give up the same way; a team launch that finds the word set at entry writes
//     NaN and leaves at once (a replayed graph, launches already in the queue): nothing of the exchange area is trusted after a time-out -- a late
//     member may have read the tag counter after it was rewritten and left future-valued tags behind;
//   * the host reads the mirror without a synchronise: dp_temporal_status(), and the NEXT dp_temporal_predict fails with DP_ERR_TIMEOUT, after
//     which the handle launches no more teams (one workgroup per sequence from then on).
// Hardware assumption of the granule (stated, soaked by tests/test_hip_temporal.py::test_team_soak): a 16-byte-aligned global_store_dwordx4 of one
// lane becomes visible as a whole -- the tag in word 3 never before the three sums.  The ISA's memory model promises single-copy atomicity up to 8
// bytes only; on gfx950 the 16 bytes travel as one write request inside one 32-byte sector and no tear has ever been observed
// (profiles/r05_team_soak.txt: 27 k launches; the gated soak test repeats it every GPU run).
DEV int load_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void store_granule(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
template <int N> DEV void load_granules(f4 (&v)[16], const f4* p);
template <> DEV void load_granules<2>(f4 (&v)[16], const f4* p)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(p) : "memory");
}
template <> DEV void load_granules<4>(f4 (&v)[16], const f4* p)
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
}
template <> DEV void load_granules<8>(f4 (&v)[16], const f4* p)
{
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p) : "memory");
}
template <> DEV void load_granules<16>(f4 (&v)[16], const f4* p)
{
    asm volatile("global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %16, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %16, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %16, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %16, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %16, off offset:112 sc1\n\t"
                 "global_load_dwordx4 %8, %16, off offset:128 sc1\n\tglobal_load_dwordx4 %9, %16, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %10, %16, off offset:160 sc1\n\tglobal_load_dwordx4 %11, %16, off offset:176 sc1\n\t"
                 "global_load_dwordx4 %12, %16, off offset:192 sc1\n\tglobal_load_dwordx4 %13, %16, off offset:208 sc1\n\t"
                 "global_load_dwordx4 %14, %16, off offset:224 sc1\n\tglobal_load_dwordx4 %15, %16, off offset:240 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]),
                   "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : "v"(p) : "memory");
}
struct Team { // (uniform per workgroup)
    f4* xch;       // this team's exchange area: [2 slots][XCH_GRANULES][XCH_GMAX] granules
    int* status;   // the handle's status word
    unsigned tag;  // tag of the NEXT exchange (counts up: every workgroup of a team makes the same calls in the same order)
    int* hstatus;  // ... its host mirror
    int g, G;      // this workgroup's rank in its team, the team's size (2, 4, 8 or 16)
    int* dead;     // (LDS) an exchange of this workgroup has timed out (1) / the handle was dead at entry (2): no further waiting
    int poll_limit;
    bool mute;     // private test hook: this member never publishes
};
DEV void team_timeout(Team* tm)
{ // latch the handle's status word (device + host mirror), stop waiting for good
    __hip_atomic_store(tm->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tm->hstatus, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *tm->dead = 1;
}

// a thread's three outputs: re-read the GG members' granules until all carry the awaited tag, then add them up in member order
template <int GG> DEV void gather(float (&sum)[3], const f4* src, Team* tm)
{
    f4 v[16];
    const bool dead = *tm->dead != 0;
    for (int tries = 0;;) {
        load_granules<GG>(v, src);
        bool all = true;
#pragma unroll
        for (int gg = 0; gg < GG; ++gg) all = all && __float_as_uint(v[gg][3]) == tm->tag;
        if (all || dead) break;
        if (++tries >= tm->poll_limit || ((tries & 255) == 0 && load_agent(tm->status) != 0)) { team_timeout(tm); break; } // (gives up: see "time-out" above)
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int gg = 0; gg < GG; ++gg) sum[j] += v[gg][j];
    }
}

// ---- the feed-forward block in SPLIT PRECISION on the bf16 matrix pipe (round 5; the arithmetic of dp_w16.h) -------------------------------
// 94 % of the block's FLOPs are the two products of the feed-forward layers; on v_mfma_f32_16x16x4_f32 they kept the matrix pipe -- and with it,
// fp32 MFMAs and the vector ALU being ONE issue resource on gfx950, the whole SIMD -- busy for 2/3 of the kernel at 1024 sequences (an ablation
// with half of them: -34 %).  v_mfma_f32_16x16x32_bf16 retires K = 32 per 16 cycles and leaves the vector ALU to the other waves.  Every fp32
// operand -- weight or activation -- is the exact sum of three bf16 terms (round-to-nearest at every stage, the remainders exact fp32 differences),
// and a product keeps the six term pairs above 2^-24 relative (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi), accumulated in fp32 by the MFMA: an fp32
// product up to its last bit or two.  A tile = 32 hidden units:
//   product 1  H^T[16 hidden x 16 tokens] (two M-tiles) = W1[16 x 48] X^T[48 x 16]: A = the weight rows (terms split by the host), B = the tokens'
//              channels (split ONCE per call into LDS, split_tokens).  Channels 0 .. 31 are one K-block: six MFMAs.  Channels 32 .. 47 are half
//              a K-block, so each of their MFMAs carries TWO term pairs -- slots 0 .. 15 one, slots 16 .. 31 another: A = [hi | hi], [mid | mid],
//              [lo | hi] against B = [hi | mid], [hi | mid], [hi | lo] -- three MFMAs, not six: 2 x 9 = 18 MFMAs;
//   bias + ReLU in registers; lane (token l16, group g) holds hidden units 4 g + r of both M-tiles: EIGHT values = its K-slots j = 4 t + r of
//   product 2  OUT[16 tokens x 16 columns] (three column tiles) += H[16 x 32] W2^T[32 x 16]: A = the three terms of those eight values (split in
//              registers: no LDS, no transposition), B = the weight terms, K-slot (g, j) = hidden unit 16 (j >> 2) + 4 g + (j & 3) by the host's
//              packing -- 3 x 6 = 18 MFMAs.  The result layout is the fp32 kernel's (lane = column l16 of tile ct, register r = token 4 g + r).
// 36 MFMAs of 16 cycles per 32 hidden units and token tile, where the fp32 form took 48 of 32.
DEV unsigned cvt_pk(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{lo, hi}, bf2)); }
DEV void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l)
{ // x = h + m + l exactly
    h = cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk(s0, s1);
}
struct T3 { u4 t[3]; }; // the three bf16 terms of eight values (one lane's share of a K-block)
DEV T3 split_block(f4 t0, f4 t1)
{
    unsigned h[4], m[4], l[4];
    split_pair(t0.x, t0.y, h[0], m[0], l[0]);
    split_pair(t0.z, t0.w, h[1], m[1], l[1]);
    split_pair(t1.x, t1.y, h[2], m[2], l[2]);
    split_pair(t1.z, t1.w, h[3], m[3], l[3]);
    T3 b;
    b.t[0] = u4{h[0], h[1], h[2], h[3]};
    b.t[1] = u4{m[0], m[1], m[2], m[3]};
    b.t[2] = u4{l[0], l[1], l[2], l[3]};
    return b;
}
DEV f4 mm(u4 a, u4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0); }
DEV f4 six(f4 acc, const u4 (&a)[3], const u4 (&b)[3])
{ // acc += A B over one K-block: the six term pairs, small ones first
    acc = mm(a[2], b[0], acc);
    acc = mm(a[1], b[1], acc);
    acc = mm(a[0], b[2], acc);
    acc = mm(a[1], b[0], acc);
    acc = mm(a[0], b[1], acc);
    acc = mm(a[0], b[0], acc);
    return acc;
}

// the tokens of a call as product 1's B operand, once per call, five rows of 64 lanes x 16 bytes per token tile (stride six): rows 0 .. 2 = the
// hi / mid / lo terms of channels 8 g .. 8 g + 7 of token l16 (lane = (l16, g)); row 3 = [hi | mid], row 4 = [hi | lo] of channels 32 .. 47
// (lanes g < 2: the first term of channels 32 + 8 g ..; lanes g >= 2: the second term of channels 32 + 8 (g - 2) ..).  Zero for rows without a
// token.  Every thread of the workgroup; ends with a barrier.
template <int NS>
DEV void split_tokens(u4* xs, const float* x, int T, int R)
{
    const int ntt = n_ttiles<NS>(T, R);
    int tid = ltid();
    asm volatile("" : "+v"(tid)); // (as in lin: the per-thread addresses are recomputed per call, not hoisted out of the layer loops and held)
    for (int item = tid; item < ntt * 96; item += NT) { // per token tile: 64 lane slots of the full K-block, 32 of the half one
        const int tt = item / 96, r = item - tt * 96, l16 = r & 15, g = r >> 4, ch = 8 * g; // g = 0 .. 3: channels 0 .. 31; g = 4, 5: channels 32 .. 47
        f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (row_valid<NS>(16 * tt + l16, T, R)) {
            v0 = *(const f4*)(x + (16 * tt + l16) * D + ch);
            v1 = *(const f4*)(x + (16 * tt + l16) * D + ch + 4);
        }
        const T3 b = split_block(v0, v1);
        u4* row = xs + tt * 6 * 64;
        if (g < 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k) row[k * 64 + 16 * g + l16] = b.t[k];
        } else {
            const int ls = 16 * (g - 4) + l16;
            row[3 * 64 + ls] = b.t[0]; row[3 * 64 + 32 + ls] = b.t[1];
            row[4 * 64 + ls] = b.t[0]; row[4 * 64 + 32 + ls] = b.t[2];
        }
    }
    __syncthreads();
}

// a tile's image, one lane's share (FFN_IMG_V 16-byte words; pack layout in dp_temporal_create), in the parts the products read
struct FW1 { u4 w[2][3]; f4 b1; };       // product 1, ONE M-tile (16 hidden units): [K-block][term], bias row
struct FW2 { u4 w[3][3]; };              // product 2: [column tile][term]
DEV void ffn_load1(FW1& im, const u4* img, int nt, int ntiles, int t)
{ // (img: the layer's image + lane; t: the M-tile, a compile-time constant at every call)
    if (nt < ntiles) {
        const u4* p = img + (size_t)nt * FFN_IMG_V * 64;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int k = 0; k < 3; ++k) im.w[kb][k] = p[((t * 2 + kb) * 3 + k) * 64];
        im.b1 = __builtin_bit_cast(f4, p[(21 + t) * 64]);
    }
}
DEV void ffn_load2(FW2& im, const u4* img, int nt, int ntiles)
{
    if (nt < ntiles) {
        const u4* p = img + (size_t)nt * FFN_IMG_V * 64;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int k = 0; k < 3; ++k) im.w[ct][k] = p[(12 + ct * 3 + k) * 64];
    }
}
// product 1 of one M-tile (16 hidden units) on one token tile, bias, ReLU: lane (token l16, group g) gets hidden units 4 g + r of the M-tile.
// xs = that token tile's operand rows in LDS (+ lane)
DEV f4 ffn_p1(const FW1& im, const u4* xs)
{
    f4 h = {0.f, 0.f, 0.f, 0.f};
    { // channels 32 .. 47: two term pairs per MFMA (small ones first)
        const u4 b1 = xs[3 * 64], b3 = xs[4 * 64];
        h = mm(im.w[1][2], b3, h); // lo.hi + hi.lo
        h = mm(im.w[1][1], b1, h); // mid.hi + mid.mid
        h = mm(im.w[1][0], b1, h); // hi.hi + hi.mid
    }
    {
        const u4 xb[3] = {xs[0], xs[64], xs[2 * 64]};
        h = six(h, im.w[0], xb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r] + im.b1[r], 0.f);
    return h;
}
// product 2: the two M-tiles' values are the lane's eight K-slots; their three terms are made here (kept as fp32 until now: 8 registers, not 12)
DEV void ffn_p2(const FW2& im, f4 h0, f4 h1, f4 (&acc)[3])
{
    const T3 ht = split_block(h0, h1);
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) acc[ct] = six(acc[ct], ht.t, im.w[ct]);
}

// o[t] = linear2(relu(linear1(x[t]))) + bias2  (header comment; pack layout in dp_temporal_create).
// (PREFETCH: no longer a difference in the feed-forward -- every variant passes a tile's image through the registers in two parts.)
// NS = 2: every tile image serves the token tiles of both sequences.
// TEAM (NS = 1): workgroup g of G takes the tiles g * 8 + wave, + 8 G, ...; its eight waves' partial outputs are summed through LDS as ever, the G
// workgroups' sums are exchanged as granules (above) and every workgroup adds them up in the same order -- all G hold the same `o` afterwards,
// bit for bit, which is what lets them run the rest of the block redundantly and in step.
// COOP (PAIR kernels, calls over at most 8 tokens: R = 8): the 16 waves of the two halves deal the tiles among themselves and every tile image
// serves BOTH halves' token tiles (one each: two sequences of 8 rows) -- half the weight bytes through the CU's L1 per token, which is what bounds
// these calls (profiles/r05_temporal_phases_1024.txt: 43 k of 49 k cycles per layer are the images' way through the 64 B/clk L1).  `xsb` / `red`
// are then HALF 0's buffers; half h's lie XS_HALF / RED_HALF further (the kernel declares them as [2][...] arrays: contiguous).
constexpr int XS_HALF_U4 = NHD * MAXT * MAXT / 4, RED_HALF = 4 * MAXT * D;
template <bool PREFETCH, int NS, int R = 16, bool TEAM = false, bool COOP = false>
DEV void ffn(float* o, const float* x, int T, const float* w, int F, int pack, int l2b, float* red, u4* xsb, Team* tm = nullptr, const FW1* pre = nullptr)
{
    if constexpr (COOP) {
        static_assert(NS == 2 && R == 8 && !TEAM, "COOP: two halves of two sequences with 8 rows each");
        const int w16 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), half = w16 >> 3; // my wave among the workgroup's sixteen; my half (uniform)
        split_tokens<NS>(xsb + half * XS_HALF_U4, x, T, R);                  // (my half's tokens into my half's buffer; ends with the workgroup's barrier)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane)); // (as in lin)
        const int ntiles = (F + FT - 1) / FT;
        const u4* img = (const u4*)(w + pack) + lane;
        const u4* xs = xsb + lane;
        f4 acc[2][3];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[g][ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int nt = w16; nt < ntiles; nt += 2 * NWV) {
            f4 h[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                FW1 w1;
                ffn_load1(w1, img, nt, ntiles, t);
#pragma unroll
                for (int g = 0; g < 2; ++g) h[g][t] = ffn_p1(w1, xs + g * XS_HALF_U4);
                __builtin_amdgcn_sched_barrier(0); // (the next part's loads are not to be hoisted above this one's products: registers)
            }
            FW2 w2;
            ffn_load2(w2, img, nt, ntiles);
#pragma unroll
            for (int g = 0; g < 2; ++g) ffn_p2(w2, h[g][0], h[g][1], acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(23);
        // the sixteen waves' partial outputs of token tile g (= half g's), through the two halves' reduction buffers taken as one: sixteen 3 KB slots
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) *(f4*)(red + ((w16 * 3 + ct) * 64 + lane) * 4) = acc[g][ct];
            __syncthreads();
            if (half == g) { // (uniform per wave)
                for (int idx = ltid(); idx < 16 * D; idx += NT) {
                    const int tl = idx / D, c = idx - tl * D;
                    if (row_valid<NS>(tl, T, R)) {
                        const int slot = (((c >> 4) * 64) + (tl >> 2) * 16 + (c & 15)) * 4 + (tl & 3);
                        float sum = w[l2b + c];
#pragma unroll
                        for (int wv = 0; wv < 2 * NWV; ++wv) sum += red[wv * 3 * 256 + slot];
                        o[tl * D + c] = sum;
                    }
                }
            }
            __syncthreads();
        }
        STAMP(24);
        return;
    } // (xsb: LDS for the tokens' operand rows -- the attention's score buffer, idle here; pre: TEAM -- the image of this wave's first tile, requested by
  //  the caller at the head of the layer: a cold fetch hidden behind the attention; R, the rows per sequence, is a compile-time constant here: as a
  //  run-time value it cost the 128-register instantiation 8 spills)
    split_tokens<NS>(xsb, x, T, R);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in lin)
    const int wave = lwave();
    const int ntiles = (F + FT - 1) / FT;
    const u4* img = (const u4*)(w + pack) + lane;
    const u4* xs = xsb + lane;
    constexpr int NG = NS == 1 ? 1 : (NS * R + 15) / 16; // token tiles that share one pass over the weights (R = 8: ONE tile for the two sequences)
    const int ntt = n_ttiles<NS>(T, R);
    const int first = TEAM ? tm->g * NWV + wave : wave, stride = TEAM ? tm->G * NWV : NWV; // this wave's tiles
    f4* xslot = nullptr;
    if (TEAM) xslot = tm->xch + (size_t)(tm->tag & 1u) * XCH_GRANULES * XCH_GMAX;
#pragma unroll 1
    for (int tt0 = 0; tt0 < ntt; tt0 += NG) {
        constexpr int ng = NG;
        f4 acc[NG][3];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[g][ct] = f4{0.f, 0.f, 0.f, 0.f};
        // a tile's image goes through the registers in parts: product 1's two M-tiles, product 2.  256 registers (PREFETCH, TEAM): product 2's part is
        // in flight under product 1, the NEXT tile's under product 2 (TEAM: the first tile's first M-tile was requested by the caller at the head of
        // the layer).  128 registers: one part at a time, requested when the registers are free (the three other waves of the SIMD cover the round trip).
        if (TEAM || PREFETCH) {
            FW1 w1a, w1b;
            FW2 w2;
            int nt = first;
            if (TEAM && pre) w1a = *pre;
            else ffn_load1(w1a, img, nt, ntiles, 0);
            ffn_load1(w1b, img, nt, ntiles, 1);
#pragma unroll 1
            while (nt < ntiles) {
                ffn_load2(w2, img, nt, ntiles);
                f4 h[NG][2];
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ng) { h[g][0] = ffn_p1(w1a, xs + (tt0 + g) * 6 * 64); h[g][1] = ffn_p1(w1b, xs + (tt0 + g) * 6 * 64); }
                ffn_load1(w1a, img, nt + stride, ntiles, 0);
                ffn_load1(w1b, img, nt + stride, ntiles, 1);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ng) ffn_p2(w2, h[g][0], h[g][1], acc[g]);
                nt += stride;
            }
        } else {
#pragma unroll 1
            for (int nt = first; nt < ntiles; nt += stride) {
                f4 h[NG][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    FW1 w1;
                    ffn_load1(w1, img, nt, ntiles, t);
#pragma unroll
                    for (int g = 0; g < NG; ++g) if (g < ng) h[g][t] = ffn_p1(w1, xs + (tt0 + g) * 6 * 64);
                    __builtin_amdgcn_sched_barrier(0); // (the next part's loads are not to be hoisted above this one's products: registers)
                }
                FW2 w2;
                ffn_load2(w2, img, nt, ntiles);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ng) ffn_p2(w2, h[g][0], h[g][1], acc[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(23);
        // the waves' partial outputs, one token tile at a time through the reduction buffer: lane (channel l16 of tile ct,
        // token group q), register r = token 4 q + r
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng) break; // (uniform)
            const int tt = tt0 + g;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) *(f4*)(red + ((wave * 3 + ct) * 64 + lane) * 4) = acc[g][ct];
            __syncthreads();
            if (TEAM) { // this workgroup's sums of token tile tt leave as granules: thread i < 256 = outputs 3 i .. 3 i + 2 (one row: 48 = 16 x 3)
                const int i = threadIdx.x, tl = i >> 4, c0 = 3 * (i & 15);
                if (i < 16 * D / 3 && row_valid<NS>(16 * tt + tl, T, R)) {
                    f4 gr;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int c = c0 + j, slot = (((c >> 4) * 64) + (tl >> 2) * 16 + (c & 15)) * 4 + (tl & 3);
                        float sum = 0.f;
#pragma unroll
                        for (int wv = 0; wv < NWV; ++wv) sum += red[wv * 3 * 256 + slot];
                        gr[j] = sum;
                    }
                    gr[3] = __uint_as_float(tm->tag);
                    if (!tm->mute) store_granule(xslot + (size_t)(tt * (16 * D / 3) + i) * XCH_GMAX + tm->g, gr);
                }
                __syncthreads();
                continue;
            }
            for (int idx = ltid(); idx < 16 * D; idx += NT) {
                const int tl = idx / D, c = idx - tl * D;
                if (row_valid<NS>(16 * tt + tl, T, R)) {
                    const int slot = (((c >> 4) * 64) + (tl >> 2) * 16 + (c & 15)) * 4 + (tl & 3);
                    float sum = w[l2b + c];
#pragma unroll
                    for (int wv = 0; wv < NWV; ++wv) sum += red[wv * 3 * 256 + slot];
                    o[(16 * tt + tl) * D + c] = sum;
                }
            }
            __syncthreads();
        }
    }
    STAMP(24);
    if (TEAM) { // gather: thread i = granule i of the (at most two) token tiles; the G workgroups' granules of it are G x 16 contiguous bytes
        const int i = threadIdx.x, tt = i >> 8, il = i & 255, tl = il >> 4, c0 = 3 * (il & 15);
        if (tt < ntt && row_valid<NS>(16 * tt + tl, T, R)) {
            const f4* src = xslot + (size_t)i * XCH_GMAX;
            float sum[3] = {w[l2b + c0], w[l2b + c0 + 1], w[l2b + c0 + 2]};
            // (one copy of the loop per team size: sixteen "member < G" predicates kept across the layer loops cost the kernel 200 scalar spills)
            switch (tm->G) {
            case 16: gather<16>(sum, src, tm); break;
            case 8: gather<8>(sum, src, tm); break;
            case 4: gather<4>(sum, src, tm); break;
            default: gather<2>(sum, src, tm); break;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) o[(16 * tt + tl) * D + c0 + j] = sum[j];
        }
        __syncthreads();
        tm->tag += 1u;
    }
}

// OCC = waves per SIMD: 4 (two workgroups per CU, 128 registers: throughput with many sequences) or 2 (one workgroup per CU,
// 256 registers, the next part of a feed-forward tile's image in flight under the current one's products: latency with few).  NS = sequences per workgroup.
// TEAM (OCC = 2, NS = 1; few sequences): a.G workgroups per sequence.  Every one of them runs the whole block -- same instructions, same data, same
// bits -- except the feed-forward layers (94 % of the work), where each takes 1/G of the hidden units and the partial sums are exchanged (ffn).
// PAIR (OCC = 4, NS = 2): one workgroup of 2 NT threads per CU -- two halves, each what the NS = 2 workgroup is, sharing the feed-forward weight
// fetches of the calls over at most 8 tokens (ffn: COOP).
template <int OCC, int NS, bool TEAM = false, bool PAIR = false>
__global__ __launch_bounds__(PAIR ? 2 * NT : NT, PAIR ? 1 : OCC) void dp_temporal_kernel(const TArgs a)
{
    static_assert(!TEAM || NS == 1, "a team runs one sequence");
    static_assert(!PAIR || (NS == 2 && OCC == 4 && !TEAM), "PAIR: two halves of two sequences each");
    constexpr int H2 = PAIR ? 2 : 1;
    const int half = PAIR ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9) : 0; // (uniform per wave: the halves' array bases stay in scalar registers)
    __shared__ __attribute__((aligned(16))) float mem_[H2][MAXT * D], x_[H2][MAXT * D], o_[H2][MAXT * D]; // (rows are read in 16-byte words: lin)
    // q, k, v and the attention output; dead while the feed-forward block runs, whose cross-wave reduction buffer is the
    // same 24 KB (70 KB of LDS in all: two workgroups per CU -- or one of two halves)
    __shared__ __attribute__((aligned(16))) float qkva_[H2][4 * MAXT * D];
    static_assert(4 * MAXT * D >= NWV * 3 * 64 * 4 && RED_HALF == NWV * 3 * 64 * 4, "the reduction buffer fits the attention buffers (COOP: exactly, the two halves' make sixteen slots)");
    float *mem = mem_[half], *x = x_[half], *o = o_[half], *qkva = qkva_[half];
    float *q = qkva, *kb = qkva + MAXT * D, *vb = qkva + 2 * MAXT * D, *ao = qkva + 3 * MAXT * D, *red = qkva;
    __shared__ __attribute__((aligned(16))) float sc_[H2][NHD * MAXT * MAXT], tok_[H2][MAXT * LAT], enc_in_[H2][MAXT * MAX_IN], preds_[H2][NS * (MAXT + 1) * LAT];
    static_assert(XS_HALF_U4 * 4 == NHD * MAXT * MAXT, "COOP: the halves' token operand buffers are one array");
    float *sc = sc_[half], *tok = tok_[half], *enc_in = enc_in_[half], *preds = preds_[half];
    const int s0 = TEAM ? (int)blockIdx.x / a.G : PAIR ? (int)blockIdx.x * 4 + 2 * half : (int)blockIdx.x * NS, tid = ltid();
    if (!PAIR && s0 >= a.n_seq) return; // (PAIR: a half without a sequence computes on zeros and stores nothing -- its waves are needed in the shared phases)
    __shared__ int team_dead;
    Team team{};
    if (TEAM) {
        team.g = (int)blockIdx.x - s0 * a.G; team.G = a.G;
        team.xch = (f4*)a.xch + (size_t)s0 * 2 * XCH_GRANULES * XCH_GMAX;
        team.status = a.tstatus; team.hstatus = a.hstatus; team.tag = a.epochs[s0] + 1u; team.dead = &team_dead;
        team.poll_limit = a.poll_limit; team.mute = s0 == a.dbg_skip_team && team.g == a.dbg_skip_member;
        if (tid == 0) team_dead = load_agent(a.tstatus) != 0 ? 2 : 0; // (read after the barriers below)
    }
    const float* w = a.w;
    STAMP(0);
    // the layer tables (encoder layers, then decoder layers: one table) in LDS: a layer's sixteen offsets read from memory at the head of the layer
    // were scalar loads from a cold line -- 3 k cycles in front of every layer of the one-sequence launch (profiles/r05_temporal_phases_team.txt)
    constexpr int TLW = (int)(sizeof(TLayer) / sizeof(int));
    __shared__ int ltab[2 * MAXL * TLW];
    for (int i = tid; i < (a.n_enc + a.n_dec) * TLW; i += NT) ltab[i] = ((const int*)(w + a.enc_tab))[i]; // (visible after the token assembly's barrier)
    auto layer_of = [&](int li) {
        TLayer L;
        int* Lp = (int*)&L;
#pragma unroll
        for (int k = 0; k < TLW; ++k) Lp[k] = __builtin_amdgcn_readfirstlane(ltab[li * TLW + k]); // (uniform: back into scalar registers)
        return L;
    };
    // TEAM: the LayerNorm rows in LDS (17 LayerNorms of one token each stand in the decoder's latency chain: their two rows came from cold memory)
    __shared__ float lnrows[TEAM ? LN_MAX : 1];
    if constexpr (TEAM) {
        for (int i = tid; i < a.ln_len; i += NT) lnrows[i] = w[a.ln0 + i]; // (visible after the token assembly's barrier)
    }
    const float* wl = TEAM ? lnrows - a.ln0 : w; // what the LayerNorm row offsets are relative to
    const int H = a.H, step = a.step, n_past = (H + step - 1) / step, Te = n_past - 1, n_steps = a.window / step + 1;

    // ---- tokens (drag_pose.py:249-266): latent normalised | displacement accumulated over `step` frames | heights.
    //      Sequence slot sl of the workgroup fills rows 16 sl .. of the arrays (NS = 1: rows 0 .. Te-1); a slot beyond the
    //      last sequence computes on zeros and stores nothing.
    for (int idx = tid; idx < NS * Te * a.n_in; idx += NT) {
        const int sl = idx / (Te * a.n_in), r2 = idx - sl * Te * a.n_in, t = r2 / a.n_in, c = r2 - t * a.n_in, fr = t * step;
        const int s = s0 + sl;
        float v = 0.f;
        if (s < a.n_seq) {
            if (c < LAT) v = (a.latent_buf[((size_t)s * H + fr) * LAT + c] - w[a.mean + c]) / w[a.stdv + c];
            else if (c < LAT + 3) {
                for (int j = 0; j < step && fr + j < H; ++j) v += a.disp_buf[((size_t)s * H + fr + j) * 3 + (c - LAT)];
            } else v = a.heights_buf[((size_t)s * H + fr) * a.nh + (c - LAT - 3)];
        }
        enc_in[(16 * sl + t) * MAX_IN + c] = v;
    }
    if (tid < NS * LAT) {
        const int sl = tid / LAT, c = tid - sl * LAT, s = s0 + sl;
        tok[16 * sl * LAT + c] = s < a.n_seq ? (a.latent_buf[((size_t)s * H + Te * step) * LAT + c] - w[a.mean + c]) / w[a.stdv + c] : 0.f;
    }
    __syncthreads();
    STAMP(10);
    if (TEAM && team_dead == 2) { // (uniform) an earlier launch of this handle timed out: the exchange area is not to be trusted -- NaN, at once
        if (team.g == 0)
            for (int idx = tid; idx < (a.window + 1) * LAT; idx += NT) a.target[(size_t)s0 * (a.window + 1) * LAT + idx] = __builtin_nanf("");
        return;
    }

    // ---- encoder, once (the memory is the same for every autoregressive call)
    lin<MAX_IN / 4, NS>(x, D, enc_in, MAX_IN, Te, w + a.ipe_wT, w + a.ipe_b, D, a.n_in, w + a.pe, 16); // (enc_in: 16 rows per sequence)
    STAMP(11);
    for (int l = 0; l < a.n_enc; ++l) {
        const TLayer L = layer_of(l);
        FW1 pre;
        if constexpr (TEAM) ffn_load1(pre, (const u4*)(w + L.ffn_pack) + (tid & 63), team.g * NWV + (tid >> 6), (a.ff + FT - 1) / FT, 0);
        mha<NS>(o, x, Te, x, Te, w, L.sa_in_wT, L.sa_in_b, L.sa_out_wT, L.sa_out_b, q, kb, vb, ao, sc);
        add_ln<NS>(x, o, Te, wl + L.n1w, wl + L.n1b);
        STAMP(4);
        if constexpr (TEAM) ffn<true, 1, 16, true>(o, x, Te, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc, &team, &pre);
        else if (rows_per_seq<NS>(Te) == 8) { // (uniform; a history of at most 8 tokens)
            if constexpr (PAIR) ffn<false, NS, 8, false, true>(o, x, Te, w, a.ff, L.ffn_pack, L.lin2_b, qkva_[0], (u4*)sc_[0]);
            else ffn<OCC == 2, NS, 8>(o, x, Te, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc);
        }
        else ffn<OCC == 2, NS, 16>(o, x, Te, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc);
        STAMP(5);
        add_ln<NS>(x, o, Te, wl + L.n2w, wl + L.n2b);
        STAMP(4);
    }
    add_ln<NS>(x, nullptr, Te, wl + a.encn_w, wl + a.encn_b);
    for (int idx = tid; idx < MAXT * D; idx += NT) mem[idx] = x[idx];
    __syncthreads();
    STAMP(12);

    // ---- autoregressive calls (drag_pose.py:274-279): call i sees i + 1 target tokens, keeps the last position's output
    for (int it = 0; it < n_steps; ++it) {
        const int T = it + 1;
        // (the activations of a call over at most 8 tokens take 8 rows per sequence: two sequences share a tile -- rows_per_seq; the token buffer
        //  keeps 16, the memory the layout of its own token count)
        lin<LAT / 4, NS>(x, D, tok, LAT, T, w + a.ipd_wT, w + a.ipd_b, D, LAT, w + a.pe, 16);
        STAMP(13);
        for (int l = 0; l < a.n_dec; ++l) {
            const TLayer L = layer_of(a.n_enc + l);
            FW1 pre;
            if constexpr (TEAM) ffn_load1(pre, (const u4*)(w + L.ffn_pack) + (tid & 63), team.g * NWV + (tid >> 6), (a.ff + FT - 1) / FT, 0);
            mha<NS>(o, x, T, x, T, w, L.sa_in_wT, L.sa_in_b, L.sa_out_wT, L.sa_out_b, q, kb, vb, ao, sc);
            add_ln<NS>(x, o, T, wl + L.n1w, wl + L.n1b);
            STAMP(4);
            mha<NS>(o, x, T, mem, Te, w, L.ca_in_wT, L.ca_in_b, L.ca_out_wT, L.ca_out_b, q, kb, vb, ao, sc);
            add_ln<NS>(x, o, T, wl + L.n2w, wl + L.n2b);
            STAMP(4);
            if constexpr (TEAM) ffn<true, 1, 16, true>(o, x, T, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc, &team, &pre);
            else if (rows_per_seq<NS>(T) == 8) { // (uniform)
                if constexpr (PAIR) ffn<false, NS, 8, false, true>(o, x, T, w, a.ff, L.ffn_pack, L.lin2_b, qkva_[0], (u4*)sc_[0]);
                else ffn<OCC == 2, NS, 8>(o, x, T, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc);
            }
            else ffn<OCC == 2, NS, 16>(o, x, T, w, a.ff, L.ffn_pack, L.lin2_b, red, (u4*)sc);
            STAMP(5);
            add_ln<NS>(x, o, T, wl + L.n3w, wl + L.n3b);
            STAMP(4);
        }
        add_ln<NS>(x, nullptr, T, wl + a.decn_w, wl + a.decn_b);
        STAMP(4);
        next_token<NS>(tok, preds, x, T, it, it + 1 < n_steps, w + a.op_wT, w + a.op_b);
        STAMP(14);
    }

    // ---- de-normalise, then the reference's "lerp" with weight 1 (drag_pose.py:283-291): frame k of the window holds the
    //      NEXT sampled prediction, the last frame its own
    const int W = a.window;
    if (TEAM && team_dead != 0) { // (uniform: the last write to it lies behind several barriers) this member gave up waiting: see "time-out"
        for (int idx = tid; idx < (W + 1) * LAT; idx += NT) a.target[(size_t)s0 * (W + 1) * LAT + idx] = __builtin_nanf("");
        return;
    }
    if (TEAM && team.g != 0) return; // (every workgroup of the team holds the result; the first stores it)
    if (TEAM && tid == 0) a.epochs[s0] = team.tag - 1u; // (the team's last tag: every member has read the word long ago -- it has published since)
    for (int idx = tid; idx < NS * (W + 1) * LAT; idx += NT) {
        const int sl = idx / ((W + 1) * LAT), r2 = idx - sl * (W + 1) * LAT, k = r2 / LAT, c = r2 - k * LAT, m = k < W ? k / step + 1 : W / step;
        if (s0 + sl < a.n_seq)
            a.target[((size_t)(s0 + sl) * (W + 1) + k) * LAT + c] = preds[(sl * (MAXT + 1) + m) * LAT + c] * w[a.stdv + c] + w[a.mean + c];
    }
    STAMP(15);
}

thread_local std::string g_terr;

} // namespace

struct dp_temporal {
    int device = -1, n_cu = 256;
    int forced_variant = 0; // dp_temporal_debug_force_variant (private test hook, below): 21, 41 or 42 (waves per SIMD, sequences
                            // per workgroup) = that kernel variant whatever the batch; 0 = chosen from the batch (the product)
    float* d_w = nullptr;
    float* d_xch = nullptr;  // the teams' exchange area: [n_cu / 2 teams][2][XCH_GRANULES][XCH_GMAX] granules, the teams' tag counters, the status word
    size_t xch_granule_bytes = 0;
    int* h_status = nullptr; // page-locked host mirror of the status word (written by the device on a team time-out, read here without a synchronise)
    int status_seen = 0;     // DP_TEMPORAL_* bits ever seen in it (sticky)
    bool teams_off = false;  // no team launches any more: a time-out was reported, or the TEAM kernel does not fit a CU of this device
    int poll_limit = XCH_POLL_LIMIT, dbg_skip_team = -1, dbg_skip_member = -1; // dp_temporal_debug_team_fault (private test hook)
    TArgs args{};
    std::string err;
};

static int tfail(dp_temporal* t, int code, const std::string& msg)
{
    if (t) t->err = msg;
    else g_terr = msg;
    return code;
}

// float -> three bf16 terms with x = t0 + t1 + t2 exactly: the host's copy of the device's split_pair (round to nearest even at every stage, what
// v_cvt_pk_bf16_f32 does; the remainders are exact fp32 differences).  Weights are finite.
static unsigned host_bf16_rne(float x)
{
    unsigned u;
    std::memcpy(&u, &x, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static float host_bf16_val(unsigned h)
{
    const unsigned u = h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static void host_split3(float x, unsigned (&t)[3])
{
    t[0] = host_bf16_rne(x);
    const float r = x - host_bf16_val(t[0]);
    t[1] = host_bf16_rne(r);
    const float q = r - host_bf16_val(t[1]);
    t[2] = host_bf16_rne(q);
}
// private test hook (host arithmetic only; a CPU test holds it to numpy): the three bf16 terms of x as 16-bit patterns
extern "C" void dp_temporal_debug_split3(float x, unsigned short* out3)
{
    unsigned t[3];
    host_split3(x, t);
    for (int k = 0; k < 3; ++k) out3[k] = (unsigned short)t[k];
}

extern "C" const char* dp_temporal_last_error(const dp_temporal* t) { return t ? t->err.c_str() : g_terr.c_str(); }

extern "C" int dp_temporal_create(dp_temporal** out, const dp_temporal_model* m, int device)
{
    if (!out) return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: out is NULL");
    *out = nullptr;
    if (!m) return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: model is NULL");
    if (m->n_heights < 0 || m->n_heights > DP_MAX_HEIGHT_JOINTS || m->dim_feedforward < 1 || m->sample_step < 1 || m->max_len < 1 ||
        m->n_encoder_layers < 1 || m->n_encoder_layers > MAXL || m->n_decoder_layers < 1 || m->n_decoder_layers > MAXL)
        return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: architecture out of range (layers 1..8, heights <= 8)");
    const int n_in = LAT + 3 + m->n_heights, F = m->dim_feedforward;
    std::vector<float> buf;
    bool null_seen = false;
    auto put = [&](const float* p, size_t n) { // plain copy
        const int off = (int)buf.size();
        if (!p) { null_seen = true; buf.resize(buf.size() + n, 0.f); return off; }
        buf.insert(buf.end(), p, p + n);
        return off;
    };
    auto putT = [&](const float* p, int rows_out, int cols_in, int ldk = 0) { // Linear.weight [out][in] as it is, rows padded with zeros to a multiple of 4 (or to ldk)
        while (buf.size() % 4) buf.push_back(0.f);               // floats and 16-byte aligned (lin: a lane reads KS consecutive floats of a row)
        const int off = (int)buf.size();
        if (ldk == 0) ldk = (cols_in + 3) / 4 * 4;
        buf.resize(buf.size() + (size_t)rows_out * ldk, 0.f);
        if (!p) { null_seen = true; return off; }
        for (int r = 0; r < rows_out; ++r)
            for (int c = 0; c < cols_in; ++c) buf[off + (size_t)r * ldk + c] = p[(size_t)r * cols_in + c];
        return off;
    };
    // feed-forward image (split precision: the comment above ffn_tile), per tile of 32 hidden units and lane (l16 = lane & 15, g = lane >> 4),
    // FFN_IMG_V 16-byte words of eight bf16 each (element j in bits 16 (j & 1) of word j >> 1), three words per operand = its hi / mid / lo terms:
    //   v = (t 2) 3 + term:           W1[32 nt + 16 t + l16][8 g + j]                                  (A of product 1: M-tile t, channels 0 .. 31)
    //   v = (t 2 + 1) 3 + {0, 1, 2}:  [hi | hi], [mid | mid], [lo | hi] of W1[32 nt + 16 t + l16][32 + 8 (g & 1) + j]: the first term in lanes g < 2,
    //                                 the second in lanes g >= 2                                       (channels 32 .. 47: two term pairs per MFMA)
    //   v = 12 + ct 3 + term:         W2[16 ct + l16][32 nt + 16 (j >> 2) + 4 g + (j & 3)]             (B of product 2: column tile ct)
    //   v = 21 + t (four floats):     bias1[32 nt + 16 t + 4 g + r]
    // hidden units beyond F and input channels beyond 47 are zeros (ReLU(0) = 0 contributes nothing)
    auto split3 = [&](float x, unsigned (&t)[3]) { host_split3(x, t); };
    auto pack_ffn = [&](const float* w1, const float* b1, const float* w2) {
        while (buf.size() % 4) buf.push_back(0.f); // 16-byte alignment of the image
        const int off = (int)buf.size(), ntiles = (F + FT - 1) / FT;
        buf.resize(buf.size() + (size_t)ntiles * FFN_TILE_FLOATS, 0.f);
        if (!w1 || !b1 || !w2) { null_seen = true; return off; }
        for (int nt = 0; nt < ntiles; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int l16 = lane & 15, g = lane >> 4;
                float* dst = buf.data() + off + (size_t)nt * FFN_TILE_FLOATS;
                auto put8 = [&](int v0, const float (&val)[8]) { // eight values -> the words v0 (hi), v0 + 1 (mid), v0 + 2 (lo) of this lane
                    unsigned words[3][4] = {};
                    for (int j = 0; j < 8; ++j) {
                        unsigned t[3];
                        split3(val[j], t);
                        for (int k = 0; k < 3; ++k) words[k][j >> 1] |= t[k] << (16 * (j & 1));
                    }
                    for (int k = 0; k < 3; ++k) std::memcpy(dst + ((v0 + k) * 64 + lane) * 4, words[k], 16);
                };
                for (int t = 0; t < 2; ++t) {
                    float val[8];
                    const int h = FT * nt + 16 * t + l16;
                    for (int j = 0; j < 8; ++j) val[j] = h < F ? w1[(size_t)h * D + 8 * g + j] : 0.f;
                    put8((t * 2) * 3, val);
                    // channels 32 .. 47: which TERM a lane holds depends on its half of the K-block
                    unsigned words[3][4] = {};
                    for (int j = 0; j < 8; ++j) {
                        unsigned tm[3];
                        split3(h < F ? w1[(size_t)h * D + 32 + 8 * (g & 1) + j] : 0.f, tm);
                        const unsigned pick[3] = {tm[0], tm[1], g < 2 ? tm[2] : tm[0]}; // [hi | hi], [mid | mid], [lo | hi]
                        for (int k = 0; k < 3; ++k) words[k][j >> 1] |= pick[k] << (16 * (j & 1));
                    }
                    for (int k = 0; k < 3; ++k) std::memcpy(dst + (((t * 2 + 1) * 3 + k) * 64 + lane) * 4, words[k], 16);
                }
                for (int ct = 0; ct < 3; ++ct) {
                    float val[8];
                    for (int j = 0; j < 8; ++j) {
                        const int h = FT * nt + 16 * (j >> 2) + 4 * g + (j & 3);
                        val[j] = h < F ? w2[(size_t)(16 * ct + l16) * F + h] : 0.f;
                    }
                    put8(12 + ct * 3, val);
                }
                for (int t = 0; t < 2; ++t)
                    for (int r = 0; r < 4; ++r) {
                        const int h = FT * nt + 16 * t + 4 * g + r;
                        dst[((21 + t) * 64 + lane) * 4 + r] = h < F ? b1[h] : 0.f;
                    }
            }
        return off;
    };
    std::vector<float> lnbuf; // the LayerNorm rows, appended to buf as one block below (offsets are relative until then)
    auto put_ln = [&](const float* p) {
        const int off = (int)lnbuf.size();
        if (!p) { null_seen = true; lnbuf.resize(lnbuf.size() + D, 0.f); return off; }
        lnbuf.insert(lnbuf.end(), p, p + D);
        return off;
    };
    TArgs a{};
    a.n_enc = m->n_encoder_layers; a.n_dec = m->n_decoder_layers; a.ff = F; a.n_in = n_in; a.nh = m->n_heights;
    a.max_len = m->max_len; a.step = m->sample_step;
    a.ipe_wT = putT(m->in_proj_encoder_w, D, n_in, MAX_IN); a.ipe_b = put(m->in_proj_encoder_b, D); // (the kernel's K-steps cover MAX_IN inputs)
    a.ipd_wT = putT(m->in_proj_decoder_w, D, LAT); a.ipd_b = put(m->in_proj_decoder_b, D);
    a.op_wT = putT(m->out_proj_w, LAT, D); a.op_b = put(m->out_proj_b, LAT);
    a.pe = put(m->pos_encoding, (size_t)m->max_len * D);
    a.encn_w = put_ln(m->enc_norm_w); a.encn_b = put_ln(m->enc_norm_b);
    a.decn_w = put_ln(m->dec_norm_w); a.decn_b = put_ln(m->dec_norm_b);
    a.mean = put(m->means_latent, LAT); a.stdv = put(m->stds_latent, LAT);
    if (!m->enc || !m->dec) return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: NULL layer array");
    auto layer = [&](const dp_temporal_layer& L, bool dec) {
        TLayer o{};
        o.sa_in_wT = putT(L.sa_in_w, 3 * D, D); o.sa_in_b = put(L.sa_in_b, 3 * D);
        o.sa_out_wT = putT(L.sa_out_w, D, D); o.sa_out_b = put(L.sa_out_b, D);
        if (dec) {
            o.ca_in_wT = putT(L.ca_in_w, 3 * D, D); o.ca_in_b = put(L.ca_in_b, 3 * D);
            o.ca_out_wT = putT(L.ca_out_w, D, D); o.ca_out_b = put(L.ca_out_b, D);
        }
        o.ffn_pack = pack_ffn(L.lin1_w, L.lin1_b, L.lin2_w);
        o.lin2_b = put(L.lin2_b, D);
        o.n1w = put_ln(L.norm1_w); o.n1b = put_ln(L.norm1_b);
        o.n2w = put_ln(L.norm2_w); o.n2b = put_ln(L.norm2_b);
        if (dec) { o.n3w = put_ln(L.norm3_w); o.n3b = put_ln(L.norm3_b); }
        return o;
    };
    std::vector<TLayer> tabs;
    for (int l = 0; l < a.n_enc; ++l) tabs.push_back(layer(m->enc[l], false));
    for (int l = 0; l < a.n_dec; ++l) tabs.push_back(layer(m->dec[l], true));
    while (buf.size() % 4) buf.push_back(0.f);
    a.ln0 = (int)buf.size(); a.ln_len = (int)lnbuf.size();
    buf.insert(buf.end(), lnbuf.begin(), lnbuf.end());
    for (TLayer& t : tabs) { t.n1w += a.ln0; t.n1b += a.ln0; t.n2w += a.ln0; t.n2b += a.ln0; t.n3w += a.ln0; t.n3b += a.ln0; } // (n3*: decoder layers only; unused otherwise)
    a.encn_w += a.ln0; a.encn_b += a.ln0; a.decn_w += a.ln0; a.decn_b += a.ln0;
    static_assert(sizeof(TLayer) % sizeof(float) == 0, "layer tables live in the float buffer");
    a.enc_tab = (int)buf.size();
    a.dec_tab = a.enc_tab + a.n_enc * (int)(sizeof(TLayer) / sizeof(float));
    buf.resize(buf.size() + tabs.size() * sizeof(TLayer) / sizeof(float));
    std::memcpy(buf.data() + a.enc_tab, tabs.data(), tabs.size() * sizeof(TLayer));
    if (null_seen) return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: NULL tensor pointer in model");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tfail(nullptr, DP_ERR_DEVICE, "dp_temporal_create: no HIP device (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return tfail(nullptr, DP_ERR_INVALID, "dp_temporal_create: bad device index");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return tfail(nullptr, DP_ERR_DEVICE, "dp_temporal_create: hipSetDevice failed");
    dp_temporal* t = new dp_temporal;
    t->device = device;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) t->n_cu = cu; }
    hipError_t e = hipMalloc((void**)&t->d_w, buf.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->d_w, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice);
    const int max_teams = t->n_cu / 2 > 0 ? t->n_cu / 2 : 1;
    t->xch_granule_bytes = (size_t)max_teams * 2 * XCH_GRANULES * XCH_GMAX * sizeof(f4);
    const size_t xch_bytes = t->xch_granule_bytes + (size_t)max_teams * sizeof(unsigned) + 16;
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_xch, xch_bytes);
    if (e == hipSuccess) e = hipMemset(t->d_xch, 0, xch_bytes); // (tag 0 = never written; a team's first exchange carries tag 1)
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_status, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        *t->h_status = 0;
        // a team member needs a CU's worth of LDS and registers: co-residency is sized from what the RUNTIME says fits, not from this file's arithmetic
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_temporal_kernel<2, 1, true>, NT, 0) != hipSuccess || per_cu < 1) t->teams_off = true;
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (t->d_w) (void)hipFree(t->d_w);
        if (t->d_xch) (void)hipFree(t->d_xch);
        if (t->h_status) (void)hipHostFree(t->h_status);
        delete t;
        return tfail(nullptr, DP_ERR_DEVICE, std::string("dp_temporal_create: ") + hipGetErrorString(e));
    }
    a.w = t->d_w;
    t->args = a;
    *out = t;
    return DP_OK;
}

extern "C" int dp_temporal_destroy(dp_temporal* t)
{
    if (!t) return DP_ERR_INVALID;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(t->device);
    if (t->d_w) (void)hipFree(t->d_w);
    if (t->d_xch) (void)hipFree(t->d_xch);
    if (t->h_status) (void)hipHostFree(t->h_status);
    if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
    delete t;
    return DP_OK;
}

#ifdef DPT_STAMPS
// diagnostic build only: the stamps of the launches since the last call (synchronises the device)
extern "C" int dp_temporal_debug_read_stamps(unsigned long long* out, int cap)
{
    int n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_nstamps), sizeof(int)) != hipSuccess) return -1;
    n = n < cap ? n : cap;
    if (n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) != hipSuccess) return -1;
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nstamps), &zero, sizeof(int)) != hipSuccess) return -1;
    return n;
}
#endif

// private test hook (not in include/dragposer.h; the product reads no environment variable): pin the kernel variant of later predictions
extern "C" int dp_temporal_debug_force_variant(dp_temporal* t, int variant)
{
    // (102, 104, 108, 116: a team of 2 / 4 / 8 / 16 workgroups per sequence where the launch fits the device, else as 0)
    if (!t || (variant != 0 && variant != 21 && variant != 41 && variant != 42 && variant != 44 && variant != 102 && variant != 104 && variant != 108 && variant != 116)) return DP_ERR_INVALID;
    t->forced_variant = variant;
    return DP_OK;
}

// The team size the library picks for n_seq sequences on a device of n_cu usable CUs (1: no teams).  Host arithmetic only (a CPU test holds it): the
// largest power of two up to 16 with every workgroup on a CU of its own (team members wait for each other: all of them must be resident, and the
// TEAM kernel's 86 KB of LDS allow one workgroup per CU), ALL TEAMS TOGETHER ON AT MOST HALF THE CUs (round 6: the other half is what keeps a second
// handle's teams, or another stream's kernel, from starving a member -- a launch that filled the device left no slack at all) and at least one
// feed-forward tile per wave; 16 pays with a quarter of the device at most (profiles/r05_team_latency.txt), 8 beyond.
extern "C" int dp_temporal_debug_team_size(int n_cu, int n_seq, int dim_feedforward)
{
    if (n_cu <= 0 || n_seq <= 0 || dim_feedforward <= 0) return 1;
    int G = 1;
    while (G < 16 && n_seq * (2 * G) <= n_cu / 2 && (dim_feedforward + FT - 1) / FT >= 2 * G * NWV) G *= 2;
    if (G == 16 && n_seq * 64 > n_cu) G = 8;
    return G;
}

// private test hook: make the team exchange fail on purpose -- member `member` of sequence `team`'s team never publishes its partial sums (-1: nobody),
// and a member gives up after `poll_limit` re-reads instead of ~1 s (0: the default).  What the product promises then is in "time-out" above.
extern "C" int dp_temporal_debug_team_fault(dp_temporal* t, int team, int member, int poll_limit)
{
    if (!t) return DP_ERR_INVALID;
    t->dbg_skip_team = team; t->dbg_skip_member = member;
    t->poll_limit = poll_limit > 0 ? poll_limit : XCH_POLL_LIMIT;
    return DP_OK;
}

// Health of the handle, WITHOUT a synchronise: DP_TEMPORAL_TEAM_TIMEOUT once a team member of an earlier launch has given up waiting (the device
// writes the word into page-locked host memory the moment it happens; sticky).  The targets of that launch's affected sequences are NaN.
extern "C" int dp_temporal_status(const dp_temporal* t)
{
    if (!t) return DP_ERR_INVALID;
    int v = t->status_seen;
    if (t->h_status && *(volatile const int*)t->h_status != 0) v |= DP_TEMPORAL_TEAM_TIMEOUT;
    return v;
}

// private test hook: the teams' status word (0: every exchange completed; 1: a workgroup waited XCH_POLL_LIMIT reads for its team -- the launch's
// predictions are garbage); synchronises the device
extern "C" int dp_temporal_debug_team_status(dp_temporal* t)
{
    if (!t || !t->d_xch) return -1;
    int v = -1;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&v, (char*)t->d_xch + t->xch_granule_bytes + (size_t)(t->n_cu / 2 > 0 ? t->n_cu / 2 : 1) * sizeof(unsigned), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

extern "C" int dp_temporal_predict(dp_temporal* t, int n_seq, const dp_seq_state* st, int window, float* target_buf, void* stream)
{
    if (!t) return DP_ERR_INVALID;
    if (n_seq <= 0 || !st || !target_buf) return tfail(t, DP_ERR_INVALID, "dp_temporal_predict: bad arguments");
    if (!st->latent_buf || !st->disp_buf || !st->heights_buf) return tfail(t, DP_ERR_INVALID, "dp_temporal_predict: NULL history buffer");
    const TArgs& m = t->args;
    if (st->n_heights != m.nh) return tfail(t, DP_ERR_INVALID, "dp_temporal_predict: state.n_heights differs from the model's");
    if (window < 0 || window % m.step != 0) return tfail(t, DP_ERR_INVALID, "dp_temporal_predict: window must be a non-negative multiple of sample_step");
    const int n_past = (st->history + m.step - 1) / m.step, n_steps = window / m.step + 1;
    if (st->history < 2 * m.step || n_past - 1 > MAXT || n_steps > MAXT || n_past - 1 > m.max_len || n_steps > m.max_len)
        return tfail(t, DP_ERR_UNSUPPORTED, "dp_temporal_predict: more than 32 encoder or decoder tokens (or more than max_len positions)");
    if (!t->teams_off && *(volatile const int*)t->h_status != 0) { // a team member of an EARLIER launch gave up waiting (see "time-out")
        t->teams_off = true;
        t->status_seen |= DP_TEMPORAL_TEAM_TIMEOUT;
        return tfail(t, DP_ERR_TIMEOUT, "dp_temporal_predict: a team of workgroups of an earlier launch of this handle timed out waiting for a member that was not "
                                        "resident (another stream's kernel or a CU mask held its CUs?); that launch wrote NaN into the targets of the affected "
                                        "sequences.  Nothing was launched now; the handle runs one workgroup per sequence from here on -- call again");
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != t->device && hipSetDevice(t->device) != hipSuccess) return tfail(t, DP_ERR_DEVICE, "cannot select the predictor's device");
    TArgs a = m;
    a.latent_buf = st->latent_buf; a.disp_buf = st->disp_buf; a.heights_buf = st->heights_buf;
    a.H = st->history; a.n_seq = n_seq; a.window = window; a.target = target_buf;
    // variant: few sequences -> latency (one workgroup per CU, prefetch); many -> two workgroups per CU, and two sequences per
    // workgroup when each has at most 16 tokens (every weight fetch then serves both)
    const bool pair_ok = n_past - 1 <= 16 && n_steps <= 16;
    // (44 = PAIR, one 1024-thread workgroup of two NS = 2 halves per CU that share the weight fetches of the feed-forward layers in calls over at
    //  most 8 tokens.  Measured at 1024 / 4096 sequences, profiles/r06_temporal_pair_ab.txt: window 16 (five decoder calls of 1 .. 5 tokens) -12.8 %
    //  / -12.8 %; window 0 -3.1 % / -1.2 %; window 60 -2.7 % / -2.4 % -- there most of what the shared fetches save is given back by the halves'
    //  lock-step: two independent workgroups on a CU drift apart and run one's small phases under the other's tile loop.  Taken wherever variant 42
    //  would put two workgroups on a CU anyway.)
    int variant = n_seq <= t->n_cu ? 21 : (pair_ok ? (n_seq > 2 * t->n_cu ? 44 : 42) : 41);
    // few sequences: a TEAM of G workgroups per sequence (the largest power of two up to 16 with every workgroup on a CU of its own -- they wait for
    // each other, so all of them must be resident -- and at least one feed-forward tile per wave)
    // (the CUs this launch may use: the device's, or the stream's CU mask when it has one -- hipExtStreamGetCUMask reports the effective mask)
    int n_cu = t->n_cu;
    {
        uint32_t mask[16] = {};
        if (hipExtStreamGetCUMask((hipStream_t)stream, 16, mask) == hipSuccess) {
            int bits = 0;
            for (uint32_t w : mask) bits += __builtin_popcount(w);
            if (bits > 0 && bits < n_cu) n_cu = bits;
        } else (void)hipGetLastError(); // (not an error of this call)
    }
    int G = t->teams_off ? 1 : dp_temporal_debug_team_size(n_cu, n_seq, m.ff);
    if (t->forced_variant >= 100 && !t->teams_off) { // (a forced size: the largest the launch fits -- here the whole device may be used --, whatever pays)
        G = 1;
        while (G < 16 && n_seq * (2 * G) <= n_cu && (m.ff + FT - 1) / FT >= 2 * G * NWV) G *= 2;
        const int want = t->forced_variant - 100;
        G = G >= want ? want : 1;
    }
    if (G >= 2 && (t->forced_variant == 0 || t->forced_variant >= 100)) variant = 100 + G;
    if (t->forced_variant == 21 || t->forced_variant == 41 || ((t->forced_variant == 42 || t->forced_variant == 44) && pair_ok)) variant = t->forced_variant;
    if (variant >= 100) {
        const int max_teams = t->n_cu / 2 > 0 ? t->n_cu / 2 : 1;
        a.G = G; a.xch = t->d_xch; a.epochs = (unsigned*)((char*)t->d_xch + t->xch_granule_bytes); a.tstatus = (int*)(a.epochs + max_teams);
        a.hstatus = t->h_status; a.poll_limit = t->poll_limit; a.dbg_skip_team = t->dbg_skip_team; a.dbg_skip_member = t->dbg_skip_member;
        hipLaunchKernelGGL((dp_temporal_kernel<2, 1, true>), dim3(n_seq * G), dim3(NT), 0, (hipStream_t)stream, a);
    } else if (variant == 21) hipLaunchKernelGGL((dp_temporal_kernel<2, 1>), dim3(n_seq), dim3(NT), 0, (hipStream_t)stream, a);
    else if (variant == 41) hipLaunchKernelGGL((dp_temporal_kernel<4, 1>), dim3(n_seq), dim3(NT), 0, (hipStream_t)stream, a);
    else if (variant == 44) hipLaunchKernelGGL((dp_temporal_kernel<4, 2, false, true>), dim3((n_seq + 3) / 4), dim3(2 * NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dp_temporal_kernel<4, 2>), dim3((n_seq + 1) / 2), dim3(NT), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return tfail(t, DP_ERR_LAUNCH, std::string("dp_temporal_predict: ") + hipGetErrorString(e));
    return DP_OK;
}

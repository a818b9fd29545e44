// coop_probe.hip -- round-4 probe behind "a CU-cooperative kernel for <= 4096 frames" (DESIGN.md 5.3).
//
// At 4096 frames an MI355X has 16 frames per CU.  dp_w4 gives each of the CU's four waves 4 of them, start to finish (fp32 MFMA,
// which blocks the vector ALU); dp_w16's arithmetic (v_mfma_f32_16x16x32_bf16 in three-term split precision, 16 frames = one tile
// dimension) needs all 16 frames in one wave and leaves three SIMDs idle.  The cooperative shape: ONE workgroup of four waves owns
// the 16 frames; every product's OUTPUT TILES are dealt to the waves (L0: 3 tiles, L1: 4, L2: 6 -> one or two per wave), each wave
// keeps the weights of its own tiles in registers (no weight image in LDS), applies bias / LeakyReLU / the three-term split to its
// own tiles only, and PUBLISHES the bf16 terms through LDS in the next product's B-operand layout (dp_w16.h: a pair of output
// tiles IS the next K-block); one workgroup barrier per layer.
//
// What this probe measures (s_memtime, per iteration of a dependent loop, steady state):
//   coop   forward L0 -> L1 -> L2 on four waves with the three exchanges (activations L0 -> L1, L1 -> L2, and "next latent" -> L0),
//          stamped per phase (matrix / element-wise + publish / barrier + operand read) and unstamped
//   mono   the same arithmetic in ONE wave (dp_w16's shape: every tile in the wave, nothing exchanged), weights in registers
//   xonly  coop with the MFMAs removed: what the exchanges alone cost
// and checks that coop == mono bit for bit (same accumulation order per tile).  The numbers go next to dp_w4's stamped forward
// (profiles/r03_phase_cycles.txt: L0 926 + L1 508 + L2 1292 = 2726 cycles per iteration for 16 frames per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/coop_probe tools/ubench/coop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__

// products: input K-blocks / output tiles (dp_w16.h)
constexpr int NT[3] = {3, 4, 6};
constexpr int NKB[3] = {1, 2, 2};
// weight image: [layer][tile][kb][term][lane] u4; tiles padded to 4 / 4 / 8 so that every wave finds a tile (dummy: zeros)
constexpr int TPAD[3] = {4, 4, 8};
constexpr int W_OFF[3] = {0, 4 * 1 * 3, 4 * 1 * 3 + 4 * 2 * 3};
constexpr int W_U4 = (4 * 1 + 4 * 2 + 8 * 2) * 3; // per lane

struct B3 { u4 t[3]; };
struct W3 { u4 h, m, l; };
DEV unsigned cvt_pk(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{lo, hi}, bf2)); }
DEV void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l)
{
    h = cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk(s0, s1);
}
struct T3 { u2 t[3]; }; // the three terms of ONE tile (half a K-block)
DEV T3 split_tile(f4 x)
{
    unsigned h[2], m[2], l[2];
    split_pair(x.x, x.y, h[0], m[0], l[0]);
    split_pair(x.z, x.w, h[1], m[1], l[1]);
    T3 r;
    r.t[0] = u2{h[0], h[1]}; r.t[1] = u2{m[0], m[1]}; r.t[2] = u2{l[0], l[1]};
    return r;
}
DEV B3 join(const T3& a, const T3& b)
{
    B3 r;
    for (int t = 0; t < 3; ++t) r.t[t] = u4{a.t[t].x, a.t[t].y, b.t[t].x, b.t[t].y};
    return r;
}
template <bool MM> DEV f4 mm(u4 a, u4 b, f4 c)
{
    if constexpr (MM) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    else return f4{__uint_as_float(__float_as_uint(c.x) ^ (a.x & 1u)), c.y, c.z, __uint_as_float(__float_as_uint(c.w) ^ (b.w & 1u))}; // (keeps the dependencies: one VALU op per operand)
}
template <bool MM> DEV f4 pair_mm(f4 acc, const W3& w, const B3& b)
{ // six term products, small ones first (dp_w16_impl.h)
    acc = mm<MM>(w.l, b.t[0], acc);
    acc = mm<MM>(w.m, b.t[1], acc);
    acc = mm<MM>(w.h, b.t[2], acc);
    acc = mm<MM>(w.m, b.t[0], acc);
    acc = mm<MM>(w.h, b.t[1], acc);
    acc = mm<MM>(w.h, b.t[0], acc);
    return acc;
}
DEV f4 lrelu(f4 x)
{
    const f4 t = {x.x * 0x1p127f, x.y * 0x1p127f, x.z * 0x1p127f, x.w * 0x1p127f};
    return f4{x.x * __builtin_amdgcn_fmed3f(t.x, 0.2f, 1.f), x.y * __builtin_amdgcn_fmed3f(t.y, 0.2f, 1.f), x.z * __builtin_amdgcn_fmed3f(t.z, 0.2f, 1.f),
              x.w * __builtin_amdgcn_fmed3f(t.w, 0.2f, 1.f)};
}
DEV W3 wload(const u4* img, int layer, int tile, int kb, int lane)
{
    const u4* p = img + ((size_t)(W_OFF[layer] + (tile * NKB[layer] + kb) * 3)) * 64 + lane;
    return W3{p[0], p[64], p[128]};
}
DEV f4 next_latent(f4 z, f4 y) { return f4{z.x + 1e-3f * y.x, z.y + 1e-3f * y.y, z.z + 1e-3f * y.z, z.w + 1e-3f * y.w}; } // stands for kinematics + backward + Adam

// LDS exchange areas, in u4 units per lane: [area][kb][term][lane]; area 0 = latent (1 K-block), 1 = a0 (2), 2 = a1 (2)
constexpr int X_OFF[3] = {0, 3 * 64, 3 * 64 + 6 * 64};
constexpr int X_U4 = 3 * 64 + 6 * 64 + 6 * 64;
DEV void publish(u4* lds, int area, int tile, int lane, const T3& s)
{ // tile n is half (n & 1) of K-block n >> 1: two dwords per term
    u2* p = (u2*)(lds + X_OFF[area] + ((tile >> 1) * 3) * 64 + lane) + (tile & 1);
    p[0] = s.t[0]; p[2 * 64] = s.t[1]; p[4 * 64] = s.t[2];
}
DEV B3 operand(const u4* lds, int area, int kb, int lane)
{
    const u4* p = lds + X_OFF[area] + (kb * 3) * 64 + lane;
    return B3{{p[0], p[64], p[128]}};
}

#define STAMP(k) do { if (STAMPS) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (it >= 8) acc_cyc[k] += t_ - t_prev; t_prev = t_; } } while (0)

// ---------------------------------------------------------------- cooperative: four waves, 16 frames
template <bool MM, bool STAMPS> __global__ __launch_bounds__(256, 1) void k_coop(const u4* img, const f4* z0, f4* yout, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) u4 lds[X_U4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // my tiles: L0 tile `wave` (wave 3: a zero tile, the pad of a0's second K-block), L1 tile `wave`, L2 tiles `wave` and `wave + 4`
    const W3 w0 = wload(img, 0, wave, 0, lane);
    const W3 w1a = wload(img, 1, wave, 0, lane), w1b = wload(img, 1, wave, 1, lane);
    const W3 w2a = wload(img, 2, wave, 0, lane), w2b = wload(img, 2, wave, 1, lane);
    const W3 w2c = wload(img, 2, wave + 4, 0, lane), w2d = wload(img, 2, wave + 4, 1, lane);
    const f4 bias = {0.01f * (lane & 15), -0.02f, 0.03f, 0.005f * wave};
    f4 z = z0[(blockIdx.x * 4 + (wave & 1)) * 64 + lane]; // waves 0, 1 own the latent's two tiles (2, 3: copies, never published)
    if (wave < 2) publish(lds, 0, wave, lane, split_tile(z));
    __syncthreads();
    unsigned long long acc_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, t_loop = 0;
    f4 y0 = {0, 0, 0, 0}, y1 = y0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t_loop = __builtin_amdgcn_s_memtime();
        if (STAMPS) t_prev = __builtin_amdgcn_s_memtime();
        const B3 bz = operand(lds, 0, 0, lane);
        STAMP(0); // operand read (latent)
        f4 a0 = pair_mm<MM>(bias, w0, bz);
        STAMP(1); // L0 matrix
        publish(lds, 1, wave, lane, split_tile(lrelu(a0)));
        STAMP(2); // L0 element-wise + publish
        __syncthreads();
        const B3 b00 = operand(lds, 1, 0, lane), b01 = operand(lds, 1, 1, lane);
        STAMP(3); // barrier + operand read
        f4 a1 = pair_mm<MM>(pair_mm<MM>(bias, w1a, b00), w1b, b01);
        STAMP(1);
        publish(lds, 2, wave, lane, split_tile(lrelu(a1)));
        STAMP(2);
        __syncthreads();
        const B3 b10 = operand(lds, 2, 0, lane), b11 = operand(lds, 2, 1, lane);
        STAMP(3);
        y0 = pair_mm<MM>(pair_mm<MM>(bias, w2a, b10), w2b, b11);
        y1 = pair_mm<MM>(pair_mm<MM>(bias, w2c, b10), w2d, b11);
        STAMP(4); // L2 matrix (two tiles)
        z = next_latent(z, y0);
        if (wave < 2) publish(lds, 0, wave, lane, split_tile(z));
        STAMP(5); // stand-in for the rest of the iteration + publish of the next latent
        __syncthreads();
        STAMP(6); // barrier
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    yout[((blockIdx.x * 4 + wave) * 2 + 0) * 64 + lane] = y0;
    yout[((blockIdx.x * 4 + wave) * 2 + 1) * 64 + lane] = y1;
    if (lane == 0) {
        unsigned long long* c = cyc + (blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 7; ++k) c[k] = acc_cyc[k];
        c[7] = t_end - t_loop;
    }
}

// ---------------------------------------------------------------- mono: one wave does every tile (dp_w16's shape)
__global__ __launch_bounds__(64, 1) void k_mono(const u4* img, const f4* z0, f4* yout, int iters, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    W3 w0[3], w1[4][2], w2[6][2];
    for (int n = 0; n < 3; ++n) w0[n] = wload(img, 0, n, 0, lane);
    for (int n = 0; n < 4; ++n) for (int k = 0; k < 2; ++k) w1[n][k] = wload(img, 1, n, k, lane);
    for (int n = 0; n < 6; ++n) for (int k = 0; k < 2; ++k) w2[n][k] = wload(img, 2, n, k, lane);
    f4 z[2] = {z0[(blockIdx.x * 4 + 0) * 64 + lane], z0[(blockIdx.x * 4 + 1) * 64 + lane]};
    f4 y[6];
    unsigned long long t_loop = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t_loop = __builtin_amdgcn_s_memtime();
        const B3 bz = join(split_tile(z[0]), split_tile(z[1]));
        T3 s0[4], s1[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { // (tile 3 of a0 = the zero pad: its weights are zeros, its wave-3 bias is not -- as in coop)
            const f4 bias = {0.01f * (lane & 15), -0.02f, 0.03f, 0.005f * n};
            const W3 w = n < 3 ? w0[n] : wload(img, 0, 3, 0, lane);
            s0[n] = split_tile(lrelu(pair_mm<true>(bias, w, bz)));
        }
        const B3 b00 = join(s0[0], s0[1]), b01 = join(s0[2], s0[3]);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f4 bias = {0.01f * (lane & 15), -0.02f, 0.03f, 0.005f * n};
            s1[n] = split_tile(lrelu(pair_mm<true>(pair_mm<true>(bias, w1[n][0], b00), w1[n][1], b01)));
        }
        const B3 b10 = join(s1[0], s1[1]), b11 = join(s1[2], s1[3]);
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const f4 bias = {0.01f * (lane & 15), -0.02f, 0.03f, 0.005f * (n & 3)};
            y[n] = pair_mm<true>(pair_mm<true>(bias, w2[n][0], b10), w2[n][1], b11);
        }
        z[0] = next_latent(z[0], y[0]);
        z[1] = next_latent(z[1], y[1]);
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < 6; ++n) yout[(blockIdx.x * 8 + n) * 64 + lane] = y[n];
    if (lane == 0) cyc[blockIdx.x] = t_end - t_loop;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 58;
    std::vector<unsigned> img((size_t)W_U4 * 64 * 4);
    srand(7);
    // random bf16 pairs of moderate size; zero tiles where a product has fewer tiles than the padding
    auto bf = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    for (int layer = 0; layer < 3; ++layer)
        for (int tile = 0; tile < TPAD[layer]; ++tile)
            for (int kb = 0; kb < NKB[layer]; ++kb)
                for (int term = 0; term < 3; ++term)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const float sc = term == 0 ? 0.3f : term == 1 ? 0.3f / 256 : 0.3f / 65536;
                            const float a = sc * (rand() / (float)RAND_MAX - 0.5f), b = sc * (rand() / (float)RAND_MAX - 0.5f);
                            const bool real = tile < NT[layer];
                            img[((size_t)(W_OFF[layer] + (tile * NKB[layer] + kb) * 3 + term) * 64 + lane) * 4 + j] = real ? (bf(a) | (bf(b) << 16)) : 0u;
                        }
    std::vector<float> z0((size_t)blocks * 4 * 64 * 4);
    for (auto& v : z0) v = 0.6f * (rand() / (float)RAND_MAX - 0.5f);
    unsigned* d_img; float *d_z, *d_y1, *d_y2; unsigned long long *d_c1, *d_c2;
    CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_z, z0.size() * 4));
    CK(hipMalloc(&d_y1, (size_t)blocks * 8 * 64 * 16)); CK(hipMalloc(&d_y2, (size_t)blocks * 8 * 64 * 16));
    CK(hipMalloc(&d_c1, (size_t)blocks * 4 * 8 * 8)); CK(hipMalloc(&d_c2, (size_t)blocks * 8));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_z, z0.data(), z0.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned long long> c1((size_t)blocks * 32), c2(blocks);
    std::vector<float> y1((size_t)blocks * 8 * 64 * 4), y2(y1.size());
    const int n = iters - 8;
    auto report = [&](const char* name, bool stamps) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c1.data(), d_c1, c1.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0, ph[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { tot += c1[(b * 4 + w) * 8 + 7]; for (int k = 0; k < 7; ++k) ph[k] += c1[(b * 4 + w) * 8 + k]; }
        const double den = (double)blocks * 4 * n;
        printf("%-28s %7.0f cycles per iteration (16 frames, forward L0 + L1 + L2 + three exchanges; mean over %d workgroups x 4 waves)\n", name, tot / den, blocks);
        if (stamps)
            printf("    per phase: latent operand read %.0f | L0 + L1 matrix %.0f | element-wise + split + publish (x2) %.0f | barrier + operand read (x2) %.0f | L2 matrix, two tiles %.0f | "
                   "next latent + publish %.0f | closing barrier %.0f\n", ph[0] / den, ph[1] / den, ph[2] / den, ph[3] / den, ph[4] / den, ph[5] / den, ph[6] / den);
    };
    for (int rep = 0; rep < 2; ++rep) { // (first round: warm-up)
        hipLaunchKernelGGL((k_coop<true, false>), dim3(blocks), dim3(256), 0, 0, (const u4*)d_img, (const f4*)d_z, (f4*)d_y1, iters, d_c1);
        if (rep) report("coop (unstamped)", false);
        hipLaunchKernelGGL((k_coop<true, true>), dim3(blocks), dim3(256), 0, 0, (const u4*)d_img, (const f4*)d_z, (f4*)d_y2, iters, d_c1);
        if (rep) report("coop (stamped)", true);
        hipLaunchKernelGGL((k_coop<false, false>), dim3(blocks), dim3(256), 0, 0, (const u4*)d_img, (const f4*)d_z, (f4*)d_y2, iters, d_c1);
        if (rep) report("xonly (no MFMAs)", false);
        hipLaunchKernelGGL(k_mono, dim3(blocks), dim3(64), 0, 0, (const u4*)d_img, (const f4*)d_z, (f4*)d_y2, iters, d_c2);
        CK(hipDeviceSynchronize());
        if (rep) {
            CK(hipMemcpy(c2.data(), d_c2, c2.size() * 8, hipMemcpyDeviceToHost));
            double tot = 0;
            for (auto v : c2) tot += v;
            printf("%-28s %7.0f cycles per iteration (the same arithmetic in ONE wave, weights in registers, nothing exchanged)\n", "mono", tot / ((double)blocks * n));
        }
    }
    CK(hipMemcpy(y1.data(), d_y1, y1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y2.data(), d_y2, y2.size() * 4, hipMemcpyDeviceToHost));
    // coop stores [block][wave][2][lane] = tiles wave, wave + 4; mono [block][8][lane] = tiles 0..5
    size_t bad = 0, cnt = 0;
    for (int b = 0; b < blocks; ++b)
        for (int t = 0; t < 6; ++t)
            for (int l = 0; l < 64 * 4; ++l) {
                const int w = t & 3, h = t >> 2;
                bad += memcmp(&y1[(((size_t)b * 4 + w) * 2 + h) * 256 + l], &y2[((size_t)b * 8 + t) * 256 + l], 4) != 0;
                ++cnt;
            }
    printf("coop == mono bit for bit on %zu of %zu outputs after %d iterations%s\n", cnt - bad, cnt, iters, bad ? "  ** MISMATCH **" : "");
    printf("beside it: dp_w4's forward on the same 16 frames per CU, stamped (profiles/r03_phase_cycles.txt): L0 926 + L1 508 + L2 1292 = 2726 cycles per iteration\n");
    return bad != 0;
}

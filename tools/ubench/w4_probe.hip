// w4_probe.hip -- micro-probes behind the wave-private ("w4") kernel design:
//   1. operand / result lane maps of v_mfma_f32_4x4x1_16b_f32, with and without the A-block broadcast (CBSZ/ABID)
//   2. its issue rate on one wave per SIMD: 1 / 2 / 4 accumulators, weights from registers or streamed from LDS
//   3. lone-wave issue cost of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (1 and 2 waves per SIMD)
//   4. the 4x4 register<->lane transpose inside lane quads (two DPP + select stages): result check and cost
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/w4_probe tools/ubench/w4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__
#include <utility>
template <class F, int... I> DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---------------------------------------------------------------- 1. lane maps
__global__ void k_layout(float* out)
{
    const int l = threadIdx.x;
    const float a = (float)(l + 1), b = (float)(100 * (l + 1));
    f4 c = {0.f, 0.f, 0.f, 0.f};
    f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0); // every block takes A from block 3
    f4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 13, 0);
    f4 d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, 1, 0); // groups of 4 blocks take A from the group's block 1
    for (int r = 0; r < 4; ++r) {
        out[(0 * 64 + l) * 4 + r] = d0[r];
        out[(1 * 64 + l) * 4 + r] = d1[r];
        out[(2 * 64 + l) * 4 + r] = d2[r];
        out[(3 * 64 + l) * 4 + r] = d3[r];
    }
}

// ---------------------------------------------------------------- 2. MFMA rate
// NACC accumulators, 96 steps per pass; weights (B operand) either loop-invariant registers (LDSW = 0) or streamed from
// LDS with one ds_read_b128 per four steps (LDSW = 1); A operand = 4 registers, ABID walks the 16 blocks.
template <int K> struct Step {
    template <int NACC> static DEV void run(f4 (&acc)[NACC], const f4& x, const f4 (&w)[24])
    {
        constexpr int g = K >> 2, m = K & 3;
        acc[K % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[m], w[g][m], acc[K % NACC], 4, g & 15, 0);
        if constexpr (K + 1 < 96) Step<K + 1>::template run<NACC>(acc, x, w);
    }
};

template <int NACC, int LDSW> __global__ __launch_bounds__(256, 1) void k_mfma_rate(float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float wl[24 * 64 * 4];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 24 * 64 * 4; i += blockDim.x) wl[i] = 1e-3f * (float)(i % 97);
    __syncthreads();
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    f4 x = {1.f + l, 2.f, 3.f, 4.f};
    f4 w[24];
#pragma unroll
    for (int g = 0; g < 24; ++g) w[g] = *(const f4*)(wl + (g * 64 + l) * 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (LDSW) { // weights streamed from LDS two groups (8 steps) ahead of their use
            int o4 = l;
            asm volatile("" : "+v"(o4)); // opaque per iteration: the reads cannot be hoisted out of the loop
            const f4* wp = (const f4*)wl + o4;
            f4 wa = wp[0], wb = wp[64];
            static_for<12>([&](auto gi) {
                constexpr int g = 2 * decltype(gi)::value;
                f4 na = wa, nb = wb;
                if (g + 2 < 24) { na = wp[(g + 2) * 64]; nb = wp[(g + 3) * 64]; }
                static_for<4>([&](auto mi) { constexpr int m = decltype(mi)::value; acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[m], wa[m], acc[m % NACC], 4, g & 15, 0); });
                static_for<4>([&](auto mi) { constexpr int m = decltype(mi)::value; acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[m], wb[m], acc[m % NACC], 4, (g + 1) & 15, 0); });
                wa = na; wb = nb;
            });
        } else {
            Step<0>::template run<NACC>(acc, x, w);
        }
        x[0] += acc[0][0] * 1e-30f; // keep the iterations dependent, as the real chain is
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------- 2b. the bf16 sibling: v_mfma_f32_4x4x4_16b_bf16 (K = 4 per instruction)
typedef short s4v __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(256, 1) void k_bf16_rate(float* out, int iters, unsigned long long* cyc)
{
    const int l = threadIdx.x & 63;
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    s4v x = {(short)(0x3f80 + l), 0x3f80, 0x4000, 0x4040};
    s4v w[24];
#pragma unroll
    for (int g = 0; g < 24; ++g) w[g] = s4v{(short)(0x3c00 + g), (short)(0x3c10 + l), 0x3c20, 0x3c30};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<96>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x, w[k % 24], acc[k % NACC], 4, k & 15, 0);
        });
        x[0] += (short)(acc[0][0] * 1e-30f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------- 2c. can the VALU run under the matrix pipe?  One wave per SIMD:
// a chain of 4x4x1 MFMAs on two accumulators with NV independent v_pk_fma_f32 (four separate dependency chains) issued
// behind every MFMA -- the question behind software-pipelining two frame groups in one wave
template <int ABID> DEV void ov_mfma(f4& acc, float x, float w) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(acc) : "v"(x), "v"(w), "i"(ABID)); }
DEV void ov_valu(f2& v, f2 m, f2 c) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c)); }
template <int NV> __global__ __launch_bounds__(256, 1) void k_overlap(float* out, int iters, unsigned long long* cyc)
{
    const int l = threadIdx.x & 63;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float x = 1.f + l, w0 = 1e-3f * l, w1 = 2e-3f * l;
    f2 v[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<48>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            ov_mfma<(k & 15)>(acc0, x, w0);
            if constexpr (NV >= 1) ov_valu(v[0], m, c);
            if constexpr (NV >= 2) ov_valu(v[1], m, c);
            ov_mfma<(k & 15)>(acc1, x, w1);
            if constexpr (NV >= 1) ov_valu(v[2], m, c);
            if constexpr (NV >= 2) ov_valu(v[3], m, c);
        });
        x += acc0[0] * 1e-30f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + v[0].x + v[1].y + v[2].x + v[3].y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------- 3. VALU issue: scalar vs packed fp32
template <int MODE, int ILP> __global__ void k_valu(float* out, int iters, unsigned long long* cyc)
{
    f2 a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------- 4. quad transpose
// lane 4q+i holds r[0..3] = T[i][0..3]; afterwards r[j] = T[j][i]
DEV void quad_transpose(f4& r, int lane)
{
    const bool p1 = lane & 1, p2 = lane & 2;
    auto dpp = [](float v, int ctrl) {
        return __int_as_float(ctrl == 0xB1 ? __builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)
                                           : __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    };
    { // exchange with lane ^ 1: (r0, r1) and (r2, r3)
        const float x0 = dpp(r[1], 0xB1), x1 = dpp(r[0], 0xB1), x2 = dpp(r[3], 0xB1), x3 = dpp(r[2], 0xB1);
        r = f4{p1 ? x0 : r[0], p1 ? r[1] : x1, p1 ? x2 : r[2], p1 ? r[3] : x3};
    }
    { // exchange with lane ^ 2: (r0, r2) and (r1, r3)
        const float x0 = dpp(r[2], 0x4E), x2 = dpp(r[0], 0x4E), x1 = dpp(r[3], 0x4E), x3 = dpp(r[1], 0x4E);
        r = f4{p2 ? x0 : r[0], p2 ? x1 : r[1], p2 ? r[2] : x2, p2 ? r[3] : x3};
    }
}

// the same with the DPP folded into the select (hipcc does not combine them): 8 VALU + 4 SALU
DEV void quad_transpose_asm(f4& r, unsigned long long even1, unsigned long long even2)
{
    float n0, n1, n2, n3;
    asm volatile("s_mov_b64 vcc, %8\n\t"
                 "s_nop 0\n\t" // a VALU write of an input just ahead of the statement -> DPP read: 2 wait states
                 "v_cndmask_b32_dpp %0, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %2, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %1, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %3, %6, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "s"(even1)
                 : "vcc");
    float m0, m1, m2, m3;
    asm volatile("s_mov_b64 vcc, %8\n\t"
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

// the same on the matrix pipe: D_b[i][j] = sum_r A_r[i] * e_r[j]  (exact: products with 1, sums with 0)
DEV void quad_transpose_mfma(f4& r, const f4& e)
{
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(r[0], e[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(r[1], e[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(r[2], e[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(r[3], e[3], d, 0, 0, 0);
    r = d;
}

template <int MODE> __global__ void k_transpose(float* out, int iters, unsigned long long* cyc)
{
    const int l = threadIdx.x;
    f4 r = {(float)(l * 4 + 0), (float)(l * 4 + 1), (float)(l * 4 + 2), (float)(l * 4 + 3)};
    const unsigned long long even1 = 0x5555555555555555ull, even2 = 0x3333333333333333ull;
    const f4 e = {(l & 3) == 0 ? 1.f : 0.f, (l & 3) == 1 ? 1.f : 0.f, (l & 3) == 2 ? 1.f : 0.f, (l & 3) == 3 ? 1.f : 0.f};
    if (MODE == 0) quad_transpose(r, l);
    if (MODE == 1) quad_transpose_asm(r, even1, even2);
    if (MODE == 2) quad_transpose_mfma(r, e);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
    f4 s = r;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) quad_transpose(s, l);
        if (MODE == 1) quad_transpose_asm(s, even1, even2);
        if (MODE == 2) quad_transpose_mfma(s, e);
        s[0] += 1.f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[256 + l] = s[0] + s[1] + s[2] + s[3];
    if (l == 0) *cyc = t1 - t0;
}

// ----------------------------------------------------------------
int main()
{
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 22);
    hipMalloc(&cyc, 8);
    unsigned long long h;

    { // 1. lane maps: find (block, row, col) hypotheses that match
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
        std::vector<float> d(4 * 64 * 4);
        hipMemcpy(d.data(), out, d.size() * 4, hipMemcpyDeviceToHost);
        auto A = [](int lane) { return (double)(lane + 1); };
        auto B = [](int lane) { return 100.0 * (lane + 1); };
        int bad[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int b = l >> 2, j = l & 3;
                // hypothesis: D_b[i = r][j] in lane 4b + j register r;  A_b[i] in lane 4b + i;  B_b[j] in lane 4b + j
                const double e0 = A(4 * b + r) * B(4 * b + j);
                const double e1 = A(4 * 3 + r) * B(4 * b + j);
                const double e2 = A(4 * 13 + r) * B(4 * b + j);
                const double e3 = A(4 * ((b & ~3) + 1) + r) * B(4 * b + j);
                bad[0] += d[(0 * 64 + l) * 4 + r] != (float)e0;
                bad[1] += d[(1 * 64 + l) * 4 + r] != (float)e1;
                bad[2] += d[(2 * 64 + l) * 4 + r] != (float)e2;
                bad[3] += d[(3 * 64 + l) * 4 + r] != (float)e3;
            }
        printf("4x4x1 lane map  D_b[r][j] @ lane 4b+j reg r = A(lane 4b+r) * B(lane 4b+j): mismatches cbsz0 %d, cbsz4/abid3 %d, cbsz4/abid13 %d, cbsz2/abid1 %d\n",
               bad[0], bad[1], bad[2], bad[3]);
        if (bad[0] || bad[1] || bad[2] || bad[3]) {
            printf(" raw lane 5 (b=1,j=1): cbsz0 %.0f %.0f %.0f %.0f | abid3 %.0f %.0f %.0f %.0f | cbsz2 %.0f %.0f %.0f %.0f\n", d[5 * 4], d[5 * 4 + 1],
                   d[5 * 4 + 2], d[5 * 4 + 3], d[(64 + 5) * 4], d[(64 + 5) * 4 + 1], d[(64 + 5) * 4 + 2], d[(64 + 5) * 4 + 3],
                   d[(3 * 64 + 5) * 4], d[(3 * 64 + 5) * 4 + 1], d[(3 * 64 + 5) * 4 + 2], d[(3 * 64 + 5) * 4 + 3]);
        }
    }

    const int iters = 500;
#define RUN_MFMA(NACC, LDSW, BLOCKS, THREADS)                                                                               \
    hipLaunchKernelGGL((k_mfma_rate<NACC, LDSW>), dim3(BLOCKS), dim3(THREADS), 0, 0, out, iters, cyc);                       \
    hipDeviceSynchronize();                                                                                                 \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                                           \
    printf("mfma 4x4x1: %d acc, weights %s, %d blocks x %d waves: %.2f cycles per MFMA\n", NACC, LDSW ? "LDS" : "reg", BLOCKS, \
           THREADS / 64, (double)h / (iters * 96.0));
    RUN_MFMA(1, 0, 1, 64) RUN_MFMA(2, 0, 1, 64) RUN_MFMA(4, 0, 1, 64)
    RUN_MFMA(1, 0, 256, 256) RUN_MFMA(2, 0, 256, 256) RUN_MFMA(4, 0, 256, 256)
    RUN_MFMA(1, 1, 256, 256) RUN_MFMA(2, 1, 256, 256) RUN_MFMA(4, 1, 256, 256)
    RUN_MFMA(2, 1, 1, 64)
#define RUN_BF16(NACC, BLOCKS, THREADS)                                                                                     \
    hipLaunchKernelGGL((k_bf16_rate<NACC>), dim3(BLOCKS), dim3(THREADS), 0, 0, out, iters, cyc);                             \
    hipDeviceSynchronize();                                                                                                 \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                                           \
    printf("mfma 4x4x4 bf16 (K = 4): %d acc, %d blocks x %d waves: %.2f cycles per MFMA\n", NACC, BLOCKS, THREADS / 64, (double)h / (iters * 96.0));
    RUN_BF16(1, 1, 64) RUN_BF16(2, 1, 64) RUN_BF16(2, 256, 256) RUN_BF16(4, 256, 256)

#define RUN_OVL(NV)                                                                                                         \
    hipLaunchKernelGGL((k_overlap<NV>), dim3(256), dim3(256), 0, 0, out, iters, cyc);                                       \
    hipDeviceSynchronize();                                                                                                 \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                                           \
    printf("mfma 4x4x1 chain with %d independent v_pk_fma_f32 behind every MFMA (256 blocks x 4 waves): %.2f cycles per MFMA\n", NV, (double)h / (iters * 96.0));
    RUN_OVL(0) RUN_OVL(1) RUN_OVL(2)

    const int vit = 2000;
    const char* names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32"};
#define RUN_VALU(MODE, ILP, WAVES)                                                                                       \
    hipLaunchKernelGGL((k_valu<MODE, ILP>), dim3(256), dim3(64 * WAVES), 0, 0, out, vit, cyc);                            \
    hipDeviceSynchronize();                                                                                              \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                                        \
    printf("valu %-13s ILP %d, %d waves/CU: %.2f cycles per instruction per wave\n", names[MODE], ILP, WAVES, (double)h / (vit * 16.0 * ILP));
#define RUN_VALU_ALL(MODE) RUN_VALU(MODE, 1, 4) RUN_VALU(MODE, 4, 4) RUN_VALU(MODE, 8, 4) RUN_VALU(MODE, 1, 8) RUN_VALU(MODE, 8, 8) RUN_VALU(MODE, 8, 16)
    RUN_VALU_ALL(0) RUN_VALU_ALL(1) RUN_VALU_ALL(2) RUN_VALU_ALL(3) RUN_VALU_ALL(4)

    { // 3b. chip-wide VALU rate by wall clock (HIP events): 256 blocks x 16 waves, ILP 8
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            const int it = 20000;
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL((k_valu<0, 8>), dim3(256), dim3(1024), 0, 0, out, it, cyc);
            else hipLaunchKernelGGL((k_valu<1, 8>), dim3(256), dim3(1024), 0, 0, out, it, cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double instr = 256.0 * 16 * it * 16.0 * 8; // wave-instructions
            const double flop = instr * 64 * 2 * (mode ? 2 : 1);
            printf("wall clock %s: %.3f ms, %.1f TFLOP/s, %.2f wave-instr per ns chip-wide, s_memtime ticks %.0f -> %.3f GHz tick rate\n", names[mode], ms,
                   flop / ms * 1e-9, instr / ms * 1e-6, (double)h, (double)h / ms * 1e-6);
        }
    }
    for (int mode = 0; mode < 3; ++mode) { // 4. transpose
        if (mode == 0) hipLaunchKernelGGL(k_transpose<0>, dim3(1), dim3(64), 0, 0, out, 1000, cyc);
        if (mode == 1) hipLaunchKernelGGL(k_transpose<1>, dim3(1), dim3(64), 0, 0, out, 1000, cyc);
        if (mode == 2) hipLaunchKernelGGL(k_transpose<2>, dim3(1), dim3(64), 0, 0, out, 1000, cyc);
        hipDeviceSynchronize();
        std::vector<float> d(256);
        hipMemcpy(d.data(), out, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) bad += d[l * 4 + j] != (float)(((l & ~3) + j) * 4 + (l & 3));
        printf("quad transpose (%s): %d mismatches, %.1f cycles per transpose (dependent, one wave)\n", mode == 0 ? "mov_dpp + cndmask" : mode == 1 ? "cndmask_dpp asm" : "mfma", bad, (double)h / 1000.0);
    }
    return 0;
}

// pair_probe.hip -- round-3 probes behind "can anything overlap the matrix pipe of the optimise kernel?" (DESIGN.md 5.3):
//   A. SELF: one wave per SIMD, a dependent-free MFMA chain on two accumulators with NV independent fillers behind every
//      MFMA, for three shapes: v_mfma_f32_4x4x1_16b_f32 (the product's instruction, 8 cycles), v_mfma_f32_16x16x4_f32
//      (32 cycles, 16 frames per wave) and v_mfma_f32_16x16x32_bf16 (16 cycles); fillers v_fma_f32 / v_pk_fma_f32 /
//      ds_read_b128.  r02's probe used the packed form only, which the hardware notes call an anti-lever beside MFMAs.
//   B. PAIR: two waves per SIMD (512-thread workgroups; waves w and w + 4 share a SIMD -- checked through HW_ID), waves 0..3
//      issue the MFMA chain, waves 4..7 a vector stream (the instruction mix of kinematics stage T: fma / pk_fma / mul,
//      ILP 4); each role timed alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/pair_probe tools/ubench/pair_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s8v __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__
template <class F, int... I> DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

enum { S_4x4x1 = 0, S_16x16x4 = 1, S_BF16 = 2 };
enum { F_FMA = 0, F_PKFMA = 1, F_LDS = 2, F_MIX = 3 };

template <int SHAPE, int ABID> DEV void mfma(f4& acc, float x, float w, const s8v& xb, const s8v& wb)
{
    if constexpr (SHAPE == S_4x4x1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(acc) : "v"(x), "v"(w), "i"(ABID));
    if constexpr (SHAPE == S_16x16x4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(w));
    if constexpr (SHAPE == S_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(xb), "v"(wb));
}

struct Fill {
    f2 v[8];
    f4 l[2];
};
template <int FILL, int I> DEV void filler(Fill& s, f2 m, f2 c, unsigned lp)
{
    constexpr int i = I & 7;
    if constexpr (FILL == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i].x) : "v"(m.x), "v"(c.x));
    if constexpr (FILL == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i]) : "v"(m), "v"(c));
    if constexpr (FILL == F_LDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s.l[I & 1]) : "v"(lp), "i"((I & 15) * 1024));
    if constexpr (FILL == F_MIX) { // stage T's mix: 2 fma : 1 mul : 1 pk_fma
        if constexpr ((I & 3) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i]) : "v"(m), "v"(c));
        else if constexpr ((I & 3) == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s.v[i].x) : "v"(m.x));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i].x) : "v"(m.x), "v"(c.x));
    }
}

// ---------------------------------------------------------------- A. one wave per SIMD, fillers in the MFMA wave itself
template <int SHAPE, int FILL, int NV> __global__ __launch_bounds__(256, 1) void k_self(float* out, int iters, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float wl[16 * 256 + 64 * 4];
    for (int i = threadIdx.x; i < 16 * 256 + 256; i += blockDim.x) wl[i] = 1e-3f * (float)(i % 97);
    __syncthreads();
    const int l = threadIdx.x & 63;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float x = 1.f + l, w0 = 1e-3f * l, w1 = 2e-3f * l;
    s8v xb, wb;
    for (int i = 0; i < 8; ++i) { xb[i] = (short)(0x3c00 + l + i); wb[i] = (short)(0x3a00 + 3 * l + i); }
    Fill s;
    for (int i = 0; i < 8; ++i) s.v[i] = f2{1.f + i, 2.f + i};
    s.l[0] = s.l[1] = f4{0.f, 0.f, 0.f, 0.f};
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) float*)wl + 16u * l; // LDS byte address
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<24>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            mfma<SHAPE, (k & 15)>(acc0, x, w0, xb, wb);
            static_for<NV>([&](auto vi) { filler<FILL, (2 * k) * NV + decltype(vi)::value>(s, m, c, lp); });
            mfma<SHAPE, (k & 15)>(acc1, x, w1, xb, wb);
            static_for<NV>([&](auto vi) { filler<FILL, (2 * k + 1) * NV + decltype(vi)::value>(s, m, c, lp); });
        });
        if (FILL == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x += acc0[0] * 1e-30f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc0[0] + acc1[1] + s.l[0][0] + s.l[1][1];
    for (int i = 0; i < 8; ++i) r += s.v[i].x + s.v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------- B. two waves per SIMD: matrix role and vector role
// roles: bit 0 = waves 0..3 run the MFMA chain (iters x 48 MFMAs), bit 1 = waves 4..7 run the vector stream (iters x nvec x 8)
template <int SHAPE, int FILL> __global__ __launch_bounds__(512, 2) void k_pair(float* out, int iters, int nvec, int roles, unsigned long long* cyc, unsigned* hwid)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if (blockIdx.x == 0 && l == 0) hwid[w] = id;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float x = 1.f + l, w0 = 1e-3f * l, w1 = 2e-3f * l;
    s8v xb, wb;
    for (int i = 0; i < 8; ++i) { xb[i] = (short)(0x3c00 + l + i); wb[i] = (short)(0x3a00 + 3 * l + i); }
    Fill s;
    for (int i = 0; i < 8; ++i) s.v[i] = f2{1.f + i, 2.f + i};
    s.l[0] = s.l[1] = f4{0.f, 0.f, 0.f, 0.f};
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        if (roles & 1)
            for (int it = 0; it < iters; ++it) {
                static_for<24>([&](auto ki) {
                    constexpr int k = decltype(ki)::value;
                    mfma<SHAPE, (k & 15)>(acc0, x, w0, xb, wb);
                    mfma<SHAPE, (k & 15)>(acc1, x, w1, xb, wb);
                });
                x += acc0[0] * 1e-30f;
            }
    } else {
        if (roles & 2)
            for (int it = 0; it < iters; ++it)
                for (int j = 0; j < nvec; ++j) static_for<8>([&](auto vi) { filler<FILL, decltype(vi)::value>(s, m, c, 0u); });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc0[0] + acc1[1];
    for (int i = 0; i < 8; ++i) r += s.v[i].x + s.v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && l == 0) cyc[w] = t1 - t0;
}

// ----------------------------------------------------------------
static float* out;
static unsigned long long* cyc;
static unsigned* hwid;

template <int SHAPE, int FILL, int NV> void run_self(const char* shape, const char* fill, int iters)
{
    unsigned long long h;
    hipLaunchKernelGGL((k_self<SHAPE, FILL, NV>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("self %-16s + %d x %-12s per MFMA: %7.2f cycles per MFMA\n", shape, NV, fill, (double)h / (iters * 48.0));
}

template <int SHAPE, int FILL> void run_pair(const char* shape, const char* fill, int iters, int nvec)
{
    unsigned long long h[3][8];
    unsigned ids[8];
    for (int roles = 1; roles <= 3; ++roles) {
        hipLaunchKernelGGL((k_pair<SHAPE, FILL>), dim3(256), dim3(512), 0, 0, out, iters, nvec, roles, cyc, hwid);
        hipDeviceSynchronize();
        hipMemcpy(h[roles - 1], cyc, 64, hipMemcpyDeviceToHost);
    }
    hipMemcpy(ids, hwid, 32, hipMemcpyDeviceToHost);
    double a_alone = 0, b_alone = 0, a_pair = 0, b_pair = 0;
    for (int w = 0; w < 4; ++w) { a_alone += h[0][w] / 4.0; a_pair += h[2][w] / 4.0; b_alone += h[1][w + 4] / 4.0; b_pair += h[2][w + 4] / 4.0; }
    const double nm = iters * 48.0, nv = (double)iters * nvec * 8.0;
    printf("pair %-16s | %-12s: matrix wave %6.2f -> %6.2f cycles per MFMA; vector wave %5.2f -> %5.2f cycles per instruction (%.0f MFMA, %.0f vector instr per wave; "
           "serial %.0f, paired %.0f cycles)  simd ids", shape, fill, a_alone / nm, a_pair / nm, b_alone / nv, b_pair / nv, nm, nv, a_alone + b_alone,
           a_pair > b_pair ? a_pair : b_pair);
    for (int w = 0; w < 8; ++w) printf(" %u", (ids[w] >> 4) & 3);
    printf("\n");
}

int main()
{
    hipMalloc(&out, 1 << 22);
    hipMalloc(&cyc, 64);
    hipMalloc(&hwid, 32);
    const int iters = 400;
#define SELF_ALL(SHAPE, NAME)                                                                                                      \
    run_self<SHAPE, F_FMA, 0>(NAME, "-", iters);                                                                                   \
    run_self<SHAPE, F_FMA, 1>(NAME, "v_fma_f32", iters);                                                                           \
    run_self<SHAPE, F_FMA, 2>(NAME, "v_fma_f32", iters);                                                                           \
    run_self<SHAPE, F_FMA, 4>(NAME, "v_fma_f32", iters);                                                                           \
    run_self<SHAPE, F_FMA, 6>(NAME, "v_fma_f32", iters);                                                                           \
    run_self<SHAPE, F_FMA, 8>(NAME, "v_fma_f32", iters);                                                                           \
    run_self<SHAPE, F_PKFMA, 1>(NAME, "v_pk_fma_f32", iters);                                                                      \
    run_self<SHAPE, F_PKFMA, 2>(NAME, "v_pk_fma_f32", iters);                                                                      \
    run_self<SHAPE, F_PKFMA, 4>(NAME, "v_pk_fma_f32", iters);                                                                      \
    run_self<SHAPE, F_MIX, 4>(NAME, "stage-T mix", iters);                                                                         \
    run_self<SHAPE, F_MIX, 6>(NAME, "stage-T mix", iters);                                                                         \
    run_self<SHAPE, F_LDS, 1>(NAME, "ds_read_b128", iters);
    SELF_ALL(S_4x4x1, "4x4x1_16b_f32")
    SELF_ALL(S_16x16x4, "16x16x4_f32")
    SELF_ALL(S_BF16, "16x16x32_bf16")

    // pairs: the vector stream sized to about the matrix chain's lone duration (8.5 / 32 / 16 cycles per MFMA, 5 per vector instruction)
    run_pair<S_4x4x1, F_FMA>("4x4x1_16b_f32", "v_fma_f32", iters, 10);
    run_pair<S_4x4x1, F_PKFMA>("4x4x1_16b_f32", "v_pk_fma_f32", iters, 10);
    run_pair<S_4x4x1, F_MIX>("4x4x1_16b_f32", "stage-T mix", iters, 10);
    run_pair<S_16x16x4, F_FMA>("16x16x4_f32", "v_fma_f32", iters, 38);
    run_pair<S_16x16x4, F_PKFMA>("16x16x4_f32", "v_pk_fma_f32", iters, 38);
    run_pair<S_16x16x4, F_MIX>("16x16x4_f32", "stage-T mix", iters, 38);
    run_pair<S_BF16, F_FMA>("16x16x32_bf16", "v_fma_f32", iters, 19);
    run_pair<S_BF16, F_MIX>("16x16x32_bf16", "stage-T mix", iters, 19);
    return 0;
}

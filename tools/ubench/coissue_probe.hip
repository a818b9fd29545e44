// coissue_probe.hip -- which vector instructions issue under v_mfma_f32_16x16x32_bf16?  (round 3, behind dp_w16.hip)
//   SELF: one wave per SIMD, an MFMA chain on two accumulators with TWO independent instructions of one kind behind every MFMA
//         (16.5 cycles per MFMA alone; an instruction that co-issues leaves that unchanged)
//   PAIR: two waves per SIMD, waves 0..3 the MFMA chain, waves 4..7 a stream of that instruction (8 independent chains)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/coissue_probe tools/ubench/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s8v __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__
template <class F, int... I> DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

enum { K_FMA, K_MUL, K_ADD, K_SUB, K_AND, K_LSHL, K_CVTPK, K_CNDMASK, K_MOV, K_ACCRD, K_PKMUL, K_PKFMA, K_RSQ, K_MED3, K_BPERM, K_FMAC, K_MULE64, K_PERM32, K_DOT2C, K_DOT2, K_PERMB, K_CVTF32, K_PACK, K_ANDOR, K_BFI, K_NKINDS };
static const char* NAMES[] = {"v_fma_f32", "v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_and_b32", "v_lshlrev_b32", "v_cvt_pk_bf16_f32", "v_cndmask_b32",
                              "v_mov_b32", "v_accvgpr_read_b32", "v_pk_mul_f32", "v_pk_fma_f32", "v_rsq_f32", "v_med3_f32", "ds_bpermute_b32", "v_fmac_f32_e32",
                              "v_mul_f32_e64 (neg)", "v_permlane32_swap", "v_dot2c_f32_bf16", "v_dot2_f32_bf16", "v_perm_b32", "v_cvt_f32_bf16", "v_pack_b32_f16 (hi,hi)", "v_and_or_b32", "v_bfi_b32"};
struct St { f2 v[8]; unsigned u[8]; f4 a; };
template <int KIND, int I> DEV void op(St& s, f2 m, f2 c)
{
    constexpr int i = I & 7;
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i].x) : "v"(m.x), "v"(c.x));
    if constexpr (KIND == K_MUL) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(s.v[i].x) : "v"(m.x));
    if constexpr (KIND == K_ADD) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(s.v[i].x) : "v"(c.x));
    if constexpr (KIND == K_SUB) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(s.v[i].x) : "v"(c.x));
    if constexpr (KIND == K_AND) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(s.u[i]) : "v"(0xfffffff0u));
    if constexpr (KIND == K_LSHL) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(s.u[i]));
    if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[i]) : "v"(s.v[i].x), "v"(s.v[i].y));
    if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(s.v[i].x) : "v"(c.x));
    if constexpr (KIND == K_MOV) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(s.v[i].x) : "v"(s.v[i].y));
    if constexpr (KIND == K_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s.v[i].x) : "a"(s.a[i & 3]));
    if constexpr (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(s.v[i]) : "v"(m));
    if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i]) : "v"(m), "v"(c));
    if constexpr (KIND == K_RSQ) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(s.v[i].x));
    if constexpr (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s.v[i].x) : "v"(m.x), "v"(c.x));
    if constexpr (KIND == K_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(s.v[i].x) : "v"(s.u[i]), "v"(s.v[i].y));
    if constexpr (KIND == K_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s.v[i].x) : "v"(m.x), "v"(c.x));
    if constexpr (KIND == K_MULE64) asm volatile("v_mul_f32_e64 %0, -%1, %0" : "+v"(s.v[i].x) : "v"(m.x));
    if constexpr (KIND == K_PERM32) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(s.v[i].x), "+v"(s.v[i].y));
    if constexpr (KIND == K_DOT2C) asm volatile("v_dot2c_f32_bf16_e32 %0, %1, %2" : "+v"(s.v[i].x) : "v"(0xbf80u), "v"(s.u[i]));
    if constexpr (KIND == K_DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(s.v[i].x) : "v"(0xbf80u), "v"(s.u[i]));
    if constexpr (KIND == K_PERMB) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s.u[i]) : "v"(s.v[i].x), "v"(s.v[i].y), "v"(0x07060302u));
    if constexpr (KIND == K_PACK) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(s.u[i]) : "v"(s.v[i].x), "v"(s.v[i].y));
    if constexpr (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(s.u[i]) : "v"(s.v[i].x), "v"(0xffff0000u));
    if constexpr (KIND == K_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(s.u[i]) : "v"(0xffff0000u), "v"(s.v[i].x));
    if constexpr (KIND == K_CVTF32) asm volatile("v_cvt_f32_bf16_e32 %0, %1" : "=v"(s.v[i].x) : "v"(s.u[i]));
}
DEV void mfma(f4& acc, const s8v& a, const s8v& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
DEV void init(St& s, int l)
{
    for (int i = 0; i < 8; ++i) { s.v[i] = f2{1.f + i + l * 1e-3f, 2.f + i}; s.u[i] = 4u * (unsigned)((l + i) & 63); }
    s.a = f4{1.f, 2.f, 3.f, 4.f};
}
DEV float fin(const St& s) { float r = 0.f; for (int i = 0; i < 8; ++i) r += s.v[i].x + s.v[i].y + (float)s.u[i]; return r; }

template <int KIND, int NV> __global__ __launch_bounds__(256, 1) void k_self(float* out, int iters, unsigned long long* cyc)
{
    const int l = threadIdx.x & 63;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    s8v xb, wb;
    for (int i = 0; i < 8; ++i) { xb[i] = (short)(0x3c00 + l + i); wb[i] = (short)(0x3a00 + 3 * l + i); }
    St s; init(s, l);
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<24>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            mfma(acc0, xb, wb);
            static_for<NV>([&](auto vi) { op<KIND, (2 * k) * NV + decltype(vi)::value>(s, m, c); });
            mfma(acc1, xb, wb);
            static_for<NV>([&](auto vi) { op<KIND, (2 * k + 1) * NV + decltype(vi)::value>(s, m, c); });
        });
        if (KIND == K_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + fin(s);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// roles bit 0: waves 0..3 MFMA chain (iters x 48); bit 1: waves 4..7 vector stream (iters x 64 instructions, unrolled)
template <int KIND> __global__ __launch_bounds__(512, 2) void k_pair(float* out, int iters, int roles, unsigned long long* cyc)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    s8v xb, wb;
    for (int i = 0; i < 8; ++i) { xb[i] = (short)(0x3c00 + l + i); wb[i] = (short)(0x3a00 + 3 * l + i); }
    St s; init(s, l);
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        if (roles & 1)
            for (int it = 0; it < iters; ++it) static_for<24>([&](auto) { mfma(acc0, xb, wb); mfma(acc1, xb, wb); });
    } else if (roles & 2) {
        for (int it = 0; it < iters; ++it) {
            static_for<160>([&](auto vi) { op<KIND, decltype(vi)::value>(s, m, c); });
            if (KIND == K_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + fin(s);
    if (blockIdx.x == 0 && l == 0) cyc[w] = t1 - t0;
}

// does v_pack_b32_f16 with op_sel:[1,1,0] pass the two high halves through bit for bit (bf16 patterns are not f16 values)?
__global__ void k_packbits(unsigned* bad)
{
    unsigned n = 0;
    for (unsigned hi = threadIdx.x + blockIdx.x * blockDim.x; hi < 65536u; hi += blockDim.x * gridDim.x) {
        const unsigned a = (hi << 16) | 0x1234u, b = ((hi ^ 0x5a5au) << 16) | 0xbeefu;
        unsigned r;
        asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(r) : "v"(a), "v"(b));
        n += r != ((a >> 16) | (b & 0xffff0000u));
    }
    if (n) atomicAdd(bad, n);
}
static float* out; static unsigned long long* cyc;
template <int KIND> void run(int iters)
{
    unsigned long long h, hp[3][8];
    double self[3];
    hipLaunchKernelGGL((k_self<KIND, 0>), dim3(256), dim3(256), 0, 0, out, iters, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); self[0] = (double)h / (iters * 48.0);
    hipLaunchKernelGGL((k_self<KIND, 2>), dim3(256), dim3(256), 0, 0, out, iters, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); self[1] = (double)h / (iters * 48.0);
    hipLaunchKernelGGL((k_self<KIND, 4>), dim3(256), dim3(256), 0, 0, out, iters, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); self[2] = (double)h / (iters * 48.0);
    for (int roles = 1; roles <= 3; ++roles) {
        hipLaunchKernelGGL((k_pair<KIND>), dim3(256), dim3(512), 0, 0, out, iters, roles, cyc);
        hipDeviceSynchronize();
        hipMemcpy(hp[roles - 1], cyc, 64, hipMemcpyDeviceToHost);
    }
    double a_alone = 0, b_alone = 0, a_pair = 0, b_pair = 0;
    for (int w = 0; w < 4; ++w) { a_alone += hp[0][w] / 4.0; a_pair += hp[2][w] / 4.0; b_alone += hp[1][w + 4] / 4.0; b_pair += hp[2][w + 4] / 4.0; }
    const double nm = iters * 48.0, nv = iters * 160.0;
    printf("%-22s self: %5.2f / %5.2f / %5.2f cycles per MFMA with 0 / 2 / 4 behind each | pair: MFMA wave %5.2f -> %5.2f per MFMA, vector wave %5.2f -> %5.2f per instruction; serial %7.0f paired %7.0f ideal %7.0f\n",
           NAMES[KIND], self[0], self[1], self[2], a_alone / nm, a_pair / nm, b_alone / nv, b_pair / nv, a_alone + b_alone, a_pair > b_pair ? a_pair : b_pair, a_alone > b_alone ? a_alone : b_alone);
}
template <int K> void run_all(int iters) { run<K>(iters); if constexpr (K + 1 < K_NKINDS) run_all<K + 1>(iters); }
int main()
{
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    run_all<0>(200);
    unsigned* bad; unsigned hb = 0;
    hipMalloc(&bad, 4); hipMemcpy(bad, &hb, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_packbits, dim3(64), dim3(256), 0, 0, bad); hipDeviceSynchronize(); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("v_pack_b32_f16 op_sel:[1,1,0]: %u of 65536 high-half patterns (both operands) not passed through bit for bit\n", hb);
    return 0;
}

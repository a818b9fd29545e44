// micro-benchmark: v_mfma_f32_16x16x4_f32 issue interval with 1 / 2 / 4 independent accumulator chains, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CH> __global__ void k(float* out, int iters, unsigned long long* cyc)
{
    f4 acc[CH];
    for (int i = 0; i < CH; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        unsigned long long h;
#define RUN(CH)                                                                                   \
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);               \
        (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);            \
        printf("waves/CU %d chains %d: %.2f ticks per MFMA per wave\n", waves, CH, (double)h / (iters * 16.0));
        RUN(1) RUN(2) RUN(4) RUN(8)
    }
    return 0;
}

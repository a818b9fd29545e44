// micro-benchmark: VALU issue rate on gfx950 for 1..4 waves per SIMD, dependent vs independent v_fma_f32 streams
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP> __global__ void k(float* out, int iters, unsigned long long* cyc)
{
    float a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves = 1; waves <= 16; waves *= 2) { // waves per workgroup, one workgroup per CU (256 blocks)
        unsigned long long h;
#define RUN(ILP)                                                                                     \
        hipLaunchKernelGGL(k<ILP>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);                  \
        hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                           \
        printf("waves/CU %2d (per SIMD %.2f) ILP %d: %.2f cycles per v_fma per wave, %.2f per SIMD-instr\n", waves, waves / 4.0, ILP, \
               (double)h / (iters * 16.0 * ILP), (double)h / (iters * 16.0 * ILP) / (waves > 4 ? waves / 4.0 : 1.0));
        RUN(1) RUN(2) RUN(4) RUN(8)
    }
    return 0;
}

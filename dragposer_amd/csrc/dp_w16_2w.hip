// dp_w16_2w.hip -- the 16-frames-per-wavefront optimise kernel (dp_w16_impl.h), TWO waves per SIMD: 8 waves = 128 frames per
// workgroup, 256 registers per wave, weights read where they are used (the partner wave covers the LDS latency).  Compiled with the
// default instruction scheduler (dp_w16.hip says why).
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_2w(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + 8 * FPW - 1) / (8 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<8, 2>), dim3(grid), dim3(512), 0, stream, *args);
    return hipGetLastError();
}

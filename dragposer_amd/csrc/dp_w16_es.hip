// dp_w16_es.hip -- the 16-frames-per-wave kernel with the reference's per-frame while-condition (early stop), one wave per SIMD.
// Its own translation unit like its siblings (dp_w16.hip, dp_w16_2w.hip): each instantiation takes the compiler flags it measured best with.
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_es(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + 4 * FPW - 1) / (4 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<4, 1, true>), dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
}

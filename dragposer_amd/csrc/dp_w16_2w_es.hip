// dp_w16_2w_es.hip -- the early-stop instantiation with two waves per SIMD (see dp_w16_es.hip, dp_w16_2w.hip).
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_2w_es(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + 8 * FPW - 1) / (8 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<8, 2, true>), dim3(grid), dim3(512), 0, stream, *args);
    return hipGetLastError();
}

// dp_w16_2w_es_long.hip -- dp_w16_2w_es.hip's instantiation (early stop, two waves per SIMD) for n_iter > 256 (dp_w16_impl.h: LONG).
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_2w_es_long(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + 8 * FPW - 1) / (8 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<8, 2, true, true>), dim3(grid), dim3(512), 0, stream, *args);
    return hipGetLastError();
}

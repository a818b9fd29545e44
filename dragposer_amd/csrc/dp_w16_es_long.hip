// dp_w16_es_long.hip -- dp_w16_es.hip's instantiation (early stop, one wave per SIMD) for n_iter > 256 (dp_w16_impl.h: LONG).
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_es_long(const KArgs* args, hipStream_t stream)
{
    const int grid = (args->n_frames + 4 * FPW - 1) / (4 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<4, 1, true, true>), dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
}

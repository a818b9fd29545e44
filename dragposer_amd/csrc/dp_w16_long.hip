// dp_w16_long.hip -- dp_w16.hip's instantiation (one wave per SIMD) for n_iter beyond the 256-entry argument table of Adam scalars
// (dp_w16_impl.h: LONG), and the dispatch of the four LONG units.  Same flags as dp_w16.hip.
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_2w_long(const KArgs* args, hipStream_t stream);
extern "C" hipError_t dp_launch_w16_es_long(const KArgs* args, hipStream_t stream);
extern "C" hipError_t dp_launch_w16_2w_es_long(const KArgs* args, hipStream_t stream);

extern "C" hipError_t dp_launch_w16_long(const KArgs* args, hipStream_t stream, int waves)
{
    if (args->early_stop) return waves == 8 ? dp_launch_w16_2w_es_long(args, stream) : dp_launch_w16_es_long(args, stream);
    if (waves == 8) return dp_launch_w16_2w_long(args, stream);
    const int grid = (args->n_frames + 4 * FPW - 1) / (4 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<4, 1, false, true>), dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
}

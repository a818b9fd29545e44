// dp_w16.hip -- the 16-frames-per-wavefront optimise kernel (dp_w16_impl.h), ONE wave per SIMD: 4 waves = 64 frames per workgroup,
// the full register file per wave, weight reads issued a pair ahead.  Compiled with -amdgpu-sched-strategy=iterative-ilp (measured
// against max-ilp / the default: -6 % / -9 % kernel time; the two-waves-per-SIMD instantiation loses 17 % with it, hence its own file).
#include "dp_w16_impl.h"

extern "C" hipError_t dp_launch_w16_2w(const KArgs* args, hipStream_t stream);

// One wave per SIMD (4 waves, 64 frames per workgroup) until every SIMD of the chip has a wave; beyond that two waves per SIMD
// (8 waves, 128 frames per workgroup): the matrix phases of one wave run under the vector phases of the other.
extern "C" hipError_t dp_launch_w16_es(const KArgs* args, hipStream_t stream);
extern "C" hipError_t dp_launch_w16_2w_es(const KArgs* args, hipStream_t stream);

extern "C" hipError_t dp_launch_w16_long(const KArgs* args, hipStream_t stream, int waves); // n_iter beyond the argument table (dp_w16_long.hip)

extern "C" hipError_t dp_launch_w16(const KArgs* args, hipStream_t stream, int waves)
{
    if (args->n_iter > MAX_ITERS) return dp_launch_w16_long(args, stream, waves);
    if (args->early_stop) return waves == 8 ? dp_launch_w16_2w_es(args, stream) : dp_launch_w16_es(args, stream);
    if (waves == 8) return dp_launch_w16_2w(args, stream);
    const int grid = (args->n_frames + 4 * FPW - 1) / (4 * FPW);
    hipLaunchKernelGGL((dp_w16_kernel<4, 1>), dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
}

extern "C" int dp_w16_lds_bytes(void) { return L16_END * 4; }
extern "C" int dp_w16_frames_per_wave(void) { return FPW; }

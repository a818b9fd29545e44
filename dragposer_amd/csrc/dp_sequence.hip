// dp_sequence.hip -- the per-frame epilogue of a sequence (reference: DragPose.run, python/src/drag_pose.py:369-402):
// what happens between the optimise loop and `return`, for S sequences advancing in lock-step.  A handful of scalars
// and three 60-deep history buffers per sequence: one 64-thread workgroup per sequence, HBM-trivial; it exists so
// that a frame costs two launches (dp_optimize + this) instead of two dozen framework ops.
#include <hip/hip_runtime.h>
#include "dp_sequence.h"

__global__ __launch_bounds__(64) void dp_sequence_advance_kernel(const SeqArgs a)
{
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= a.n_seq) return;
    // every thread derives the frame's scalars itself (a few flops) instead of exchanging them
    float gp[3], dsp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gp[k] = a.global_pos[s * 3 + k] + a.world_disp[s * 3 + k]; // drag_pose.py:370
        dsp[k] = a.disp[s * 3 + k];
    }
    if (a.adjust_joint >= 0) { // drag_pose.py:377-384
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float adj = (a.tgt_pos[(s * 22 + a.adjust_target_joint) * 3 + k] - a.pos[(s * 22 + a.adjust_joint) * 3 + k]) * a.adjust_weight;
            gp[k] += adj;
            dsp[k] += adj;
        }
    }
    // history buffers: one thread per channel column shifts its column by one frame and appends (drag_pose.py:386-397)
    const int H = a.history, NH = a.n_heights;
    if (tid < 24 + 3 + NH) {
        float* col;
        int stride;
        float newest;
        if (tid < 24) {
            col = a.latent_buf + (size_t)s * H * 24 + tid; stride = 24;
            newest = a.z_pre[s * 24 + tid];
        } else if (tid < 27) {
            col = a.disp_buf + (size_t)s * H * 3 + (tid - 24); stride = 3;
            newest = dsp[tid - 24];
        } else {
            const int h = tid - 27;
            col = a.heights_buf + (size_t)s * H * NH + h; stride = NH;
            newest = a.pos[(s * 22 + a.height_joints[h]) * 3 + 1] + gp[1]; // y of (pos + current_global_pos)
        }
        for (int t = 0; t + 1 < H; ++t) col[(size_t)t * stride] = col[(size_t)(t + 1) * stride];
        col[(size_t)(H - 1) * stride] = newest;
    }
    if (a.pose_ret) { // returned pose: root channels = normalised global rotation (drag_pose.py:399-402)
        for (int c = tid; c < 88; c += 64)
            a.pose_ret[s * 88 + c] = c < 4 ? (a.world_rot[s * 4 + c] - a.mean_q0[c]) / a.std_q0[c] : a.pose[s * 88 + c];
    }
    __syncthreads(); // every read of the old state above precedes its replacement below
    if (tid < 3) {
        a.global_pos[s * 3 + tid] = gp[tid];
        if (a.pos_ret) a.pos_ret[s * 3 + tid] = gp[tid];
    }
    if (tid < 4) a.global_rot[s * 4 + tid] = a.world_rot[s * 4 + tid];
}

extern "C" hipError_t dp_launch_sequence_advance(const SeqArgs* args, hipStream_t stream)
{
    hipLaunchKernelGGL(dp_sequence_advance_kernel, dim3(args->n_seq), dim3(64), 0, stream, *args);
    return hipGetLastError();
}

// The history buffers after n_steps frames (drag_pose.py:384-391, n_steps times): every column shifted by n = min(T, H) and the
// last n steps appended.  One thread per (sequence, column); reads run ahead of writes.
__global__ __launch_bounds__(64) void dp_sequence_history_kernel(const HistArgs a)
{
    const int s = blockIdx.x, c = threadIdx.x, NH = a.n_heights, W = 24 + 3 + NH;
    if (s >= a.n_seq || c >= W) return;
    const int H = a.history, T = a.n_steps, n = T < H ? T : H;
    float* col;
    int stride;
    if (c < 24) { col = a.latent_buf + (size_t)s * H * 24 + c; stride = 24; }
    else if (c < 27) { col = a.disp_buf + (size_t)s * H * 3 + (c - 24); stride = 3; }
    else { col = a.heights_buf + (size_t)s * H * NH + (c - 27); stride = NH; }
    for (int t = 0; t < H; ++t)
        col[(size_t)t * stride] = t + n < H ? col[(size_t)(t + n) * stride] : a.scratch[((size_t)(T - H + t) * a.n_seq + s) * W + c];
}

extern "C" hipError_t dp_launch_sequence_history(const HistArgs* args, hipStream_t stream)
{
    hipLaunchKernelGGL(dp_sequence_history_kernel, dim3(args->n_seq), dim3(64), 0, stream, *args);
    return hipGetLastError();
}

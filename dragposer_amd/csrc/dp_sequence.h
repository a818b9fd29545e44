// dp_sequence.h -- argument block of the sequence-advance kernel (dp_sequence.hip), filled by dp_host.cpp
#pragma once
#include <hip/hip_runtime.h>

struct SeqArgs {
    int n_seq, history, n_heights;
    int height_joints[8];
    int adjust_joint, adjust_target_joint;
    float adjust_weight;
    float mean_q0[4], std_q0[4]; // de-normalisation of the root quaternion channels
    // frame results (dp_result of this frame)
    const float *z_pre, *pose, *disp, *world_disp, *world_rot, *pos;
    const float* tgt_pos;
    // state
    float *global_pos, *global_rot, *latent_buf, *disp_buf, *heights_buf;
    float *pose_ret, *pos_ret;
};

extern "C" hipError_t dp_launch_sequence_advance(const SeqArgs* args, hipStream_t stream);

struct HistArgs { // appends the rows of n_steps frames to the three history buffers (dp_optimize_sequence's second launch)
    int n_seq, n_steps, history, n_heights;
    const float* scratch; // [T][S][24 + 3 + NH]: z_pre | displacement | heights of every step
    float *latent_buf, *disp_buf, *heights_buf;
};
extern "C" hipError_t dp_launch_sequence_history(const HistArgs* args, hipStream_t stream);

// dp_kernel.h -- argument block shared by the host side (dp_host.cpp) and the kernel (dp_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "dp_layout.h"
#include "dp_w4.h"

struct AdamTab { // per-iteration scalars torch's single-tensor Adam computes in Python doubles
    float step[dpl::MAX_ITERS]; // lr / (1 - beta1^t)
    float bc2s[dpl::MAX_ITERS]; // 1 / sqrt(1 - beta2^t)
};

// debug dump (iteration 0 only), floats per frame
constexpr int DBG_Y = 0;     // [104] decoder output quads (sum of the two K-half planes)
constexpr int DBG_GY = 104;  // [104] dL/dy quads (23.. = virtual items)
constexpr int DBG_GZ = 208;  // [24] dL/dz (incl. temporal term)
constexpr int DBG_STRIDE = 240;

struct KArgs {
    // model (device)
    const float* wfrag;          // [NWAVE][W_REGS][64]  per-wave, per-lane MFMA A operands
    const float* bias;           // [2][64] padded bias rows of L0 (c0) and L1 (b1)
    const dpl::ItemConst* items; // [32]
    const float* w4img;          // [dpw4::N_GROUPS][64][4]  weight image of the wave-private kernel (dp_w4.h)
    const float* w4bias;         // [4][64] accumulator seeds of L0, L1, L2A, L2B
    const dpw4::Pair* w4pairs;   // [16] kinematics constants per lane quad
    // batch (device)
    const float *z0, *z_tgt, *cur_rot, *tgt_pos, *tgt_rot, *w;
    const unsigned char* tracked;
    // results (device, nullable)
    float *z, *z_pre, *pose, *disp, *world_disp, *world_rot, *pos, *rot, *loss;
    int* iters;
    float* dbg;
    int n_frames, n_iter, mode; // mode 0: optimise, 1: forward only (n_iter = 1)
    int early_stop;             // per-frame while-condition of drag_pose.py:300-304
    float stop_eps_pos, stop_eps_rot, min_loss_incr;
    float lam_rot, lam_tmp, ctmp; // ctmp = 2 lam_tmp / 24
    float beta2, one_m_b1, one_m_b2, eps;
    unsigned smask[dpl::NWAVE][dpl::NGEMM]; // bit i: step i of the wave's chain has a non-zero weight block
    AdamTab tab;
};

extern "C" hipError_t dp_launch_optimize(const KArgs* args, hipStream_t stream);
extern "C" int dp_kernel_lds_bytes(void);
// dp_w4.hip: wave-private kernel, 4 frames per wave, no workgroup barrier inside the loop
extern "C" hipError_t dp_launch_w4(const KArgs* args, hipStream_t stream);
extern "C" int dp_w4_lds_bytes(void);
extern "C" int dp_w4_frames_per_block(void);

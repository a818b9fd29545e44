// dp_kernel.h -- argument block shared by the host side (dp_host.cpp) and the kernel (dp_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dragposer.h"
#include "dp_layout.h"
#include "dp_w4.h"
#include "dp_w16.h"

struct AdamTab { // per-iteration scalars torch's single-tensor Adam computes in Python doubles
    float step[dpl::MAX_ITERS]; // lr / (1 - beta1^t)
    float bc2s[dpl::MAX_ITERS]; // 1 / sqrt(1 - beta2^t)
};
// ... for the first MAX_ITERS iterations (the table rides in the kernel arguments: 2 KB of their 4).  A frame that iterates longer -- the reference
// has no cap on max_iter -- continues the two products in double on the device (adam_beyond, dp_device.h), one uniform f64 sequence per iteration
// beyond the table: what the host would have computed, at ~10 % more per such iteration.
struct AdamCont {
    double beta1, beta2, lr;
    double b1t, b2t; // beta1^MAX_ITERS, beta2^MAX_ITERS
};

// debug dump (iteration 0 only), floats per frame
constexpr int DBG_Y = 0;     // [104] decoder output quads (sum of the two K-half planes)
constexpr int DBG_GY = 104;  // [104] dL/dy quads (23.. = virtual items)
constexpr int DBG_GZ = 208;  // [24] dL/dz (incl. temporal term)
constexpr int DBG_STRIDE = 240;

struct SeqK { // whole-sequence launches (dp_optimize_sequence): the frame loop of a sequence set inside ONE launch of dp_w4
    int n_steps;               // 0: an ordinary launch
    int z_tgt_step, z_tgt_seq; // strides (floats) of z_tgt between steps / between sequences
    const float* tgt_root;     // [T][S][3] or NULL: position targets of step t are tgt_pos + (tgt_root[t] - current global position)
    float* global_pos;         // [S][3] in/out
    float* global_rot;         // [S][4] out (in: KArgs::cur_rot)
    float* hist;               // [T][S][24 + 3 + NH] history rows of every step: z_pre | root-space displacement (joint adjustment included) | heights
    float* pos_ret;            // [T][S][3]  returned global position, nullable
    int n_heights, height_joints[8];
    int adjust_joint, adjust_target_joint;
    float adjust_weight;
    float mean_q0[4], std_q0[4]; // normalisation of the returned pose's root channels
};

struct KArgs {
    // model (device)
    const float* wfrag;          // [NWAVE][W_REGS][64]  per-wave, per-lane MFMA A operands
    const float* bias;           // [2][64] padded bias rows of L0 (c0) and L1 (b1)
    const dpl::ItemConst* items; // [32]
    const float* w4img;          // [dpw4::N_GROUPS][64][4]  weight image of the wave-private kernel (dp_w4.h)
    const float* w4bias;         // [4][64] accumulator seeds of L0, L1, L2A, L2B
    const dpw4::Pair* w4pairs;   // [16] kinematics constants per lane quad
    const unsigned* w16img;      // [dpw16::IMG_U32] split-precision weight image of the 16-frames-per-wave kernel (dp_w16.h)
    const float* w16bias;        // [dpw16::BIAS_FLOATS]
    const dpw16::SlotConst* w16slots; // [6 tiles][4 groups]
    // batch (device)
    const float *z0, *z_tgt, *cur_rot, *tgt_pos, *tgt_rot, *w;
    const unsigned char* tracked;
    // results (device, nullable)
    float *z, *z_pre, *pose, *disp, *world_disp, *world_rot, *pos, *rot, *loss;
    int* iters;
    int* status;             // [B] DP_STATUS_* bits, nullable
    unsigned long long* clk; // [2] shader cycles / 100 MHz ticks of workgroup 0's iteration loop, nullable
    float* dbg;
    int n_frames, n_iter, mode; // mode 0: optimise, 1: forward only (n_iter = 1)
    int early_stop;             // per-frame while-condition of drag_pose.py:300-304
    float stop_eps_pos, stop_eps_rot, min_loss_incr;
    float lam_rot, lam_tmp, ctmp; // ctmp = 2 lam_tmp / 24
    float beta2, one_m_b1, one_m_b2, eps;
    SeqK seq;
    unsigned smask[dpl::NWAVE][dpl::NGEMM]; // bit i: step i of the wave's chain has a non-zero weight block
    AdamTab tab;
    AdamCont cont;
};

extern "C" hipError_t dp_launch_optimize(const KArgs* args, hipStream_t stream);
extern "C" int dp_kernel_lds_bytes(void);
// dp_w4.hip: wave-private kernel, 4 frames per wave, no workgroup barrier inside the loop
extern "C" hipError_t dp_launch_w4(const KArgs* args, hipStream_t stream);
extern "C" int dp_w4_lds_bytes(void);
extern "C" int dp_w4_frames_per_block(void);
// dp_w16*.hip: 16 frames per wave, decoder on v_mfma_f32_16x16x32_bf16 in split precision (fixed iteration count, or KArgs.early_stop: the per-frame while-condition)
extern "C" hipError_t dp_launch_w16(const KArgs* args, hipStream_t stream, int waves /* 4 or 8 per workgroup */);
extern "C" int dp_w16_lds_bytes(void);
extern "C" int dp_w16_frames_per_wave(void);
extern "C" int dp_w16_supported(const dp_model* m);
extern "C" int dp_debug_pack_w16(const dp_folded* f, const dp_model* m, unsigned* img, float* bias, void* slots_out);

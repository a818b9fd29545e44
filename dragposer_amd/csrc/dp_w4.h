// dp_w4.h -- layout of the wave-private kernel (dp_w4.hip), shared with the host packer (dp_host.cpp).
//
// One wavefront owns FOUR frames for the whole launch; nothing is exchanged between waves.  Every product of the folded
// decoder runs on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1, 8 cycles): with CBSZ = 4 all 16 blocks take their
// A operand from block ABID, so
//     D_b[i][j] += A_ABID[i] * B_b[j]          lane 4b+j, register i   (b = block, probe: tools/ubench/w4_probe.hip)
// is a rank-1 update of a (64 output channels) x (4 frames) tile:  i = frame, 4b+j = output channel,
//     A = activations, ONE register holds 16 K-steps: lane 4b+i = act[frame i][channel k(b)]
//     B = weights of K-step k: lane l = W[row l][k]   (streamed from LDS, one ds_read_b128 per 4 steps)
// Register layouts of a 64-channel x 4-frame tile:
//     "D" (what a product leaves):   lane 4b+j, register r  = value[channel 4b+j][frame r]
//     "X" (what a product consumes): lane 4b+i, register m  = value[channel 4b+m][frame i]
// D <-> X is a 4x4 transpose of (register, lane-in-quad), done with 8 DPP selects, so activations never leave the
// register file between layers.  X is also the natural layout of the kinematics phase: lane 4b+i holds the 4 channels
// (one quaternion) of item b of frame i.
#pragma once
#include "dp_layout.h"

namespace dpw4 {

constexpr int FPW = 4; // frames per wave

// steps (K = 1 each) of the weight image, in program order; every product's count is a multiple of 4
constexpr int S_L0 = 0;            // 24: rows = 40 channels of a0
constexpr int S_L1 = S_L0 + 24;    // 40: rows = 60 channels of a1
constexpr int S_L2A = S_L1 + 40;   // 60: rows = items 0..15 of y (4 channels each)
constexpr int S_L2B = S_L2A + 60;  // 60: rows = items 16..31 of y
constexpr int S_B2 = S_L2B + 60;   // 104: K = 4 channels of items 0..25 of dL/dy, rows = 60 channels of d1
constexpr int S_B1 = S_B2 + 104;   // 60: rows = 40 channels of d0
constexpr int S_B0 = S_B1 + 60;    // 40: rows = 24 channels of dL/dz
constexpr int N_STEPS = S_B0 + 40; // 388
constexpr int N_GROUPS = N_STEPS / 4;
// device image: [group][lane][4 steps] floats  (a lane's four steps are one ds_read_b128)
constexpr int IMG_FLOATS = N_GROUPS * 64 * 4;
// bias image [4][64]: L0, L1, L2A, L2B rows (the accumulators start from it)
constexpr int BIAS_FLOATS = 4 * 64;

constexpr int ITEMS_A = 16; // items 0..15 ride in block A of layer 2, items 16..31 in block B

} // namespace dpw4

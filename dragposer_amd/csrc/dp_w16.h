// dp_w16.h -- layout of the 16-frames-per-wave optimise kernel (dp_w16.hip), shared with the host packer (dp_w16_host.cpp).
//
// Why a second kernel: fp32 MFMAs (every shape) and the vector ALU are ONE issue resource on gfx950 -- measured, within a wave and
// between the two waves of a SIMD (profiles/r03_pair_probe.txt) -- so dp_w4.hip's matrix half and vector half can never overlap.
// v_mfma_f32_16x16x32_bf16 does leave the vector ALU free (two v_fma_f32 per MFMA hide completely), and it retires K = 32 per 16
// cycles.  This kernel therefore runs the folded decoder on it in SPLIT PRECISION: every fp32 operand -- weight or activation -- is
// the exact sum of three bf16 terms (round-to-nearest splits), and a product keeps the six term pairs above 2^-24 relative
// (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) accumulated in fp32 by the MFMA: the arithmetic of an fp32 product up to its last
// bit or two.  BASELINE config 5 (bf16 decoder weights on CDNA4 MFMA) is its home; the weights it splits are the folded matrices
// built from the bf16-rounded tensors, the same ones dp_w4 multiplies in fp32.
//
// Orientation of every product: D[16 out-channels][16 frames] += A[16][32] * B[32][16]
//   A = weights (LDS image, one ds_read_b128 per term), lane l: A[m = l & 15][k = 8 (l >> 4) + j], j = 0..7
//   B = activations, lane l: B[k = 8 (l >> 4) + j][n = l & 15]
//   D: lane l holds rows 4 (l >> 4) + r, r = 0..3, of column l & 15
// so lane (f = l & 15, g = l >> 4) holds FOUR CONSECUTIVE CHANNELS of frame f of every 16-channel output tile -- after layer 2
// a whole quaternion per tile -- and, because the summation index of the next product may be ordered freely, a pair of output
// tiles IS the next product's B operand: slots j = 0..3 of K-block p are channels 4g..4g+3 of tile 2p, slots 4..7 those of tile
// 2p + 1.  No transposition anywhere; the host orders the K index of every weight image accordingly (in_channel()).
#pragma once
#include "dp_layout.h"

namespace dpw16 {

constexpr int FPW = 16;  // frames per wave
constexpr int NTY = 6;   // tiles of the decoder output: 24 slots of 4 channels = 22 joints + displacement + 1 idle

// Slot map: tile t of lane group g.  Tiles 0..3 = a chain of four joints (A), tiles 4, 5 = a chain of two (B); each chain hangs
// off an EXTERNAL parent (the root, or a joint of another group, fetched across lane groups): the same template for all four
// groups, so that every group runs the same instruction stream.
//   g = 0: A = left leg 1-4,    B = (root, displacement)     -- not a kinematic chain: zero offsets, selected by lane
//   g = 1: A = right leg 5-8,   B = spine 9, 10
//   g = 2: A = left arm 14-17,  B = spine 11, 12  (ext: 10 of group 1);  A's ext: 11 (own B)
//   g = 3: A = right arm 18-21, B = spine 13, idle (ext: 12 of group 2); A's ext: 11 (group 2)
constexpr int ITEM_PAD = -1;
DP_HD constexpr int slot_item(int t, int g)
{
    if (t < 4) return g == 0 ? 1 + t : g == 1 ? 5 + t : g == 2 ? 14 + t : 18 + t;
    if (g == 0) return t == 4 ? 0 : dpl::ITEM_DISP;
    if (g == 1) return 9 + (t - 4);
    if (g == 2) return 11 + (t - 4);
    return t == 4 ? 13 : ITEM_PAD;
}
// the skeleton this slot map is for (the reference's: python/data/example/eval/example.bvh)
constexpr int PARENTS[dpl::NJ] = {0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 11, 18, 19, 20};

// the six products: input channels (padded to K-blocks of 32), output tiles, K-blocks
enum { L0 = 0, L1, L2, B2, B1, B0, NL };
constexpr int N_IN[NL] = {24, 40, 60, 96, 60, 40};
constexpr int N_OUT[NL] = {40, 60, 96, 60, 40, 24};
constexpr int NT_OUT[NL] = {3, 4, 6, 4, 3, 2};
constexpr int NKB[NL] = {1, 2, 2, 3, 2, 2};
constexpr int PAIR0[NL] = {0, 3, 11, 23, 35, 41}; // first (tile, K-block) pair; pair = PAIR0 + n * NKB + kb
constexpr int N_PAIRS = 45;
constexpr int N_TERMS = 3;
constexpr int IMG_U32 = N_PAIRS * N_TERMS * 64 * 4; // image in 32-bit words (two bf16 each): [pair][term][lane][4]
constexpr int BIAS_TILE0[3] = {0, 3, 7};            // bias rows of L0, L1, L2: [tile][group][4] floats
constexpr int BIAS_FLOATS = 13 * 16;
// input channel that K slot (kb, kg = lane >> 4, j) of a product carries: channel 4 kg + (j & 3) of tile 2 kb + (j >> 2)
DP_HD constexpr int in_channel(int kb, int kg, int j) { return 16 * (2 * kb + (j >> 2)) + 4 * kg + (j & 3); }

struct SlotConst { // per (tile, group): what the kinematics and the epilogue need; [NTY][4] on the device
    float off[3];  // offset of the slot's joint from its parent (zero: root, displacement, idle)
    int item;      // joint id, ITEM_DISP, or ITEM_PAD
    float mu[4], sd[4]; // de-normalisation of the slot's four decoder channels (epilogue: returned pose)
};
static_assert(sizeof(SlotConst) == 48, "SlotConst is 12 words");

} // namespace dpw16

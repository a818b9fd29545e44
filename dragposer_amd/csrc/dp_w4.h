// dp_w4.h -- layout of the wave-private kernel (dp_w4.hip), shared with the host packer (dp_host.cpp).
//
// One wavefront owns FOUR frames for the whole launch; nothing is exchanged between waves.  Every product of the folded
// decoder runs on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1, 8 cycles): with CBSZ = 4 all 16 blocks take their
// A operand from block ABID, so
//     D_b[i][j] += A_ABID[i] * B_b[j]          lane 4b+j, register i   (b = block, probe: tools/ubench/w4_probe.hip)
// is a rank-1 update of a (64 output channels) x (4 frames) tile:  i = frame, 4b+j = output channel,
//     A = activations, ONE register holds 16 K-steps: lane 4b+i = act[frame i][channel k(b)]
//         (CBSZ = 3: blocks 0..7 take block ABID, blocks 8..15 block 8 + ABID -- two K-steps per instruction for products of
//          at most 32 rows, used by bL0)
//     B = weights of K-step k: lane l = W[row l][k]   (streamed from LDS, one ds_read_b128 per 4 steps)
// Register layouts of a 64-channel x 4-frame tile:
//     "D" (what a product leaves):   lane 4b+j, register r  = value[channel 4b+j][frame r]
//     "X" (what a product consumes): lane 4b+i, register m  = value[channel 4b+m][frame i]
// D <-> X is a 4x4 transpose of (register, lane-in-quad), done with four MFMAs against unit rows, so activations never leave the
// register file between layers.  X is also the natural layout of the kinematics phase: lane 4b+i holds the 4 channels
// (one quaternion) of item b of frame i.
#pragma once
#include "dp_layout.h"

namespace dpw4 {

constexpr int FPW = 4; // frames per wave

// steps (K = 1 each) of the weight image, in program order; every product's count is a multiple of 4
constexpr int S_L0 = 0;            // 24: rows = 40 channels of a0
constexpr int S_L1 = S_L0 + 24;    // 40: rows = 60 channels of a1
constexpr int S_L2A = S_L1 + 40;   // 60: rows = channels 0, 1 of both items of every quad (l2_side / l2_channel below)
constexpr int S_L2B = S_L2A + 60;  // 60: rows = channels 2, 3 of both items of every quad
constexpr int S_B2 = S_L2B + 60;   // 104: K = 4 channels of the 16 side-A items, then of side-B quads 1..10; rows = 60 channels of d1
constexpr int S_B1 = S_B2 + 104;   // 60: rows = 40 channels of d0
constexpr int S_B0 = S_B1 + 60;    // 20: rows = 24 channels of dL/dz, K split over the two lane halves (below)
constexpr int N_STEPS = S_B0 + 20; // 368
constexpr int N_GROUPS = N_STEPS / 4;
// device image: [group][lane][4 steps] floats  (a lane's four steps are one ds_read_b128)
constexpr int IMG_FLOATS = N_GROUPS * 64 * 4;
// bias image [4][64]: L0, L1, L2A, L2B rows (the C operand of each chain's first step)
constexpr int BIAS_FLOATS = 4 * 64;

// The 40 channels of the first hidden layer sit in rows (= lanes of layout D, = quads of layout X) 0..19 and 32..51:
// channel c < 20 in row c, channel c >= 20 in row 12 + c.  That puts one half of them in quads 0..4 and the other in
// quads 8..12, which is what lets bL0 (24 output rows: half a register of weights per K-step) run TWO K-steps per
// instruction: with CBSZ = 3, blocks 0..7 take their A operand from quad ABID and blocks 8..15 from quad 8 + ABID, so lanes
// 0..31 accumulate K-steps 0..19 and lanes 32..63 K-steps 20..39 of the same rows; one half-wave swap adds them.
DP_HD constexpr int h0_row(int c) { return c < 20 ? c : 12 + c; }
DP_HD constexpr int h0_channel(int row) { return row < 20 ? row : (row >= 32 && row < 52 ? row - 12 : -1); }

// Rows of layer 2's two 64-row blocks: row 4b + r of block `blk` is channel 2 blk + (r >> 1) of the side-(r & 1) item of quad
// b.  The transposed result of a block then has the SAME channel of the quad's two items in each even/odd register pair,
// which is what the packed kinematics arithmetic (both items of a quad per instruction) reads -- no moves in between.
DP_HD constexpr int l2_side(int r) { return r & 1; }
DP_HD constexpr int l2_channel(int blk, int r) { return 2 * blk + (r >> 1); }

// Kinematics items of a lane quad b: side A = item b (the 16 joints 0..15), side B = item 15 + b for b = 1..10 (joints
// 16..21, the root displacement 22, virtual child-bone copies 23..25); quad 0's side B is idle, so that the root (side A
// of quad 0) shares its lanes with nothing.  Block A / B of layer 2 and the K order of its transpose follow this map.
constexpr int ITEMS_A = 16;
DP_HD constexpr int item_of(int side, int b) { return side == 0 ? b : (b >= 1 && b <= 10 ? 15 + b : -1); }
constexpr int B2_GROUPS_A = 16, B2_GROUPS_B = 10, B2_ABID0_B = 1; // bL2: K groups of side A (ABID 0..15), side B (ABID 1..10)

// per-quad constants of the two items, side A / side B interleaved (the kernel keeps them as packed register pairs)
struct Pair {
    float sd[4][2], mu[4][2]; // de-normalisation of the item's 4 decoder channels (idle: 0 / (1,0,0,0))
    float off[3][2];          // offset of the child bone the item handles (zero when none)
    float sgn[2];             // +1: dL/dq = (0, 2 tau) (x) q;  -1 (root): q (x) (0, 2 tau)
    float rho[2];             // 1 (root): tau = sum of the trackers' root torques; 0: tau = bone x S + own torque
    int item[2];              // item id, -1 idle
    int kind[2];              // dpl::KIND_*
    int bone_slot[2];         // bone slot the item writes (child joint id, or a trash slot)
    unsigned ch_sub[2];       // trackers summed by the item: subtree of its child (root, displacement: every joint)
    int pad[2];
};
static_assert(sizeof(Pair) == 36 * 4, "Pair is 36 words");

} // namespace dpw4

// dp_layout.h -- compile-time description of how the optimise kernel maps the folded decoder
// onto v_mfma_f32_16x16x4_f32 tiles.  Shared by the host packer (dp_host.cpp) and the device
// code (dp_kernel.hip) so that the two cannot disagree.
//
// Orientation of every product: D[M=16 out-channels][N=16 frames] += A[M][K=4] * B[K=4][N]
//   A = weights  (loop-invariant, one 32-bit register per MFMA step, packed on the host per wave/lane)
//   B = activations of the 16 frames of the workgroup, read from LDS
//   lane l: A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], D[row = 4*(l>>4)+r][col = l&15], r=0..3
// so a lane holds 4 consecutive output channels of ONE frame: a whole quaternion after layer 2,
// and the B operand of step p is the single float  act[frame = l&15][4p + (l>>4)].
// K runs in natural order: MFMA step p covers input channels 4p..4p+3 (= one joint of the pooled
// skeleton), which is what makes whole (tile, step) blocks structurally zero and skippable.
#pragma once

#if defined(__HIPCC__)
#define DP_HD __host__ __device__
#else
#define DP_HD
#endif

namespace dpl {

constexpr int NJ = 22;       // joints
constexpr int LAT = 24;      // latent
constexpr int FPB = 16;      // frames per workgroup (= MFMA N)
constexpr int NWAVE = 8;     // waves per workgroup
constexpr int NTHREADS = NWAVE * 64;
constexpr int MAX_ITERS = 256;

// the six products of one iteration
enum { G_L0 = 0, G_L1, G_L2, G_B2, G_B1, G_B0, NGEMM };
//                               L0  L1  L2  bL2  bL1 bL0
constexpr int G_ROWS[NGEMM] = {40, 60, 92, 60, 40, 24};  // real output channels
constexpr int G_KREAL[NGEMM] = {24, 40, 60, 92, 60, 40}; // real input channels
constexpr int G_NT[NGEMM] = {3, 4, 6, 4, 3, 2};          // 16-row output tiles
constexpr int G_NS[NGEMM] = {6, 10, 16, 26, 16, 10};     // MFMA steps over K (4 channels each)
// Tile t of a product is computed by wave t (two interleaved accumulators: even / odd steps).
// Exception: layer 2 has 6 tiles; tiles 4 and 5 are split in two K-halves of 8 steps over waves
// 4..7 (wave 4+i+2*half -> tile 4+i), written to two partial planes that the reader sums.
constexpr int L2_SPLIT_TILE0 = 4;
constexpr int L2_HALF_STEPS = 8;

// LDS row strides (floats); stride/4 odd spreads the rows of a 16-byte access over the banks
constexpr int S_Z = 28, S_A0 = 52, S_A1 = 68, S_Y = 108;
constexpr int L2_ONE_COL = 60; // layer-2 bias rides on a constant-1 input column (K padding 60..63)

// per-wave weight image on the device: [L0:6][L1:10][L2:16][bL2:26][bL1:16][bL0:10]
constexpr int W_OFF_L0 = 0, W_OFF_L1 = 6, W_OFF_L2 = 16, W_OFF_B2 = 32, W_OFF_B1 = 58, W_OFF_B0 = 74;
constexpr int W_REGS = 84;
constexpr int W_FWD = W_OFF_B2;          // forward weights stay in VGPRs
constexpr int W_BWD = W_REGS - W_OFF_B2; // backward weights of waves 0..3 are parked in LDS (52 per lane)

// which (wave, product) computes which tile / step range
DP_HD constexpr int wave_tile(int g, int w) {
    if (g == G_L2) return w < 4 ? w : L2_SPLIT_TILE0 + (w & 1);
    return w < G_NT[g] ? w : -1;
}
// K-half of the split tiles: tile 4 -> waves 6 (half 0), 4 (half 1); tile 5 -> waves 5 (half 0), 7 (half 1).  With the
// shipped model's zero-block pattern this pairs the long halves with the short full tiles on each SIMD
// (SIMD = wave % 4: 7+8, 8+4, 13+2, 9+5 masked steps instead of 7+2, 8+4, 13+8, 9+5).
DP_HD constexpr int l2_half(int w) { return ((w >> 1) & 1) ^ ((w & 1) ? 0 : 1); }
DP_HD constexpr int wave_step0(int g, int w) { return (g == G_L2 && w >= 4) ? L2_HALF_STEPS * l2_half(w) : 0; }
DP_HD constexpr int wave_nsteps(int g, int w) {
    if (g == G_L2) return w < 4 ? G_NS[G_L2] : L2_HALF_STEPS;
    return w < G_NT[g] ? G_NS[g] : 0;
}

// P3 (kinematics / loss / backward): wave w handles frames 2w, 2w+1; 32 lanes per frame; one
// item per lane.  Every item reads one 4-channel quad of y, handles at most ONE child bone and
// writes one quad of dL/dy:
//   0..21  joints (item 0 = root; the root's children have constant root-frame bones)
//   22     root displacement
//   23..25 "virtual" copies of a non-root joint that has more than one child: each extra child
//          gets its own lane; the copy reads the joint's quad and writes its gradient to quad
//          23.., and the bL2 product carries duplicated weight rows for those quads, so the
//          matrix product itself sums the contributions (deterministic, no extra exchange)
//   rest   idle
constexpr int ITEM_DISP = 22;
constexpr int ITEM_VIRT0 = 23;
constexpr int MAX_VIRT = 3;    // quads 23..25 of the 26-quad gy row (K = 104)
constexpr int NQUAD_GY = 26;
constexpr int SLOT_ZERO = 23;  // bone slot that always reads zero
constexpr int SLOT_TRASH = 24; // first of 8 write-only slots
constexpr int MAX_ROOT_CH = 3;
constexpr int MAX_PATH = 7;    // bones from the root to the deepest joint
enum { KIND_JOINT = 0, KIND_ROOT = 1, KIND_DISP = 2, KIND_VIRT = 3, KIND_IDLE = 4 };

struct ItemConst { // one per item id; lives in LDS, read where needed
    float sd[4], mu[4];        // de-normalisation of the quad's 4 decoder channels
    float ch_off[3];           // offset of the child bone this item handles (zero when none)
    int ch_id;                 // bone slot to write: child joint id, or a trash slot
    unsigned ch_sub;           // bit t set <=> joint t in subtree(child)   (displacement: every joint)
    unsigned path_lo, path_hi; // 7 x 5-bit bone ids root->joint (SLOT_ZERO padded): 6 in lo, 1 in hi
    int src_quad;              // quad of y to read
    int dst_quad;              // quad of dL/dy to write (-1: none)
    int kind;
    int init_id;               // items 0..2: bone slot of the root's k-th child (constant bone), else trash
    float init_off[3];
    int pad[10];
};
static_assert(sizeof(ItemConst) == 32 * 4, "ItemConst is 32 words");

struct TrackIn { // per (frame, joint) tracker inputs, pre-scaled on load; lives in LDS
    float tp[3], cgp; // target position, 2 w_pos / (3E)
    float tR[9], cgr; // target rotation, 2 lam_rot w_rot / (9E)
    float clp, clr;   // w_pos / (3E), lam_rot w_rot / (9E)
};
static_assert(sizeof(TrackIn) == 16 * 4, "TrackIn is 16 words");

} // namespace dpl

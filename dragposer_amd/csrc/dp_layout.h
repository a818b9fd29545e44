// dp_layout.h -- compile-time description of how the optimise kernel maps the folded decoder
// onto v_mfma_f32_16x16x4_f32 tiles.  Shared by the host packer (dp_host.cpp) and the device
// code (dp_kernel.hip) so that the two cannot disagree.
//
// Orientation of every product: D[M=16 out-channels][N=16 frames] += A[M][K=4] * B[K=4][N]
//   A = weights  (loop-invariant, one VGPR per MFMA, packed on the host per wave/lane)
//   B = activations of the 16 frames of the workgroup, read from LDS
//   lane l: A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], D[row = 4*(l>>4)+r][col = l&15], r=0..3
// so a lane holds 4 consecutive output channels of ONE frame: a whole quaternion after layer 2.
//
// K ordering.  The dot product is order-free, so MFMA step i of a product takes, for lane
// group h = l>>4, the column kcol(K, i, h) below: 16-column blocks are read from LDS as one
// float4 per lane (4 steps), a trailing 8-column block as one float2 per lane (2 steps).
#pragma once

#if defined(__HIPCC__) || defined(__CUDACC__)
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
//                      L0   L1   L2   bL2  bL1  bL0
constexpr int G_ROWS[NGEMM] = {40, 60, 92, 60, 40, 24};  // real output channels
constexpr int G_KREAL[NGEMM] = {24, 40, 60, 92, 60, 40}; // real input channels
constexpr int G_K[NGEMM] = {24, 40, 64, 104, 64, 40};    // columns read from LDS (zero padded)
constexpr int G_NT[NGEMM] = {3, 4, 6, 4, 3, 2};          // 16-row tiles
constexpr int G_NM[NGEMM] = {6, 10, 16, 26, 16, 10};     // MFMA steps per tile (= K/4)
constexpr int G_HALF0[NGEMM] = {3, 5, 8, 12, 8, 5};      // steps in K-half 0 (half 1 takes the rest)

// LDS row strides (floats) of the buffers; stride/4 odd keeps float4 row reads spread over banks
constexpr int S_A0 = 52, S_A1 = 68, S_Y = 108, S_D1 = 68, S_D0 = 52, S_GZ = 36, S_AD = 28;
constexpr int L2_ONE_COL = 60; // layer-2 bias rides on a constant-1 input column (K padding 60..63)

DP_HD constexpr int kcol(int K, int i, int h) {
    return (i < (K / 16) * 4) ? 16 * (i / 4) + 4 * h + (i % 4) : (K / 16) * 16 + 2 * h + (i - (K / 16) * 4);
}

// chunk = (tile, K-half).  Wave w owns chunk slot 0 (every product) and, for L2 only, slot 1.
// Returns tile (or -1 when the wave is idle in that product); *half receives the K-half.
DP_HD constexpr int chunk_tile(int g, int w, int slot, int* half) {
    int nt = G_NT[g];
    int c = (slot == 0) ? w : w + 8;
    if (slot == 1 && !(g == G_L2 && w < 4)) return -1;
    if (c >= 2 * nt) return -1;
    *half = c / nt;
    return c % nt;
}

// per-wave weight registers: [L0:3][L1:5][L2 slot0:8][L2 slot1:8][bL2:14][bL1:8][bL0:5]
constexpr int W_OFF_L0 = 0, W_OFF_L1 = 3, W_OFF_L2A = 8, W_OFF_L2B = 16, W_OFF_B2 = 24, W_OFF_B1 = 38, W_OFF_B0 = 46;
constexpr int W_REGS = 51;

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

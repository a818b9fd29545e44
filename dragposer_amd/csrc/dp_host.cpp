// dp_host.cpp -- host side of libdragposer_hip.so: decoder folding, MFMA fragment packing,
// skeleton tables, context management and the C ABI declared in include/dragposer.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dragposer.h"
#include "dp_kernel.h"
#include "dp_sequence.h"
#include "dp_w4.h"

using namespace dpl;

struct dp_ctx {
    int device = -1;
    int n_cu = 256;
    float* d_wfrag = nullptr;
    float* d_bias = nullptr;
    float* d_w4img = nullptr;
    float* d_w4bias = nullptr;
    dpw4::Pair* d_w4pairs = nullptr;
    unsigned* d_w16img = nullptr; // 16-frames-per-wave kernel (dp_w16.hip); NULL when the skeleton is not the one its slot map is for
    float* d_w16bias = nullptr;
    dpw16::SlotConst* d_w16slots = nullptr;
    int weight_dtype = DP_WEIGHTS_FP32;
    ItemConst* d_items = nullptr;
    dp_folded folded;
    std::vector<unsigned> smask;
    std::string err;
    float mean_q0[4] = {0, 0, 0, 0}, std_q0[4] = {1, 1, 1, 1}; // root quaternion channels (sequence epilogue)
    int last_kernel = 0;   // what the last launch used: 4 = dp_w4.hip (8 in the test-only library, below)
};

// The product library has two optimise kernels: dp_w4.hip (4 frames per wave, fp32 MFMA) and dp_w16*.hip (16 frames per wave, bf16
// MFMA in split precision, for batches beyond one round of the former): include/dragposer.h, DP_KERNEL_*.  Round 1's 8-wave kernel (dp_kernel.hip: the reference's matrix
// chain taken literally, 16x16x4 tiles) survives as an independent second implementation for cross-checks in a TEST-ONLY
// library, libdragposer_hip_ref8.so = this file compiled with -DDP_REF8_BUILD + dp_kernel.o; no environment variable is read.
#ifdef DP_REF8_BUILD
constexpr int KERNEL_CHOICE = 8;
#else
constexpr int KERNEL_CHOICE = 4;
#endif

static thread_local std::string g_create_err;

// Every entry point that touches the device runs on the context's device, whatever the calling thread's current device
// is, and leaves the caller's current device as it found it (a NULL stream would otherwise launch on the wrong GPU).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
        else prev = -1; // nothing to restore
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define DEVICE_GUARD(ctx)                                                                        \
    DeviceGuard guard_((ctx)->device);                                                           \
    if (!guard_.ok) return fail(ctx, DP_ERR_DEVICE, "cannot select the context's device")

static int fail(dp_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

extern "C" int dp_version(void) { return DP_VERSION; }

extern "C" const char* dp_last_error(const dp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// ------------------------------------------------------------------------------------------------
static float round_bf16(float x)
{ // nearest-even; NaN/Inf do not occur in checkpoint weights
    unsigned u;
    std::memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    std::memcpy(&x, &u, 4);
    return x;
}

static bool model_ptrs_ok(const dp_model* m)
{
    if (!m || !m->f_latent_w || !m->f_latent_b || !m->mean_q || !m->std_q || !m->mean_disp || !m->std_disp ||
        !m->parents || !m->offsets)
        return false;
    for (int l = 0; l < 3; ++l)
        if (!m->unpool_w[l] || !m->conv_w[l] || !m->conv_mask[l] || !m->conv_b[l]) return false;
    return true;
}

// A0 = (W0*M0) U0 Wf, c0 = (W0*M0) U0 bf + b0, A1 = (W1*M1) U1, A2 = (W2*M2) U2
// (reference: autoencoder.py:228-234, skeleton.py:120,245 -- no non-linearity between these steps)
extern "C" int dp_fold_decoder(const dp_model* m, dp_folded* out)
{
    if (!model_ptrs_ok(m) || !out) return fail(nullptr, DP_ERR_INVALID, "dp_fold_decoder: NULL pointer in model");
    const int dims[4] = {24, 40, 60, 92};
    const bool bf = m->weight_dtype == DP_WEIGHTS_BF16;
    if (m->weight_dtype != DP_WEIGHTS_FP32 && !bf) return fail(nullptr, DP_ERR_INVALID, "dp_fold_decoder: unknown weight_dtype");
    auto wq = [&](float x) { return bf ? round_bf16(x) : x; };
    // T = U0 Wf (40x24), tb = U0 bf
    std::vector<double> T(40 * 24), tb(40);
    for (int i = 0; i < 40; ++i) {
        for (int k = 0; k < 24; ++k) {
            double s = 0;
            for (int j = 0; j < 24; ++j) s += (double)m->unpool_w[0][i * 24 + j] * (double)wq(m->f_latent_w[j * 24 + k]);
            T[i * 24 + k] = s;
        }
        double s = 0;
        for (int j = 0; j < 24; ++j) s += (double)m->unpool_w[0][i * 24 + j] * (double)m->f_latent_b[j];
        tb[i] = s;
    }
    auto wm = [&](int l, int i, int j) { return (double)(wq(m->conv_w[l][i * dims[l + 1] + j]) * m->conv_mask[l][i * dims[l + 1] + j]); };
    for (int i = 0; i < 40; ++i) {
        for (int k = 0; k < 24; ++k) {
            double s = 0;
            for (int j = 0; j < 40; ++j) s += wm(0, i, j) * T[j * 24 + k];
            out->A0[i * 24 + k] = (float)s;
        }
        double s = m->conv_b[0][i];
        for (int j = 0; j < 40; ++j) s += wm(0, i, j) * tb[j];
        out->c0[i] = (float)s;
    }
    for (int i = 0; i < 60; ++i) {
        for (int k = 0; k < 40; ++k) {
            double s = 0;
            for (int j = 0; j < 60; ++j) s += wm(1, i, j) * (double)m->unpool_w[1][j * 40 + k];
            out->A1[i * 40 + k] = (float)s;
        }
        out->b1[i] = m->conv_b[1][i];
    }
    for (int i = 0; i < 92; ++i) {
        for (int k = 0; k < 60; ++k) {
            double s = 0;
            for (int j = 0; j < 92; ++j) s += wm(2, i, j) * (double)m->unpool_w[2][j * 60 + k];
            out->A2[i * 60 + k] = (float)s;
        }
        out->b2[i] = m->conv_b[2][i];
    }
    return DP_OK;
}

// ------------------------------------------------------------------------------------------------
// Skeleton-derived layout of the P3 items (see dp_layout.h)
struct ItemPlan {
    int nvirt = 0;
    int virt_parent[MAX_VIRT] = {0, 0, 0}; // joint whose quad virtual item v copies
    int virt_child[MAX_VIRT] = {0, 0, 0};  // the extra child bone it handles
    int first_child[NJ];                   // child handled by the joint's own item (-1: leaf)
    int root_child[MAX_ROOT_CH] = {-1, -1, -1};
};

static int plan_items(const int* par, ItemPlan& pl, std::string& err)
{
    if (par[0] != 0) { err = "parents[0] must be 0"; return DP_ERR_INVALID; }
    for (int j = 1; j < NJ; ++j)
        if (par[j] < 0 || par[j] >= j) { err = "parents must be topologically ordered (parents[j] < j)"; return DP_ERR_INVALID; }
    for (int j = 0; j < NJ; ++j) pl.first_child[j] = -1;
    int nroot = 0;
    for (int k = 1; k < NJ; ++k) {
        const int p = par[k];
        if (p == 0) {
            if (nroot == MAX_ROOT_CH) { err = "root has more than 3 children"; return DP_ERR_UNSUPPORTED; }
            pl.root_child[nroot++] = k;
        } else if (pl.first_child[p] < 0) {
            pl.first_child[p] = k;
        } else {
            if (pl.nvirt == MAX_VIRT) { err = "more than 3 extra child bones on non-root joints"; return DP_ERR_UNSUPPORTED; }
            pl.virt_parent[pl.nvirt] = p;
            pl.virt_child[pl.nvirt] = k;
            ++pl.nvirt;
        }
    }
    return DP_OK;
}

// weight of product g at (output row, input column); zero outside the real matrix
static float gemm_w(const dp_folded& f, const ItemPlan& pl, int g, int row, int col)
{
    if (row >= G_ROWS[g]) return 0.f;
    if (g == G_L2 && col == L2_ONE_COL) return f.b2[row]; // bias rides on the constant-1 column
    if (g == G_B2 && col >= 4 * ITEM_VIRT0) {             // duplicated rows for the virtual quads
        const int v = (col - 4 * ITEM_VIRT0) / 4;
        if (v >= pl.nvirt) return 0.f;
        return f.A2[(4 * pl.virt_parent[v] + (col & 3)) * 60 + row];
    }
    if (col >= G_KREAL[g]) return 0.f;
    switch (g) {
    case G_L0: return f.A0[row * 24 + col];
    case G_L1: return f.A1[row * 40 + col];
    case G_L2: return f.A2[row * 60 + col];
    case G_B2: return (col == 91) ? 0.f : f.A2[col * 60 + row]; // A2^T; channel 91 is unused padding
    case G_B1: return f.A1[col * 40 + row]; // A1^T
    case G_B0: return f.A0[col * 24 + row]; // A0^T
    }
    return 0.f;
}

static float gemm_bias(const dp_folded& f, int g, int row)
{
    if (row >= G_ROWS[g]) return 0.f;
    switch (g) {
    case G_L0: return f.c0[row];
    case G_L1: return f.b1[row];
    }
    return 0.f;
}

// host-only, exported for the CPU tests: per-wave/per-lane MFMA operand images
//   wfrag [NWAVE][W_REGS][64], bias [2][64] (rows of c0 / b1, zero padded; b1 row 60 = 1 feeds the
//   constant-1 column that carries b2), smask [NWAVE][NGEMM] (bit i: step i of the wave's chain is non-zero)
extern "C" int dp_debug_pack(const dp_folded* f, const int* parents, float* wfrag, float* bias, unsigned* smask)
{
    if (!f || !parents || !wfrag || !bias || !smask) return DP_ERR_INVALID;
    ItemPlan pl;
    std::string err;
    int rc = plan_items(parents, pl, err);
    if (rc != DP_OK) return fail(nullptr, rc, err);
    std::memset(wfrag, 0, sizeof(float) * NWAVE * W_REGS * 64);
    std::memset(smask, 0, sizeof(unsigned) * NWAVE * NGEMM);
    for (int r = 0; r < 64; ++r) { bias[r] = gemm_bias(*f, G_L0, r); bias[64 + r] = gemm_bias(*f, G_L1, r); }
    bias[64 + L2_ONE_COL] = 1.0f; // lrelu(1) = 1: a1[:, 60] == 1
    const int woff[NGEMM] = {W_OFF_L0, W_OFF_L1, W_OFF_L2, W_OFF_B2, W_OFF_B1, W_OFF_B0};
    for (int w = 0; w < NWAVE; ++w) {
        for (int g = 0; g < NGEMM; ++g) {
            const int tile = wave_tile(g, w), s0 = wave_step0(g, w), n = wave_nsteps(g, w);
            if (tile < 0) continue;
            for (int i = 0; i < n; ++i) {
                bool any = false;
                for (int l = 0; l < 64; ++l) {
                    const float v = gemm_w(*f, pl, g, 16 * tile + (l & 15), 4 * (s0 + i) + (l >> 4));
                    wfrag[(w * W_REGS + woff[g] + i) * 64 + l] = v;
                    any = any || v != 0.f;
                }
                if (any) smask[w * NGEMM + g] |= 1u << i;
            }
        }
    }
    return DP_OK;
}

// host-only, exported for the CPU tests: weight image of the wave-private kernel (dp_w4.h)
//   img [N_GROUPS][64][4]: step s = 4 g + m of lane l at img[(g * 64 + l) * 4 + m];  bias [4][64]
// Rows of layer 2 / columns of its transpose are indexed by P3 item: row 4 * item + c is channel c of joint `item`
// (items 0..21), of the root displacement (22) or of a virtual copy of a joint with a second / third child (23..25).
static int w4_src_row(const ItemPlan& pl, int item, int c)
{ // row of A2 that feeds channel c of `item`, or -1
    if (item < 0) return -1;
    if (item < NJ) return 4 * item + c;
    if (item == ITEM_DISP) return 4 * ITEM_DISP + c; // 88..91 (91: the decoder's unused fourth displacement channel)
    const int v = item - ITEM_VIRT0;
    if (v >= 0 && v < pl.nvirt) return 4 * pl.virt_parent[v] + c;
    return -1;
}

extern "C" int dp_debug_pack_w4(const dp_folded* f, const dp_model* m, float* img, float* bias)
{
    if (!f || !model_ptrs_ok(m) || !img || !bias) return DP_ERR_INVALID;
    ItemPlan pl;
    std::string err;
    int rc = plan_items(m->parents, pl, err);
    if (rc != DP_OK) return fail(nullptr, rc, err);
    std::memset(img, 0, sizeof(float) * dpw4::IMG_FLOATS);
    std::memset(bias, 0, sizeof(float) * dpw4::BIAS_FLOATS);
    auto put = [&](int step, int lane, float v) { img[((step >> 2) * 64 + lane) * 4 + (step & 3)] = v; };
    // The de-normalisation of the decoder's last layer (drag_pose.py:84-85: r = y * sigma + mu) is folded into it:
    // rows of A2 scaled by sigma, bias sigma * b2 + mu; its transpose carries the same scaling (dL/dy = sigma * dL/dr).
    auto sd_of = [&](int item, int c) -> double {
        if (item == ITEM_DISP) return c < 3 ? (double)m->std_disp[c] : 0.0;
        const int r = w4_src_row(pl, item, c);
        return r >= 0 ? (double)m->std_q[r] : 0.0;
    };
    auto mu_of = [&](int item, int c) -> double {
        if (item == ITEM_DISP) return c < 3 ? (double)m->mean_disp[c] : 0.0;
        const int r = w4_src_row(pl, item, c);
        return r >= 0 ? (double)m->mean_q[r] : (c == 0 ? 1.0 : 0.0); // idle rows decode to the unit quaternion
    };
    for (int l = 0; l < 64; ++l) {
        const int c0ch = dpw4::h0_channel(l); // the first hidden layer's channel in row l (dp_w4.h), or -1
        for (int k = 0; k < 24; ++k) put(dpw4::S_L0 + k, l, c0ch >= 0 ? f->A0[c0ch * 24 + k] : 0.f);
        for (int k = 0; k < 40; ++k) put(dpw4::S_L1 + k, l, l < 60 ? f->A1[l * 40 + k] : 0.f); // (K-step k = channel k: quads 0..4, 8..12)
        // layer 2: row l of block blk = channel l2_channel(blk, l & 3) of the side-l2_side(l & 3) item of quad l >> 2 (dp_w4.h)
        const int it2 = dpw4::item_of(dpw4::l2_side(l & 3), l >> 2);
        const int ch2[2] = {dpw4::l2_channel(0, l & 3), dpw4::l2_channel(1, l & 3)};
        const int r2[2] = {w4_src_row(pl, it2, ch2[0]), w4_src_row(pl, it2, ch2[1])};
        for (int k = 0; k < 60; ++k) {
            put(dpw4::S_L2A + k, l, r2[0] >= 0 ? (float)(sd_of(it2, ch2[0]) * (double)f->A2[r2[0] * 60 + k]) : 0.f);
            put(dpw4::S_L2B + k, l, r2[1] >= 0 ? (float)(sd_of(it2, ch2[1]) * (double)f->A2[r2[1] * 60 + k]) : 0.f);
        }
        for (int k = 0; k < 104; ++k) { // column k = channel k & 3 of an item of dL/dr (side A quads 0..15, then side B quads 1..10)
            const int item = k < 64 ? dpw4::item_of(0, k >> 2) : dpw4::item_of(1, dpw4::B2_ABID0_B + ((k - 64) >> 2));
            const int r = w4_src_row(pl, item, k & 3);
            put(dpw4::S_B2 + k, l, (l < 60 && r >= 0) ? (float)(sd_of(item, k & 3) * (double)f->A2[r * 60 + l]) : 0.f);
        }
        for (int k = 0; k < 60; ++k) put(dpw4::S_B1 + k, l, c0ch >= 0 ? f->A1[k * 40 + c0ch] : 0.f);
        for (int k = 0; k < 20; ++k) // bL0, K split: lanes 0..31 carry K-step k, lanes 32..63 K-step 20 + k, of rows l & 31
            put(dpw4::S_B0 + k, l, (l & 31) < 24 ? f->A0[((l < 32 ? 0 : 20) + k) * 24 + (l & 31)] : 0.f);
        bias[l] = c0ch >= 0 ? f->c0[c0ch] : 0.f;
        bias[64 + l] = l < 60 ? f->b1[l] : 0.f;
        for (int blk = 0; blk < 2; ++blk) // (idle items: sigma 0, mu (1, 0, 0, 0) -- they decode to the unit quaternion)
            bias[128 + 64 * blk + l] = (float)(sd_of(it2, ch2[blk]) * (r2[blk] >= 0 ? (double)f->b2[r2[blk]] : 0.0) + mu_of(it2, ch2[blk]));
    }
    return DP_OK;
}

extern "C" int dp_debug_items(const dp_model* m, void* out_items);

// host-only, exported for the CPU tests: kinematics constants of the wave-private kernel, one dpw4::Pair per lane quad
extern "C" int dp_debug_pairs_w4(const dp_model* m, void* out_pairs /* 16 x 144 B */)
{
    if (!model_ptrs_ok(m) || !out_pairs) return fail(nullptr, DP_ERR_INVALID, "dp_debug_pairs_w4: NULL pointer");
    std::vector<ItemConst> items(32);
    int rc = dp_debug_items(m, items.data());
    if (rc != DP_OK) return rc;
    dpw4::Pair* pr = (dpw4::Pair*)out_pairs;
    std::memset(pr, 0, sizeof(dpw4::Pair) * 16);
    for (int b = 0; b < 16; ++b)
        for (int s = 0; s < 2; ++s) {
            dpw4::Pair& p = pr[b];
            const int item = dpw4::item_of(s, b);
            p.item[s] = item;
            p.kind[s] = KIND_IDLE;
            p.mu[0][s] = 1.f; // idle: a unit quaternion, whatever the (zero) decoder channels say
            p.sgn[s] = 1.f;
            p.bone_slot[s] = SLOT_TRASH + ((2 * b + s) & 7);
            if (item < 0) continue;
            const ItemConst& c = items[item];
            p.kind[s] = c.kind;
            if (c.kind == KIND_IDLE) continue;
            for (int k = 0; k < 4; ++k) { p.sd[k][s] = c.sd[k]; p.mu[k][s] = c.mu[k]; }
            for (int k = 0; k < 3; ++k) p.off[k][s] = c.ch_off[k];
            p.bone_slot[s] = c.ch_id;
            p.ch_sub[s] = c.ch_sub;
            if (c.kind == KIND_ROOT) { p.sgn[s] = -1.f; p.rho[s] = 1.f; p.ch_sub[s] = (1u << NJ) - 1u; }
        }
    return DP_OK;
}

// host-only, exported for the CPU tests: P3 per-item constants [32]
extern "C" int dp_debug_items(const dp_model* m, void* out_items /* 32 x 128 B */)
{
    if (!model_ptrs_ok(m) || !out_items) return fail(nullptr, DP_ERR_INVALID, "dp_debug_items: NULL pointer");
    ItemConst* it = (ItemConst*)out_items;
    std::memset(it, 0, sizeof(ItemConst) * 32);
    const int* par = m->parents;
    ItemPlan pl;
    std::string err;
    int rc = plan_items(par, pl, err);
    if (rc != DP_OK) return fail(nullptr, rc, err);
    unsigned sub[NJ]; // subtree masks
    for (int j = 0; j < NJ; ++j) sub[j] = 1u << j;
    for (int j = NJ - 1; j >= 1; --j) sub[par[j]] |= sub[j];
    auto set_child = [&](ItemConst& c, int k) {
        c.ch_id = k;
        c.ch_sub = sub[k];
        for (int a = 0; a < 3; ++a) c.ch_off[a] = m->offsets[3 * k + a];
    };
    for (int id = 0; id < 32; ++id) {
        ItemConst& c = it[id];
        c.kind = KIND_IDLE;
        c.ch_id = SLOT_TRASH + (id & 7);
        c.init_id = SLOT_TRASH + (id & 7);
        c.src_quad = ITEM_VIRT0; // a quad layer 2 always writes as zeros (channels 92..95)
        c.dst_quad = (id < NQUAD_GY) ? id : -1;
        unsigned long long path = 0;
        for (int i = 0; i < MAX_PATH; ++i) path |= (unsigned long long)SLOT_ZERO << (5 * i);
        if (id < NJ) {
            c.kind = id == 0 ? KIND_ROOT : KIND_JOINT;
            c.src_quad = id;
            for (int k = 0; k < 4; ++k) { c.sd[k] = m->std_q[4 * id + k]; c.mu[k] = m->mean_q[4 * id + k]; }
            if (id > 0 && pl.first_child[id] >= 0) set_child(c, pl.first_child[id]);
            int chain[NJ], n = 0; // bones on the path root -> id (joint ids, root excluded)
            for (int k = id; k != 0; k = par[k]) chain[n++] = k;
            if (n > MAX_PATH) return fail(nullptr, DP_ERR_UNSUPPORTED, "kinematic chain deeper than 7 bones");
            path = 0;
            for (int i = 0; i < MAX_PATH; ++i) path |= (unsigned long long)(i < n ? chain[i] : SLOT_ZERO) << (5 * i);
        } else if (id == ITEM_DISP) {
            c.kind = KIND_DISP;
            c.src_quad = ITEM_DISP;
            for (int k = 0; k < 3; ++k) { c.sd[k] = m->std_disp[k]; c.mu[k] = m->mean_disp[k]; }
            c.ch_sub = (1u << NJ) - 1u; // the displacement's "subtree" is every joint
        } else if (id - ITEM_VIRT0 < pl.nvirt) {
            const int v = id - ITEM_VIRT0, j = pl.virt_parent[v];
            c.kind = KIND_VIRT;
            c.src_quad = j;
            for (int k = 0; k < 4; ++k) { c.sd[k] = m->std_q[4 * j + k]; c.mu[k] = m->mean_q[4 * j + k]; }
            set_child(c, pl.virt_child[v]);
        }
        if (id < MAX_ROOT_CH && pl.root_child[id] >= 0) {
            c.init_id = pl.root_child[id];
            for (int a = 0; a < 3; ++a) c.init_off[a] = m->offsets[3 * pl.root_child[id] + a];
        }
        c.path_lo = (unsigned)(path & 0x3FFFFFFFull);
        c.path_hi = (unsigned)(path >> 30);
    }
    return DP_OK;
}

// ------------------------------------------------------------------------------------------------
#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(ctx, DP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int dp_create(dp_ctx** out, const dp_model* model, int device)
{
    if (!out) return fail(nullptr, DP_ERR_INVALID, "dp_create: out is NULL");
    *out = nullptr;
    if (!model_ptrs_ok(model)) return fail(nullptr, DP_ERR_INVALID, "dp_create: NULL pointer in model");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DP_ERR_DEVICE, "dp_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, DP_ERR_INVALID, "dp_create: bad device index");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DP_ERR_DEVICE, std::string("dp_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    dp_ctx* ctx = new dp_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int k = 0; k < 4; ++k) { ctx->mean_q0[k] = model->mean_q[k]; ctx->std_q0[k] = model->std_q[k]; }
    int rc = dp_fold_decoder(model, &ctx->folded);
    std::vector<float> wfrag(NWAVE * W_REGS * 64), bfrag(128);
    ctx->smask.assign(NWAVE * NGEMM, 0u);
    std::vector<ItemConst> items(32);
    std::vector<float> w4img(dpw4::IMG_FLOATS), w4bias(dpw4::BIAS_FLOATS);
    if (rc == DP_OK) rc = dp_debug_pack(&ctx->folded, model->parents, wfrag.data(), bfrag.data(), ctx->smask.data());
    if (rc == DP_OK) rc = dp_debug_pack_w4(&ctx->folded, model, w4img.data(), w4bias.data());
    if (rc == DP_OK) rc = dp_debug_items(model, items.data());
    std::vector<dpw4::Pair> pairs(16);
    if (rc == DP_OK) rc = dp_debug_pairs_w4(model, pairs.data());
    const bool w16 = rc == DP_OK && dp_w16_supported(model);
    std::vector<unsigned> w16img(w16 ? dpw16::IMG_U32 : 0);
    std::vector<float> w16bias(dpw16::BIAS_FLOATS);
    std::vector<dpw16::SlotConst> w16slots(dpw16::NTY * 4);
    if (w16) rc = dp_debug_pack_w16(&ctx->folded, model, w16img.data(), w16bias.data(), w16slots.data());
    ctx->weight_dtype = model->weight_dtype;
    if (rc != DP_OK) { delete ctx; return rc; }
    int prev = 0;
    hipGetDevice(&prev);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_wfrag, wfrag.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_bias, bfrag.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_items, items.size() * sizeof(ItemConst));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w4img, w4img.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w4bias, w4bias.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w4pairs, pairs.size() * sizeof(dpw4::Pair));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_w4pairs, pairs.data(), pairs.size() * sizeof(dpw4::Pair), hipMemcpyHostToDevice);
    if (w16) {
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w16img, w16img.size() * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w16bias, w16bias.size() * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_w16slots, w16slots.size() * sizeof(dpw16::SlotConst));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_w16img, w16img.data(), w16img.size() * sizeof(unsigned), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ctx->d_w16bias, w16bias.data(), w16bias.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ctx->d_w16slots, w16slots.data(), w16slots.size() * sizeof(dpw16::SlotConst), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(ctx->d_w4img, w4img.data(), w4img.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_w4bias, w4bias.data(), w4bias.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_wfrag, wfrag.data(), wfrag.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_bias, bfrag.data(), bfrag.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_items, items.data(), items.size() * sizeof(ItemConst), hipMemcpyHostToDevice);
    hipSetDevice(prev);
    if (e != hipSuccess) {
        std::string msg = std::string("dp_create: ") + hipGetErrorString(e);
        hipFree(ctx->d_wfrag); hipFree(ctx->d_bias); hipFree(ctx->d_items); hipFree(ctx->d_w4img); hipFree(ctx->d_w4bias); hipFree(ctx->d_w4pairs);
        hipFree(ctx->d_w16img); hipFree(ctx->d_w16bias); hipFree(ctx->d_w16slots);
        delete ctx;
        return fail(nullptr, DP_ERR_DEVICE, msg);
    }
    *out = ctx;
    return DP_OK;
}

extern "C" int dp_destroy(dp_ctx* ctx)
{
    if (!ctx) return DP_ERR_INVALID;
    DeviceGuard guard_(ctx->device);
    hipFree(ctx->d_wfrag);
    hipFree(ctx->d_bias);
    hipFree(ctx->d_items);
    hipFree(ctx->d_w4img);
    hipFree(ctx->d_w4bias);
    hipFree(ctx->d_w4pairs);
    hipFree(ctx->d_w16img);
    hipFree(ctx->d_w16bias);
    hipFree(ctx->d_w16slots);
    delete ctx;
    return DP_OK;
}

extern "C" int dp_io_alloc(dp_ctx* ctx, unsigned long long bytes, void** dev_ptr)
{
    if (!ctx || !dev_ptr) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipMalloc(dev_ptr, bytes));
    HIP_TRY(ctx, hipMemset(*dev_ptr, 0, bytes));
    return DP_OK;
}

extern "C" int dp_io_free(dp_ctx* ctx, void* dev_ptr)
{
    if (!ctx) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipFree(dev_ptr));
    return DP_OK;
}

extern "C" int dp_io_upload(dp_ctx* ctx, void* dev_dst, const void* host_src, unsigned long long bytes, void* stream)
{
    if (!ctx || !dev_dst || !host_src) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return DP_OK;
}

extern "C" int dp_io_download(dp_ctx* ctx, void* host_dst, const void* dev_src, unsigned long long bytes, void* stream)
{
    if (!ctx || !host_dst || !dev_src) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return DP_OK;
}

extern "C" int dp_io_alloc_host(dp_ctx* ctx, unsigned long long bytes, void** host_ptr)
{
    if (!ctx || !host_ptr) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    std::memset(*host_ptr, 0, bytes);
    return DP_OK;
}

extern "C" int dp_io_free_host(dp_ctx* ctx, void* host_ptr)
{
    if (!ctx) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipHostFree(host_ptr));
    return DP_OK;
}

extern "C" int dp_stream_sync(dp_ctx* ctx, void* stream)
{
    if (!ctx) return DP_ERR_INVALID;
    DEVICE_GUARD(ctx);
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return DP_OK;
}

extern "C" int dp_kernel_geometry(const dp_ctx* ctx, int* frames_per_block, int* threads_per_block, int* lds_bytes)
{ // of the kernel the context's launches use
    (void)ctx;
#ifdef DP_REF8_BUILD
    if (frames_per_block) *frames_per_block = FPB;
    if (threads_per_block) *threads_per_block = NTHREADS;
    if (lds_bytes) *lds_bytes = dp_kernel_lds_bytes();
#else
    const int w16w = ctx && ctx->last_kernel >= 16 ? ctx->last_kernel / 4 : 0; // waves per workgroup of the last dp_w16 launch
    const bool k16 = w16w != 0;
    if (frames_per_block) *frames_per_block = k16 ? w16w * dp_w16_frames_per_wave() : dp_w4_frames_per_block();
    if (threads_per_block) *threads_per_block = k16 ? w16w * 64 : 256;
    if (lds_bytes) *lds_bytes = k16 ? dp_w16_lds_bytes() : dp_w4_lds_bytes();
#endif
    return DP_OK;
}

static void fill_model_args(const dp_ctx* ctx, KArgs& k)
{
    std::memset(&k, 0, sizeof(k));
    k.wfrag = ctx->d_wfrag;
    k.bias = ctx->d_bias;
    k.items = ctx->d_items;
    k.w4img = ctx->d_w4img;
    k.w4bias = ctx->d_w4bias;
    k.w4pairs = ctx->d_w4pairs;
    k.w16img = ctx->d_w16img;
    k.w16bias = ctx->d_w16bias;
    k.w16slots = ctx->d_w16slots;
    std::memcpy(k.smask, ctx->smask.data(), sizeof(k.smask));
}

// dp_params / dp_result as the caller compiled them (include/dragposer.h: struct_size): a copy of the first struct_size bytes over a zeroed
// struct of THIS build -- a field the caller's header did not have reads as its default, a caller built against a pre-0.5 header is refused
constexpr unsigned PARAMS_SIZE_V500 = offsetof(dp_params, kernel) + sizeof(int);
constexpr unsigned RESULT_SIZE_V500 = offsetof(dp_result, clock) + sizeof(void*);
static int take_params(dp_ctx* ctx, const dp_params* p, dp_params& o, const char* who)
{
    if (p->struct_size < PARAMS_SIZE_V500 || p->struct_size > 4096u)
        return fail(ctx, DP_ERR_INVALID, std::string(who) + ": dp_params.struct_size is " + std::to_string(p->struct_size) + ", this library (DP_VERSION " +
                                             std::to_string(DP_VERSION) + ") expects at least " + std::to_string(PARAMS_SIZE_V500) +
                                             " -- was the caller compiled against a pre-0.5 dragposer.h?  (dp_params p = DP_PARAMS_INIT;)");
    // A 0.4 caller's struct is 52 bytes and starts with n_iter: one that asks for 56 ... 4096 iterations passes the size test above.  Its SECOND word is
    // `lr`, whose bits read as an iteration count are beyond DP_MAX_ITERS for any learning rate above 1.4e-39 -- tested HERE, on the two words every
    // layout has, before anything is copied: the 52-byte struct is never read past.
    if (p->n_iter < 1 || p->n_iter > DP_MAX_ITERS)
        return fail(ctx, DP_ERR_INVALID, std::string(who) + ": n_iter " + std::to_string(p->n_iter) + " out of range [1, DP_MAX_ITERS] (dp_params.struct_size " +
                                             std::to_string(p->struct_size) + ": if that is the iteration count you meant, the caller was compiled against a pre-0.5 "
                                             "dragposer.h, whose dp_params starts with n_iter; dp_params p = DP_PARAMS_INIT;)");
    std::memset(&o, 0, sizeof(o));
    std::memcpy(&o, p, std::min<size_t>(p->struct_size, sizeof(o)));
    return DP_OK;
}
static int take_result(dp_ctx* ctx, const dp_result* r, dp_result& o, const char* who)
{
    std::memset(&o, 0, sizeof(o));
    if (!r) return DP_OK;
    if (r->struct_size < RESULT_SIZE_V500 || r->struct_size > 4096u || r->reserved0 != 0u)
        return fail(ctx, DP_ERR_INVALID, std::string(who) + ": dp_result.struct_size is " + std::to_string(r->struct_size) + " (reserved0 " +
                                             std::to_string(r->reserved0) + "), this library (DP_VERSION " + std::to_string(DP_VERSION) + ") expects at least " +
                                             std::to_string(RESULT_SIZE_V500) + " and reserved0 = 0 -- was the caller compiled against a pre-0.5 dragposer.h?  (dp_result r = DP_RESULT_INIT;)");
    std::memcpy(&o, r, std::min<size_t>(r->struct_size, sizeof(o)));
    return DP_OK;
}

// Adam's per-iteration scalars, as torch computes them in Python doubles: a table for the first MAX_ITERS iterations (in the kernel arguments), and
// what the kernels need to continue the two products on the device beyond it (dp_kernel.h: AdamCont)
static void fill_adam(KArgs& k, const dp_params& p)
{
    double b1t = 1.0, b2t = 1.0;
    for (int t = 0; t < MAX_ITERS; ++t) {
        b1t *= (double)p.beta1;
        b2t *= (double)p.beta2;
        if (t < p.n_iter) {
            k.tab.step[t] = (float)((double)p.lr / (1.0 - b1t));
            k.tab.bc2s[t] = (float)(1.0 / std::sqrt(1.0 - b2t));
        }
    }
    k.cont.beta1 = (double)p.beta1; k.cont.beta2 = (double)p.beta2; k.cont.lr = (double)p.lr;
    k.cont.b1t = b1t; k.cont.b2t = b2t;
}

static void fill_results(const dp_result* out, KArgs& k)
{
    if (!out) return;
    k.status = out->status; k.clk = out->clock;
    k.z = out->z; k.z_pre = out->z_pre; k.pose = out->pose; k.disp = out->disp; k.world_disp = out->world_disp;
    k.world_rot = out->world_rot; k.pos = out->pos; k.rot = out->rot; k.loss = out->loss; k.iters = out->iters;
}

// Which kernel runs a launch: the wave-private kernel of dp_w4.hip (4 frames per wave, no workgroup barrier in the
// loop); in the test-only library the previous decomposition (dp_kernel.hip: 16 frames per 8-wave workgroup).  Both
// implement the same operator within the tolerance of tests/test_hip_w4.py.
static int launch(dp_ctx* ctx, KArgs& k, void* stream, int kernel = DP_KERNEL_W4)
{
    DEVICE_GUARD(ctx);
    ctx->last_kernel = KERNEL_CHOICE;
#ifdef DP_REF8_BUILD
    (void)kernel;
    hipError_t e = dp_launch_optimize(&k, (hipStream_t)stream);
#else
    // dp_w16: one wave per SIMD (4 waves, 64 frames per workgroup) until every SIMD of the chip has a wave; beyond that two
    // (8 waves, 128 frames per workgroup): one wave's matrix phases under the other's vector phases
    const int w16_waves = k.n_frames > ctx->n_cu * 4 * dp_w16_frames_per_wave() ? 8 : 4;
    if (kernel == DP_KERNEL_W16) ctx->last_kernel = 16 * (w16_waves / 4);
    hipError_t e = kernel == DP_KERNEL_W16 ? dp_launch_w16(&k, (hipStream_t)stream, w16_waves) : dp_launch_w4(&k, (hipStream_t)stream);
#endif
    if (e != hipSuccess) return fail(ctx, DP_ERR_LAUNCH, std::string("kernel launch: ") + hipGetErrorString(e));
    return DP_OK;
}

// private extension used by the tests: same as dp_optimize, plus an optional debug dump
// [B][240] = y(104) | dL/dy(104) | dL/dz(24) | pad, all of iteration 0.
extern "C" int dp_optimize_debug(dp_ctx* ctx, const dp_batch* in, const dp_params* p_in, const dp_result* out_in, float* dbg, void* stream)
{
    if (!ctx) return DP_ERR_INVALID;
    if (!in || !p_in) return fail(ctx, DP_ERR_INVALID, "dp_optimize: NULL batch/params");
    dp_params pv; dp_result ov;
    if (int rc = take_params(ctx, p_in, pv, "dp_optimize")) return rc;
    if (int rc = take_result(ctx, out_in, ov, "dp_optimize")) return rc;
    const dp_params* p = &pv;
    const dp_result* out = &ov;
    if (in->n_frames <= 0) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_frames must be positive");
    if (!in->z0 || !in->z_tgt || !in->cur_rot || !in->tgt_pos || !in->tgt_rot || !in->w || !in->tracked)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: NULL input array");
#ifdef DP_REF8_BUILD
    if (p->n_iter < 1 || p->n_iter > MAX_ITERS) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_iter out of range [1,256] (the test-only kernel reads the argument table only)");
#else
    if (p->n_iter < 1 || p->n_iter > DP_MAX_ITERS) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_iter out of range [1, DP_MAX_ITERS]");
#endif
    if (!(p->lr > 0.f) || !(p->beta1 >= 0.f && p->beta1 < 1.f) || !(p->beta2 >= 0.f && p->beta2 < 1.f))
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: bad Adam hyper-parameters");
    if (!(p->eps > 0.f)) return fail(ctx, DP_ERR_INVALID, "dp_optimize: Adam eps must be > 0 (include/dragposer.h: dp_params.eps)");
    KArgs k;
    fill_model_args(ctx, k);
    k.z0 = in->z0; k.z_tgt = in->z_tgt; k.cur_rot = in->cur_rot; k.tgt_pos = in->tgt_pos; k.tgt_rot = in->tgt_rot;
    k.w = in->w; k.tracked = in->tracked;
    fill_results(out, k);
    k.dbg = dbg;
    k.n_frames = in->n_frames; k.n_iter = p->n_iter; k.mode = 0;
    k.lam_rot = p->lambda_rot; k.lam_tmp = p->lambda_tmp; k.ctmp = 2.f * p->lambda_tmp / 24.f;
    // torch passes (1-beta) as Python doubles into fp32 tensor ops
    k.beta2 = p->beta2; k.one_m_b1 = (float)(1.0 - (double)p->beta1); k.one_m_b2 = (float)(1.0 - (double)p->beta2);
    k.eps = p->eps;
    k.early_stop = p->early_stop ? 1 : 0;
    k.stop_eps_pos = p->stop_eps_pos; k.stop_eps_rot = p->stop_eps_rot; k.min_loss_incr = p->min_loss_incr;
    fill_adam(k, *p);
    // which kernel (include/dragposer.h: DP_KERNEL_*)
    const bool w16_can = ctx->d_w16img != nullptr;
    if (p->kernel != DP_KERNEL_AUTO && p->kernel != DP_KERNEL_W4 && p->kernel != DP_KERNEL_W16)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: unknown kernel selector");
    if (p->kernel == DP_KERNEL_W16 && !w16_can)
        return fail(ctx, DP_ERR_UNSUPPORTED, "dp_optimize: DP_KERNEL_W16 is laid out for the reference's 22-joint skeleton only");
    // (beyond the argument table of Adam scalars, n_iter > 256, both kernels have LONG instantiations that continue them on the device)
    const int kernel = p->kernel == DP_KERNEL_AUTO ? dp_auto_kernel(ctx, in->n_frames) : p->kernel;
    return launch(ctx, k, stream, kernel);
}

extern "C" int dp_auto_kernel(const dp_ctx* ctx, int n_frames)
{
    if (!ctx || n_frames <= 0) return DP_ERR_INVALID;
#ifdef DP_REF8_BUILD
    return DP_KERNEL_W4;
#else
    // more than TWO rounds of dp_w4's 16 frames per CU: dp_w16 (where its slot map fits the skeleton).  Up to two rounds dp_w4 is the faster one
    // since round 5 (its waves' staggered start: 0.252 ms for 8192 frames against dp_w16's 0.273 at the steady clock, 0.251 against 0.274 at 6144,
    // profiles/r05_batch_sweep.txt; equal from an idle GPU); from three rounds on dp_w16 wins by 1.3x and more
    return ctx->d_w16img != nullptr && n_frames > ctx->n_cu * 32 ? DP_KERNEL_W16 : DP_KERNEL_W4;
#endif
}

extern "C" int dp_optimize(dp_ctx* ctx, const dp_batch* in, const dp_params* p, const dp_result* out, void* stream)
{
    return dp_optimize_debug(ctx, in, p, out, nullptr, stream);
}

extern "C" int dp_forward(dp_ctx* ctx, int n_frames, const float* z, const float* cur_rot, const dp_result* out, void* stream)
{
    if (!ctx) return DP_ERR_INVALID;
    if (n_frames <= 0 || !z || !cur_rot || !out) return fail(ctx, DP_ERR_INVALID, "dp_forward: bad arguments");
    dp_result ov;
    if (int rc = take_result(ctx, out, ov, "dp_forward")) return rc;
    out = &ov;
    KArgs k;
    fill_model_args(ctx, k);
    k.z0 = z; k.cur_rot = cur_rot;
    fill_results(out, k);
    k.z = nullptr; k.z_pre = nullptr; k.loss = nullptr; k.iters = nullptr; k.clk = nullptr;
    k.n_frames = n_frames; k.n_iter = 1; k.mode = 1;
    return launch(ctx, k, stream);
}

// ------------------------------------------------------------------------------------------------
// n_steps frames of S sequences in one launch (+ one for the history buffers), see include/dragposer.h
extern "C" int dp_optimize_sequence(dp_ctx* ctx, int n_seq, float* latent, const dp_seq_frames* fr, const dp_params* p_in, const dp_seq_state* st,
                                    const dp_seq_step* adj, const dp_seq_results* out, void* stream)
{
    if (!ctx) return DP_ERR_INVALID;
#ifdef DP_REF8_BUILD
    return fail(ctx, DP_ERR_UNSUPPORTED, "dp_optimize_sequence: not part of the test-only library");
#else
    if (n_seq <= 0 || !latent || !fr || !p_in || !st || !out) return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: bad arguments");
    dp_params pv;
    if (int rc = take_params(ctx, p_in, pv, "dp_optimize_sequence")) return rc;
    const dp_params* p = &pv;
    dp_seq_results ov;
    {
        constexpr unsigned SEQ_RESULTS_SIZE_V500 = offsetof(dp_seq_results, status) + sizeof(void*);
        if (out->struct_size < SEQ_RESULTS_SIZE_V500 || out->struct_size > 4096u || out->reserved0 != 0u)
            return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: dp_seq_results.struct_size is " + std::to_string(out->struct_size) + ", this library (DP_VERSION " +
                                                 std::to_string(DP_VERSION) + ") expects at least " + std::to_string(SEQ_RESULTS_SIZE_V500) +
                                                 " and reserved0 = 0 -- was the caller compiled against a pre-0.5 dragposer.h?  (dp_seq_results r = DP_SEQ_RESULTS_INIT;)");
        std::memset(&ov, 0, sizeof(ov));
        std::memcpy(&ov, out, std::min<size_t>(out->struct_size, sizeof(ov)));
        out = &ov;
    }
    if (fr->n_steps <= 0 || !fr->tgt_pos || !fr->tgt_rot || !fr->w || !fr->tracked || !fr->z_tgt)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: NULL input array / n_steps must be positive");
    if (!st->global_pos || !st->global_rot || !st->latent_buf || !st->disp_buf || !st->heights_buf || !out->hist_scratch)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: NULL state array / hist_scratch");
    if (st->history < 1 || st->n_heights < 0 || st->n_heights > DP_MAX_HEIGHT_JOINTS)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: history / n_heights out of range");
    for (int h = 0; h < st->n_heights; ++h)
        if (st->height_joints[h] < 0 || st->height_joints[h] >= NJ) return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: bad height joint");
    if (adj && (adj->adjust_joint >= NJ || (adj->adjust_joint >= 0 && (adj->adjust_target_joint < 0 || adj->adjust_target_joint >= NJ))))
        return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: bad joint adjustment");
    if (p->n_iter < 1 || p->n_iter > DP_MAX_ITERS) return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: n_iter out of range [1, DP_MAX_ITERS]");
    if (!(p->lr > 0.f) || !(p->beta1 >= 0.f && p->beta1 < 1.f) || !(p->beta2 >= 0.f && p->beta2 < 1.f))
        return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: bad Adam hyper-parameters");
    if (!(p->eps > 0.f)) return fail(ctx, DP_ERR_INVALID, "dp_optimize_sequence: Adam eps must be > 0 (include/dragposer.h: dp_params.eps)");
    KArgs k;
    fill_model_args(ctx, k);
    k.z0 = latent; k.z_tgt = fr->z_tgt; k.cur_rot = st->global_rot; k.tgt_pos = fr->tgt_pos; k.tgt_rot = fr->tgt_rot; k.w = fr->w; k.tracked = fr->tracked;
    k.z = latent; k.pose = out->pose_ret; k.world_rot = out->world_rot; k.iters = out->iters; k.loss = out->loss; k.status = out->status;
    k.n_frames = n_seq; k.n_iter = p->n_iter; k.mode = 0;
    k.lam_rot = p->lambda_rot; k.lam_tmp = p->lambda_tmp; k.ctmp = 2.f * p->lambda_tmp / 24.f;
    k.beta2 = p->beta2; k.one_m_b1 = (float)(1.0 - (double)p->beta1); k.one_m_b2 = (float)(1.0 - (double)p->beta2);
    k.eps = p->eps;
    k.early_stop = 1;
    k.stop_eps_pos = p->stop_eps_pos; k.stop_eps_rot = p->stop_eps_rot; k.min_loss_incr = p->min_loss_incr;
    fill_adam(k, *p);
    SeqK& q = k.seq;
    q.n_steps = fr->n_steps; q.z_tgt_step = fr->z_tgt_step; q.z_tgt_seq = fr->z_tgt_seq; q.tgt_root = fr->tgt_root;
    q.global_pos = st->global_pos; q.global_rot = st->global_rot; q.hist = out->hist_scratch; q.pos_ret = out->pos_ret;
    q.n_heights = st->n_heights;
    for (int h = 0; h < st->n_heights; ++h) q.height_joints[h] = st->height_joints[h];
    q.adjust_joint = adj ? adj->adjust_joint : -1;
    q.adjust_target_joint = adj ? adj->adjust_target_joint : -1;
    q.adjust_weight = adj ? adj->adjust_weight : 0.f;
    for (int c = 0; c < 4; ++c) { q.mean_q0[c] = ctx->mean_q0[c]; q.std_q0[c] = ctx->std_q0[c]; }
    int rc = launch(ctx, k, stream, DP_KERNEL_W4);
    if (rc != DP_OK) return rc;
    DEVICE_GUARD(ctx);
    HistArgs h;
    h.n_seq = n_seq; h.n_steps = fr->n_steps; h.history = st->history; h.n_heights = st->n_heights;
    h.scratch = out->hist_scratch; h.latent_buf = st->latent_buf; h.disp_buf = st->disp_buf; h.heights_buf = st->heights_buf;
    hipError_t e = dp_launch_sequence_history(&h, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, DP_ERR_LAUNCH, std::string("history launch: ") + hipGetErrorString(e));
    return DP_OK;
#endif
}

// ------------------------------------------------------------------------------------------------
// per-frame epilogue of S sequences (reference drag_pose.py:369-402), see include/dragposer.h
extern "C" int dp_sequence_advance(dp_ctx* ctx, int n_seq, const dp_result* res, const dp_seq_state* st, const dp_seq_step* step, void* stream)
{
    if (!ctx) return DP_ERR_INVALID;
    if (n_seq <= 0 || !res || !st || !step) return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: bad arguments");
    dp_result rv;
    if (int rc = take_result(ctx, res, rv, "dp_sequence_advance")) return rc;
    res = &rv;
    if (!res->z_pre || !res->pose || !res->disp || !res->world_disp || !res->world_rot || !res->pos)
        return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: the frame result needs z_pre, pose, disp, world_disp, world_rot and pos");
    if (!st->global_pos || !st->global_rot || !st->latent_buf || !st->disp_buf || !st->heights_buf)
        return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: NULL state array");
    if (st->history < 1 || st->n_heights < 0 || st->n_heights > DP_MAX_HEIGHT_JOINTS)
        return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: history / n_heights out of range");
    for (int h = 0; h < st->n_heights; ++h)
        if (st->height_joints[h] < 0 || st->height_joints[h] >= NJ) return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: bad height joint");
    if (step->adjust_joint >= NJ || (step->adjust_joint >= 0 && (step->adjust_target_joint < 0 || step->adjust_target_joint >= NJ || !step->tgt_pos)))
        return fail(ctx, DP_ERR_INVALID, "dp_sequence_advance: bad joint adjustment");
    SeqArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_seq = n_seq; a.history = st->history; a.n_heights = st->n_heights;
    for (int h = 0; h < st->n_heights; ++h) a.height_joints[h] = st->height_joints[h];
    a.adjust_joint = step->adjust_joint < 0 ? -1 : step->adjust_joint;
    a.adjust_target_joint = step->adjust_target_joint; a.adjust_weight = step->adjust_weight;
    for (int k = 0; k < 4; ++k) { a.mean_q0[k] = ctx->mean_q0[k]; a.std_q0[k] = ctx->std_q0[k]; }
    a.z_pre = res->z_pre; a.pose = res->pose; a.disp = res->disp; a.world_disp = res->world_disp; a.world_rot = res->world_rot; a.pos = res->pos;
    a.tgt_pos = step->tgt_pos;
    a.global_pos = st->global_pos; a.global_rot = st->global_rot; a.latent_buf = st->latent_buf; a.disp_buf = st->disp_buf;
    a.heights_buf = st->heights_buf; a.pose_ret = step->pose_ret; a.pos_ret = step->pos_ret;
    DEVICE_GUARD(ctx);
    hipError_t e = dp_launch_sequence_advance(&a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, DP_ERR_LAUNCH, std::string("sequence kernel launch: ") + hipGetErrorString(e));
    return DP_OK;
}

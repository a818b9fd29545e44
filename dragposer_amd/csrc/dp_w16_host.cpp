// dp_w16_host.cpp -- host side of the 16-frames-per-wave kernel (dp_w16.hip): the split-precision weight image.
// Every fp32 entry of the folded decoder matrices (dp_fold_decoder; for BASELINE config 5 folded from the bf16-rounded tensors)
// is written as the exact sum of three bf16 numbers (round-to-nearest-even at every stage), laid out as the A operands of
// v_mfma_f32_16x16x32_bf16 in the order the kernel consumes them (dp_w16.h).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/dragposer.h"
#include "dp_w16.h"

using namespace dpw16;

namespace {

inline uint16_t bf16_rne(float v)
{
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}
// v = t[0] + t[1] + t[2] exactly (fp32 has 24 significant bits, three bf16 terms carry 8 each; the remainders are exact fp32 subtractions)
inline void split3(float v, uint16_t t[3])
{
    t[0] = bf16_rne(v);
    const float r1 = v - bf16_to_f32(t[0]);
    t[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(t[1]);
    t[2] = bf16_rne(r2);
}

// decoder row behind row rho = 16 t + 4 g + r of the kernel's output order (-1: none), and its de-normalisation
inline int dec_row(int rho)
{
    const int item = slot_item(rho >> 4, (rho >> 2) & 3), r = rho & 3;
    if (item == ITEM_PAD) return -1;
    if (item == dpl::ITEM_DISP) return r < 3 ? 88 + r : -1;
    return 4 * item + r;
}
inline double sd_of(const dp_model* m, int rho)
{
    const int d = dec_row(rho);
    return d < 0 ? 0.0 : d >= 88 ? (double)m->std_disp[d - 88] : (double)m->std_q[d];
}
inline double mu_of(const dp_model* m, int rho)
{
    const int d = dec_row(rho);
    if (d < 0) return (rho & 3) == 0 && slot_item(rho >> 4, (rho >> 2) & 3) == ITEM_PAD ? 1.0 : 0.0; // the idle slot decodes to the unit quaternion
    return d >= 88 ? (double)m->mean_disp[d - 88] : (double)m->mean_q[d];
}

// entry (out, in) of product `layer` as the kernel multiplies it; the de-normalisation r = y sigma + mu of the decoder's last
// layer (drag_pose.py:84-85) is folded into layer 2 (rows scaled by sigma), and into its transpose (dL/dy = sigma dL/dr)
float entry(const dp_folded* f, const dp_model* m, int layer, int out, int in)
{
    if (out >= N_OUT[layer] || in >= N_IN[layer]) return 0.f;
    switch (layer) {
    case L0: return f->A0[out * 24 + in];
    case L1: return f->A1[out * 40 + in];
    case L2: { const int d = dec_row(out); return d < 0 ? 0.f : (float)(sd_of(m, out) * (double)f->A2[d * 60 + in]); }
    case B2: { const int d = dec_row(in); return d < 0 ? 0.f : (float)(sd_of(m, in) * (double)f->A2[d * 60 + out]); }
    case B1: return f->A1[in * 40 + out];
    default: return f->A0[in * 24 + out];
    }
}

} // namespace

extern "C" int dp_w16_supported(const dp_model* m)
{ // the slot map of dp_w16.h is the reference skeleton's
    if (!m || !m->parents) return 0;
    for (int j = 0; j < dpl::NJ; ++j)
        if (m->parents[j] != PARENTS[j]) return 0;
    return 1;
}

// host-only, exported for the CPU tests: img [IMG_U32] words, bias [BIAS_FLOATS], slots [NTY * 4] SlotConst
extern "C" int dp_debug_pack_w16(const dp_folded* f, const dp_model* m, unsigned* img, float* bias, void* slots_out)
{
    if (!f || !m || !img || !bias || !slots_out || !m->std_q || !m->mean_q || !m->std_disp || !m->mean_disp || !m->offsets) return DP_ERR_INVALID;
    if (!dp_w16_supported(m)) return DP_ERR_UNSUPPORTED;
    std::memset(img, 0, sizeof(unsigned) * IMG_U32);
    for (int layer = 0; layer < NL; ++layer)
        for (int n = 0; n < NT_OUT[layer]; ++n)
            for (int kb = 0; kb < NKB[layer]; ++kb) {
                const int pair = PAIR0[layer] + n * NKB[layer] + kb;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        uint16_t t[3];
                        split3(entry(f, m, layer, 16 * n + (lane & 15), in_channel(kb, lane >> 4, j)), t);
                        for (int term = 0; term < N_TERMS; ++term)
                            img[((pair * N_TERMS + term) * 64 + lane) * 4 + (j >> 1)] |= (unsigned)t[term] << (16 * (j & 1));
                    }
            }
    for (int tile = 0; tile < 13; ++tile)
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                float v;
                if (tile < 3) { const int c = 16 * tile + 4 * g + r; v = c < 40 ? f->c0[c] : 0.f; }
                else if (tile < 7) { const int c = 16 * (tile - 3) + 4 * g + r; v = c < 60 ? f->b1[c] : 0.f; }
                else {
                    const int rho = 16 * (tile - 7) + 4 * g + r, d = dec_row(rho);
                    v = (float)(sd_of(m, rho) * (d >= 0 ? (double)f->b2[d] : 0.0) + mu_of(m, rho));
                }
                bias[(tile * 4 + g) * 4 + r] = v;
            }
    SlotConst* sc = (SlotConst*)slots_out;
    std::memset(sc, 0, sizeof(SlotConst) * NTY * 4);
    for (int t = 0; t < NTY; ++t)
        for (int g = 0; g < 4; ++g) {
            SlotConst& s = sc[t * 4 + g];
            s.item = slot_item(t, g);
            for (int r = 0; r < 4; ++r) { s.mu[r] = (float)mu_of(m, 16 * t + 4 * g + r); s.sd[r] = (float)sd_of(m, 16 * t + 4 * g + r); }
            if (s.item > 0 && s.item < dpl::NJ) // (the root's own offset is forced to zero: train.py:340)
                for (int k = 0; k < 3; ++k) s.off[k] = m->offsets[3 * s.item + k];
        }
    return DP_OK;
}

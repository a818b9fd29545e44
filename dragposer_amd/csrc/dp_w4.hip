// dp_w4.hip -- wave-private variant of the fused latent-optimisation kernel for gfx950 (MI355X).
//
// Same problem as dp_kernel.hip (reference: DragPose.run's while loop, python/src/drag_pose.py:296-355), other
// decomposition: ONE wavefront owns FOUR frames from the first decoder layer to the Adam step, so an iteration has no
// workgroup barrier and no cross-wave traffic at all.
//   * Decoder forward / backward: v_mfma_f32_4x4x1_16b_f32 with the A-block broadcast (dp_w4.h): a step is a rank-1
//     update of a 64-channel x 4-frame tile; activations stay in registers between layers (a 4x4 register <-> lane
//     transpose inside lane quads turns a product's result into the next product's operand).  The weights of L0, L1,
//     L2, bL1 (accumulator half of the register file), bL0 (vector half) and the first 3-5 groups of bL2 (what is left of
//     both) stay resident for the whole launch; the rest of bL2 is streamed from LDS (shared by the waves of a workgroup),
//     requested a phase ahead of its use.
//   * Kinematics: three stages per iteration, two wave-level LDS exchanges between them, all inside the wave.
//       J  lane 4b+i = the two items of quad b (dp_w4.h) of frame i, both in one packed (v_pk_*) instruction stream:
//          de-normalise, normalise, root-frame bone of the item's child (quaternion sandwich, no matrix);
//       T  lane 4u+i = tracker of rank u of frame i: root-frame position from the bones, position error, rotation error
//          as a quaternion product, their gradients as TORQUES (3-vectors in the tangent space of the rotations);
//       G  lanes as in J: subtree sums of the trackers' gradients, torque -> dL/dq = (0, 2 tau) (x) q -> dL/dy.
//     The formulation is the reference's loss (drag_pose.py:66-194) and the gradient autograd derives from it, restated
//     where that is cheaper and equal in real arithmetic: rotations compose as M(cur (x) q) =
//     M(cur) M(q), so the loop works in the frame of `cur_rot` with targets rotated once; |M - T|_F^2 = 8 (1 - <q_M,
//     q_T>^2) for rotations; and the gradient of a normalised quaternion lives in its tangent space, where it is the
//     torque of the loss.  Target rotations must therefore be rotation matrices (the reference builds them with
//     to_matrix from unit quaternions, eval_drag.py:199, run_drag.py:136).
//   * Adam: element-wise in the layout the last product leaves dL/dz in (lane = latent dim, register = frame).
// A workgroup is NW waves that share nothing but the streamed weight image; grid = ceil(B / (4 NW)).
#include "dp_device.h"
#include "dp_w4.h"
#include <utility>

using namespace dpw4;

template <class F, int... I> DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------------
// LDS map (floats)
constexpr int W4_R = 24; // tracker capacity per frame (>= NJ: every joint may carry one)
// one block per frame
constexpr int FB_QS = 0;                     // [32][4] unit quaternion by item id; 22: root displacement (x,y,z,-); 30: (1,0,0,0); 31: trash
constexpr int FB_BN = FB_QS + 128;           // [32][4] root-frame bone by child joint id; SLOT_ZERO: zero; 24..31: trash
constexpr int FB_GP = FB_BN + 128;           // [R][4]  tracker position gradient by rank
constexpr int FB_RT = FB_GP + 4 * W4_R;      // [R][4]  tracker torque on the root by rank
constexpr int FB_WT = FB_RT + 4 * W4_R;      // [32][4] own rotation torque by joint id (zero where untracked); 30: zero; 31: trash
constexpr int FB_LP = FB_WT + 128;           // [R][2]  tracker loss terms (last iteration)
constexpr int FB_TI = FB_LP + 2 * W4_R;      // [3][R][4] tracker inputs by rank: tp', cgp | qT' | k8, clp, clr8, joint
constexpr int FB_ZPRE = FB_TI + 12 * W4_R;   // [24] latent of the last forward pass (epilogue only)
constexpr int FB_ZT = FB_ZPRE + LAT;         // [24] z_tgt (epilogue only)
constexpr int FB_LT = FB_ZT + LAT;           // [24] early stop: (z - z_tgt)^2 per latent dim of the current latent
constexpr int FB_ES = FB_LT + LAT;           // [4]  early stop: losses of the frame's last executed iteration (pos, rot, tmp, -)
constexpr int FB_CUR = FB_ES + 4;            // [4]  cur_rot of the frame (for the epilogue)
// whole-sequence launches (SeqK): what the per-step state update reads from the epilogue's lanes, and the running state
constexpr int FB_GPOS = FB_CUR + 4;          // [4]  current global position of the sequence
constexpr int FB_SWD = FB_GPOS + 4;          // [4]  this step's world displacement
constexpr int FB_SD = FB_SWD + 4;            // [4]  this step's root-space displacement
constexpr int FB_SQW = FB_SD + 4;            // [4]  this step's world rotation
constexpr int FB_SPOS = FB_SQW + 4;          // [22][3] this step's joint positions (+ 2 pad)
constexpr int FB_END = FB_SPOS + 68;
constexpr int QS_DISP = ITEM_DISP, QS_IDENT = 30, QS_TRASH = 31, WT_ZERO = 30, WT_TRASH = 31;
// the four frames of a wave sit in the four lanes of every quad: block stride = 16 banks (mod 64) apart, so that the
// quad's 16-byte accesses to the same row of four blocks never share a bank
#ifndef W4_FB_RES
#define W4_FB_RES 16 // (diagnostic: other residues of the block stride mod 64 banks; multiples of 4)
#endif
constexpr int FB_STRIDE = ((FB_END - W4_FB_RES + 63) / 64) * 64 + W4_FB_RES;
static_assert(FB_STRIDE >= FB_END && FB_STRIDE % 64 == W4_FB_RES && W4_FB_RES % 4 == 0 && FB_LP % 4 == 0 && FB_TI % 4 == 0 && FB_ZPRE % 4 == 0 && FB_ES % 4 == 0, "frame block layout");

constexpr int GR_B2 = S_B2 / 4, NG_B2 = 26;     // the streamed product: first group, groups
constexpr int L_IMG2 = 0;                       // bL2 image [26][64][4]
constexpr int L_TAB = L_IMG2 + NG_B2 * 256;     // per-iteration Adam scalars [MAX_ITERS][2]: step, 1/sqrt(1-beta2^t)
constexpr int L_OC = L_TAB + 2 * MAX_ITERS;     // [16 quads][20] what the epilogue needs per quad: sd[4][2], mu[4][2], path words of both items
constexpr int L_ARGS = L_OC + 16 * 20;          // whole-sequence launches: the fields of the argument block a step reads (StepArgs)
constexpr int L_ARGS_WORDS = 80;
constexpr int L_FR = L_ARGS + L_ARGS_WORDS;     // frame blocks [NW * 4][FB_STRIDE]
// ... and behind them (so that nothing else moves: shifting the frame blocks by 32 words cost the headline 1 %) [NW waves][2 doubles]: Adam's
// running products beyond the argument table (adam_beyond; LONG instantiations only)
template <int NW> constexpr int lds_adx() { return (L_FR + NW * FPW * FB_STRIDE + 1) & ~1; }
template <int NW> constexpr int lds_total() { return lds_adx<NW>() + NW * 4; }

// ------------------------------------------------------------------------------------------------
// One group = four K-steps on two accumulators: step m multiplies x[m] (channel 4 ABID + m of the X-layout operand, block
// ABID broadcast to all 16 blocks) with the weight register w[m].  Written as ONE asm statement so that the operand
// classes are ours: weights the kernel keeps resident live in the ACCUMULATOR half of the register file ("a": an MFMA
// reads its B operand from either half; the kinematics arithmetic cannot use that half anyway), streamed weights and the
// accumulators in the vector half.  Hazards the compiler would pad for a builtin: a dependent accumulate (SrcC) needs 2
// wait states behind a 2-pass MFMA -> the s_nop between the pairs; the readers of the result: chain_end().
// W4_TAIL: what stands behind a group's last pair.  The next statement's first MFMA accumulates into acc0 again: M3(acc0) M4(acc1) | M1'(acc0)
// needs two wait states between M3 and M1' -- M4 and ONE more.  hipcc puts an `s_nop 0` of its own between any two asm statements, and
// that one is the second wait state: a nop of ours on top of it (rounds 2 and 3 had one) is a THIRD, and two back-to-back nops are not
// hidden behind the 8-cycle MFMA -- 4 cycles per group of four, 99 groups per iteration: dropping it measured -3.8 % kernel time
// (0.1614 -> 0.1553 ms, A/B inside one gpurun call, outputs bit-identical).  The compiler's nop is not ours to rely on blindly:
// tools/check_mfma_hazards.py walks the generated ISA and tests/test_build_quality.py fails if any dependent pair ends up closer
// than two wait states.  -DW4_TRAIL_NOP restores the belt-and-braces form.
#ifdef W4_TRAIL_NOP
#define W4_TAIL "\n\ts_nop 0"
#else
#define W4_TAIL ""
#endif
#define W4_GROUP_ASM(WC)                                                                                                  \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0 cbsz:4 abid:%10\n\t"                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, %1 cbsz:4 abid:%10\n\t"                                           \
                 "s_nop 0\n\t"                                                                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%10\n\t"                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%10" W4_TAIL                                                  \
                 : "+v"(acc0), "+v"(acc1)                                                                                 \
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), WC(w[0]), WC(w[1]), WC(w[2]), WC(w[3]), "i"(ABID))
// the same with CBSZ = 3 (two K-steps per instruction, dp_w4.h), weights in vector registers, starting from zero or not
#define W4_GROUP3_ASM(C0, C1, OUT)                                                                                        \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, " C0 " cbsz:3 abid:%10\n\t"                                       \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, " C1 " cbsz:3 abid:%10\n\t"                                       \
                 "s_nop 0\n\t"                                                                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:3 abid:%10\n\t"                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:3 abid:%10" W4_TAIL                                                  \
                 : OUT(acc0), OUT(acc1)                                                                                   \
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "i"(ABID))
template <int ABID> DEV void group3_v(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_GROUP3_ASM("%0", "%1", "+v"); }
template <int ABID> DEV void first3_v_zero(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_GROUP3_ASM("0", "0", "=&v"); }
template <int ABID> DEV void group_a(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_GROUP_ASM("a"); }
template <int ABID> DEV void group_v(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_GROUP_ASM("v"); }
// first group of a chain: the accumulators START here -- acc1 (and acc0, when the product has no bias row) take the inline
// constant 0 as their C operand instead of being cleared by eight v_mov first
#define W4_FIRST_ASM(WC, C0, OUT0)                                                                                        \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, " C0 " cbsz:4 abid:%10\n\t"                                       \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, 0 cbsz:4 abid:%10\n\t"                                            \
                 "s_nop 0\n\t"                                                                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%10\n\t"                                           \
                 "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%10" W4_TAIL                                                  \
                 : OUT0(acc0), "=&v"(acc1)                                                                                \
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), WC(w[0]), WC(w[1]), WC(w[2]), WC(w[3]), "i"(ABID))
// ... or a bias row: C = a register tuple that holds it for the whole launch, D = the accumulator (no copy per iteration)
template <int ABID> DEV void first_a_biased(f4& acc0, f4& acc1, const f4& x, const f4& w, const f4& c)
{
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %11 cbsz:4 abid:%10\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, 0 cbsz:4 abid:%10\n\t"
                 "s_nop 0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%10\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%10" W4_TAIL
                 : "=&v"(acc0), "=&v"(acc1)
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "a"(w[0]), "a"(w[1]), "a"(w[2]), "a"(w[3]), "i"(ABID), "v"(c));
}
template <int ABID> DEV void first_a_zero(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_FIRST_ASM("a", "0", "=&v"); }
template <int ABID> DEV void first_v_zero(f4& acc0, f4& acc1, const f4& x, const f4& w) { W4_FIRST_ASM("v", "0", "=&v"); }
// a VALU result (transpose, kinematics) feeding the first MFMA of a chain / the chain's result feeding the VALU
// (needed: VALU -> MFMA operand 2 wait states, MFMA -> MFMA operand 4 -- what hipcc itself pads builtins with.  Behind a transpose
//  nothing is missing: its own `s_nop 3` and the `s_nop 0` hipcc puts between two asm statements make 5.  Behind VALU code this
//  statement's one wait state and that same compiler nop make 2.  tools/check_mfma_hazards.py holds the generated ISA to all of it.)
DEV void chain_begin() { asm volatile("s_nop 0"); }
// chain_end: MFMA -> VALU read needs 4 wait states; hipcc adds one `s_nop 0` of its own behind an asm statement whose outputs the next
// instructions read (seen in every build; tools/check_mfma_hazards.py fails the build-quality test if it ever does not), so three here.
DEV void chain_end(f4& acc0, f4& acc1) { asm volatile("s_nop 2" : "+v"(acc0), "+v"(acc1)); }

// NG groups from resident weights (accumulator registers) / from weights in vector registers
// START: 0 continues a chain; 1 starts one from the bias row `bias`; 2 starts one from zero
// SKIP: bit (ABID0 + g) set = group g is left out (bL2: the K-steps of an item whose dL/dy is zero in all four frames of the wave, below)
template <int NG, int ABID0, int START = 0, unsigned SKIP = 0u> DEV void chain_a(f4& acc0, f4& acc1, const f4& x, const f4* wv, const f4& bias = f4{0.f, 0.f, 0.f, 0.f})
{
    static_assert(START == 0 || !(SKIP & (1u << ABID0)), "the group that starts a chain is not skipped");
    static_for<NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        if constexpr ((SKIP >> (ABID0 + g)) & 1u) {}
        else if constexpr (g == 0 && START == 1) first_a_biased<ABID0>(acc0, acc1, x, wv[0], bias);
        else if constexpr (g == 0 && START == 2) first_a_zero<ABID0>(acc0, acc1, x, wv[0]);
        else group_a<ABID0 + g>(acc0, acc1, x, wv[g]);
    });
}
template <int NG, int ABID0, int START = 0, unsigned SKIP = 0u> DEV void chain_v(f4& acc0, f4& acc1, const f4& x, const f4* wv)
{
    static_assert(START == 0 || !(SKIP & (1u << ABID0)), "the group that starts a chain is not skipped");
    static_for<NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        if constexpr ((SKIP >> (ABID0 + g)) & 1u) {}
        else if constexpr (g == 0 && START == 2) first_v_zero<ABID0>(acc0, acc1, x, wv[0]);
        else group_v<ABID0 + g>(acc0, acc1, x, wv[g]);
    });
}
template <int NG> DEV void chain3_v_zero(f4& acc0, f4& acc1, const f4& x, const f4* wv)
{ // NG groups of 2 x 4 K-steps from zero: quads 0..NG-1 against lanes 0..31, quads 8..8+NG-1 against lanes 32..63
    static_for<NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        if constexpr (g == 0) first3_v_zero<0>(acc0, acc1, x, wv[0]);
        else group3_v<g>(acc0, acc1, x, wv[g]);
    });
}
DEV f4 add_halves(f4 v)
{ // lanes l and l ^ 32 both get v[l] + v[l ^ 32] (one v_permlane32_swap + one add per register)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[r]), false, false);
        v[r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    return v;
}
// ---- Adam's state in HALF the registers (round 6; the fixed-count kernel).  The latent lives in layout D -- lane = latent dim (24 of 64 lanes),
// register = frame -- so z, m, v, z_tgt and the gradient are four registers each of which 40 lanes idle, and every step of the update (13 packed
// instructions and eight 16-cycle sqrt / rcp) is issued twice.  bL0 leaves the gradient as two half sums in lanes l and l ^ 32 (above): instead of
// giving BOTH halves the sum of all four frames (add_halves: four swaps), one v_permlane32_swap per frame PAIR gives lanes 0..31 the sums of frames
// 0 | 1 and lanes 32..63 those of frames 2 | 3 -- "packed": register p, lane l = frame p + 2 (l >> 5), dim l & 31.  The whole update runs on two
// registers per quantity (half the instructions, same arithmetic per element: outputs bit-identical), and the new latent goes back to layout D for
// the next L0 with two swaps.
DEV f2 pack_frames(const f4& v)
{ // lanes 0..31 keep frames 0 | 1 of their dim, lanes 32..63 take frames 2 | 3 of dim l - 32
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
    return f2{__uint_as_float(a[0]), __uint_as_float(b[0])};
}
DEV f4 unpack_frames(const f2& p, const f4& old)
{ // layout D again, valid in lanes 0..31 -- all a product reads (K-rows 0..23); the upper halves hold finite leftovers.  Frames 0 | 1 ARE the packed
  // registers' lower halves; frames 2 | 3 come down with one swap each, into the registers of the previous iteration's copy (`old`: dead, finite)
  // -- the swap's other operand is a throw-away copy of the packed register, so it costs one move, not two
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[0]), __float_as_uint(old[2]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[1]), __float_as_uint(old[3]), false, false);
    return f4{p[0], p[1], __uint_as_float(a[1]), __uint_as_float(b[1])};
}
DEV f2 add_halves_packed(const f4& v)
{ // the two half sums of bL0 joined per frame pair: lanes 0..31 = frames 0 | 1, lanes 32..63 = frames 2 | 3 (lo + hi, as add_halves adds them)
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
    return f2{__uint_as_float(a[0]), __uint_as_float(b[0])} + f2{__uint_as_float(a[1]), __uint_as_float(b[1])};
}
#ifndef W4_PACKED_ADAM
#define W4_PACKED_ADAM 1
#endif
template <int NG, int ABID0 = 0, unsigned SKIP = 0u> DEV void load_w(f4* wv, const f4* w)
{ // (SKIP as in chain_a / chain_v: the weights of a group that is left out are not read)
    static_for<NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        if constexpr (!((SKIP >> (ABID0 + g)) & 1u)) wv[g] = w[g * 64];
    });
}

// an empty asm that "uses" NG weight groups (accumulator / vector registers): pins where the wait for their loads stands
DEV void touch_a1(f4& w) { asm volatile("" : "+a"(w)); }
DEV void touch_v1(f4& w) { asm volatile("" : "+v"(w)); }
template <int NG> DEV void touch_a(f4* wv) { static_for<NG>([&](auto gi) { touch_a1(wv[decltype(gi)::value]); }); }
template <int NG> DEV void touch_v(f4* wv) { static_for<NG>([&](auto gi) { touch_v1(wv[decltype(gi)::value]); }); }

// D <-> X: transpose of (register index, lane-in-quad), on the matrix pipe: D_b[i][j] = sum_r x_r[i] * e_r[j] with
// e_r[j] = (j == r) -- products with 1, sums with 0: exact.  Four dependent 2-pass MFMAs; the result may feed an MFMA
// (4 wait states) or the VALU (4).  (The VALU form -- two exchange stages of v_cndmask_b32_dpp -- is in
// tools/ubench/w4_probe.hip; it measured 1.3 % slower in the loop and keeps the VALU busy between two products.)
// Head and tail of a transpose: its input comes from VALU code (2 wait states), its result feeds an MFMA chain or the VALU (4).  In every
// build hipcc puts an `s_nop 0` in front of and behind these statements (rounds 2 and 3 added `s_nop 1` / `s_nop 3` of their own on top:
// 3 and 5 wait states); one fewer of ours each, the generated ISA held to the requirement by tools/check_mfma_hazards.py.
#ifdef W4_TRAIL_NOP
#define W4_QT_HEAD "s_nop 1\n\t"
#define W4_QT_TAIL "s_nop 3"
#else
#define W4_QT_HEAD "s_nop 0\n\t"
#define W4_QT_TAIL "s_nop 2"
#endif
DEV void quad_transpose_mfma(f4& r, const f4& e)
{
    f4 d;
    asm volatile(W4_QT_HEAD
                 "v_mfma_f32_4x4x1_16b_f32 %0, %1, %5, 0\n\t"
                 "s_nop 1\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0\n\t"
                 "s_nop 1\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %3, %7, %0\n\t"
                 "s_nop 1\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0\n\t"
                 W4_QT_TAIL
                 : "=&v"(d)
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
    r = d;
}
#ifdef W4_ABLATE_QT
#define QT(x) asm volatile("s_nop 1" : "+v"(x))
#else
#define QT(x) quad_transpose_mfma(x, eT)
#endif
// two independent transposes, their dependent chains interleaved (a dependent 2-pass MFMA needs 2 wait states: the other
// chain's MFMA and one s_nop)
DEV void quad_transpose_mfma2(f4& r0, f4& r1, const f4& e)
{
    f4 d0, d1;
    asm volatile(W4_QT_HEAD
                 "v_mfma_f32_4x4x1_16b_f32 %0, %2, %10, 0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %6, %10, 0\n\t"
                 "s_nop 0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %3, %11, %0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %7, %11, %1\n\t"
                 "s_nop 0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %4, %12, %0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %8, %12, %1\n\t"
                 "s_nop 0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %0, %5, %13, %0\n\t"
                 "v_mfma_f32_4x4x1_16b_f32 %1, %9, %13, %1\n\t"
                 W4_QT_TAIL
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]), "v"(e[0]), "v"(e[1]),
                   "v"(e[2]), "v"(e[3]));
    r0 = d0;
    r1 = d1;
}

DEV f4 splat(float v) { return f4{v, v, v, v}; }
// LeakyReLU(0.2) as a per-element factor (1 or 0.2): forward a = x * f, backward d = g * f -- the factor is what the
// backward needs, and both multiplications are packed
// x > 0 ? 1 : 0.2 as med3(x * 2^127, 0.2, 1): any normal x > 0 scales to >= 2, any x <= 0 to <= 0 -- two packed multiplies and
// four v_med3_f32 for a register quad instead of four compares and four selects.  (A positive DENORMAL pre-activation, below
// 1.2e-38, would get a slope between 0.2 and 1 instead of 1; the activation it scales is below 1.2e-38 either way.)
DEV f4 lrelu_factor(f4 x)
{
    const f4 t = x * 0x1p127f;
    return f4{__builtin_amdgcn_fmed3f(t.x, 0.2f, 1.f), __builtin_amdgcn_fmed3f(t.y, 0.2f, 1.f), __builtin_amdgcn_fmed3f(t.z, 0.2f, 1.f),
              __builtin_amdgcn_fmed3f(t.w, 0.2f, 1.f)};
}
DEV f2 splat2(float v) { return f2{v, v}; }

// ------------------------------------------------------------------------------------------------
// Kinematics.  Quaternions are (w, v) = (w, x, y, z), Hamilton; R(q) a = a + 2 (w (v x a) + v x (v x a)) is the rotation
// to_matrix_4 (utils.py:49-74) encodes for unit q.
struct PairC { // loop-invariant constants of my quad's two items, side A | side B packed (registers)
    f2 off[3], sgn, rho, sel[6];
    unsigned subA, subB; // tracker subsets of the two items (general path: more than 6 trackers in a frame)
    int qsA, qsB;        // float index of my items' quaternion slots in the frame block
    int bnA, bnB;        //                          child-bone slots
    int wtA, wtB;        //                          own-torque slots
    int tab;             // FB_RT on the root's quad, FB_GP elsewhere: the table my items sum over the trackers
    int kindB;           // KIND_* of side B (side A is always a joint)
    int itemA, itemB;    // item ids (-1: idle)
};

struct TRec { // a tracker as its T-stage lane sees it
    bool act;
    int qs, wt, rank;  // float index of the tracked joint's quaternion slot / own-torque slot; rank
    unsigned plo, phi; // bone slots on the path root -> joint (dp_layout.h: 7 x 5 bits)
    V3 tp;             // target position in the frame of cur_rot
    f2 qT0, qT1;       // target rotation in the frame of cur_rot: (w, x), (y, z) -- register pairs for the packed products
    float cgp, clp, k8, clr8; // 2 w_pos / (3E), w_pos / (3E), -8 lam w_rot / (9E), 8 lam w_rot / (9E)
};

DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV V3 rot_conj(Q4 q, V3 a)
{ // R(conj q) a
    const V3 v = {q.x, q.y, q.z};
    const V3 t = cross(v, a), c = cross(v, t);
    return {a.x + 2.f * (c.x - q.w * t.x), a.y + 2.f * (c.y - q.w * t.y), a.z + 2.f * (c.z - q.w * t.z)};
}

// Pointers that come out of the argument block re-read per step (whole-sequence launches) have lost their address space: the compiler
// would use FLAT loads and stores, which count against BOTH memory counters -- every later wait for an LDS read then also waits for
// the results' stores.  Every pointer of the argument block is a device-memory pointer: say so.
typedef float __attribute__((address_space(1))) gfloat;
typedef int __attribute__((address_space(1))) gint;
DEV gfloat* GM(float* p) { return (gfloat*)p; }
DEV const gfloat* GM(const float* p) { return (const gfloat*)p; }
DEV gint* GM(int* p) { return (gint*)p; }
DEV Q4 quat_from_rotmat(const float* m)
{ // row-major 3x3 rotation -> unit quaternion (once per tracker, before the loop).  Shepperd's four branches pick the
  // largest of (w, x, y, z) to divide by; since the result is normalised anyway, each branch is just four sums scaled by a
  // common factor -- the candidates are built without a division or a square root and ONE is selected (the branch
  // conditions of the textbook form), then normalised.
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float tr = m00 + m11 + m22;
    const float a = m21 - m12, b = m02 - m20, c = m10 - m01, d = m01 + m10, e = m02 + m20, f = m12 + m21;
    const bool s0 = tr > 0.f, s1 = m00 > m11 && m00 > m22, s2 = m11 > m22;
    Q4 q;
    q.w = s0 ? 1.f + tr : (s1 ? a : (s2 ? b : c));
    q.x = s0 ? a : (s1 ? 1.f + m00 - m11 - m22 : (s2 ? d : e));
    q.y = s0 ? b : (s1 ? d : (s2 ? 1.f + m11 - m00 - m22 : f));
    q.z = s0 ? c : (s1 ? e : (s2 ? f : 1.f + m22 - m00 - m11));
    const float n = 1.f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return {q.w * n, q.x * n, q.y * n, q.z * n};
}

// Whole-sequence launches: what a step reads from the argument block.  Its fields held in scalar registers across the iteration loop
// would be some 140 of them, spilled; re-read from the kernarg segment per step they come one dependent uncached load at a time (the
// compiler cannot batch loads it must assume the results' stores may alias: a dozen round trips per step, tools/seq_step_stamps.sh).
// So the set-up copies them into LDS once and a step reads them from there.
struct StepArgs {
    const dpl::ItemConst* items;
    const float* w4img;
    const float *z_tgt, *tgt_pos, *tgt_rot, *w;
    float *z, *z_pre, *pose, *disp, *world_disp, *world_rot, *pos, *rot, *loss;
    int *iters, *status;
    int n_iter;
    float lam_rot, lam_tmp;
    SeqK seq;
};
static_assert(sizeof(StepArgs) <= L_ARGS_WORDS * 4 && alignof(StepArgs) <= 16, "StepArgs fits its LDS slot");
DEV void stage_step_args(float* lds, const KArgs& a)
{
    StepArgs* d = (StepArgs*)(lds + L_ARGS);
    d->items = a.items; d->w4img = a.w4img;
    d->z_tgt = a.z_tgt; d->tgt_pos = a.tgt_pos; d->tgt_rot = a.tgt_rot; d->w = a.w;
    d->z = a.z; d->z_pre = a.z_pre; d->pose = a.pose; d->disp = a.disp; d->world_disp = a.world_disp; d->world_rot = a.world_rot;
    d->pos = a.pos; d->rot = a.rot; d->loss = a.loss;
    d->iters = a.iters; d->status = a.status;
    d->n_iter = a.n_iter;
    d->lam_rot = a.lam_rot; d->lam_tmp = a.lam_tmp;
    d->seq = a.seq;
}
template <bool SEQ> DEV decltype(auto) step_args_of(const KArgs& a, const float* lds)
{
    if constexpr (SEQ) {
        int o = L_ARGS;
        asm volatile("" : "+v"(o)); // opaque per use: the reads stay where a step needs them
        return *(const StepArgs*)(lds + o);
    } else {
        return (a);
    }
}

// tracker of rank `rank` of the frame whose tracked-joint mask is `tmask` (E of them), in two halves so that the setup can
// have the inputs in flight while other loads are issued: tracker_fetch issues the global loads, tracker_finish rotates the
// targets into the frame of `cur`, stores the record in the frame block (general path, epilogue) and returns it
// position of the (rank + 1)-th set bit of m (31 if there is none): the largest p with popcount(m below p) <= rank
DEV int nth_set_bit(unsigned m, int rank)
{
    int pos = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1) {
        const int c = __popc(m & ((1u << (pos + step)) - 1u));
        pos += c <= rank ? step : 0;
    }
    return pos;
}
struct TRaw {
    bool act;
    int rank, j;
    unsigned plo, phi;
    float p[3], wp, wr, m[9];
};
template <class A> DEV TRaw tracker_fetch(const A& a, bool optimise, int gf, unsigned tmask, int E, int rank, int gf_tgt = -1)
{ // (!optimise: no tracker arrays -- E = 0, every lane inactive; the loads read the weight image instead and are ignored)
  // gf_tgt: row of the targets when it differs from the row of the weights (whole-sequence launches: step * S + sequence)
    TRaw r;
    r.act = rank < E;
    r.rank = rank;
    const int j = r.act ? nth_set_bit(tmask, rank) : 0;
    r.j = j;
    {
        const auto* it = (const dpl::ItemConst __attribute__((address_space(1)))*)a.items + j;
        r.plo = it->path_lo;
        r.phi = it->path_hi;
    }
    const int row = optimise ? gf * NJ + j : 0; // (inactive lanes read joint 0's inputs and ignore them)
    const size_t rowt = optimise ? (size_t)(gf_tgt >= 0 ? gf_tgt : gf) * NJ + j : 0;
    const gfloat* p = GM(optimise ? a.tgt_pos : a.w4img) + rowt * 3;
    const gfloat* rm = GM(optimise ? a.tgt_rot : a.w4img) + rowt * 9;
    const gfloat* wv = GM(optimise ? a.w : a.w4img) + (size_t)row * 2;
    r.p[0] = p[0]; r.p[1] = p[1]; r.p[2] = p[2];
    r.wp = wv[0];
    r.wr = wv[1];
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = rm[k];
    return r;
}
template <class A> DEV TRec tracker_finish(const A& a, float* fb, const TRaw& r, int E, Q4 cur, V3 shift = V3{0.f, 0.f, 0.f})
{ // shift: added to the position target (whole-sequence launches: tgt_root[t] - current global position)
    TRec t;
    t.act = r.act;
    t.rank = r.rank;
    const int j = r.j;
    t.qs = FB_QS + 4 * (j == 0 ? QS_IDENT : j); // the root is the identity in its own frame
    t.wt = FB_WT + 4 * (j == 0 ? WT_TRASH : j); // ... and has no torque of its own: everything goes to the root sum
    t.plo = r.plo;
    t.phi = r.phi;
    t.tp = {0.f, 0.f, 0.f};
    t.qT0 = f2{1.f, 0.f}; t.qT1 = f2{0.f, 0.f};
    t.cgp = t.clp = t.k8 = t.clr8 = 0.f;
    if (t.act) {
        const float invE = 1.f / (float)E;
        t.tp = rot_conj(cur, V3{r.p[0] + shift.x, r.p[1] + shift.y, r.p[2] + shift.z});
        const Q4 qT = quat_mul(Q4{cur.w, -cur.x, -cur.y, -cur.z}, quat_from_rotmat(r.m));
        t.qT0 = f2{qT.w, qT.x}; t.qT1 = f2{qT.y, qT.z};
        t.clp = r.wp * invE * (1.f / 3.f);                       // loss_pos coefficient  w_pos / (3E)
        const float clr = a.lam_rot * r.wr * invE * (1.f / 9.f); // loss_rot coefficient  lam w_rot / (9E)
        t.cgp = 2.f * t.clp;
        t.k8 = -8.f * clr;
        t.clr8 = 8.f * clr;
        float* ti = fb + FB_TI + r.rank * 4;
        *(f4*)(ti) = f4{t.tp.x, t.tp.y, t.tp.z, t.cgp};
        *(f4*)(ti + 4 * W4_R) = f4{qT.w, qT.x, qT.y, qT.z};
        *(f4*)(ti + 8 * W4_R) = f4{t.k8, t.clp, t.clr8, __int_as_float(j)};
    }
    return t;
}
// ---- input screening (include/dragposer.h: DP_STATUS_*).  The four frames of a wave meet in the D <-> X transposes, which are matrix products
// with unit rows: a NaN or Inf in one frame times the 0 of another frame's row is NaN -- one tracker drop-out would take the three neighbouring
// frames (three other sequences, in a whole-sequence launch) with it.  The reference has no such coupling (it runs one frame at a time), so the
// kernel screens its inputs once per launch (per step): a frame with a non-finite or absurd (> DP_INPUT_LIMIT) input is computed on neutral
// values -- weights 0, z_tgt = z, so that its gradient is exactly zero and nothing in it ever leaves the finite range -- and its RESULTS are set
// to what the reference returns for it (NaN; drag_pose.py:300-304,342-344), with the reason in dp_result.status.
DEV bool out_of_range(float x) { return !(fabsf(x) <= DP_INPUT_LIMIT); } // NaN, Inf, or beyond the limit
DEV bool raw_bad(const TRaw& r)
{
    bool bad = out_of_range(r.p[0]) || out_of_range(r.p[1]) || out_of_range(r.p[2]) || out_of_range(r.wp) || out_of_range(r.wr);
#pragma unroll
    for (int k = 0; k < 9; ++k) bad = bad || out_of_range(r.m[k]);
    return r.act && bad;
}
DEV void raw_neutral(TRaw& r)
{
    r.p[0] = r.p[1] = r.p[2] = 0.f;
    r.wp = r.wr = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = (k % 4 == 0) ? 1.f : 0.f;
}
DEV unsigned frames_of(unsigned long long votes)
{ // bit i: some lane with (lane & 3) == i voted -- the lanes of frame i
    votes |= votes >> 32; votes |= votes >> 16; votes |= votes >> 8; votes |= votes >> 4;
    return (unsigned)votes & 0xFu;
}
DEV float poisoned(bool p, float v) { return p ? __builtin_nanf("") : v; }
template <class A> DEV TRec make_tracker(const A& a, float* fb, int gf, unsigned tmask, int E, int rank, Q4 cur, int gf_tgt = -1, V3 shift = V3{0.f, 0.f, 0.f}, bool neutral = false)
{
    TRaw r = tracker_fetch(a, true, gf, tmask, E, rank, gf_tgt);
    if (neutral) raw_neutral(r);
    return tracker_finish(a, fb, r, E, cur, shift);
}

DEV TRec load_tracker(const KArgs& a, const float* fb, int E, int rank)
{ // the same from the frame block (ranks beyond the first 16 of a frame: rare)
    TRec t;
    t.act = rank < E;
    t.rank = rank;
    const float* ti = fb + FB_TI + (t.act ? rank : 0) * 4;
    const f4 i0 = *(const f4*)(ti), i1 = *(const f4*)(ti + 4 * W4_R), i2 = *(const f4*)(ti + 8 * W4_R);
    const int j = t.act ? __float_as_int(i2.w) : 0;
    t.qs = FB_QS + 4 * (j == 0 ? QS_IDENT : j);
    t.wt = FB_WT + 4 * (j == 0 ? WT_TRASH : j);
    t.plo = a.items[j].path_lo;
    t.phi = a.items[j].path_hi;
    t.tp = {i0.x, i0.y, i0.z};
    t.cgp = i0.w;
    t.qT0 = f2{i1.x, i1.y}; t.qT1 = f2{i1.z, i1.w};
    t.k8 = i2.x; t.clp = i2.y; t.clr8 = i2.z;
    return t;
}

// four / three separate registers -> consecutive LDS words.  (Plain stores get merged into ds_write_b128 / b96, whose data
// operand is a register tuple: the packed arithmetic leaves every value in a pair with the OTHER item's, so each such
// store costs four v_mov.  ds_write2_b32 takes two unrelated registers.)  LDS operations of a wave execute in order.
DEV void lds_store4(float* p, float a, float b, float c, float d)
{
    const unsigned ad = (unsigned)(size_t)p;
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write2_b32 %0, %3, %4 offset0:2 offset1:3" : : "v"(ad), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
DEV void lds_store3(float* p, float a, float b, float c)
{
    const unsigned ad = (unsigned)(size_t)p;
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write_b32 %0, %3 offset:8" : : "v"(ad), "v"(a), "v"(b), "v"(c) : "memory");
}

// ---- stage J: both items of my quad, packed
struct JOut { f2 q[4], u[3], inv; };
DEV void j_stage(const PairC& c, float* fb, const f4 y01, const f4 y23, JOut& o)
{ // y01 / y23: the transposed blocks of layer 2 -- channels (0, 1) / (2, 3), each as a side A | side B register pair (dp_w4.h)
    const f2 rq[4] = {f2{y01[0], y01[1]}, f2{y01[2], y01[3]}, f2{y23[0], y23[1]}, f2{y23[2], y23[3]}}; // (de-normalised by layer 2 itself)
    const f2 nn = rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3];
    const float invA = __builtin_amdgcn_rsqf(nn.x);
    const float invB = c.kindB == KIND_DISP ? 1.f : (c.kindB == KIND_IDLE ? 0.f : __builtin_amdgcn_rsqf(nn.y));
    o.inv = f2{invA, invB};
#pragma unroll
    for (int k = 0; k < 4; ++k) o.q[k] = rq[k] * o.inv; // (the displacement item passes its de-normalised channels through)
    // child bone u = R(q) off = off + 2 (w t + v x t), t = v x off
    const f2 w = o.q[0], vx = o.q[1], vy = o.q[2], vz = o.q[3];
    const f2 tx = vy * c.off[2] - vz * c.off[1], ty = vz * c.off[0] - vx * c.off[2], tz = vx * c.off[1] - vy * c.off[0];
    const f2 cx = vy * tz - vz * ty, cy = vz * tx - vx * tz, cz = vx * ty - vy * tx;
    o.u[0] = c.off[0] + 2.f * (w * tx + cx);
    o.u[1] = c.off[1] + 2.f * (w * ty + cy);
    o.u[2] = c.off[2] + 2.f * (w * tz + cz);
    lds_store4(fb + c.qsA, o.q[0].x, o.q[1].x, o.q[2].x, o.q[3].x);
    lds_store4(fb + c.qsB, o.q[0].y, o.q[1].y, o.q[2].y, o.q[3].y);
    lds_store3(fb + c.bnA, o.u[0].x, o.u[1].x, o.u[2].x);
    lds_store3(fb + c.bnB, o.u[0].y, o.u[1].y, o.u[2].y);
}

// Quaternion products on register pairs (w, x), (y, z): eight packed instructions each, the operand swaps and signs of the
// Hamilton product (and of the conjugate) expressed as op_sel / neg modifiers -- the compiler builds them with moves.
//   (w, x) = aw (bw, bx) + ax (-bx, bw) + ay (-by, bz) + az (-bz, -by)
//   (y, z) = aw (by, bz) + ax (-bz, by) + ay (bw, -bx) + az (bx, bw)
// The ax term is the rounded product and the aw term the first fused one, then ay, az: the order the compiler gives
// quat_mul (dp_device.h), so that the packed form is bit-identical to the scalar one.
DEV void quat_mul_conj_a(f2 a0, f2 a1, f2 b0, f2 b1, f2& o0, f2& o1)
{ // conj(a) (x) b
    asm("v_pk_mul_f32 %0, %2, %4 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %2, %5, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %3, %5, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %3, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
        : "=&v"(o0), "=&v"(o1)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
DEV void quat_mul_conj_b(f2 a0, f2 a1, f2 b0, f2 b1, f2& o0, f2& o1)
{ // a (x) conj(b)
    asm("v_pk_mul_f32 %0, %2, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %2, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %5, %1 op_sel_hi:[0,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %3, %5, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %3, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(o0), "=&v"(o1)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}

// ---- stage T: one tracker per lane
DEV void t_stage(const TRec& t, float* fb, bool losses)
{
    if (!t.act) return;
    const f4 q0v = *(const f4*)(fb + FB_QS), qtv = *(const f4*)(fb + t.qs), dv = *(const f4*)(fb + FB_QS + 4 * QS_DISP);
    f2 pxy = f2{dv.x, dv.y}; // root-frame position: displacement + the bones on the path (short paths end in the zero slot)
    float pz = dv.z;
    {
        const unsigned plo = t.plo, phi = t.phi;
        f4 bn[MAX_PATH]; // all reads in flight together: one LDS latency, not seven
#pragma unroll
        for (int k = 0; k < MAX_PATH; ++k) bn[k] = *(const f4*)(fb + FB_BN + 4 * ((k < 6) ? ((plo >> (5 * k)) & 31u) : (phi & 31u)));
#pragma unroll
        for (int k = 0; k < MAX_PATH; ++k) { pxy += f2{bn[k].x, bn[k].y}; pz += bn[k].z; }
    }
    const Q4 q0 = {q0v.x, q0v.y, q0v.z, q0v.w};
    const V3 at = rot_conj(q0, t.tp); // target position in the root frame
    // (x, y) components in packed registers, z beside them: the same operations, fewer instructions
    const f2 exy = pxy - f2{at.x, at.y};
    const float ez = pz - at.z;
    const f2 gpxy = t.cgp * exy;
    const float gpz = t.cgp * ez;
    const V3 e = {exy.x, exy.y, ez}, gp = {gpxy.x, gpxy.y, gpz};
    // rotation error  s = conj(q0) (x) qT (x) conj(qt):  scalar part c = <q0 (x) qt, qT>,  |M - T|_F^2 = 8 |vec s|^2
    f2 r0, r1, s0, s1;
    quat_mul_conj_a(f2{q0v.x, q0v.y}, f2{q0v.z, q0v.w}, t.qT0, t.qT1, r0, r1);
    quat_mul_conj_b(r0, r1, f2{qtv.x, qtv.y}, f2{qtv.z, qtv.w}, s0, s1);
    const Q4 s = {s0.x, s0.y, s1.x, s1.y};
    const float k = t.k8 * s.w;
    const float ownx = k * s.x;
    const f2 ownyz = k * s1; // torque on the tracked joint (and on the root)
    const V3 own = {ownx, ownyz.x, ownyz.y};
    const V3 ag = cross(at, gp);
    const f2 rtyz = f2{ag.y, ag.z} + ownyz;
    const V3 rt = {ag.x + own.x, rtyz.x, rtyz.y};
    // (12-byte stores: the readers take 16 bytes and ignore the fourth word)
    typedef float f3 __attribute__((ext_vector_type(3)));
    *(f3*)(fb + FB_GP + 4 * t.rank) = f3{gp.x, gp.y, gp.z};
    *(f3*)(fb + FB_RT + 4 * t.rank) = f3{rt.x, rt.y, rt.z};
    *(f3*)(fb + t.wt) = f3{own.x, own.y, own.z};
    if (losses) // (uniform) read by the epilogue
        *(f2*)(fb + FB_LP + 2 * t.rank) = f2{t.clp * (e.x * e.x + e.y * e.y + e.z * e.z), t.clr8 * (s.x * s.x + s.y * s.y + s.z * s.z)};
}

// ---- stage G: both items of my quad, packed -> their quads of dL/dy
DEV void g_stage(const PairC& c, const float* fb, const JOut& j, unsigned tmask, int Emax, f4& gyA, f4& gyB)
{
    f2 S[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};
    {
        const float* tab = fb + c.tab;
        f4 g[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) g[u] = *(const f4*)(tab + 4 * u);
#pragma unroll
        for (int u = 0; u < 6; ++u) asm volatile("" : "+v"(g[u])); // (keeps the reads 16 bytes wide: the fourth word is unused, and hipcc would
                                                                    //  narrow them to ds_read_b96 -- 8 LDS cycles in 8-lane groups against ds_read_b128's 4)
#pragma unroll
        for (int u = 0; u < 6; ++u) { S[0] += c.sel[u] * splat2(g[u].x); S[1] += c.sel[u] * splat2(g[u].y); S[2] += c.sel[u] * splat2(g[u].z); }
        if (Emax > 6) { // more than 6 trackers in a frame of this wave (uniform, rare): general path
            unsigned m = tmask;
#pragma unroll
            for (int u = 0; u < 6; ++u) m &= m - 1u;
            for (int e0 = 6; e0 < Emax; ++e0) {
                const f4 ge = *(const f4*)(tab + 4 * e0);
                const int t = __builtin_ctz(m | 0x80000000u); // joint id of this rank (31 when exhausted)
                m &= m - 1u;
                const f2 sl = f2{(float)((c.subA >> t) & 1u), (float)((c.subB >> t) & 1u)};
                S[0] += sl * splat2(ge.x); S[1] += sl * splat2(ge.y); S[2] += sl * splat2(ge.z);
            }
        }
    }
    f4 wa = *(const f4*)(fb + c.wtA), wb = *(const f4*)(fb + c.wtB);
    asm volatile("" : "+v"(wa), "+v"(wb)); // (16-byte reads, as above)
    // torque: bone x S (+ own rotation torque); on the root the sum over the trackers' root torques itself
    f2 t0 = j.u[1] * S[2] - j.u[2] * S[1] + c.rho * S[0];
    f2 t1 = j.u[2] * S[0] - j.u[0] * S[2] + c.rho * S[1];
    f2 t2 = j.u[0] * S[1] - j.u[1] * S[0] + c.rho * S[2];
    // (scalar adds on the halves: pairing wa / wb components for three packed adds costs four moves)
    t0.x += wa.x; t0.y += wb.x; t1.x += wa.y; t1.y += wb.y; t2.x += wa.z; t2.y += wb.z;
    // dL/dq = (0, a) (x) q = (-a.v, w a + a x v), a = 2 tau   (root: q (x) (0, a): the cross product changes sign)
    const f2 a0 = t0 + t0, a1 = t1 + t1, a2 = t2 + t2;
    const f2 w = j.q[0], vx = j.q[1], vy = j.q[2], vz = j.q[3];
    const f2 g0 = -(a0 * vx + a1 * vy + a2 * vz);
    const f2 g1 = w * a0 + c.sgn * (a1 * vz - a2 * vy);
    const f2 g2 = w * a1 + c.sgn * (a2 * vx - a0 * vz);
    const f2 g3 = w * a2 + c.sgn * (a0 * vy - a1 * vx);
    // through the normalisation (already tangent: no projection); the de-normalisation is part of bL2's weights
    const f2 si = j.inv;
    f2 y0 = g0 * si, y1 = g1 * si, y2 = g2 * si, y3 = g3 * si;
    if (c.kindB == KIND_DISP) { y0.y = S[0].y; y1.y = S[1].y; y2.y = S[2].y; y3.y = 0.f; }
    gyA = f4{y0.x, y1.x, y2.x, y3.x};
    gyB = f4{y0.y, y1.y, y2.y, y3.y};
}

// ---- outputs of the LAST forward pass of (item, frame gf) from the frame block (reference: drag_pose.py:84-113 and what
// run() returns); kept simple, it runs once
struct OutC { int item, kind; f4 sd, mu; unsigned plo, phi; }; // what an item's outputs need from global memory
DEV void out_consts(const float* oc, int itemA, int kindA, int itemB, int kindB, OutC& oA, OutC& oB)
{ // from the quad's row of L_OC (staged by the set-up: no global round trip between the last iteration and the stores)
    const f4* t = (const f4*)oc; // sd[4][2], mu[4][2]: the two sides interleaved; then path_lo/hi of side A, of side B
    const f4 s0 = t[0], s1 = t[1], m0 = t[2], m1 = t[3], pw = t[4];
    oA.item = itemA; oA.kind = kindA;
    oB.item = itemB; oB.kind = kindB;
    oA.sd = f4{s0.x, s0.z, s1.x, s1.z}; oB.sd = f4{s0.y, s0.w, s1.y, s1.w};
    oA.mu = f4{m0.x, m0.z, m1.x, m1.z}; oB.mu = f4{m0.y, m0.w, m1.y, m1.w};
    oA.plo = __float_as_uint(pw.x); oA.phi = __float_as_uint(pw.y);
    oB.plo = __float_as_uint(pw.z); oB.phi = __float_as_uint(pw.w);
}
template <bool SEQ = false, class A>
DEV void w4_outputs(const A& a, const OutC& oc, float* fb, int gf, bool optimise, Q4 cur, unsigned tmask, bool early, bool pall = false, bool ploss = false)
{ // pall / ploss: this frame failed the input screening -- every result / the loss is NaN (what stays in the frame block for the state update is not) // SEQ (whole-sequence launches): gf = step * S + sequence; the state update's inputs are also left in the frame block, and the
  // pose written is the one run() RETURNS (root channels = the normalised world rotation, drag_pose.py:394-396)
    const int item = oc.item, kind = oc.kind;
    const auto P = [pall](float v) { return poisoned(pall, v); };
    if (item < 0 || kind == KIND_IDLE || kind == KIND_VIRT) return;
    const f4 sd = oc.sd, mu = oc.mu;
    const f4 qv = *(const f4*)(fb + FB_QS + 4 * item); // what stage J of the last forward pass left: unit quaternion / displacement
    const Q4 rq = {qv.x, qv.y, qv.z, qv.w};
    const f4 q0v = *(const f4*)(fb + FB_QS);
    const Q4 qw = quat_mul(cur, Q4{q0v.x, q0v.y, q0v.z, q0v.w}); // world rotation (drag_pose.py:88)
    const M3 R0 = quat_to_mat(qw);
    if (kind == KIND_DISP) {
        if (a.disp) { gfloat* o = GM(a.disp) + (size_t)gf * 3; o[0] = P(rq.w); o[1] = P(rq.x); o[2] = P(rq.y); }
        if (a.world_disp || SEQ) {
            const V3 wd = mat_vec(R0, V3{rq.w, rq.x, rq.y});
            if (a.world_disp) { gfloat* o = GM(a.world_disp) + (size_t)gf * 3; o[0] = P(wd.x); o[1] = P(wd.y); o[2] = P(wd.z); }
            if (SEQ) { *(f4*)(fb + FB_SWD) = f4{wd.x, wd.y, wd.z, 0.f}; *(f4*)(fb + FB_SD) = f4{rq.w, rq.x, rq.y, 0.f}; }
        }
        return;
    }
    const Q4 q = rq;
    if (a.pose) {
        gfloat* o = GM(a.pose) + (size_t)gf * 88 + 4 * item;
        if (SEQ && kind == KIND_ROOT) {
            o[0] = P((qw.w - a.seq.mean_q0[0]) / a.seq.std_q0[0]); o[1] = P((qw.x - a.seq.mean_q0[1]) / a.seq.std_q0[1]);
            o[2] = P((qw.y - a.seq.mean_q0[2]) / a.seq.std_q0[2]); o[3] = P((qw.z - a.seq.mean_q0[3]) / a.seq.std_q0[3]);
        } else {
            o[0] = P((q.w - mu.x) / sd.x); o[1] = P((q.x - mu.y) / sd.y); o[2] = P((q.y - mu.z) / sd.z); o[3] = P((q.z - mu.w) / sd.w);
        }
    }
    if (a.pos || SEQ) {
        const f4 dv = *(const f4*)(fb + FB_QS + 4 * QS_DISP);
        V3 pr = {dv.x, dv.y, dv.z};
        const unsigned plo = oc.plo, phi = oc.phi;
        f4 bn[MAX_PATH];
#pragma unroll
        for (int k = 0; k < MAX_PATH; ++k) bn[k] = *(const f4*)(fb + FB_BN + 4 * ((k < 6) ? ((plo >> (5 * k)) & 31u) : (phi & 31u)));
#pragma unroll
        for (int k = 0; k < MAX_PATH; ++k) { pr.x += bn[k].x; pr.y += bn[k].y; pr.z += bn[k].z; }
        const V3 pw = mat_vec(R0, pr);
        if (a.pos) { gfloat* o = GM(a.pos) + ((size_t)gf * NJ + item) * 3; o[0] = P(pw.x); o[1] = P(pw.y); o[2] = P(pw.z); }
        if (SEQ) { float* o = fb + FB_SPOS + 3 * item; o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; }
    }
    if (a.rot) {
        M3 M = quat_to_mat(q);
        if (kind == KIND_ROOT) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        const M3 G = mat_mat(R0, M);
        gfloat* o = GM(a.rot) + ((size_t)gf * NJ + item) * 9;
        o[0] = P(G.m00); o[1] = P(G.m01); o[2] = P(G.m02); o[3] = P(G.m10); o[4] = P(G.m11); o[5] = P(G.m12); o[6] = P(G.m20); o[7] = P(G.m21); o[8] = P(G.m22);
    }
    if (kind == KIND_ROOT) {
        if (a.world_rot) { gfloat* o = GM(a.world_rot) + (size_t)gf * 4; o[0] = P(qw.w); o[1] = P(qw.x); o[2] = P(qw.y); o[3] = P(qw.z); }
        if (SEQ) *(f4*)(fb + FB_SQW) = f4{qw.w, qw.x, qw.y, qw.z};
        if (optimise && a.loss && early) { // losses of the frame's last executed iteration, as the stop test saw them
            const f4 es = *(const f4*)(fb + FB_ES);
            gfloat* o = GM(a.loss) + (size_t)gf * 3;
            o[0] = poisoned(ploss, es.x); o[1] = poisoned(ploss, es.y); o[2] = poisoned(ploss, es.z);
        } else if (optimise && a.loss) {
            float lsum_p = 0.f, lsum_r = 0.f, lt = 0.f;
            const int E = min(__popc(tmask), W4_R);
            for (int e0 = 0; e0 < E; ++e0) { const f2 l = *(const f2*)(fb + FB_LP + 2 * e0); lsum_p += l.x; lsum_r += l.y; }
            for (int k = 0; k < LAT; k += 4) {
                const f4 dz = *(const f4*)(fb + FB_ZPRE + k) - *(const f4*)(fb + FB_ZT + k);
                lt += dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
            }
            gfloat* o = GM(a.loss) + (size_t)gf * 3;
            o[0] = poisoned(ploss, lsum_p);
            o[1] = poisoned(ploss, lsum_r);
            o[2] = poisoned(ploss, lt * a.lam_tmp * (1.f / 24.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// EARLY: the reference's per-frame while-condition (drag_pose.py:298-304, 351-355).  A stopped frame keeps its pre-step
// latent, so the forward passes it still takes part in reproduce its last one; a wave leaves the loop once all four of its
// frames have stopped.
// Head of bL2 kept resident beside the other products' weights: every streamed group costs the wave ~12 issue cycles per
// iteration for its ds_read_b128 (measured: 3 / 4 / 5 resident groups = -0.8 / -1.1 / -1.3 % kernel time).  B2_RES_A groups
// fill what is left of the accumulator half (244 + 12 = 256), B2_RES_V go to vector registers: 2 is what the fixed-count
// kernel holds without spilling, the early-stop kernel (more live state) none.
#ifndef W4_B2_RES_A
#define W4_B2_RES_A 3
#endif
#ifndef W4_B2_RES_V
#define W4_B2_RES_V 2
#endif
#ifndef W4_B2_RES_A_SEQ
#define W4_B2_RES_A_SEQ 1
#endif
// SEQ (whole-sequence launches, dp_optimize_sequence): frames are SEQUENCES; the kernel loops over a.seq.n_steps frame indices,
// carrying every sequence's state (latent, global position / rotation) from step to step in registers and LDS -- the per-frame
// epilogue of drag_pose.py:369-402 included -- and writes each step's results to the step's slab of the output arrays.
// LONG: n_iter beyond the kernel-argument table of Adam scalars (MAX_ITERS: the reference has no cap on max_iter) -- the same kernel with one
// uniform branch per iteration that continues the two bias corrections in double (adam_beyond).  Its own instantiations, because that branch,
// never taken, costs the ordinary launches 1.5-3 % (measured: it perturbs the loop's schedule); they are what every launch with n_iter <= 256 runs.
template <int NW, bool EARLY, bool SEQ = false, bool LONG = false>
__global__ __launch_bounds__(NW * 64, 1) void dp_w4_kernel(const KArgs a)
{
    static_assert(!SEQ || EARLY, "sequences run the reference's while-condition");
    static_assert(W4_B2_RES_A_SEQ >= 1 && W4_B2_RES_A >= 1, "the first resident group of bL2 starts the chain from zero");
    __shared__ __attribute__((aligned(16))) float lds[lds_total<NW>()];

#ifdef DP_PROFILE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    unsigned long long t_setup[5] = {0, 0, 0, 0, 0};
#define SETUP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); t_setup[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0) // (no drain: when the wave gets here)
#else
#define SETUP_STAMP(i)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, i = lane & 3; // quad (item pair / block of the products / tracker rank), lane in quad = frame
    const int nB = a.n_frames;
    const int f0 = (blockIdx.x * NW + wave) * FPW;
    const bool optimise = (a.mode == 0);

    const int gfi = min(f0 + i, nB - 1); // my frame as the lane of a quad (clamped: ragged tails compute a copy)
    const bool fvalid = f0 + i < nB;
    float* fb0 = lds + L_FR + wave * FPW * FB_STRIDE; // this wave's four frame blocks
    float* fb = fb0 + i * FB_STRIDE;

    // ---- Set-up.  Two things have to arrive: per wave, 71 KB of resident weights (the same for every wave), and a chain of
    //      dependent global round trips (tracked flags -> which joints -> their targets).  A wave may have 63 vector loads
    //      outstanding (vmcnt), so 71 weight loads per lane block it, and four waves pulling the same 71 KB through one
    //      64 B/clk L1 take 4.6 k cycles.  Instead the workgroup fetches the weight image ONCE -- 23 loads per lane into LDS
    //      (over the area the frame blocks will occupy; the streamed part straight to its final place) -- and every wave
    //      fills its registers from LDS (128 B/clk) while its tracker loads are in flight.  Three barriers instead of one.
    // ---- the whole weight image, once per workgroup: global -> registers -> LDS, requested before anything else (it is what
    //      the longest part of the set-up -- filling the registers -- waits for).  Groups of the streamed product (bL2) go to
    //      L_IMG2, where they stay; the others to the staging area [L_TAB, ...), in image order without them.
    constexpr int N_IMG = N_GROUPS * 64, N_LD = (N_IMG + NW * 64 - 1) / (NW * 64); // float4s of the image, loads per thread
    static_assert((N_GROUPS - NG_B2) * 256 <= lds_total<NW>() - L_TAB, "the staging area holds the resident groups");
    f4 im[N_LD];
#pragma unroll
    for (int k = 0; k < N_LD; ++k)
        if (tid + k * NW * 64 < N_IMG) im[k] = ((const f4*)a.w4img)[tid + k * NW * 64];
    __builtin_amdgcn_sched_barrier(0);

    // (dp_forward has no trackers and passes no tracker arrays: the loads then read the weight image and are ignored)
    typedef unsigned u32_any __attribute__((aligned(1), may_alias));
    typedef unsigned short u16_any __attribute__((aligned(1), may_alias));
    const unsigned char* trow = optimise ? a.tracked + (size_t)gfi * NJ : (const unsigned char*)a.w4img;
    unsigned tflag[6]; // the 22 flags of my frame: five unaligned words and a half
#pragma unroll
    for (int k = 0; k < 5; ++k) tflag[k] = ((const u32_any*)trow)[k];
    tflag[5] = *(const u16_any*)(trow + 20);
    f4 cv = *(const f4*)(a.cur_rot + (size_t)gfi * 4);
    // per-lane accumulator seeds (bias rows of L0, L1, L2A, L2B)
    const float bias0 = a.w4bias[lane], bias1 = a.w4bias[64 + lane], bias2a = a.w4bias[128 + lane], bias2b = a.w4bias[192 + lane];
    // latent and Adam state in the D layout of the last product: lane = latent dim, register r = frame f0 + r
    f4 zD = {0.f, 0.f, 0.f, 0.f}, ztD = zD, mD = zD, vD = zD;
    if (lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            const int gf = min(f0 + r, nB - 1);
            zD[r] = a.z0[(size_t)gf * LAT + lane];
            if (optimise) ztD[r] = a.z_tgt[(size_t)gf * (SEQ ? a.seq.z_tgt_seq : LAT) + lane];
        }
    }
    // kinematics constants of my quad's two items
    const Pair* pp = a.w4pairs + b;
    PairC pc;
#pragma unroll
    for (int k = 0; k < 3; ++k) pc.off[k] = f2{pp->off[k][0], pp->off[k][1]};
    pc.sgn = f2{pp->sgn[0], pp->sgn[1]};
    pc.rho = f2{pp->rho[0], pp->rho[1]};
    pc.subA = pp->ch_sub[0]; pc.subB = pp->ch_sub[1];
    pc.itemA = pp->item[0]; pc.itemB = pp->item[1];
    pc.kindB = pp->kind[1];
    const int kindA = pp->kind[0], slotA = pp->bone_slot[0], slotB = pp->bone_slot[1];
    const ItemConst* ic = a.items + min(b, MAX_ROOT_CH - 1); // constant root-frame bones of the root's children (quads 0..2 store them)
    const int init_id = ic->init_id;
    const f4 init_off = {ic->init_off[0], ic->init_off[1], ic->init_off[2], 0.f};
    // what the epilogue will need per quad (wave 0, one lane per quad): parked in LDS now, while loads are cheap
    const bool oc_lane = wave == 0 && i == 0;
    f4 ocv[5] = {};
    if (oc_lane) {
        const f4* t = (const f4*)pp;
#pragma unroll
        for (int k = 0; k < 4; ++k) ocv[k] = t[k];
        const ItemConst* ia = a.items + item_of(0, b); // (the item ids of a quad are a function of b: no dependent load)
        const ItemConst* ib = a.items + max(item_of(1, b), 0);
        ocv[4] = f4{__uint_as_float(ia->path_lo), __uint_as_float(ia->path_hi), __uint_as_float(ib->path_lo), __uint_as_float(ib->path_hi)};
    }
    const f2 adam_row = tid < min(a.n_iter, MAX_ITERS) ? f2{a.tab.step[tid], a.tab.bc2s[tid]} : f2{0.f, 0.f}; // (one row of the table per thread)
    __builtin_amdgcn_sched_barrier(0);

    // the image into LDS (the small loads above stay in flight: loads return in issue order and the image came first)
#pragma unroll
    for (int k = 0; k < N_LD; ++k) {
        const int e = tid + k * NW * 64, g = e >> 6; // float4 index, group
        if (e < N_IMG) {
            const int dst = g < GR_B2 ? L_TAB + 4 * e : g < GR_B2 + NG_B2 ? L_IMG2 + 4 * (e - GR_B2 * 64) : L_TAB + 4 * (e - NG_B2 * 64);
            *(f4*)(lds + dst) = im[k];
        }
    }
    __syncthreads();

    // resident weights from the staging area (MFMA B operands, loop-invariant: L0, L1, L2A, L2B, bL1 = 61 groups = 244
    // accumulator registers; bL0's 5 groups in VECTOR registers -- measured +2 % over streaming them; the head of bL2 in
    // what is left of both halves).  First the forward layers (the small loads have that long to come back) ...
    f4 wL0[6], wL1[10], wL2A[15], wL2B[15], wB1[15], wz[5];
    // (SEQ: the step loop around the iteration loop keeps more values alive; with every accumulator register holding a weight the
    //  allocator starts copying weight groups around inside the iteration loop -- behind the hand-padded MFMA groups' backs.  The
    //  head of bL2 is streamed like the rest of it there.)
    constexpr int B2_RES_A = SEQ ? W4_B2_RES_A_SEQ : W4_B2_RES_A, B2_RES_V = EARLY ? 0 : W4_B2_RES_V, B2_RES = B2_RES_A + B2_RES_V;
    f4 wB2a[B2_RES_A > 0 ? B2_RES_A : 1], wB2v[B2_RES_V > 0 ? B2_RES_V : 1];
    const f4* wst = (const f4*)(lds + L_TAB) + lane; // group g of the image at wst[64 g] (g < GR_B2), wst[64 (g - NG_B2)] beyond bL2
    load_w<6>(wL0, wst + (S_L0 / 4) * 64);
    load_w<10>(wL1, wst + (S_L1 / 4) * 64);
    load_w<15>(wL2A, wst + (S_L2A / 4) * 64);
    load_w<15>(wL2B, wst + (S_L2B / 4) * 64);
    __builtin_amdgcn_sched_barrier(0);

    // ... then the second level of the tracker chain (lane 4u+i: rank u of frame i), in flight under the rest
    // bit j = (flag byte j != 0): per word, OR every byte down into its bit 0, then gather the four bits with one multiply
    unsigned tmask = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned v = tflag[k];
        v |= v >> 4; v |= v >> 2; v |= v >> 1;
        tmask |= ((((v & 0x01010101u) * 0x01020408u) >> 24) & (k < 5 ? 0xFu : 0x3u)) << (4 * k);
    }
    if (!optimise) tmask = 0;
    const int E = min(__popc(tmask), W4_R);
    const int Emax = max(max(__builtin_amdgcn_readlane(E, 0), __builtin_amdgcn_readlane(E, 1)),
                         max(__builtin_amdgcn_readlane(E, 2), __builtin_amdgcn_readlane(E, 3)));
    TRaw raw = tracker_fetch(a, optimise, gfi, tmask, E, b);
    SETUP_STAMP(0);
    __builtin_amdgcn_sched_barrier(0);

    load_w<15>(wB1, wst + (S_B1 / 4 - NG_B2) * 64);
    load_w<5>(wz, wst + (S_B0 / 4 - NG_B2) * 64);
    {
        const f4* w2 = (const f4*)(lds + L_IMG2) + lane;
        load_w<B2_RES_A>(wB2a, w2);
        load_w<B2_RES_V>(wB2v, w2 + B2_RES_A * 64);
    }
    touch_a<6>(wL0); touch_a<10>(wL1); touch_a<15>(wL2A); touch_a<15>(wL2B); touch_a<15>(wB1);
    touch_a<B2_RES_A>(wB2a); touch_v<B2_RES_V>(wB2v); touch_v<5>(wz); // (the reads have completed: the area is free)
    __syncthreads();

    // ---- under the stream: frame blocks, trackers, LDS image
    // what has to read as zero in a frame block: the own-torque slots of untracked joints, the two tracker tables (stage G
    // reads rows beyond a frame's tracker count) and the bones' zero slot
    // (... and the tracker loss terms, which the stop test sums over all ranks: FB_LP follows FB_WT)
    constexpr int NW_LP = 32 + W4_R / 2, NZ = NW_LP + 2 * W4_R + 1;
    static_assert(FB_LP == FB_WT + 128 && W4_R % 2 == 0, "the loss terms follow the own-torque slots");
    for (int k = lane; k < FPW * NZ; k += 64) {
        const int fr = k / NZ, r = k % NZ;
        *(f4*)(fb0 + fr * FB_STRIDE + (r < NW_LP ? FB_WT + 4 * r : r < NW_LP + 2 * W4_R ? FB_GP + 4 * (r - NW_LP) : FB_BN + 4 * SLOT_ZERO)) = f4{0.f, 0.f, 0.f, 0.f};
    }
    static_assert(MAX_ITERS <= NW * 64, "one row of the Adam table per thread");
    if (tid < min(a.n_iter, MAX_ITERS)) *(f2*)(lds + L_TAB + 2 * tid) = adam_row;
    if (oc_lane) {
#pragma unroll
        for (int k = 0; k < 5; ++k) *(f4*)(lds + L_OC + 20 * b + 4 * k) = ocv[k];
    }
    if (SEQ && tid == 0) stage_step_args(lds, a);
    // ---- input screening (out_of_range above): which of my wave's four frames cannot be optimised, and their neutral stand-ins
    unsigned bad_state = 0u, bad_tgt = 0u; // bit r: frame f0 + r (uniform per wave).  state: z0 / cur_rot; tgt: targets, weights, z_tgt
    unsigned not_rot = 0u;                 // bit r: a tracked target of frame f0 + r is not a rotation matrix (DP_STATUS_TARGET_NOT_ROTATION: reported, computed as given)
    {
        bool tb = !SEQ && raw_bad(raw); // (SEQ: every step screens its own targets, in the step loop)
        bool nr = !SEQ && raw.act && not_rotation(raw.m);
        if (!SEQ)
            for (int base = 16; base < Emax; base += 16) { // (uniform, rare)
                const TRaw rx = tracker_fetch(a, true, gfi, tmask, E, base + b);
                tb = tb || raw_bad(rx);
                nr = nr || (rx.act && not_rotation(rx.m));
            }
        not_rot = frames_of(__ballot(nr && !tb)); // (a non-finite target is DP_STATUS_BAD_TARGETS, not this)
        const bool cb = out_of_range(cv.x) || out_of_range(cv.y) || out_of_range(cv.z) || out_of_range(cv.w);
        bad_tgt = frames_of(__ballot(tb));
        bad_state = frames_of(__ballot(cb));
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            bad_state |= (__ballot(lane < LAT && out_of_range(zD[r])) != 0ull ? 1u : 0u) << r;
            if (!SEQ) bad_tgt |= (__ballot(lane < LAT && out_of_range(ztD[r])) != 0ull ? 1u : 0u) << r;
        }
        bad_tgt &= ~bad_state;
        if (bad_state | bad_tgt) { // (uniform, rare)
            if (((bad_state | bad_tgt) >> i) & 1u) raw_neutral(raw);
            if ((bad_state >> i) & 1u) cv = f4{1.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < FPW; ++r) {
                if ((bad_state >> r) & 1u) zD[r] = 0.f;
                if (((bad_state | bad_tgt) >> r) & 1u) ztD[r] = zD[r]; // no pull: with the weights at zero the frame's gradient is exactly zero
            }
        }
    }
    if (b == 0) *(f4*)(fb + FB_CUR) = cv;
    wave_sync(); // (the zero fill above and the tracker records below touch the same frame blocks from different lanes)

    pc.qsA = FB_QS + 4 * pc.itemA;
    pc.qsB = FB_QS + 4 * (pc.itemB >= 0 ? pc.itemB : QS_TRASH);
    pc.bnA = FB_BN + 4 * slotA;
    pc.bnB = FB_BN + 4 * slotB;
    pc.wtA = FB_WT + 4 * (kindA == KIND_JOINT ? pc.itemA : WT_ZERO);                      // the root takes its torque from the root sum,
    pc.wtB = FB_WT + 4 * (pc.kindB == KIND_JOINT && pc.itemB >= 0 ? pc.itemB : WT_ZERO); // virtual copies only carry a bone
    pc.tab = kindA == KIND_ROOT ? FB_RT : FB_GP;
    {
        unsigned m = tmask;
#pragma unroll
        for (int u = 0; u < 6; ++u) { // first 6 ranks: the fast path of stage G
            const int t = __builtin_ctz(m | 0x80000000u);
            pc.sel[u] = (u < E) ? f2{(float)((pc.subA >> t) & 1u), (float)((pc.subB >> t) & 1u)} : splat2(0.f);
            m &= m - 1u;
        }
    }
    // bL2's dead K-groups (see `bl2` in the loop): side-A item b is live when its joint, or a joint below its child bone, is tracked in one of
    // the wave's four frames.  Two skip patterns are compiled: items 4 and 8 (mode 1), items 1..8 (mode 2); a mode is taken only when every item
    // of its pattern is dead -- whatever the skeleton and the tracker sets are, a skipped group multiplies zeros.
    constexpr unsigned B2_SKIP_1 = (1u << 4) | (1u << 8), B2_SKIP_2 = 0x1FEu;
    int b2_mode = 0;
    {
        const unsigned tm_any = (unsigned)__builtin_amdgcn_readlane((int)tmask, 0) | (unsigned)__builtin_amdgcn_readlane((int)tmask, 1) |
                                (unsigned)__builtin_amdgcn_readlane((int)tmask, 2) | (unsigned)__builtin_amdgcn_readlane((int)tmask, 3);
        const bool liveA = kindA != KIND_JOINT || ((pc.subA | (1u << pc.itemA)) & tm_any) != 0u; // (the root always is)
        const unsigned long long lv = __ballot(liveA);
        unsigned live16 = 0; // bit b: side-A item of quad b
#pragma unroll
        for (int q = 0; q < 16; ++q) live16 |= (unsigned)((lv >> (4 * q)) & 1ull) << q;
        b2_mode = (live16 & B2_SKIP_2) == 0u ? 2 : (live16 & B2_SKIP_1) == 0u ? 1 : 0;
        if (!optimise) b2_mode = 0;
    }
    TRec trk;
    {
        const Q4 cur = {cv.x, cv.y, cv.z, cv.w};
        V3 shift = {0.f, 0.f, 0.f};
        if (SEQ) {
            const float* gpp = a.seq.global_pos + (size_t)gfi * 3;
            const V3 gp0 = {gpp[0], gpp[1], gpp[2]};
            if (a.seq.tgt_root) { const float* rp = a.seq.tgt_root + (size_t)gfi * 3; shift = {rp[0] - gp0.x, rp[1] - gp0.y, rp[2] - gp0.z}; }
            if (b == 0) *(f4*)(fb + FB_GPOS) = f4{gp0.x, gp0.y, gp0.z, 0.f};
        }
        // (SEQ: every step, the first included, builds its tracker records in the step loop below -- ONE code path, so that a
        //  sequence cut into launches of any lengths gives the same bits)
        if (!SEQ) {
            trk = tracker_finish(a, fb, raw, E, cur, shift);
            for (int base = 16; base < Emax; base += 16) make_tracker(a, fb, gfi, tmask, E, base + b, cur, -1, shift, ((bad_state | bad_tgt) >> i) & 1u); // (uniform, rare)
        } else {
            trk = TRec{};
        }
    }
    if (b == 0) *(f4*)(fb + FB_QS + 4 * QS_IDENT) = f4{1.f, 0.f, 0.f, 0.f};
    if (b < MAX_ROOT_CH) *(f4*)(fb + FB_BN + 4 * init_id) = init_off;
    SETUP_STAMP(1);

    f4 zfinD = zD;               // early stop: latent after a frame's last step
    float es_prev = 10000000.f;  // early stop, lanes 0..3 (quad 0 = the root's) = frames: previous total loss (drag_pose.py:297),
    bool es_act = true;          //   still iterating,
    int es_iters = 0;            //   iterations executed
    if (EARLY && lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) { const float dz = zD[r] - ztD[r]; fb0[r * FB_STRIDE + FB_LT + lane] = dz * dz; }
    }

    __syncthreads();
    if (f0 >= nB) return; // (uniform per wave) no barrier below this line

    SETUP_STAMP(2);
    wave_sync();

    const f4 bias0T = splat(bias0), bias1T = splat(bias1), bias2aT = splat(bias2a), bias2bT = splat(bias2b); // C operands of the chains' first steps
    const f4 eT = {i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f, i == 3 ? 1.f : 0.f}; // unit rows of the transposes
    // ---- Stagger (round 5).  The four waves of a workgroup sit on four SIMDs and share nothing in the loop -- except the LDS: ~100 reads and
    //      writes of 1 KB per wave and iteration (the frame blocks' J / T / G exchanges, bL2's streamed weights), half of the LDS's cycles.  Released
    //      from one barrier the waves run the same instructions in lockstep, every burst of one collides with the same burst of the other three
    //      (a dependent read that takes ~70 cycles alone waits for three others' kilobytes), and nothing ever separates them.  Started W4_STAGGER x
    //      64 cycles apart -- about a quarter of an iteration: each wave's kinematics window falls into the others' matrix phases -- the launch is
    //      5-7 % shorter on every configuration (profiles/r05_stagger_sweep.txt: 12 ... 36, seven workloads, six rounds each; 18 ... 22 is the
    //      flat optimum, 20 the most even), the three sleeps included.  Outputs unchanged (a wave's arithmetic does not know when it runs).
//      Round 6, after Adam's packing shortened the iteration by ~130 cycles: 18 beats 20 on all seven workloads (-0.1 ... -0.7 %, four rounds each,
//      two boxes; 16 and 22 lose on the 100-iteration ones): profiles/r06_stagger_sweep.txt.
#ifndef W4_STAGGER
#define W4_STAGGER 18
#endif
    if (W4_STAGGER > 0 && optimise)
        for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(W4_STAGGER);
    JOut jo;
    Prof prof;
    prof.start();
#ifdef DP_PROFILE
    prof.t[10] = prof.prev - t_entry; // kernel entry -> first iteration
#ifndef W4_STAMP_L0 // (-DW4_STAMP_L0: slots 12..15 take the sub-phases of L0 instead -- Adam's tail | D -> X transpose | 24 K-steps | LeakyReLU; slot 0 is then empty)
    prof.t[12] = t_setup[0] - t_entry;    // image fetched into LDS (barrier), flags decoded, tracker loads issued
    prof.t[13] = t_setup[1] - t_setup[0]; // registers filled from LDS (barrier), frame blocks, trackers
    prof.t[14] = t_setup[2] - t_setup[1]; // LDS image, barrier
    prof.t[15] = prof.prev - t_setup[2];  // resident weights arrived
#endif
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), mt0 = prof.prev;
#endif
    // dp_result.clock: shader cycles and 100 MHz ticks workgroup 0 spends from here to its last store (four scalar instructions per launch)
    const bool clk_on = a.clk != nullptr && blockIdx.x == 0 && wave == 0; // (uniform)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_r0 = __builtin_amdgcn_s_memrealtime(); clk_c0 = __builtin_amdgcn_s_memtime(); }
    int step = 0;
#ifdef DP_SEQ_STAMPS // diagnostic build (tools/seq_step_stamps.sh): where a step of a whole-sequence launch spends its cycles
    unsigned long long sq_t[5] = {0, 0, 0, 0, 0}, sq_p = __builtin_amdgcn_s_memtime();
#define SQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sq_t[i] += n_ - sq_p; sq_p = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SQ_STAMP(i)
#endif
    double* adx = (double*)(lds + lds_adx<NW>()) + 2 * wave;
    // packed Adam state (pack_frames above): the fixed-count kernel only -- the while-condition's per-frame selects act on whole registers
    constexpr bool PK = W4_PACKED_ADAM && !EARLY;
    f2 zP = {0.f, 0.f}, ztP = zP, mP = zP, vP = zP;
    f4 xz = zD; // the latent in layout D for the next L0
    const int lane5 = lane & 31, fhalf = lane >> 5; // packed: my dim, my frame pair (frames fhalf * 2 + register ... see pack_frames: frame p + 2 fhalf)
    if constexpr (PK) { zP = pack_frames(zD); ztP = pack_frames(ztD); }
    do { // (one pass unless SEQ)
    if (LONG) { adx[0] = a.cont.b1t; adx[1] = a.cont.b2t; } // (every frame / step starts Adam afresh, drag_pose.py:218)
    if (SEQ) {
        const auto& as = step_args_of<SEQ>(a, lds);
        V3 step_shift = {0.f, 0.f, 0.f};
        int gft = step * nB + gfi;
        asm volatile("" : "+v"(gft)); // (opaque: keeps this step's address arithmetic out of the registers live across the iteration loop)
        if (as.seq.tgt_root) {
            const gfloat* rp = GM(as.seq.tgt_root) + (size_t)gft * 3;
            const f4 gpv = *(const f4*)(fb + FB_GPOS);
            step_shift = {rp[0] - gpv.x, rp[1] - gpv.y, rp[2] - gpv.z};
        }
        { // this frame of every sequence: its targets, a warm-started latent, a fresh Adam state (drag_pose.py:218)
            f4 cvs = *(const f4*)(fb + FB_CUR);
            TRaw rs = tracker_fetch(as, true, gfi, tmask, E, b, gft);
            if (step > 0) zD = zfinD;
            mD = f4{0.f, 0.f, 0.f, 0.f}; vD = mD;
            if (lane < LAT) {
#pragma unroll
                for (int r = 0; r < FPW; ++r) {
                    const int gf = min(f0 + r, nB - 1);
                    ztD[r] = GM(as.z_tgt)[(size_t)step * as.seq.z_tgt_step + (size_t)gf * as.seq.z_tgt_seq + lane];
                }
            }
            // input screening of this step: a sequence whose previous step could not be optimised has a NaN latent in the reference from then
            // on (drag_pose.py:342-344 wrote it): sticky
            bad_state |= bad_tgt;
            {
                bool tb = raw_bad(rs) || out_of_range(step_shift.x) || out_of_range(step_shift.y) || out_of_range(step_shift.z);
                bool nr = rs.act && not_rotation(rs.m);
                for (int base = 16; base < Emax; base += 16) { // (uniform, rare)
                    const TRaw rx = tracker_fetch(as, true, gfi, tmask, E, base + b, gft);
                    tb = tb || raw_bad(rx);
                    nr = nr || (rx.act && not_rotation(rx.m));
                }
                bad_tgt = frames_of(__ballot(tb));
                not_rot = frames_of(__ballot(nr && !tb));
#pragma unroll
                for (int r = 0; r < FPW; ++r) bad_tgt |= (__ballot(lane < LAT && out_of_range(ztD[r])) != 0ull ? 1u : 0u) << r;
                bad_tgt &= ~bad_state;
            }
            const bool my_bad = ((bad_state | bad_tgt) >> i) & 1u;
            if (bad_state | bad_tgt) { // (uniform, rare) neutral stand-ins: the frame's state stays finite, its results are poisoned in the epilogue
                if (my_bad) { raw_neutral(rs); step_shift = V3{0.f, 0.f, 0.f}; }
                if ((bad_state >> i) & 1u) {
                    cvs = f4{1.f, 0.f, 0.f, 0.f};
                    if (b == 0) *(f4*)(fb + FB_CUR) = cvs;
                }
#pragma unroll
                for (int r = 0; r < FPW; ++r) {
                    if ((bad_state >> r) & 1u) zD[r] = 0.f;
                    if (((bad_state | bad_tgt) >> r) & 1u) ztD[r] = zD[r];
                }
            }
            zfinD = zD;
            if (lane < LAT) {
#pragma unroll
                for (int r = 0; r < FPW; ++r) { const float dz = zD[r] - ztD[r]; fb0[r * FB_STRIDE + FB_LT + lane] = dz * dz; }
            }
            const Q4 cur = {cvs.x, cvs.y, cvs.z, cvs.w};
            trk = tracker_finish(as, fb, rs, E, cur, step_shift);
            for (int base = 16; base < Emax; base += 16) make_tracker(as, fb, gfi, tmask, E, base + b, cur, gft, step_shift, my_bad); // (uniform, rare)
            es_prev = 10000000.f; es_act = true; es_iters = 0;
            wave_sync();
        }
    }
    SQ_STAMP(0);
    for (int iter = 0; iter < a.n_iter; ++iter) {
        const bool last = (iter == a.n_iter - 1);
        const f2 adam_t = *(const f2*)(lds + L_TAB + 2 * (LONG ? min(iter, MAX_ITERS - 1) : iter)); // (an LDS broadcast read, issued a whole iteration ahead of its use)
        float step = adam_t.x, rbc2s = adam_t.y;
        if constexpr (LONG) {
            if (iter >= MAX_ITERS) adam_beyond(adx, a.cont, step, rbc2s); // (uniform: beyond the argument table)
        }
        int o = lane;
        asm volatile("" : "+v"(o)); // opaque per iteration: keeps the streamed weight reads inside the loop
        // (round 5, measured and not adopted: the streamed groups from global memory instead -- the L1 / L2 path is idle in the loop and the LDS is
        //  what the four waves contend for --: +3.3 % as it stands, +4 % with every chunk requested a phase earlier; 20 KB per wave and iteration
        //  does not live in the L1, and an L2 round trip is longer than any phase ahead the registers allow)
        const f4* w2 = (const f4*)(lds + L_IMG2) + o;

        // ================= L0: a0 = lrelu(A0 z + c0)
        f4 x = PK ? xz : zD;
#ifdef W4_STAMP_L0
        { // (a scalar read of the new latent: the stamp below cannot be taken before the Adam step's last result EXISTS -- what was still in flight is slot 12's)
            int t_;
            asm volatile("v_readfirstlane_b32 %0, %1\n\ts_nop 3\n\ts_add_u32 %0, %0, 0" : "=s"(t_) : "v"(x[2]) : "scc");
        }
        STAMP(12);
#endif
        QT(x);
#ifdef W4_STAMP_L0
        { int t_; asm volatile("s_nop 3\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 3\n\ts_add_u32 %0, %0, 0" : "=s"(t_) : "v"(x[3]) : "scc"); }
        STAMP(13);
#endif
        f4 acc0, acc1;
        chain_a<6, 0, 1>(acc0, acc1, x, wL0, bias0T);
        chain_end(acc0, acc1);
#ifdef W4_STAMP_L0
        STAMP(14);
#endif
        const f4 f0D = lrelu_factor(acc0 + acc1); // kept for the backward
#ifdef W4_STAMP_L0
        { int t_; asm volatile("v_readfirstlane_b32 %0, %1\n\ts_nop 3\n\ts_add_u32 %0, %0, 0" : "=s"(t_) : "v"(f0D[3]) : "scc"); }
        STAMP(15);
#else
        STAMP(0);
#endif
        // ================= L1: a1 = lrelu(A1 a0 + b1)
        x = (acc0 + acc1) * f0D;
        QT(x);
        chain_a<5, 0, 1>(acc0, acc1, x, wL1, bias1T); // the hidden layer's channels 0..19: quads 0..4,
        chain_a<5, 8>(acc0, acc1, x, wL1 + 5);  // 20..39: quads 8..12 (dp_w4.h)
        chain_end(acc0, acc1);
        const f4 f1D = lrelu_factor(acc0 + acc1);
        STAMP(1);
        // ================= L2: y = A2 a1 + b2, two 64-row blocks (channels 0, 1 | 2, 3 of both items of every quad)
        x = (acc0 + acc1) * f1D;
        QT(x);
        f4 y01, y23;
        {
            f4 pa0, pa1, pb0, pb1;
            chain_a<15, 0, 1>(pa0, pa1, x, wL2A, bias2aT);
            chain_a<15, 0, 1>(pb0, pb1, x, wL2B, bias2bT);
            chain_end(pa0, pa1);
            y01 = pa0 + pa1;
            y23 = pb0 + pb1;
        }
        __builtin_amdgcn_sched_barrier(0); // (both sums first: a VALU instruction between two MFMAs of one wave costs ~14 cycles)
        quad_transpose_mfma2(y01, y23, eT); // lane (b, i): the decoder channels of my two items of frame i, side A | side B in register pairs
        STAMP(2);

        // ================= kinematics
#ifndef W4_ABLATE_J
        j_stage(pc, fb, y01, y23, jo);
#else
        jo.q[0] = f2{y01[0], y01[1]}; jo.q[1] = f2{y01[2], y01[3]}; jo.q[2] = f2{y23[0], y23[1]}; jo.q[3] = f2{y23[2], y23[3]};
        jo.u[0] = jo.q[0]; jo.u[1] = jo.q[1]; jo.u[2] = jo.q[2]; jo.inv = splat2(1.f);
#endif
        wave_sync();
        STAMP(3);
        if (!optimise) break; // forward-only launch (uniform)
#ifndef W4_ABLATE_T // (diagnostic builds, tools/ablate_w4.sh: a stage left out to time the rest -- results are wrong by construction)
        t_stage(trk, fb, EARLY || last);
        for (int base = 16; base < Emax; base += 16) t_stage(load_tracker(a, fb, E, base + b), fb, EARLY || last); // (uniform, rare)
#endif
        // bL2's weights leave LDS in three chunks (a read costs the wave its issue time wherever it stands -- the four waves of
        // a workgroup want the same LDS cycles -- so the chunks only have to be requested a phase ahead of their use, and be
        // small enough for the register file): the rest of the first 8 groups across stage G, 8 ahead of the chain, 10 behind its first chunk (each
        // pinned: the scheduler would move the reads next to their use)
        constexpr int NQ = 8 - B2_RES;
        f4 wq[NQ], wr[8], ws[10];
        // (uniform; mode 2 leaves groups 1..8 out, see bl2 below: its chain goes from group 0 straight to group 9, so groups 9.. take the
        //  first chunk's registers and its place here -- requested at the head of the chain they would come back an LDS round trip late)
        if (EARLY || b2_mode != 2) load_w<NQ>(wq, w2 + B2_RES * 64);
        else load_w<NQ>(wq, w2 + 9 * 64);
        __builtin_amdgcn_sched_barrier(0);
        wave_sync();
        STAMP(4);
        unsigned actmask = 0xFu, stopmask = 0u; // bit r: frame f0 + r runs this iteration / stops after its step
        if (EARLY) {
            bool was_act = false, stop_now = false;
            { // the stop test of frame i, in EVERY lane of the frame's column (no branch: the reads and sums interleave with stage
              // G's; the state -- es_prev, es_act, es_iters -- is replicated over the column)
                // the loss terms of ALL ranks (zero beyond the frame's count: the set-up cleared them), read together -- a loop
                // over the frame's own count waits one LDS round trip per tracker -- and added in rank order
                float lp = 0.f, lr = 0.f, lt = 0.f;
                if (Emax <= 8) { // (uniform)
                    f4 l[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) l[k] = *(const f4*)(fb + FB_LP + 4 * k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { lp += l[k].x; lr += l[k].y; lp += l[k].z; lr += l[k].w; }
                } else {
                    f4 l[W4_R / 2];
#pragma unroll
                    for (int k = 0; k < W4_R / 2; ++k) l[k] = *(const f4*)(fb + FB_LP + 4 * k);
#pragma unroll
                    for (int k = 0; k < W4_R / 2; ++k) { lp += l[k].x; lr += l[k].y; lp += l[k].z; lr += l[k].w; }
                }
#pragma unroll
                for (int k = 0; k < LAT; k += 4) { const f4 d = *(const f4*)(fb + FB_LT + k); lt += (d.x + d.y) + (d.z + d.w); }
                lt *= a.lam_tmp * (1.f / 24.f);
                const float tot = (lp + lr) + lt;
                const bool cont = (lp > a.stop_eps_pos || lr > a.stop_eps_rot) && (es_prev - tot > a.min_loss_incr) && !last;
                was_act = es_act;
                es_prev = es_act ? tot : es_prev;
                es_iters += es_act ? 1 : 0;
                if (es_act && b == 0) *(f4*)(fb + FB_ES) = f4{lp, lr, lt, 0.f};
                stop_now = es_act && !cont;
                es_act = es_act && cont;
            }
            actmask = (unsigned)__ballot(was_act && b == 0) & 0xFu;
            stopmask = (unsigned)__ballot(stop_now && b == 0) & 0xFu;
        }
        f4 gyA, gyB;
#ifndef W4_ABLATE_G
        g_stage(pc, fb, jo, tmask, Emax, gyA, gyB);
#else
        gyA = f4{jo.q[0].x, jo.q[1].x, jo.q[2].x, jo.q[3].x}; gyB = f4{jo.q[0].y, jo.q[1].y, jo.q[2].y, jo.q[3].y};
#endif
        STAMP(5);

        // ================= bL2: d1 = (A2^T gy) * lrelu'(a1): K = 4 channels of the 16 side-A items (gyA), then of side-B
        // quads 1..10 (gyB)
        // A side-A item (= joint) whose dL/dy is exactly zero in all four frames of the wave -- neither it nor anything below its child bone
        // carries a tracker (the toes under the reference's 6 trackers; both legs under its 3- and 4-tracker sets) -- adds
        // nothing in its four K-steps: they are left out, with the read of their weights.  Which items is decided per wave in the set-up
        // (b2_mode); two patterns are compiled beside the general one, selected by a uniform branch per iteration.
        const auto bl2 = [&](auto skip_tag) {
            constexpr unsigned SK = decltype(skip_tag)::value;
            if constexpr (SK == B2_SKIP_2) { // group 0 | 9 .. 9 + NQ - 1 (in wq's registers, requested ahead of stage G) | the rest of side A | side B
                static_assert(NQ >= 1 && NQ <= 6, "mode 2 keeps groups 9.. in the first chunk's registers");
                load_w<7 - NQ>(wr, w2 + (9 + NQ) * 64);
                __builtin_amdgcn_sched_barrier(0);
                chain_begin();
                chain_a<B2_RES_A, 0, 2, SK>(acc0, acc1, gyA, wB2a);
                chain_v<NQ, 9>(acc0, acc1, gyA, wq);
                load_w<10>(ws, w2 + 16 * 64);
                __builtin_amdgcn_sched_barrier(0);
                chain_v<7 - NQ, 9 + NQ>(acc0, acc1, gyA, wr);
            } else {
                load_w<8, 8, SK>(wr, w2 + 8 * 64);
                __builtin_amdgcn_sched_barrier(0);
                chain_begin();
                chain_a<B2_RES_A, 0, 2, SK>(acc0, acc1, gyA, wB2a);
                chain_v<B2_RES_V, B2_RES_A, 0, SK>(acc0, acc1, gyA, wB2v);
                chain_v<NQ, B2_RES, 0, SK>(acc0, acc1, gyA, wq);
                load_w<10>(ws, w2 + 16 * 64); // (into the registers the first chunk has just released)
                __builtin_amdgcn_sched_barrier(0);
                chain_v<8, 8, 0, SK>(acc0, acc1, gyA, wr);
            }
            chain_v<B2_GROUPS_B, B2_ABID0_B>(acc0, acc1, gyB, ws);
            chain_end(acc0, acc1);
        };
        // (the early-stop and whole-sequence instantiations keep the one general chain: with their larger live state the three-way branch
        //  costs them spills inside the loop -- 3 and 39 registers, measured at compile time)
#ifdef W4_NO_B2_SKIP
        bl2(std::integral_constant<unsigned, 0u>{});
#else
        if constexpr (EARLY) bl2(std::integral_constant<unsigned, 0u>{});
        else {
            if (b2_mode == 0) bl2(std::integral_constant<unsigned, 0u>{});
            else if (b2_mode == 1) bl2(std::integral_constant<unsigned, B2_SKIP_1>{});
            else bl2(std::integral_constant<unsigned, B2_SKIP_2>{});
        }
#endif
        x = (acc0 + acc1) * f1D;
        QT(x);
        STAMP(6);
        // ================= bL1: d0 = (A1^T d1) * lrelu'(a0)
        chain_a<15, 0, 2>(acc0, acc1, x, wB1);
        chain_end(acc0, acc1);
        x = (acc0 + acc1) * f0D;
        QT(x);
        STAMP(7);
        // ================= bL0 + Adam (torch.optim.Adam, single-tensor form; m, v start at 0, t = iter + 1)
        chain3_v_zero<5>(acc0, acc1, x, wz); // two K-steps per instruction: lanes 0..31 | 32..63 hold the two halves of the sum
        chain_end(acc0, acc1);
        __builtin_amdgcn_sched_barrier(0); // (keeps the subtraction below out of the chains above: w4_probe, "independent v_pk_fma")
        f4 g = {0.f, 0.f, 0.f, 0.f};
        f2 gP = {0.f, 0.f};
        if constexpr (PK) gP = add_halves_packed(acc0 + acc1) + a.ctmp * (zP - ztP);
        else g = add_halves(acc0 + acc1) + a.ctmp * (zD - ztD);
        STAMP(8);
        if (PK && DBG_DUMP && a.dbg && iter == 0 && lane5 < LAT) {
            int ld = lane5;
            asm volatile("" : "+v"(ld));
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (f0 + r + 2 * fhalf < nB) a.dbg[(size_t)(f0 + r + 2 * fhalf) * DBG_STRIDE + DBG_GZ + ld] = gP[r];
        }
        if (!PK && DBG_DUMP && a.dbg && iter == 0 && lane < LAT) {
            int ld = lane;
            asm volatile("" : "+v"(ld)); // (opaque: the four 64-bit addresses of this once-per-launch dump are not to be formed ahead of the loop and held across it)
#pragma unroll
            for (int r = 0; r < FPW; ++r)
                if (f0 + r < nB) a.dbg[(size_t)(f0 + r) * DBG_STRIDE + DBG_GZ + ld] = g[r];
        }
#ifdef W4_ABLATE_ADAM
        if constexpr (PK) { zP = zP - 1e-6f * gP; xz = unpack_frames(zP, xz); }
        else zD = zD - 1e-6f * g;
#else
        if constexpr (PK) {
            if (last && lane5 < LAT) { // (uniform) latent of this, the last, forward pass: for the epilogue
#pragma unroll
                for (int r = 0; r < 2; ++r) fb0[(r + 2 * fhalf) * FB_STRIDE + FB_ZPRE + lane5] = zP[r];
            }
            mP = mP + a.one_m_b1 * (gP - mP);
            vP = vP * a.beta2 + a.one_m_b2 * (gP * gP);
            const f2 den = f2{__builtin_amdgcn_sqrtf(vP.x), __builtin_amdgcn_sqrtf(vP.y)} * rbc2s + a.eps;
            zP = zP - step * (mP * f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)});
            xz = unpack_frames(zP, xz);
        } else if (!EARLY) {
            if (last && lane < LAT) { // (uniform) latent of this, the last, forward pass: for the epilogue
#pragma unroll
                for (int r = 0; r < FPW; ++r) fb0[r * FB_STRIDE + FB_ZPRE + lane] = zD[r];
            }
            mD = mD + a.one_m_b1 * (g - mD);
            vD = vD * a.beta2 + a.one_m_b2 * (g * g);
            const f4 den = f4{__builtin_amdgcn_sqrtf(vD.x), __builtin_amdgcn_sqrtf(vD.y), __builtin_amdgcn_sqrtf(vD.z),
                              __builtin_amdgcn_sqrtf(vD.w)} * rbc2s + a.eps;
            zD = zD - step * (mD * f4{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                      __builtin_amdgcn_rcpf(den.w)});
        } else {
            const f4 mN = mD + a.one_m_b1 * (g - mD);
            const f4 vN = vD * a.beta2 + a.one_m_b2 * (g * g);
            const f4 den = f4{__builtin_amdgcn_sqrtf(vN.x), __builtin_amdgcn_sqrtf(vN.y), __builtin_amdgcn_sqrtf(vN.z),
                              __builtin_amdgcn_sqrtf(vN.w)} * rbc2s + a.eps;
            const f4 zN = zD - step * (mN * f4{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                               __builtin_amdgcn_rcpf(den.w)});
#pragma unroll
            for (int r = 0; r < FPW; ++r) {
                if ((actmask >> r) & 1u) { // (uniform)
                    if ((stopmask >> r) & 1u) {
                        if (lane < LAT) fb0[r * FB_STRIDE + FB_ZPRE + lane] = zD[r]; // latent of this frame's LAST forward pass
                        zfinD[r] = zN[r]; // the frame's loop ends with this step; z, m, v stay as they are
                    } else {
                        zD[r] = zN[r]; mD[r] = mN[r]; vD[r] = vN[r];
                        const float dz = zN[r] - ztD[r];
                        if (lane < LAT) fb0[r * FB_STRIDE + FB_LT + lane] = dz * dz;
                    }
                }
            }
        }
#endif
#ifdef W4_STAMP_L0
        { int t_; asm volatile("v_readfirstlane_b32 %0, %1\n\ts_nop 3\n\ts_add_u32 %0, %0, 0" : "=s"(t_) : "v"(PK ? xz[3] : zD[3]) : "scc"); } // (Adam's slot ends when its result exists)
#endif
        STAMP(9);
#ifdef DP_PROFILE
        if (iter == 0) prof.t[18] = prof.prev - mt0;               // the first iteration (cold instruction cache)
        if (iter == 1) prof.t[19] = prof.prev - mt0 - prof.t[18];  // the second
#endif
        if (EARLY && ((unsigned)__ballot(es_act && b == 0) & 0xFu) == 0u) break; // every frame of the wave has stopped
    }
#ifdef DP_PROFILE
    prof.t[16] = __builtin_amdgcn_s_memrealtime() - rt0; // 100 MHz ticks over the loop
    prof.t[17] = __builtin_amdgcn_s_memtime() - mt0;     // shader cycles over the loop
#endif
    SQ_STAMP(1);
    if constexpr (PK) { if (optimise) zD = xz; } // (layout D again: what the epilogue stores; a forward-only launch never left it)
    // ================= epilogue: outputs of the LAST forward pass (decoder quads still in registers; unit quaternions,
    // bones and the tracker loss terms in the frame blocks)
    const auto& ae = step_args_of<SEQ>(a, lds);
    // frames that failed the input screening: what the reference returns for them (include/dragposer.h: DP_STATUS_*) -- everything NaN when the
    // state was bad; z and the loss NaN when the targets were, and the pose results too unless the frame stopped after its first pass
    asm volatile("" : "+s"(bad_state), "+s"(bad_tgt), "+s"(not_rot)); // (opaque here: what the epilogue derives from them per lane is not to be computed ahead of the loop and held across it)
    const unsigned pois_z = bad_state | bad_tgt, pois_all = bad_state | ((EARLY || ae.n_iter == 1) ? 0u : bad_tgt);
    int row0 = SEQ ? step * nB : 0; // SEQ: this step's slab of the per-step output arrays
    int gfo = row0 + gfi;
    // SEQ: whatever the stores' addresses are made of is opaque per step -- the compiler otherwise hoists the step-invariant 64-bit
    // parts (item and lane offsets) out of the step loop, holds them across the iteration loop, spills them, and the epilogue then
    // sits out a dozen scratch reloads one after the other (5 k of a step's 13 k cycles outside the iterations, tools/seq_step_stamps.sh)
    int lane_e = lane, item_eA = pc.itemA, item_eB = pc.itemB, f0_e = f0, gfi_e = gfi;
    if (SEQ) {
        asm volatile("" : "+v"(gfo)); asm volatile("" : "+s"(row0));
        asm volatile("" : "+v"(lane_e), "+v"(item_eA), "+v"(item_eB), "+v"(gfi_e));
        asm volatile("" : "+s"(f0_e));
    }
    if (lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            if (!optimise) fb0[r * FB_STRIDE + FB_ZPRE + lane] = zD[r];
            fb0[r * FB_STRIDE + FB_ZT + lane] = ztD[r];
        }
    }
    wave_sync();
    {
        const f4 cve = *(const f4*)(fb + FB_CUR);
        OutC oA, oB;
        out_consts(lds + L_OC + 20 * b, item_eA, pc.tab == FB_RT ? KIND_ROOT : KIND_JOINT, item_eB, pc.kindB, oA, oB);
        if (fvalid || SEQ) { // (SEQ: the clamped copies of a ragged tail keep their own state consistent; their stores are skipped below)
            const Q4 cur = {cve.x, cve.y, cve.z, cve.w};
#ifndef W4_ABLATE_OUT // (diagnostic: the launch without its per-item result arrays -- what the epilogue costs)
            w4_outputs<SEQ>(ae, oA, fb, gfo, optimise, cur, tmask, EARLY, (pois_all >> i) & 1u, (pois_z >> i) & 1u); // (SEQ: the copies re-store the last valid frame's rows, same values)
            w4_outputs<SEQ>(ae, oB, fb, gfo, optimise, cur, tmask, EARLY, (pois_all >> i) & 1u, (pois_z >> i) & 1u);
#endif
        }
    }
    SQ_STAMP(2);
    if (optimise && lane < LAT) {
#pragma unroll
        for (int r = 0; r < FPW; ++r) {
            if (f0 + r < nB) {
                if (ae.z && (!SEQ || step == ae.seq.n_steps - 1)) GM(ae.z)[(size_t)(f0_e + r) * LAT + lane_e] = poisoned((pois_z >> r) & 1u, EARLY ? zfinD[r] : zD[r]);
                if (ae.z_pre) GM(ae.z_pre)[(size_t)(row0 + f0_e + r) * LAT + lane_e] = poisoned((pois_all >> r) & 1u, fb0[r * FB_STRIDE + FB_ZPRE + lane]);
                if (SEQ) GM(ae.seq.hist)[(size_t)(row0 + f0_e + r) * (LAT + 3 + ae.seq.n_heights) + lane_e] = poisoned((pois_all >> r) & 1u, fb0[r * FB_STRIDE + FB_ZPRE + lane]);
            }
        }
    }
    if (optimise && ae.iters && lane < FPW && f0 + lane < nB) GM(ae.iters)[row0 + f0_e + lane_e] = EARLY ? es_iters : ae.n_iter;
    if (ae.status) { // (uniform)
        unsigned nonfin = pois_z; // bit r: the latent returned for frame f0 + r is not finite
#pragma unroll
        for (int r = 0; r < FPW; ++r) nonfin |= (__ballot(optimise && lane < LAT && !(fabsf(EARLY ? zfinD[r] : zD[r]) <= 3.0e38f)) != 0ull ? 1u : 0u) << r;
        if (!optimise) nonfin = bad_state; // (dp_forward returns no latent: its frame results are NaN exactly when the state was refused)
        if (lane < FPW && f0 + lane < nB)
            GM(ae.status)[row0 + f0_e + lane_e] = (int)(((nonfin >> lane) & 1u) * DP_STATUS_NONFINITE_RESULT + ((bad_state >> lane) & 1u) * DP_STATUS_BAD_STATE +
                                                        ((bad_tgt >> lane) & 1u) * DP_STATUS_BAD_TARGETS + ((not_rot >> lane) & 1u) * DP_STATUS_TARGET_NOT_ROTATION);
    }
    SQ_STAMP(3);
    if (SEQ) { // the rest of run()'s epilogue (drag_pose.py:369-391), one lane per sequence: dp_sequence_advance's arithmetic
        wave_sync();
        if (b == 0) { // (the clamped copies of a ragged tail advance their state too -- or they would fall behind their targets,
                      //  iterate to the limit and hold the wave up --; only their stores are skipped)
            const f4 wd = *(const f4*)(fb + FB_SWD), qw = *(const f4*)(fb + FB_SQW), gp0 = *(const f4*)(fb + FB_GPOS);
            f4 ds = *(const f4*)(fb + FB_SD);
            float gp[3] = {gp0.x + wd.x, gp0.y + wd.y, gp0.z + wd.z}; // drag_pose.py:370
            float dsp[3] = {ds.x, ds.y, ds.z};
            if (ae.seq.adjust_joint >= 0) { // drag_pose.py:374-381
                const gfloat* tpp = GM(ae.tgt_pos) + ((size_t)gfo * NJ + ae.seq.adjust_target_joint) * 3;
                float sh[3] = {0.f, 0.f, 0.f}; // this step's target shift again (tgt_root[t] - the global position BEFORE this step)
                if (ae.seq.tgt_root) { const gfloat* rp = GM(ae.seq.tgt_root) + (size_t)gfo * 3; sh[0] = rp[0] - gp0.x; sh[1] = rp[1] - gp0.y; sh[2] = rp[2] - gp0.z; }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float adj = ((tpp[k] + sh[k]) - fb[FB_SPOS + 3 * ae.seq.adjust_joint + k]) * ae.seq.adjust_weight;
                    gp[k] += adj;
                    dsp[k] += adj;
                }
            }
            const bool pl = (pois_all >> i) & 1u; // (the state in the frame block stays finite; what leaves the kernel is the reference's NaN)
            if (ae.seq.pos_ret && fvalid) { gfloat* o = GM(ae.seq.pos_ret) + (size_t)gfo * 3; o[0] = poisoned(pl, gp[0]); o[1] = poisoned(pl, gp[1]); o[2] = poisoned(pl, gp[2]); }
            if (fvalid) {
                gfloat* o = GM(ae.seq.hist) + (size_t)gfo * (LAT + 3 + ae.seq.n_heights) + LAT;
                o[0] = poisoned(pl, dsp[0]); o[1] = poisoned(pl, dsp[1]); o[2] = poisoned(pl, dsp[2]);
                for (int h = 0; h < ae.seq.n_heights; ++h) o[3 + h] = poisoned(pl, fb[FB_SPOS + 3 * ae.seq.height_joints[h] + 1] + gp[1]);
            }
            *(f4*)(fb + FB_GPOS) = f4{gp[0], gp[1], gp[2], 0.f};
            *(f4*)(fb + FB_CUR) = qw; // drag_pose.py:371
            if (step == ae.seq.n_steps - 1 && fvalid) {
                gfloat* o = GM(ae.seq.global_pos) + (size_t)gfi_e * 3; o[0] = poisoned(pl, gp[0]); o[1] = poisoned(pl, gp[1]); o[2] = poisoned(pl, gp[2]);
                gfloat* oq = GM(ae.seq.global_rot) + (size_t)gfi_e * 4; oq[0] = poisoned(pl, qw.x); oq[1] = poisoned(pl, qw.y); oq[2] = poisoned(pl, qw.z); oq[3] = poisoned(pl, qw.w);
            }
        }
        wave_sync();
    }
    SQ_STAMP(4);
    ++step;
    } while (SEQ && step < a.seq.n_steps);
    if (clk_on) { // (uniform)
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c1 - clk_c0; a.clk[1] = r1 - clk_r0; }
    }
#ifdef DP_SEQ_STAMPS
    if (SEQ && tid == 0 && blockIdx.x == 0 && a.loss) { // the accumulated stamps over the first floats of `loss`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < 5; ++k) a.loss[k] = (float)sq_t[k];
    }
#endif
#ifdef DP_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the epilogue's stores have left the wave
    STAMP(11);
    prof.store(a.dbg, tid, blockIdx.x);
#endif
}

extern "C" hipError_t dp_launch_w4(const KArgs* args, hipStream_t stream)
{
    constexpr int NW = 4;
    const int grid = (args->n_frames + NW * FPW - 1) / (NW * FPW);
    const bool lng = args->n_iter > MAX_ITERS;
    if (args->seq.n_steps > 0) {
        if (lng) hipLaunchKernelGGL((dp_w4_kernel<NW, true, true, true>), dim3(grid), dim3(NW * 64), 0, stream, *args);
        else hipLaunchKernelGGL((dp_w4_kernel<NW, true, true>), dim3(grid), dim3(NW * 64), 0, stream, *args);
    } else if (args->early_stop && args->mode == 0) {
        if (lng) hipLaunchKernelGGL((dp_w4_kernel<NW, true, false, true>), dim3(grid), dim3(NW * 64), 0, stream, *args);
        else hipLaunchKernelGGL((dp_w4_kernel<NW, true>), dim3(grid), dim3(NW * 64), 0, stream, *args);
    } else {
        if (lng) hipLaunchKernelGGL((dp_w4_kernel<NW, false, false, true>), dim3(grid), dim3(NW * 64), 0, stream, *args);
        else hipLaunchKernelGGL((dp_w4_kernel<NW, false>), dim3(grid), dim3(NW * 64), 0, stream, *args);
    }
    return hipGetLastError();
}

extern "C" int dp_w4_lds_bytes(void) { return lds_total<4>() * 4; }
extern "C" int dp_w4_frames_per_block(void) { return 4 * FPW; }

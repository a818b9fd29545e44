// dp_p3.h -- the kinematics phase ("P3") of an iteration and the output epilogue of the 8-wave kernel (dp_kernel.hip):
// quaternion normalisation, root-frame FK, tracker
// loss terms, and the hand-derived backward to dL/dy (reference: DragPose.loss, python/src/drag_pose.py:66-194 with
// utils.py:80-149, and what autograd derives from it).
//
// Lane roles: 32 lanes per frame, lane = item (dp_layout.h: joints, root displacement, virtual child-bone copies).
// All cross-lane traffic goes through the frame's own LDS rows (FrameRows), written and read by one wave only, so
// the phase needs wave-level synchronisation only.  Tracker data is stored by tracker RANK (0-based position of the
// joint among the frame's tracked joints), capacity R per frame.
#pragma once
#include "dp_device.h"

constexpr int QD_S = 28; // qd row: qw[4] | d[3],0 | R0 rows (3 x [3],0) | cur_rot[4] | tracked-joint mask, -, -, -

template <int R> struct FrameRows { // LDS rows of one frame
    float* bone; // [32][4]   root-frame bone vectors by slot
    float* gpc;  // [R][4]    tracker position gradients
    float* cq;   // [R][4]    tracker contributions to dL/d(qw)
    float* lp;   // [R][2]    tracker loss terms (pos, rot)
    float* qd;   // [QD_S]
    float* trk;  // [4][R][4] tracker inputs: tp,cgp | tR0..3 | tR4..7 | tR8,cgr,clp,clr
};

struct ItemId { // what a lane needs to know about its item (registers; the float constants stay in LDS)
    int sq, dq, ch_id;
    unsigned ch_sub, plo, phi;
    bool is_joint, has_quat, is_root, is_disp;
};

DEV ItemId load_item(const ItemConst* ic)
{
    ItemId id;
    id.sq = ic->src_quad; id.dq = ic->dst_quad; id.ch_id = ic->ch_id;
    id.ch_sub = ic->ch_sub; id.plo = ic->path_lo; id.phi = ic->path_hi;
    const int kind = ic->kind;
    id.is_joint = kind == KIND_JOINT || kind == KIND_ROOT; // owns a tracker slot / outputs
    id.has_quat = kind != KIND_DISP && kind != KIND_IDLE;
    id.is_root = kind == KIND_ROOT;
    id.is_disp = kind == KIND_DISP;
    // The root lane needs no subtree sum of position gradients (M_0 is the identity in the root frame); it uses the
    // same accumulation slots to sum the trackers' contributions to dL/d(qw) instead, over every tracker.
    if (id.is_root) id.ch_sub = 0xFFFFFFFFu;
    return id;
}

// item constants: AoS in global memory (128 B per item) -> three float4 planes in LDS (sd | mu | child offset)
DEV void stage_item_tables(const KArgs& a, float* l_item, float* l_bias, int tid)
{
    if (tid < 32 * 3) {
        const int it = tid & 31, k = tid >> 5;
        const float* src = (const float*)(a.items + it) + (k == 0 ? 0 : k == 1 ? 4 : 8);
        *(f4*)(l_item + k * 128 + 4 * it) = f4{src[0], src[1], src[2], k == 2 ? 0.f : src[3]};
    }
    if (tid < 128) l_bias[tid] = a.bias[tid];
}

// Per-frame, loop-invariant setup.  Returns the lane's packed tracker word: bit 0 tracked, bits 1..5 rank,
// bits 8..13 sel6 (bit u: the tracked joint of rank u lies below my child bone; first 6 ranks = the fast path).
// Emax = max tracker count over the two frames of the wave (uniform).
template <int R>
DEV unsigned p3_setup(const KArgs& a, const ItemConst* icg, const ItemId& id, int lane, int it_id, int gfc, bool optimise,
                      const FrameRows<R>& fr, int& Emax)
{
    bool trk = false;
    if (optimise && id.is_joint) trk = a.tracked[gfc * NJ + it_id] != 0;
    unsigned long long bal = __ballot(trk);
    unsigned tmask = (lane >> 5) ? (unsigned)(bal >> 32) : (unsigned)bal; // tracked joints of my frame
    const int rank = __popc(tmask & ((1u << it_id) - 1u));
    if (R < NJ) { // beyond the capacity: ignored (memory-safe; the host never sends such batches to a kernel with R < 22)
        if (rank >= R) trk = false;
        bal = __ballot(trk);
        tmask = (lane >> 5) ? (unsigned)(bal >> 32) : (unsigned)bal;
    }
    const int E = __popc(tmask);
    unsigned sel6 = 0, m = tmask;
    for (int u = 0; u < 6; ++u) {
        const int t = __builtin_ctz(m | 0x80000000u);
        m &= m - 1u;
        sel6 |= ((id.ch_sub >> t) & 1u) << u;
    }
    Emax = max(__builtin_amdgcn_readlane(E, 0), __builtin_amdgcn_readlane(E, 32));
    if (it_id == 0) {
        *(f4*)(fr.qd + 20) = *(const f4*)(a.cur_rot + (size_t)gfc * 4);
        fr.qd[24] = __uint_as_float(tmask);
    }
    if (trk) {
        const float invE = 1.f / (float)E;
        const float* p = a.tgt_pos + (size_t)(gfc * NJ + it_id) * 3;
        const float* q = a.tgt_rot + (size_t)(gfc * NJ + it_id) * 9;
        const float wp = a.w[(gfc * NJ + it_id) * 2 + 0], wr = a.w[(gfc * NJ + it_id) * 2 + 1];
        const float clp = wp * invE * (1.f / 3.f);             // loss_pos coefficient  w_pos / (3E)
        const float clr = a.lam_rot * wr * invE * (1.f / 9.f); // loss_rot coefficient  lam w_rot / (9E)
        float* t = fr.trk + rank * 4;
        *(f4*)(t) = f4{p[0], p[1], p[2], 2.f * clp};
        *(f4*)(t + 4 * R) = f4{q[0], q[1], q[2], q[3]};
        *(f4*)(t + 8 * R) = f4{q[4], q[5], q[6], q[7]};
        *(f4*)(t + 12 * R) = f4{q[8], 2.f * clr, clp, clr};
    }
    // constant root-frame bones of the root's children
    if (it_id < MAX_ROOT_CH) *(f4*)(fr.bone + icg->init_id * 4) = f4{icg->init_off[0], icg->init_off[1], icg->init_off[2], 0.f};
    return (trk ? 1u : 0u) | ((unsigned)(rank & 31) << 1) | (sel6 << 8);
}

// One kinematics round for the lane's (frame, item).
//   icl    item's float constants in LDS (sd; +128: mu; +256: child offset)
//   y0     the frame's row of decoder outputs, plane 0; plane 1 (second K-half of tiles 4,5) at + FPB * S_Y
//   gyrow  the frame's row of dL/dy; `swz`: the row belongs to frames 8..15 (see swz4)
//   hook   called on the root lane after the tracker sums (early-stop bookkeeping of the 8-wave kernel)
template <int R, class Hook>
DEV void p3_round(const KArgs& a, const ItemId& id, unsigned pk, int Emax, const float* icl, const float* y0, float* gyrow,
                  bool swz, const FrameRows<R>& fr, int iter, int gfp, bool fvalid, Prof& prof, Hook&& hook)
{
    const bool trk = (pk & 1u) != 0u;
    const int rank = (int)((pk >> 1) & 31u);
    const float* tin = fr.trk + rank * 4;

    const f4 cv = *(const f4*)(fr.qd + 20); // cur_rot of my frame: fetched with y, used by the root lane only
    const f4 y4 = *(const f4*)(y0 + 4 * id.sq) + *(const f4*)(y0 + FPB * S_Y + 4 * id.sq);
    if (DBG_DUMP && a.dbg && iter == 0 && fvalid && id.dq == id.sq) *(f4*)(a.dbg + (size_t)gfp * DBG_STRIDE + DBG_Y + 4 * id.sq) = y4;
    const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
    const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
    const float nn = rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z;
    const float inv = id.has_quat ? __builtin_amdgcn_rsqf(nn) : 0.f;
    const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
    // one quaternion -> matrix for every lane: the root lane's is the world rotation qw = cur (x) q_0 (its own M is
    // the identity in the root frame), every other lane's is its root-space joint rotation (1 (x) q = q exactly)
    const Q4 qs = quat_mul(id.is_root ? Q4{cv.x, cv.y, cv.z, cv.w} : Q4{1.f, 0.f, 0.f, 0.f}, q);
    M3 M = quat_to_mat(qs);
    if (id.is_root) {
        *(f4*)(fr.qd) = f4{qs.w, qs.x, qs.y, qs.z};
        *(f4*)(fr.qd + 8) = f4{M.m00, M.m01, M.m02, 0.f};
        *(f4*)(fr.qd + 12) = f4{M.m10, M.m11, M.m12, 0.f};
        *(f4*)(fr.qd + 16) = f4{M.m20, M.m21, M.m22, 0.f};
    }
    if (id.is_root) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (id.is_disp) *(f4*)(fr.qd + 4) = f4{rq.w, rq.x, rq.y, 0.f};
    {
        const f4 cho = *(const f4*)(icl + 256); // child offset (x,y,z)
        const V3 u = mat_vec(M, V3{cho.x, cho.y, cho.z});
        *(f4*)(fr.bone + id.ch_id * 4) = f4{u.x, u.y, u.z, 0.f};
    }
    // tracker inputs of my joint (loop-invariant LDS data): fetched ahead of the exchange they do not depend on
    f4 t0, t1, t2, t3;
    if (trk) {
        t0 = *(const f4*)(tin);          // tp, cgp
        t1 = *(const f4*)(tin + 4 * R);  // tR[0..3]
        t2 = *(const f4*)(tin + 8 * R);  // tR[4..7]
        t3 = *(const f4*)(tin + 12 * R); // tR[8], cgr, clp, clr
    }
    wave_sync();
    prof.stamp(6);

    const f4 qwv = *(const f4*)(fr.qd);
    const f4 dv = *(const f4*)(fr.qd + 4);
    const f4 r0v = *(const f4*)(fr.qd + 8), r1v = *(const f4*)(fr.qd + 12), r2v = *(const f4*)(fr.qd + 16);
    const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
    const M3 R0 = {r0v.x, r0v.y, r0v.z, r1v.x, r1v.y, r1v.z, r2v.x, r2v.y, r2v.z};
    V3 pr = {dv.x, dv.y, dv.z}; // root-frame position: d + sum of the bones on the path
    {
        f4 b[MAX_PATH];
#pragma unroll
        for (int i = 0; i < MAX_PATH; ++i) {
            const unsigned k = (i < 6) ? ((id.plo >> (5 * i)) & 31u) : (id.phi & 31u);
            b[i] = *(const f4*)(fr.bone + k * 4);
        }
#pragma unroll
        for (int i = 0; i < MAX_PATH; ++i) { pr.x += b[i].x; pr.y += b[i].y; pr.z += b[i].z; }
    }
    M3 gM = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (trk) { // tracker terms in the root frame
        const V3 tp = {t0.x, t0.y, t0.z};
        const M3 tR = {t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x};
        const float cgp = t0.w, cgr = t3.y;
        const V3 tpr = matT_vec(R0, tp);
        const V3 e = {pr.x - tpr.x, pr.y - tpr.y, pr.z - tpr.z};
        const V3 gp = {cgp * e.x, cgp * e.y, cgp * e.z};
        const M3 tRr = matT_mat(R0, tR);
        const M3 eM = {M.m00 - tRr.m00, M.m01 - tRr.m01, M.m02 - tRr.m02, M.m10 - tRr.m10, M.m11 - tRr.m11,
                       M.m12 - tRr.m12, M.m20 - tRr.m20, M.m21 - tRr.m21, M.m22 - tRr.m22};
        gM = {cgr * eM.m00, cgr * eM.m01, cgr * eM.m02, cgr * eM.m10, cgr * eM.m11, cgr * eM.m12,
              cgr * eM.m20, cgr * eM.m21, cgr * eM.m22};
        // dL/dR0 = -(tp gp^T + tR gM^T)  ->  contribution to dL/d(qw)
        M3 C = mat_matT(tR, gM);
        C.m00 = -(C.m00 + tp.x * gp.x); C.m01 = -(C.m01 + tp.x * gp.y); C.m02 = -(C.m02 + tp.x * gp.z);
        C.m10 = -(C.m10 + tp.y * gp.x); C.m11 = -(C.m11 + tp.y * gp.y); C.m12 = -(C.m12 + tp.y * gp.z);
        C.m20 = -(C.m20 + tp.z * gp.x); C.m21 = -(C.m21 + tp.z * gp.y); C.m22 = -(C.m22 + tp.z * gp.z);
        const Q4 gqw_t = quat_mat_grad(qw, C);
        *(f4*)(fr.gpc + rank * 4) = f4{gp.x, gp.y, gp.z, 0.f};
        *(f4*)(fr.cq + rank * 4) = f4{gqw_t.w, gqw_t.x, gqw_t.y, gqw_t.z};
        const float l_p = t3.z * (e.x * e.x + e.y * e.y + e.z * e.z);
        const float l_r = t3.w * (eM.m00 * eM.m00 + eM.m01 * eM.m01 + eM.m02 * eM.m02 + eM.m10 * eM.m10 + eM.m11 * eM.m11 +
                                  eM.m12 * eM.m12 + eM.m20 * eM.m20 + eM.m21 * eM.m21 + eM.m22 * eM.m22);
        *(f2*)(fr.lp + rank * 2) = f2{l_p, l_r}; // read by the epilogue after the last iteration
    }
    wave_sync();
    prof.stamp(7);

    // sum over the trackers selected by my item: position gradients of the trackers below my child bone (S), or, on
    // the root lane, every tracker's contribution to dL/d(qw) -- same slots, same instructions, another table
    f4 S4 = {0.f, 0.f, 0.f, 0.f};
    {
        const float* tab = id.is_root ? fr.cq : fr.gpc;
        f4 g[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) g[u] = *(const f4*)(tab + u * 4);
        if (id.is_root) hook();
        const unsigned sel6 = pk >> 8;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const float b = (float)((sel6 >> u) & 1u);
            S4.x += b * g[u].x; S4.y += b * g[u].y; S4.z += b * g[u].z; S4.w += b * g[u].w;
        }
        if (Emax > 6) { // more than 6 trackers in a frame of this wave (uniform, rare): general path
            unsigned m = __float_as_uint(fr.qd[24]);
#pragma unroll
            for (int u = 0; u < 6; ++u) m &= m - 1u;
            for (int e0 = 6; e0 < Emax; ++e0) {
                const f4 ge = *(const f4*)(tab + e0 * 4);
                const int t = __builtin_ctz(m | 0x80000000u); // joint id of this rank (31 when exhausted)
                m &= m - 1u;
                const float b = (float)((id.ch_sub >> t) & 1u);
                S4.x += b * ge.x; S4.y += b * ge.y; S4.z += b * ge.z; S4.w += b * ge.w;
            }
        }
    }
    // root: d/d(q_0) through qw = cur (x) q_0 only;  joints: dL/dM_j = own rotation term + S o_child^T.  Both forms
    // are evaluated by every lane (straight-line code schedules better than a divergent if/else) and selected.
    const Q4 gq_root = quat_mul(Q4{cv.x, -cv.y, -cv.z, -cv.w}, Q4{S4.x, S4.y, S4.z, S4.w});
    const f4 cho = *(const f4*)(icl + 256);
    M3 X = gM;
    X.m00 += S4.x * cho.x; X.m01 += S4.x * cho.y; X.m02 += S4.x * cho.z;
    X.m10 += S4.y * cho.x; X.m11 += S4.y * cho.y; X.m12 += S4.y * cho.z;
    X.m20 += S4.z * cho.x; X.m21 += S4.z * cho.y; X.m22 += S4.z * cho.z;
    const Q4 gq_joint = quat_mat_grad(q, X);
    const Q4 gq = id.is_root ? gq_root : gq_joint;
    const V3 S = {S4.x, S4.y, S4.z};
    const float dot = q.w * gq.w + q.x * gq.x + q.y * gq.y + q.z * gq.z;
    f4 gyv = {sd.x * (gq.w - q.w * dot) * inv, sd.y * (gq.x - q.x * dot) * inv,
              sd.z * (gq.y - q.y * dot) * inv, sd.w * (gq.z - q.z * dot) * inv};
    if (id.is_disp) gyv = f4{sd.x * S.x, sd.y * S.y, sd.z * S.z, 0.f}; // ch_sub = every joint
    if (id.dq >= 0) {
        *(f4*)(gyrow + 4 * id.dq) = swz4(gyv, swz);
        if (DBG_DUMP && a.dbg && iter == 0 && fvalid) *(f4*)(a.dbg + (size_t)gfp * DBG_STRIDE + DBG_GY + 4 * id.dq) = gyv;
    }
}

// Outputs of the LAST forward pass of frame gfp, rebuilt from what it left in LDS (y planes, qw / d, bones, tracker
// loss terms, the pre-step latent) -- kept out of the hot loop.  `es`: the frame's early-stop record
// {-, -, iters, -, loss_pos, loss_rot, loss_tmp, -} or nullptr for fixed-iteration launches.
template <int R>
DEV void p3_outputs(const KArgs& a, const ItemId& id, const float* icl, const float* y0, const FrameRows<R>& fr, int it_id,
                    int gfp, bool optimise, const float* es, const float* zpre_row, const float* zt_row)
{
    const f4 y4 = *(const f4*)(y0 + 4 * id.sq) + *(const f4*)(y0 + FPB * S_Y + 4 * id.sq);
    const f4 sd = *(const f4*)(icl), mu = *(const f4*)(icl + 128);
    const Q4 rq = {y4.x * sd.x + mu.x, y4.y * sd.y + mu.y, y4.z * sd.z + mu.z, y4.w * sd.w + mu.w};
    const float inv = id.has_quat ? __builtin_amdgcn_rsqf(rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z) : 0.f;
    const Q4 q = {rq.w * inv, rq.x * inv, rq.y * inv, rq.z * inv};
    M3 M = quat_to_mat(q);
    if (id.is_root) M = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    const f4 qwv = *(const f4*)(fr.qd);
    const f4 dv = *(const f4*)(fr.qd + 4);
    const Q4 qw = {qwv.x, qwv.y, qwv.z, qwv.w};
    const M3 R0 = quat_to_mat(qw);
    if (id.is_joint) {
        if (a.pose) {
            float* o = a.pose + (size_t)gfp * 88 + 4 * it_id;
            o[0] = (q.w - mu.x) / sd.x; o[1] = (q.x - mu.y) / sd.y;
            o[2] = (q.y - mu.z) / sd.z; o[3] = (q.z - mu.w) / sd.w;
        }
        if (a.pos) {
            V3 pr = {dv.x, dv.y, dv.z};
            for (int i = 0; i < MAX_PATH; ++i) {
                const unsigned k = (i < 6) ? ((id.plo >> (5 * i)) & 31u) : (id.phi & 31u);
                const f4 b = *(const f4*)(fr.bone + k * 4);
                pr.x += b.x; pr.y += b.y; pr.z += b.z;
            }
            const V3 pw = mat_vec(R0, pr);
            float* o = a.pos + ((size_t)gfp * NJ + it_id) * 3;
            o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
        }
        if (a.rot) {
            const M3 G = mat_mat(R0, M);
            float* o = a.rot + ((size_t)gfp * NJ + it_id) * 9;
            o[0] = G.m00; o[1] = G.m01; o[2] = G.m02; o[3] = G.m10; o[4] = G.m11; o[5] = G.m12; o[6] = G.m20; o[7] = G.m21; o[8] = G.m22;
        }
    }
    if (id.is_root) {
        if (a.world_rot) { float* o = a.world_rot + (size_t)gfp * 4; o[0] = qw.w; o[1] = qw.x; o[2] = qw.y; o[3] = qw.z; }
        if (optimise && es) {
            for (int k = 0; k < LAT; k += 4)
                if (a.z_pre) *(f4*)(a.z_pre + (size_t)gfp * LAT + k) = *(const f4*)(zpre_row + k);
            if (a.loss) { a.loss[(size_t)gfp * 3 + 0] = es[4]; a.loss[(size_t)gfp * 3 + 1] = es[5]; a.loss[(size_t)gfp * 3 + 2] = es[6]; }
            if (a.iters) a.iters[gfp] = (int)es[2];
        } else if (optimise) {
            float lsum_p = 0.f, lsum_r = 0.f, lt = 0.f;
            const int E = __popc(__float_as_uint(fr.qd[24]));
            for (int e0 = 0; e0 < E; ++e0) { const f2 l = *(const f2*)(fr.lp + e0 * 2); lsum_p += l.x; lsum_r += l.y; }
            for (int k = 0; k < LAT; k += 4) {
                const f4 zz = *(const f4*)(zpre_row + k), zt = *(const f4*)(zt_row + k);
                if (a.z_pre) *(f4*)(a.z_pre + (size_t)gfp * LAT + k) = zz;
                const f4 dz = zz - zt;
                lt += dz.x * dz.x + dz.y * dz.y + dz.z * dz.z + dz.w * dz.w;
            }
            if (a.loss) {
                a.loss[(size_t)gfp * 3 + 0] = lsum_p;
                a.loss[(size_t)gfp * 3 + 1] = lsum_r;
                a.loss[(size_t)gfp * 3 + 2] = lt * a.lam_tmp * (1.f / 24.f);
            }
        }
    }
    if (id.is_disp) {
        if (a.disp) { float* o = a.disp + (size_t)gfp * 3; o[0] = rq.w; o[1] = rq.x; o[2] = rq.y; }
        if (a.world_disp) {
            const V3 wd = mat_vec(R0, V3{rq.w, rq.x, rq.y});
            float* o = a.world_disp + (size_t)gfp * 3; o[0] = wd.x; o[1] = wd.y; o[2] = wd.z;
        }
    }
}

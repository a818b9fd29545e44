// dp_device.h -- device-side helpers shared by the optimise kernels (dp_w4.hip: 4 frames per wave; dp_w16*.hip: 16 frames per wave;
// dp_kernel.hip: round 1's decomposition, 8 waves per 16-frame workgroup, kept in the test-only library for comparisons).
#pragma once
#include <hip/hip_runtime.h>
#include "dp_kernel.h"

using namespace dpl;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

// Diagnostic build (-DDP_PROFILE): wave 0 of every workgroup accumulates s_memtime deltas per phase
// and stores 20 x u64 per workgroup into the debug buffer.  Never part of the shipped library.
struct Prof {
#ifdef DP_PROFILE
    unsigned long long t[20], prev;
    DEV void start()
    {
        for (int i = 0; i < 20; ++i) t[i] = 0;
        prev = __builtin_amdgcn_s_memtime();
    }
    DEV void stamp(int i)
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        t[i] += now - prev;
        prev = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV void store(float* dbg, int tid, int block)
    {
#ifndef DP_PROFILE_WAVE
#define DP_PROFILE_WAVE 0 // which wave's stamps are stored (the older wave of a SIMD gets issue priority: compare 0 and 4)
#endif
        if (dbg && tid == 64 * DP_PROFILE_WAVE) {
            unsigned long long* o = (unsigned long long*)dbg + (size_t)block * 20;
            for (int i = 0; i < 20; ++i) o[i] = t[i];
        }
    }
#else
    DEV void start() {}
    DEV void stamp(int) {}
    DEV void store(float*, int, int) {}
#endif
};
#ifdef DP_PROFILE
#define DBG_DUMP 0
#else
#define DBG_DUMP 1
#endif
#define STAMP(i) prof.stamp(i)

DEV f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

DEV void wave_sync()
{ // orders this wave's LDS writes before its later LDS reads (other lanes' data); no instruction
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Adam's two per-iteration scalars beyond the kernel-argument table (iteration index >= MAX_ITERS; dp_kernel.h: AdamCont).  `st`: this wave's two
// running products in LDS (every lane reads, computes and writes the same values: nothing is held in registers from one iteration to the next).
DEV void adam_beyond(double* st, const AdamCont& c, float& step, float& rbc2s)
{
    const double b1 = st[0] * c.beta1, b2 = st[1] * c.beta2;
    st[0] = b1;
    st[1] = b2;
    step = (float)(c.lr / (1.0 - b1));
    rbc2s = (float)(1.0 / sqrt(1.0 - b2));
}

// include/dragposer.h, DP_STATUS_TARGET_NOT_ROTATION: the kernels turn a tracked joint's 3 x 3 target into a unit quaternion once per launch (the
// reference's element-wise MSE on matrices, drag_pose.py:121-124, equals the quaternion form for ROTATIONS only) -- where it is turned, R^T R is
// three norms and three dot products away: rows orthonormal within DP_ROTATION_TOL, or the frame says so in its status word.  (NaN fails every <=.)
DEV bool not_rotation(const float* m)
{
    const float n0 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2], n1 = m[3] * m[3] + m[4] * m[4] + m[5] * m[5], n2 = m[6] * m[6] + m[7] * m[7] + m[8] * m[8];
    const float d01 = m[0] * m[3] + m[1] * m[4] + m[2] * m[5], d02 = m[0] * m[6] + m[1] * m[7] + m[2] * m[8], d12 = m[3] * m[6] + m[4] * m[7] + m[5] * m[8];
    const float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const float tol = DP_ROTATION_TOL;
    return !(fabsf(n0 - 1.f) <= tol && fabsf(n1 - 1.f) <= tol && fabsf(n2 - 1.f) <= tol && fabsf(d01) <= tol && fabsf(d02) <= tol && fabsf(d12) <= tol && det > 0.f);
}

DEV float lrelu(float x) { return fmaxf(x, 0.2f * x); }
DEV float dlrelu(float a, float g) { return a > 0.f ? g : 0.2f * g; } // torch: x > 0 ? g : g*slope

struct M3 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };
struct Q4 { float w, x, y, z; };
struct V3 { float x, y, z; };

DEV M3 quat_to_mat(Q4 q)
{ // reference utils.py:49-74
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2, xy = q.x * y2, xz = q.x * z2, yz = q.y * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    M3 m;
    m.m00 = 1.f - (yy + zz); m.m01 = xy - wz;         m.m02 = xz + wy;
    m.m10 = xy + wz;         m.m11 = 1.f - (xx + zz); m.m12 = yz - wx;
    m.m20 = xz - wy;         m.m21 = yz + wx;         m.m22 = 1.f - (xx + yy);
    return m;
}

DEV Q4 quat_mat_grad(Q4 q, M3 X)
{ // g_k = sum_ab dM_ab/dq_k X_ab
    Q4 g;
    float a = X.m21 - X.m12, b = X.m02 - X.m20, c = X.m10 - X.m01; // antisymmetric part
    float s01 = X.m01 + X.m10, s02 = X.m02 + X.m20, s12 = X.m12 + X.m21;
    g.w = 2.f * (q.x * a + q.y * b + q.z * c);
    g.x = 2.f * (q.w * a + q.y * s01 + q.z * s02 - 2.f * q.x * (X.m11 + X.m22));
    g.y = 2.f * (q.w * b + q.x * s01 + q.z * s12 - 2.f * q.y * (X.m00 + X.m22));
    g.z = 2.f * (q.w * c + q.x * s02 + q.y * s12 - 2.f * q.z * (X.m00 + X.m11));
    return g;
}

DEV Q4 quat_mul(Q4 a, Q4 b)
{
    Q4 o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return o;
}

DEV V3 mat_vec(M3 m, V3 v) { return {m.m00 * v.x + m.m01 * v.y + m.m02 * v.z, m.m10 * v.x + m.m11 * v.y + m.m12 * v.z, m.m20 * v.x + m.m21 * v.y + m.m22 * v.z}; }
DEV V3 matT_vec(M3 m, V3 v) { return {m.m00 * v.x + m.m10 * v.y + m.m20 * v.z, m.m01 * v.x + m.m11 * v.y + m.m21 * v.z, m.m02 * v.x + m.m12 * v.y + m.m22 * v.z}; }

DEV M3 matT_mat(M3 a, M3 b)
{ // a^T b
    M3 c;
    c.m00 = a.m00 * b.m00 + a.m10 * b.m10 + a.m20 * b.m20; c.m01 = a.m00 * b.m01 + a.m10 * b.m11 + a.m20 * b.m21; c.m02 = a.m00 * b.m02 + a.m10 * b.m12 + a.m20 * b.m22;
    c.m10 = a.m01 * b.m00 + a.m11 * b.m10 + a.m21 * b.m20; c.m11 = a.m01 * b.m01 + a.m11 * b.m11 + a.m21 * b.m21; c.m12 = a.m01 * b.m02 + a.m11 * b.m12 + a.m21 * b.m22;
    c.m20 = a.m02 * b.m00 + a.m12 * b.m10 + a.m22 * b.m20; c.m21 = a.m02 * b.m01 + a.m12 * b.m11 + a.m22 * b.m21; c.m22 = a.m02 * b.m02 + a.m12 * b.m12 + a.m22 * b.m22;
    return c;
}

DEV M3 mat_mat(M3 a, M3 b)
{
    M3 c;
    c.m00 = a.m00 * b.m00 + a.m01 * b.m10 + a.m02 * b.m20; c.m01 = a.m00 * b.m01 + a.m01 * b.m11 + a.m02 * b.m21; c.m02 = a.m00 * b.m02 + a.m01 * b.m12 + a.m02 * b.m22;
    c.m10 = a.m10 * b.m00 + a.m11 * b.m10 + a.m12 * b.m20; c.m11 = a.m10 * b.m01 + a.m11 * b.m11 + a.m12 * b.m21; c.m12 = a.m10 * b.m02 + a.m11 * b.m12 + a.m12 * b.m22;
    c.m20 = a.m20 * b.m00 + a.m21 * b.m10 + a.m22 * b.m20; c.m21 = a.m20 * b.m01 + a.m21 * b.m11 + a.m22 * b.m21; c.m22 = a.m20 * b.m02 + a.m21 * b.m12 + a.m22 * b.m22;
    return c;
}

DEV M3 mat_matT(M3 a, M3 b)
{ // a b^T
    M3 c;
    c.m00 = a.m00 * b.m00 + a.m01 * b.m01 + a.m02 * b.m02; c.m01 = a.m00 * b.m10 + a.m01 * b.m11 + a.m02 * b.m12; c.m02 = a.m00 * b.m20 + a.m01 * b.m21 + a.m02 * b.m22;
    c.m10 = a.m10 * b.m00 + a.m11 * b.m01 + a.m12 * b.m02; c.m11 = a.m10 * b.m10 + a.m11 * b.m11 + a.m12 * b.m12; c.m12 = a.m10 * b.m20 + a.m11 * b.m21 + a.m12 * b.m22;
    c.m20 = a.m20 * b.m00 + a.m21 * b.m01 + a.m22 * b.m02; c.m21 = a.m20 * b.m10 + a.m21 * b.m11 + a.m22 * b.m12; c.m22 = a.m20 * b.m20 + a.m21 * b.m21 + a.m22 * b.m22;
    return c;
}

// B operand of steps step0 .. step0+N-1: one float per step, act[frame f16][4*step + h]
template <int N> DEV void load_b(const float* row, int h, int step0, float (&b)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = row[4 * (step0 + i) + h];
}

// N dependent-free MFMA steps on two interleaved accumulators; bit i of `mask` (wave-uniform)
// clear <=> the weight block of step i is structurally zero and the step is skipped
template <int N> DEV f4 mfma_chain(const float (&w)[N], const float (&b)[N], unsigned mask, f4 acc0)
{
    // launder the (loop-invariant) mask: otherwise the compiler hoists every bit test out of the
    // iteration loop into its own SGPR pair (~70 pairs), spills them, and the weights with them
    asm volatile("" : "+s"(mask));
    f4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if ((mask >> i) & 1u) {
            if (i & 1) acc1 = mfma4(w[i], b[i], acc1);
            else acc0 = mfma4(w[i], b[i], acc0);
        }
    }
    return acc0 + acc1;
}

// Bank swizzle of the B-operand buffers: a row is 16-byte aligned, so rows f and f+8 start on the same
// ds_read_b32 bank and the 32 lanes (16 frames x 2 K-groups) of a B read would pair up two by two.  Frames
// 8..15 therefore keep every 4-channel quad rotated by two floats: writers store swz4(v), the B read uses
// K-group h^2.  (Involution; P3 / epilogue readers of a quad apply swz4 again.)
DEV f4 swz4(f4 v, bool hi) { return hi ? f4{v.z, v.w, v.x, v.y} : v; }

// same without masks, for the dense products (no branches between the MFMAs)
template <int N> DEV f4 mfma_chain_dense(const float (&w)[N], const float (&b)[N], f4 acc0)
{
    f4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i & 1) acc1 = mfma4(w[i], b[i], acc1);
        else acc0 = mfma4(w[i], b[i], acc0);
    }
    return acc0 + acc1;
}

DEV f4 lrelu4(f4 x) { return f4{lrelu(x.x), lrelu(x.y), lrelu(x.z), lrelu(x.w)}; }
DEV f4 dlrelu4(f4 a, f4 g) { return f4{dlrelu(a.x, g.x), dlrelu(a.y, g.y), dlrelu(a.z, g.z), dlrelu(a.w, g.w)}; }


/* ORACLE (test infrastructure, not product): plain-C restatement of DragPoser's per-frame
 * latent optimisation with a HAND-DERIVED backward (no autograd).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * It follows the reference's algorithm (all under /root/reference/python/src):
 *   decoder            autoencoder.py:224-256, skeleton.py:117-130,244-245  -- folded: the linear
 *                      steps between non-linearities are pre-multiplied (exact in real arithmetic):
 *                      A0=(W0*M0) U0 Wf, c0=(W0*M0) U0 bf + b0, A1=(W1*M1) U1, A2=(W2*M2) U2
 *   denorm/normalise   autoencoder.py:242-253, drag_pose.py:84-85
 *   root composition   drag_pose.py:88-92,101-102 (quat mul / mul_vec of upc-pymotion 0.1.10,
 *                      restated as w-first Hamilton forms: the one unpinned boundary)
 *   FK                 utils.py:80-149, collapsed with the identity G_j = R0 M(q_j),
 *                      p_j = p_parent + G_parent o_j (decoder quaternions are root-space and unit)
 *   losses             drag_pose.py:116-127,185-188
 *   backward           reverse of the above w.r.t. z only (drag_pose.py:343 does it with autograd)
 *   Adam               torch.optim.Adam defaults, state reset per frame (drag_pose.py:218,344)
 *   loop / early stop  drag_pose.py:296-355
 * Checked in tests/test_oracle.py against the golden vectors produced by the real reference.
 *
 * Build: gcc -O2 -shared -fPIC -DREAL=float  -o _build/liboracle_f32.so analytic.c -lm
 *        gcc -O2 -shared -fPIC -DREAL=double -o _build/liboracle_f64.so analytic.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define NJ 22
#define D0 24
#define D1 40
#define D2 60
#define D3 92

typedef struct {
    real A0[D1 * D0], c0[D1], A1[D2 * D1], b1[D2], A2[D3 * D2], b2[D3];
    real mu4[88], sd4[88], mu_d[3], sd_d[3], offsets[NJ * 3];
    int parents[NJ];
} ora_model;

int ora_sizeof_real(void) { return (int)sizeof(real); }

/* raw decoder tensors (fp32, row-major [out][in]) -> folded model */
ora_model* ora_create(const float* Wf, const float* bf,
                      const float* U0, const float* W0, const float* M0, const float* b0,
                      const float* U1, const float* W1, const float* M1, const float* b1,
                      const float* U2, const float* W2, const float* M2, const float* b2,
                      const float* mu4, const float* sd4, const float* mu_d, const float* sd_d,
                      const int* parents, const float* offsets)
{
    ora_model* m = (ora_model*)calloc(1, sizeof(ora_model));
    double* T = (double*)calloc(D1 * D0, sizeof(double)); /* U0 Wf */
    double* tb = (double*)calloc(D1, sizeof(double));     /* U0 bf */
    for (int i = 0; i < D1; ++i) {
        for (int k = 0; k < D0; ++k) {
            double s = 0;
            for (int j = 0; j < D0; ++j) s += (double)U0[i * D0 + j] * Wf[j * D0 + k];
            T[i * D0 + k] = s;
        }
        double s = 0;
        for (int j = 0; j < D0; ++j) s += (double)U0[i * D0 + j] * bf[j];
        tb[i] = s;
    }
    for (int i = 0; i < D1; ++i) {
        for (int k = 0; k < D0; ++k) {
            double s = 0;
            for (int j = 0; j < D1; ++j) s += (double)(W0[i * D1 + j] * M0[i * D1 + j]) * T[j * D0 + k];
            m->A0[i * D0 + k] = (real)s;
        }
        double s = b0[i];
        for (int j = 0; j < D1; ++j) s += (double)(W0[i * D1 + j] * M0[i * D1 + j]) * tb[j];
        m->c0[i] = (real)s;
    }
    for (int i = 0; i < D2; ++i) {
        for (int k = 0; k < D1; ++k) {
            double s = 0;
            for (int j = 0; j < D2; ++j) s += (double)(W1[i * D2 + j] * M1[i * D2 + j]) * U1[j * D1 + k];
            m->A1[i * D1 + k] = (real)s;
        }
        m->b1[i] = b1[i];
    }
    for (int i = 0; i < D3; ++i) {
        for (int k = 0; k < D2; ++k) {
            double s = 0;
            for (int j = 0; j < D3; ++j) s += (double)(W2[i * D3 + j] * M2[i * D3 + j]) * U2[j * D2 + k];
            m->A2[i * D2 + k] = (real)s;
        }
        m->b2[i] = b2[i];
    }
    for (int i = 0; i < 88; ++i) { m->mu4[i] = mu4[i]; m->sd4[i] = sd4[i]; }
    for (int i = 0; i < 3; ++i) { m->mu_d[i] = mu_d[i]; m->sd_d[i] = sd_d[i]; }
    for (int i = 0; i < NJ; ++i) m->parents[i] = parents[i];
    for (int i = 0; i < NJ * 3; ++i) m->offsets[i] = offsets[i];
    free(T);
    free(tb);
    return m;
}

void ora_destroy(ora_model* m) { free(m); }

/* folded matrices out (as float), for comparing with the product's host-side fold */
void ora_get_folded(const ora_model* m, float* A0, float* c0, float* A1, float* b1, float* A2, float* b2)
{
    for (int i = 0; i < D1 * D0; ++i) A0[i] = (float)m->A0[i];
    for (int i = 0; i < D1; ++i) c0[i] = (float)m->c0[i];
    for (int i = 0; i < D2 * D1; ++i) A1[i] = (float)m->A1[i];
    for (int i = 0; i < D2; ++i) b1[i] = (float)m->b1[i];
    for (int i = 0; i < D3 * D2; ++i) A2[i] = (float)m->A2[i];
    for (int i = 0; i < D3; ++i) b2[i] = (float)m->b2[i];
}

static inline real lrelu(real x) { return x > 0 ? x : (real)0.2 * x; }

static void quat_to_mat(const real* q, real* M)
{ /* utils.py:49-74 */
    real w = q[0], x = q[1], y = q[2], z = q[3];
    real x2 = x + x, y2 = y + y, z2 = z + z;
    real xx = x * x2, yy = y * y2, zz = z * z2, xy = x * y2, xz = x * z2, yz = y * z2;
    real wx = w * x2, wy = w * y2, wz = w * z2;
    M[0] = (real)1 - (yy + zz); M[1] = xy - wz;             M[2] = xz + wy;
    M[3] = xy + wz;             M[4] = (real)1 - (xx + zz); M[5] = yz - wx;
    M[6] = xz - wy;             M[7] = yz + wx;             M[8] = (real)1 - (xx + yy);
}

/* gq[k] = sum_ab dM_ab/dq_k X_ab */
static void quat_mat_grad(const real* q, const real* X, real* g)
{
    real w = q[0], x = q[1], y = q[2], z = q[3];
    g[0] = 2 * (-z * X[1] + y * X[2] + z * X[3] - x * X[5] - y * X[6] + x * X[7]);
    g[1] = 2 * (y * X[1] + z * X[2] + y * X[3] - 2 * x * X[4] - w * X[5] + z * X[6] + w * X[7] - 2 * x * X[8]);
    g[2] = 2 * (-2 * y * X[0] + x * X[1] + w * X[2] + x * X[3] + z * X[5] - w * X[6] + z * X[7] - 2 * y * X[8]);
    g[3] = 2 * (-2 * z * X[0] - w * X[1] + x * X[2] + w * X[3] - 2 * z * X[4] + y * X[5] + x * X[6] + y * X[7]);
}

static void quat_mul(const real* a, const real* b, real* o)
{
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

static void matmul3(const real* A, const real* B, real* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

typedef struct {
    real loss[3];
    real grad[D0];
    real pose[88], dispn[3], wd[3], wrot[4], pos[NJ * 3], rot[NJ * 9], disp[3];
} frame_out;

/* one decode -> FK -> loss (-> backward) for one frame */
static void frame_eval(const ora_model* m, const real* z, const real* c, const real* z_tgt,
                       const real* tp, const real* tR, const real* w, const unsigned char* trk,
                       real lam_rot, real lam_tmp, int want_grad, frame_out* o)
{
    real a0[D1], a1[D2], y[D3];
    for (int i = 0; i < D1; ++i) {
        real s = m->c0[i];
        for (int k = 0; k < D0; ++k) s += m->A0[i * D0 + k] * z[k];
        a0[i] = lrelu(s);
    }
    for (int i = 0; i < D2; ++i) {
        real s = m->b1[i];
        for (int k = 0; k < D1; ++k) s += m->A1[i * D1 + k] * a0[k];
        a1[i] = lrelu(s);
    }
    for (int i = 0; i < D3; ++i) {
        real s = m->b2[i];
        for (int k = 0; k < D2; ++k) s += m->A2[i * D2 + k] * a1[k];
        y[i] = s;
    }
    real q[NJ][4], qd[NJ][4], nrm[NJ], M[NJ][9], G[NJ][9], d[3];
    for (int j = 0; j < NJ; ++j) {
        real r[4], s = 0;
        for (int k = 0; k < 4; ++k) { r[k] = y[4 * j + k] * m->sd4[4 * j + k] + m->mu4[4 * j + k]; s += r[k] * r[k]; }
        nrm[j] = (real)sqrt((double)s);
        for (int k = 0; k < 4; ++k) {
            q[j][k] = r[k] / nrm[j];
            o->pose[4 * j + k] = (q[j][k] - m->mu4[4 * j + k]) / m->sd4[4 * j + k]; /* autoencoder.py:251-253 */
            qd[j][k] = o->pose[4 * j + k] * m->sd4[4 * j + k] + m->mu4[4 * j + k]; /* drag_pose.py:84 */
        }
    }
    for (int k = 0; k < 3; ++k) { o->dispn[k] = y[88 + k]; d[k] = y[88 + k] * m->sd_d[k] + m->mu_d[k]; o->disp[k] = d[k]; }
    real qw[4], R0[9];
    quat_mul(c, qd[0], qw);
    quat_to_mat(qw, R0);
    for (int k = 0; k < 4; ++k) o->wrot[k] = qw[k];
    { /* mul_vec(qw, d) */
        real t[3] = {2 * (qw[2] * d[2] - qw[3] * d[1]), 2 * (qw[3] * d[0] - qw[1] * d[2]), 2 * (qw[1] * d[1] - qw[2] * d[0])};
        o->wd[0] = d[0] + qw[0] * t[0] + (qw[2] * t[2] - qw[3] * t[1]);
        o->wd[1] = d[1] + qw[0] * t[1] + (qw[3] * t[0] - qw[1] * t[2]);
        o->wd[2] = d[2] + qw[0] * t[2] + (qw[1] * t[1] - qw[2] * t[0]);
    }
    memcpy(G[0], R0, sizeof(R0));
    for (int k = 0; k < 3; ++k) o->pos[k] = o->wd[k];
    for (int j = 1; j < NJ; ++j) {
        int p = m->parents[j];
        quat_to_mat(qd[j], M[j]);
        matmul3(R0, M[j], G[j]);
        const real* of = &m->offsets[3 * j];
        for (int a = 0; a < 3; ++a)
            o->pos[3 * j + a] = o->pos[3 * p + a] + (G[p][3 * a] * of[0] + G[p][3 * a + 1] * of[1] + G[p][3 * a + 2] * of[2]);
    }
    memcpy(o->rot, G, sizeof(G));
    int E = 0;
    for (int j = 0; j < NJ; ++j) E += trk[j] ? 1 : 0;
    real lp = 0, lr = 0, lt = 0;
    real gp[NJ][3], gG[NJ][9];
    memset(gp, 0, sizeof(gp));
    memset(gG, 0, sizeof(gG));
    for (int j = 0; j < NJ; ++j) {
        if (!trk[j]) continue;
        for (int a = 0; a < 3; ++a) {
            real e = o->pos[3 * j + a] - tp[3 * j + a];
            lp += e * e * w[2 * j];
            gp[j][a] = 2 * w[2 * j] * e / (3 * (real)E);
        }
        for (int a = 0; a < 9; ++a) {
            real e = G[j][a] - tR[9 * j + a];
            lr += e * e * w[2 * j + 1];
            gG[j][a] = 2 * lam_rot * w[2 * j + 1] * e / (9 * (real)E);
        }
    }
    for (int k = 0; k < D0; ++k) { real e = z[k] - z_tgt[k]; lt += e * e; }
    o->loss[0] = lp / (3 * (real)E);
    o->loss[1] = lam_rot * lr / (9 * (real)E);
    o->loss[2] = lam_tmp * lt / D0;
    if (!want_grad) return;

    for (int j = NJ - 1; j >= 1; --j) { /* reverse topological: parents[j] < j */
        int p = m->parents[j];
        const real* of = &m->offsets[3 * j];
        for (int a = 0; a < 3; ++a) {
            gp[p][a] += gp[j][a];
            for (int b = 0; b < 3; ++b) gG[p][3 * a + b] += gp[j][a] * of[b];
        }
    }
    real gR0[9], gd[3], gy[D3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gR0[3 * a + b] = gG[0][3 * a + b] + gp[0][a] * d[b];
    for (int b = 0; b < 3; ++b) gd[b] = R0[b] * gp[0][0] + R0[3 + b] * gp[0][1] + R0[6 + b] * gp[0][2];
    real gq[NJ][4];
    for (int j = 1; j < NJ; ++j) {
        real X[9]; /* R0^T gG_j */
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                X[3 * a + b] = R0[a] * gG[j][b] + R0[3 + a] * gG[j][3 + b] + R0[6 + a] * gG[j][6 + b];
        quat_mat_grad(qd[j], X, gq[j]);
        for (int a = 0; a < 3; ++a) /* gR0 += gG_j M_j^T */
            for (int b = 0; b < 3; ++b)
                gR0[3 * a + b] += gG[j][3 * a] * M[j][3 * b] + gG[j][3 * a + 1] * M[j][3 * b + 1] + gG[j][3 * a + 2] * M[j][3 * b + 2];
    }
    {
        real gqw[4], cc[4] = {c[0], -c[1], -c[2], -c[3]};
        quat_mat_grad(qw, gR0, gqw);
        quat_mul(cc, gqw, gq[0]);
    }
    for (int j = 0; j < NJ; ++j) {
        real dot = 0;
        for (int k = 0; k < 4; ++k) dot += q[j][k] * gq[j][k];
        for (int k = 0; k < 4; ++k) gy[4 * j + k] = m->sd4[4 * j + k] * (gq[j][k] - q[j][k] * dot) / nrm[j];
    }
    for (int k = 0; k < 3; ++k) gy[88 + k] = m->sd_d[k] * gd[k];
    gy[91] = 0;
    real d1[D2], d0v[D1];
    for (int k = 0; k < D2; ++k) {
        real s = 0;
        for (int i = 0; i < D3; ++i) s += m->A2[i * D2 + k] * gy[i];
        d1[k] = a1[k] > 0 ? s : (real)0.2 * s;
    }
    for (int k = 0; k < D1; ++k) {
        real s = 0;
        for (int i = 0; i < D2; ++i) s += m->A1[i * D1 + k] * d1[i];
        d0v[k] = a0[k] > 0 ? s : (real)0.2 * s;
    }
    for (int k = 0; k < D0; ++k) {
        real s = 0;
        for (int i = 0; i < D1; ++i) s += m->A0[i * D0 + k] * d0v[i];
        o->grad[k] = s + 2 * lam_tmp * (z[k] - z_tgt[k]) / D0;
    }
}

/* forward only: decode + FK */
void ora_forward(const ora_model* m, int B, const float* z, const float* cur_rot,
                 float* pose, float* dispn, float* world_disp, float* world_rot, float* pos, float* rot)
{
    real zt[D0] = {0}, tp[NJ * 3] = {0}, tR[NJ * 9] = {0}, w[NJ * 2] = {0};
    unsigned char trk[NJ] = {1};
    for (int b = 0; b < B; ++b) {
        real zz[D0], c[4];
        frame_out o;
        for (int k = 0; k < D0; ++k) zz[k] = z[b * D0 + k];
        for (int k = 0; k < 4; ++k) c[k] = cur_rot[b * 4 + k];
        frame_eval(m, zz, c, zt, tp, tR, w, trk, 1, 0, 0, &o);
        for (int k = 0; k < 88; ++k) pose[b * 88 + k] = (float)o.pose[k];
        for (int k = 0; k < 3; ++k) { dispn[b * 3 + k] = (float)o.dispn[k]; world_disp[b * 3 + k] = (float)o.wd[k]; }
        for (int k = 0; k < 4; ++k) world_rot[b * 4 + k] = (float)o.wrot[k];
        for (int k = 0; k < NJ * 3; ++k) pos[b * NJ * 3 + k] = (float)o.pos[k];
        for (int k = 0; k < NJ * 9; ++k) rot[b * NJ * 9 + k] = (float)o.rot[k];
    }
}

/* losses + analytic gradient at z (one evaluation per frame) */
void ora_grad(const ora_model* m, int B, const float* z, const float* z_tgt, const float* cur_rot,
              const float* tgt_pos, const float* tgt_rot, const float* w, const unsigned char* tracked,
              float lam_rot, float lam_tmp, float* loss, float* grad)
{
    for (int b = 0; b < B; ++b) {
        real zz[D0], zt[D0], c[4], tp[NJ * 3], tR[NJ * 9], ww[NJ * 2];
        frame_out o;
        for (int k = 0; k < D0; ++k) { zz[k] = z[b * D0 + k]; zt[k] = z_tgt[b * D0 + k]; }
        for (int k = 0; k < 4; ++k) c[k] = cur_rot[b * 4 + k];
        for (int k = 0; k < NJ * 3; ++k) tp[k] = tgt_pos[b * NJ * 3 + k];
        for (int k = 0; k < NJ * 9; ++k) tR[k] = tgt_rot[b * NJ * 9 + k];
        for (int k = 0; k < NJ * 2; ++k) ww[k] = w[b * NJ * 2 + k];
        frame_eval(m, zz, c, zt, tp, tR, ww, tracked + b * NJ, lam_rot, lam_tmp, 1, &o);
        for (int k = 0; k < 3; ++k) loss[b * 3 + k] = (float)o.loss[k];
        for (int k = 0; k < D0; ++k) grad[b * D0 + k] = (float)o.grad[k];
    }
}

/* the optimise loop; outputs are those of the LAST forward pass, z_final is after the last step.
 * early_stop=0 -> exactly n_iter iterations per frame. */
void ora_optimize(const ora_model* m, int B, const float* z0, const float* z_tgt, const float* cur_rot,
                  const float* tgt_pos, const float* tgt_rot, const float* w, const unsigned char* tracked,
                  int n_iter, double lr, double beta1, double beta2, double eps, float lam_rot, float lam_tmp,
                  int early_stop, float stop_eps_pos, float stop_eps_rot, float min_loss_incr,
                  float* z_final, float* z_pre, float* pose, float* dispn, float* world_disp, float* world_rot,
                  float* pos, float* rot, float* loss, int* iters)
{
    for (int b = 0; b < B; ++b) {
        real z[D0], zt[D0], c[4], tp[NJ * 3], tR[NJ * 9], ww[NJ * 2], mm[D0] = {0}, vv[D0] = {0};
        frame_out o;
        for (int k = 0; k < D0; ++k) { z[k] = z0[b * D0 + k]; zt[k] = z_tgt[b * D0 + k]; }
        for (int k = 0; k < 4; ++k) c[k] = cur_rot[b * 4 + k];
        for (int k = 0; k < NJ * 3; ++k) tp[k] = tgt_pos[b * NJ * 3 + k];
        for (int k = 0; k < NJ * 9; ++k) tR[k] = tgt_rot[b * NJ * 9 + k];
        for (int k = 0; k < NJ * 2; ++k) ww[k] = w[b * NJ * 2 + k];
        double prev = 10000000.0, b1t = 1.0, b2t = 1.0;
        int it = 0;
        while (it < n_iter) {
            for (int k = 0; k < D0; ++k) z_pre[b * D0 + k] = (float)z[k];
            frame_eval(m, z, c, zt, tp, tR, ww, tracked + b * NJ, lam_rot, lam_tmp, 1, &o);
            ++it;
            b1t *= beta1;
            b2t *= beta2;
            real step = (real)(lr / (1.0 - b1t)), bc2s = (real)sqrt(1.0 - b2t);
            for (int k = 0; k < D0; ++k) {
                real g = o.grad[k];
                mm[k] = mm[k] + (real)(1.0 - beta1) * (g - mm[k]);
                vv[k] = vv[k] * (real)beta2 + (real)(1.0 - beta2) * g * g;
                real den = (real)sqrt((double)vv[k]) / bc2s + (real)eps;
                z[k] = z[k] - step * (mm[k] / den);
            }
            if (early_stop) {
                real totr = (o.loss[0] + o.loss[1]) + o.loss[2]; /* fp32 sum, then .item() */
                double tot = (double)totr;
                double incr = prev - tot;
                prev = tot;
                if (!((o.loss[0] > stop_eps_pos || o.loss[1] > stop_eps_rot) && incr > min_loss_incr)) break;
            }
        }
        iters[b] = it;
        for (int k = 0; k < D0; ++k) z_final[b * D0 + k] = (float)z[k];
        for (int k = 0; k < 88; ++k) pose[b * 88 + k] = (float)o.pose[k];
        for (int k = 0; k < 3; ++k) { dispn[b * 3 + k] = (float)o.dispn[k]; world_disp[b * 3 + k] = (float)o.wd[k]; loss[b * 3 + k] = (float)o.loss[k]; }
        for (int k = 0; k < 4; ++k) world_rot[b * 4 + k] = (float)o.wrot[k];
        for (int k = 0; k < NJ * 3; ++k) pos[b * NJ * 3 + k] = (float)o.pos[k];
        for (int k = 0; k < NJ * 9; ++k) rot[b * NJ * 9 + k] = (float)o.rot[k];
    }
}

/* ORACLE (test infrastructure, not product): a HOST-ONLY build of the C ABI of include/dragposer.h.
 *
 * "A host-only build of the same ABI (C++ CPU path) must exist so tests run in the GPU-less container."  This is it --
 * the SAME entry points, structs and status codes as libdragposer_hip.so (dp_create / dp_optimize / dp_forward / dp_destroy /
 * dp_last_error / dp_version / dp_auto_kernel / dp_kernel_geometry), implemented by the plain-C restatement of the reference's
 * algorithm in oracle/analytic.c (included below), with every "DEVICE pointer" of the header read as a HOST pointer and the stream
 * argument ignored.  It exists so that the CPU test suite can drive the ABI's call sequence end to end (struct layouts of
 * dragposer_amd/_lib.py, argument validation, status codes, optional result pointers) on the goldens the real reference produced.
 *
 * It is NOT a fallback: nothing under dragposer_amd/ loads, links or knows about it (the product fails loudly without a gfx950
 * device: tests/test_abi.py::test_create_fails_loudly_without_gpu); only tests/test_host_abi.py loads it, by explicit path.
 * Entry points of the header that this build does not implement return DP_ERR_UNSUPPORTED.
 *
 * Build (oracle/Makefile): gcc -O2 -shared -fPIC -I../include -o _build/libdragposer_hostonly.so host_abi.c -lm
 */
#define REAL float
#include "analytic.c"
#include "dragposer.h"
#include <stdio.h>

struct dp_ctx {
    ora_model* m;
    char err[256];
};
static __thread char g_create_err[256];

static int fail(dp_ctx* ctx, int code, const char* msg)
{
    snprintf(ctx ? ctx->err : g_create_err, 256, "%s", msg);
    return code;
}

int dp_version(void) { return DP_VERSION; }
const char* dp_last_error(const dp_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

static float bf16_round(float x)
{ /* nearest-even, as DP_WEIGHTS_BF16 asks (include/dragposer.h) */
    unsigned u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

int dp_create(dp_ctx** out, const dp_model* mo, int device)
{
    (void)device;
    if (!out || !mo) return fail(NULL, DP_ERR_INVALID, "dp_create: NULL argument");
    *out = NULL;
    if (!mo->f_latent_w || !mo->f_latent_b || !mo->mean_q || !mo->std_q || !mo->mean_disp || !mo->std_disp || !mo->parents || !mo->offsets)
        return fail(NULL, DP_ERR_INVALID, "dp_create: NULL model pointer");
    for (int l = 0; l < 3; ++l)
        if (!mo->unpool_w[l] || !mo->conv_w[l] || !mo->conv_mask[l] || !mo->conv_b[l]) return fail(NULL, DP_ERR_INVALID, "dp_create: NULL model pointer");
    if (mo->parents[0] != 0) return fail(NULL, DP_ERR_INVALID, "dp_create: parents[0] must be 0");
    for (int j = 1; j < NJ; ++j)
        if (mo->parents[j] < 0 || mo->parents[j] >= j) return fail(NULL, DP_ERR_INVALID, "dp_create: parents[j] must be < j");
    static const int dims[4] = {D0, D1, D2, D3};
    float* Wf = (float*)malloc(sizeof(float) * D0 * D0);
    float* W[3];
    memcpy(Wf, mo->f_latent_w, sizeof(float) * D0 * D0);
    for (int l = 0; l < 3; ++l) {
        const int n = dims[l + 1] * dims[l + 1];
        W[l] = (float*)malloc(sizeof(float) * n);
        memcpy(W[l], mo->conv_w[l], sizeof(float) * n);
    }
    if (mo->weight_dtype == DP_WEIGHTS_BF16) {
        for (int k = 0; k < D0 * D0; ++k) Wf[k] = bf16_round(Wf[k]);
        for (int l = 0; l < 3; ++l)
            for (int k = 0; k < dims[l + 1] * dims[l + 1]; ++k) W[l][k] = bf16_round(W[l][k]);
    } else if (mo->weight_dtype != DP_WEIGHTS_FP32) {
        free(Wf); for (int l = 0; l < 3; ++l) free(W[l]);
        return fail(NULL, DP_ERR_INVALID, "dp_create: unknown weight_dtype");
    }
    dp_ctx* ctx = (dp_ctx*)calloc(1, sizeof(dp_ctx));
    ctx->m = ora_create(Wf, mo->f_latent_b, mo->unpool_w[0], W[0], mo->conv_mask[0], mo->conv_b[0], mo->unpool_w[1], W[1], mo->conv_mask[1],
                        mo->conv_b[1], mo->unpool_w[2], W[2], mo->conv_mask[2], mo->conv_b[2], mo->mean_q, mo->std_q, mo->mean_disp, mo->std_disp,
                        mo->parents, mo->offsets);
    free(Wf);
    for (int l = 0; l < 3; ++l) free(W[l]);
    *out = ctx;
    return DP_OK;
}

int dp_destroy(dp_ctx* ctx)
{
    if (!ctx) return DP_ERR_INVALID;
    ora_destroy(ctx->m);
    free(ctx);
    return DP_OK;
}

int dp_fold_decoder(const dp_model* mo, dp_folded* out)
{
    dp_ctx* c;
    if (!out) return fail(NULL, DP_ERR_INVALID, "dp_fold_decoder: NULL argument");
    const int rc = dp_create(&c, mo, 0);
    if (rc != DP_OK) return rc;
    ora_get_folded(c->m, out->A0, out->c0, out->A1, out->b1, out->A2, out->b2);
    return dp_destroy(c);
}

/* results may be NULL: scratch rows stand in */
#define OUT_OR(p, scratch) ((p) ? (p) : (scratch))

int dp_optimize(dp_ctx* ctx, const dp_batch* in, const dp_params* p, const dp_result* out, void* stream)
{
    (void)stream;
    if (!ctx) return DP_ERR_INVALID;
    if (!in || !p) return fail(ctx, DP_ERR_INVALID, "dp_optimize: NULL batch/params");
    /* the refusals of dp_host.cpp (take_params / take_result): a struct compiled against another header is detected by its size word */
    if (p->struct_size < sizeof(dp_params) || p->struct_size > 4096u) return fail(ctx, DP_ERR_INVALID, "dp_optimize: dp_params.struct_size (pre-0.5 dragposer.h?)");
    /* (a 0.4 struct whose n_iter is 56 .. 4096 passes the size test: its second word is lr, an absurd iteration count -- refused before anything else is read) */
    if (p->n_iter < 1 || p->n_iter > DP_MAX_ITERS) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_iter out of range [1, DP_MAX_ITERS] (or dp_params.struct_size is a pre-0.5 caller's n_iter)");
    if (out && (out->struct_size < sizeof(dp_result) || out->struct_size > 4096u || out->reserved0 != 0u))
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: dp_result.struct_size (pre-0.5 dragposer.h?)");
    if (in->n_frames <= 0) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_frames must be positive");
    if (!in->z0 || !in->z_tgt || !in->cur_rot || !in->tgt_pos || !in->tgt_rot || !in->w || !in->tracked)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: NULL input array");
    if (p->n_iter < 1 || p->n_iter > DP_MAX_ITERS) return fail(ctx, DP_ERR_INVALID, "dp_optimize: n_iter out of range [1, DP_MAX_ITERS]");
    if (!(p->lr > 0.f) || !(p->beta1 >= 0.f && p->beta1 < 1.f) || !(p->beta2 >= 0.f && p->beta2 < 1.f))
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: bad Adam hyper-parameters");
    if (!(p->eps > 0.f)) return fail(ctx, DP_ERR_INVALID, "dp_optimize: Adam eps must be > 0 (include/dragposer.h: dp_params.eps)");
    if (p->kernel != DP_KERNEL_AUTO && p->kernel != DP_KERNEL_W4 && p->kernel != DP_KERNEL_W16)
        return fail(ctx, DP_ERR_INVALID, "dp_optimize: unknown kernel selector");
    const int B = in->n_frames;
    float* s = (float*)malloc(sizeof(float) * (size_t)B * (24 + 24 + 88 + 3 + 3 + 4 + NJ * 3 + NJ * 9 + 3 + 3) + sizeof(int) * (size_t)B);
    float *z = s, *zp = z + (size_t)B * 24, *pose = zp + (size_t)B * 24, *dn = pose + (size_t)B * 88, *wd = dn + (size_t)B * 3, *wr = wd + (size_t)B * 3,
          *pos = wr + (size_t)B * 4, *rot = pos + (size_t)B * NJ * 3, *loss = rot + (size_t)B * NJ * 9, *dsp = loss + (size_t)B * 3;
    int* it = (int*)(dsp + (size_t)B * 3);
    const dp_result none = DP_RESULT_INIT;
    const dp_result* o = out ? out : &none;
    ora_optimize(ctx->m, B, in->z0, in->z_tgt, in->cur_rot, in->tgt_pos, in->tgt_rot, in->w, in->tracked, p->n_iter, p->lr, p->beta1, p->beta2,
                 p->eps, p->lambda_rot, p->lambda_tmp, p->early_stop, p->stop_eps_pos, p->stop_eps_rot, p->min_loss_incr, OUT_OR(o->z, z),
                 OUT_OR(o->z_pre, zp), OUT_OR(o->pose, pose), dn, OUT_OR(o->world_disp, wd), OUT_OR(o->world_rot, wr), OUT_OR(o->pos, pos),
                 OUT_OR(o->rot, rot), OUT_OR(o->loss, loss), o->iters ? o->iters : it);
    if (o->disp) /* de-normalised root-space displacement (metres), as dp_result.disp says */
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < 3; ++k) o->disp[b * 3 + k] = dn[b * 3 + k] * (float)ctx->m->sd_d[k] + (float)ctx->m->mu_d[k];
    if (o->status) /* (the restatement has no input screening: it reports what came out) */
        for (int b = 0; b < B; ++b) {
            int bad = 0;
            for (int k = 0; k < 24; ++k) bad |= !isfinite(OUT_OR(o->z, z)[b * 24 + k]);
            o->status[b] = bad ? DP_STATUS_NONFINITE_RESULT : 0;
        }
    if (o->clock) o->clock[0] = o->clock[1] = 0; /* no shader clock on a CPU */
    free(s);
    return DP_OK;
}

int dp_forward(dp_ctx* ctx, int n_frames, const float* z, const float* cur_rot, const dp_result* out, void* stream)
{
    (void)stream;
    if (!ctx) return DP_ERR_INVALID;
    if (n_frames <= 0 || !z || !cur_rot || !out) return fail(ctx, DP_ERR_INVALID, "dp_forward: bad arguments");
    if (out->struct_size < sizeof(dp_result) || out->struct_size > 4096u || out->reserved0 != 0u)
        return fail(ctx, DP_ERR_INVALID, "dp_forward: dp_result.struct_size (pre-0.5 dragposer.h?)");
    const int B = n_frames;
    float* s = (float*)malloc(sizeof(float) * (size_t)B * (88 + 3 + 3 + 4 + NJ * 3 + NJ * 9));
    float *pose = s, *dn = pose + (size_t)B * 88, *wd = dn + (size_t)B * 3, *wr = wd + (size_t)B * 3, *pos = wr + (size_t)B * 4, *rot = pos + (size_t)B * NJ * 3;
    ora_forward(ctx->m, B, z, cur_rot, OUT_OR(out->pose, pose), dn, OUT_OR(out->world_disp, wd), OUT_OR(out->world_rot, wr), OUT_OR(out->pos, pos),
                OUT_OR(out->rot, rot));
    if (out->disp)
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < 3; ++k) out->disp[b * 3 + k] = dn[b * 3 + k] * (float)ctx->m->sd_d[k] + (float)ctx->m->mu_d[k];
    free(s);
    return DP_OK;
}

int dp_auto_kernel(const dp_ctx* ctx, int n_frames) { return !ctx || n_frames <= 0 ? DP_ERR_INVALID : DP_KERNEL_W4; }
int dp_kernel_geometry(const dp_ctx* ctx, int* f, int* t, int* l)
{
    (void)ctx;
    if (f) *f = 1;
    if (t) *t = 1;
    if (l) *l = 0;
    return DP_OK;
}

/* the rest of the header: not part of the host-only build */
#define UNSUPPORTED(ctx, name) return fail(ctx, DP_ERR_UNSUPPORTED, name ": not implemented by the host-only build")
int dp_sequence_advance(dp_ctx* ctx, int n, const dp_result* r, const dp_seq_state* s, const dp_seq_step* st, void* q)
{ (void)n; (void)r; (void)s; (void)st; (void)q; if (!ctx) return DP_ERR_INVALID; UNSUPPORTED(ctx, "dp_sequence_advance"); }
int dp_optimize_sequence(dp_ctx* ctx, int n, float* l, const dp_seq_frames* f, const dp_params* p, const dp_seq_state* s, const dp_seq_step* a,
                         const dp_seq_results* o, void* q)
{ (void)n; (void)l; (void)f; (void)p; (void)s; (void)a; (void)o; (void)q; if (!ctx) return DP_ERR_INVALID; UNSUPPORTED(ctx, "dp_optimize_sequence"); }
int dp_temporal_create(dp_temporal** out, const dp_temporal_model* m, int d) { (void)out; (void)m; (void)d; return fail(NULL, DP_ERR_UNSUPPORTED, "dp_temporal_create: not implemented by the host-only build"); }
int dp_temporal_destroy(dp_temporal* t) { (void)t; return DP_ERR_UNSUPPORTED; }
const char* dp_temporal_last_error(const dp_temporal* t) { (void)t; return g_create_err; }
int dp_temporal_predict(dp_temporal* t, int n, const dp_seq_state* s, int w, float* b, void* q) { (void)t; (void)n; (void)s; (void)w; (void)b; (void)q; return DP_ERR_UNSUPPORTED; }
int dp_temporal_status(const dp_temporal* t) { (void)t; return DP_ERR_UNSUPPORTED; }
int dp_io_alloc(dp_ctx* c, unsigned long long n, void** p) { if (!c || !p) return DP_ERR_INVALID; *p = malloc(n); return *p ? DP_OK : DP_ERR_DEVICE; }
int dp_io_free(dp_ctx* c, void* p) { if (!c) return DP_ERR_INVALID; free(p); return DP_OK; }
int dp_io_upload(dp_ctx* c, void* d, const void* s, unsigned long long n, void* q) { (void)q; if (!c || !d || !s) return DP_ERR_INVALID; memcpy(d, s, n); return DP_OK; }
int dp_io_download(dp_ctx* c, void* d, const void* s, unsigned long long n, void* q) { (void)q; if (!c || !d || !s) return DP_ERR_INVALID; memcpy(d, s, n); return DP_OK; }
int dp_stream_sync(dp_ctx* c, void* q) { (void)q; return c ? DP_OK : DP_ERR_INVALID; }
int dp_io_alloc_host(dp_ctx* c, unsigned long long n, void** p) { if (!c || !p) return DP_ERR_INVALID; *p = calloc(1, n); return *p ? DP_OK : DP_ERR_DEVICE; }
int dp_io_free_host(dp_ctx* c, void* p) { if (!c) return DP_ERR_INVALID; free(p); return DP_OK; }

/* A plain-C caller of include/dragposer.h (test program, built by tests/test_c_caller.py with gcc; no HIP headers, no torch): what a host
 * written in another language binds.  Reads the decoder tensors and one batch from raw float32 files, creates a context, moves the batch with
 * the dp_io_* helpers, runs dp_optimize and writes the results back as raw files.  Usage: c_caller DIR n_frames n_iter early_stop
 *   DIR holds f_latent_w.bin ... (the fields of dp_model, see below), parents.bin (int32), and z0 / z_tgt / cur_rot / tgt_pos / tgt_rot / w .bin,
 *   tracked.bin (uint8); outputs: DIR/out_z.bin, out_pos.bin, out_pose.bin, out_loss.bin, out_iters.bin (int32), out_status.bin (int32). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dragposer.h"

static void* slurp(const char* dir, const char* name, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void* p = malloc(bytes);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read of %s\n", path); exit(2); }
    fclose(f);
    return p;
}
static void spill(const char* dir, const char* name, const void* p, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}
#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != DP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dp_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const char* dir = argv[1];
    const int B = atoi(argv[2]), n_iter = atoi(argv[3]), early = atoi(argv[4]);
    dp_ctx* ctx = NULL;
    if (dp_version() != DP_VERSION) { fprintf(stderr, "header %d, library %d\n", DP_VERSION, dp_version()); return 1; }

    dp_model m;
    memset(&m, 0, sizeof m);
    m.f_latent_w = slurp(dir, "f_latent_w", 4 * 24 * 24);
    m.f_latent_b = slurp(dir, "f_latent_b", 4 * 24);
    static const int width[4] = {24, 40, 60, 92};
    for (int l = 0; l < 3; ++l) {
        char name[32];
        snprintf(name, sizeof name, "unpool_w%d", l);  m.unpool_w[l] = slurp(dir, name, 4u * width[l + 1] * width[l]);
        snprintf(name, sizeof name, "conv_w%d", l);    m.conv_w[l] = slurp(dir, name, 4u * width[l + 1] * width[l + 1]);
        snprintf(name, sizeof name, "conv_mask%d", l); m.conv_mask[l] = slurp(dir, name, 4u * width[l + 1] * width[l + 1]);
        snprintf(name, sizeof name, "conv_b%d", l);    m.conv_b[l] = slurp(dir, name, 4u * width[l + 1]);
    }
    m.mean_q = slurp(dir, "mean_q", 4 * 88); m.std_q = slurp(dir, "std_q", 4 * 88);
    m.mean_disp = slurp(dir, "mean_disp", 4 * 3); m.std_disp = slurp(dir, "std_disp", 4 * 3);
    m.parents = slurp(dir, "parents", 4 * DP_NUM_JOINTS);
    m.offsets = slurp(dir, "offsets", 4 * DP_NUM_JOINTS * 3);
    m.weight_dtype = DP_WEIGHTS_FP32;
    if (dp_create(&ctx, &m, 0) != DP_OK) { fprintf(stderr, "dp_create: %s\n", dp_last_error(NULL)); return 1; }

    /* a struct compiled against another header is refused, not read past */
    {
        dp_params old = DP_PARAMS_INIT;
        dp_batch none;
        memset(&none, 0, sizeof none);
        none.n_frames = 1;
        old.struct_size = 50; /* a 0.4 caller's first word: n_iter */
        if (dp_optimize(ctx, &none, &old, NULL, NULL) != DP_ERR_INVALID || !strstr(dp_last_error(ctx), "struct_size")) {
            fprintf(stderr, "a short dp_params was not refused\n");
            return 1;
        }
    }

    /* the batch: host -> device with the library's own helpers (a caller without a HIP binding) */
    const struct { const char* name; size_t per_frame; } in[] = {{"z0", 4 * 24}, {"z_tgt", 4 * 24}, {"cur_rot", 4 * 4}, {"tgt_pos", 4 * 22 * 3},
                                                                 {"tgt_rot", 4 * 22 * 9}, {"w", 4 * 22 * 2}, {"tracked", 22}};
    void* dev_in[7];
    for (int k = 0; k < 7; ++k) {
        void* h = slurp(dir, in[k].name, in[k].per_frame * B);
        CHECK(dp_io_alloc(ctx, in[k].per_frame * B, &dev_in[k]));
        CHECK(dp_io_upload(ctx, dev_in[k], h, in[k].per_frame * B, NULL));
        CHECK(dp_stream_sync(ctx, NULL));
        free(h);
    }
    dp_batch b;
    memset(&b, 0, sizeof b);
    b.n_frames = B;
    b.z0 = dev_in[0]; b.z_tgt = dev_in[1]; b.cur_rot = dev_in[2]; b.tgt_pos = dev_in[3]; b.tgt_rot = dev_in[4]; b.w = dev_in[5]; b.tracked = dev_in[6];

    dp_params p = DP_PARAMS_INIT;
    p.n_iter = n_iter; p.lr = 1e-2f; p.beta1 = 0.9f; p.beta2 = 0.999f; p.eps = 1e-8f;
    p.lambda_rot = 1.f; p.lambda_tmp = 0.02f;
    p.early_stop = early; p.stop_eps_pos = early ? 1e-4f : 0.f; p.stop_eps_rot = early ? 1e-2f : 0.f;
    p.min_loss_incr = early ? 1e-5f : -3.0e38f;
    p.kernel = DP_KERNEL_AUTO;

    const struct { const char* name; size_t per_frame; } out[] = {{"out_z", 4 * 24}, {"out_pos", 4 * 22 * 3}, {"out_pose", 4 * 88}, {"out_loss", 4 * 3},
                                                                  {"out_iters", 4}, {"out_status", 4}};
    void* dev_out[6];
    for (int k = 0; k < 6; ++k) CHECK(dp_io_alloc(ctx, out[k].per_frame * B, &dev_out[k]));
    dp_result r = DP_RESULT_INIT;
    r.z = dev_out[0]; r.pos = dev_out[1]; r.pose = dev_out[2]; r.loss = dev_out[3]; r.iters = dev_out[4]; r.status = dev_out[5];
    CHECK(dp_optimize(ctx, &b, &p, &r, NULL));
    for (int k = 0; k < 6; ++k) {
        void* h = malloc(out[k].per_frame * B);
        CHECK(dp_io_download(ctx, h, dev_out[k], out[k].per_frame * B, NULL));
        CHECK(dp_stream_sync(ctx, NULL));
        spill(dir, out[k].name, h, out[k].per_frame * B);
        free(h);
    }
    int fpb = 0, tpb = 0, lds = 0;
    CHECK(dp_kernel_geometry(ctx, &fpb, &tpb, &lds));
    printf("c_caller: %d frames x %d iterations, kernel geometry %d frames / %d threads per workgroup, %d bytes of LDS\n", B, n_iter, fpb, tpb, lds);
    for (int k = 0; k < 7; ++k) CHECK(dp_io_free(ctx, dev_in[k]));
    for (int k = 0; k < 6; ++k) CHECK(dp_io_free(ctx, dev_out[k]));
    return dp_destroy(ctx) == DP_OK ? 0 : 1;
}

/* dragposer.h -- C ABI of libdragposer_hip.so: DragPoser's per-frame latent optimisation on
 * MI355X (gfx950), batched over independent frames.
 *
 * What each entry point replaces in the reference (UPC-ViRVIG/DragPoser, python/src):
 *   dp_create / dp_destroy   the model state DragPose.__init__ keeps (drag_pose.py:13-45): frozen
 *                            Decoder tensors (autoencoder.py:146-222, skeleton.py:44-52,231-242),
 *                            dataset mean/std (drag_pose.py:27-34), skeleton parents/offsets
 *   dp_optimize              the optimise loop of DragPose.run (drag_pose.py:296-355): Decoder.forward
 *                            (autoencoder.py:224-256), DragPose.loss (drag_pose.py:66-194) with
 *                            from_root_quat_to_rotmat + fk_rotmat (utils.py:80-149), loss.backward()
 *                            and optim.Adam.step() (drag_pose.py:218,342-344) -- for B frames at once
 *   dp_forward               Decoder.forward + the FK part of DragPose.loss (drag_pose.py:84-113) with
 *                            no loss/backward: pose, world root transform and joint positions of z
 *   dp_sequence_advance      the per-frame epilogue of DragPose.run (drag_pose.py:369-402): global pose update,
 *                            joint adjustment, history buffers -- for S sequences in lock-step
 *   dp_optimize_sequence     the frame loop around both (eval_drag.py:204-222 calling DragPose.run per frame): T consecutive frames
 *                            of S sequences in one launch, the state carried on the device
 *   dp_temporal_*            Temporal (temporal_transformer.py:7-77) and the temporal target block of DragPose.run
 *                            (drag_pose.py:234-294): the source of z_tgt
 *   dp_fold_decoder          host-only helper: the algebra the reference re-does every call
 *                            (W*mask, skeleton.py:120; unpool matmul, skeleton.py:245) done once
 *
 * Conventions (the reference's: SURVEY.md 8.2): quaternions are (w,x,y,z) Hamilton; all arrays are
 * fp32, row-major, frame-major; joint count 22, latent 24, decoder widths 24->40->60->92.
 * Every function returns DP_OK (0) or a negative dp_status and never throws; the message of the
 * last failure on a context is dp_last_error(ctx) (dp_last_error(NULL): last failure of dp_create
 * on this thread).  The caller owns every buffer; a context owns only its device copy of the
 * model.  dp_optimize / dp_forward are asynchronous on the given HIP stream and perform no
 * allocation, no host synchronisation and no host<->device copy of caller data (graph-capturable).
 * One context per device; a context is not thread-safe; distinct contexts are independent.
 * There is NO CPU fallback: without a usable gfx950 device dp_create fails with DP_ERR_DEVICE.
 *
 * Skeletons the kernels are laid out for (dp_create returns DP_ERR_UNSUPPORTED with a message otherwise): 22 joints with
 * parents[j] < j; at most 3 children of the root; at most 3 child bones beyond the first on all non-root joints together
 * (Xsens: the two shoulders and the neck branching off the upper spine use 2); kinematic chains of at most 7 bones from the
 * root.  The skeleton of the reference's data (the Xsens hierarchy every .bvh under python/data has, train.py:75-97)
 * satisfies all four.  What kind of limit each one is (tests/test_hip_topology.py runs two other trees inside them through DP_KERNEL_W4):
 *   22 joints               the MODEL: the decoder's widths 24 -> 40 -> 60 -> 92 = 22 x 4 + 4 are the checkpoint's (a layout constant of every kernel)
 *   <= 3 root children      a table size (the root's children have constant root-frame bones, stored by item quads 0..2): raising it costs nothing
 *                           in the loop
 *   <= 3 extra child bones  a layout constant WITH A PRICE: every extra child bone is a "virtual" item whose decoder rows are a copy of its joint's,
 *                           i.e. four more K-steps of the transposed last layer (K = 92 + 12 today); the item map has room for 8
 *   chains of <= 7 bones    a table size with a price: a tracker sums the bones of its path in one pass of 7 LDS reads (two 32-bit path words hold 12)
 * DP_KERNEL_W16's register slot map IS the reference's tree (four lane groups = left leg | right leg + lower spine | left arm + upper spine | right
 * arm + head): any other parents array gets DP_ERR_UNSUPPORTED from it and DP_KERNEL_W4 from DP_KERNEL_AUTO.
 */
#ifndef DRAGPOSER_H
#define DRAGPOSER_H

#ifdef __cplusplus
extern "C" {
#endif

#define DP_VERSION 510 /* 0.5.1 (compatible with 0.5.0 callers: nothing moved, nothing grew): dp_temporal_status, DP_ERR_TIMEOUT,
                          DP_STATUS_TARGET_NOT_ROTATION, DP_KERNEL_W16 for n_iter > 256.
                          0.5.0: ABI BREAK, the last one of this kind: dp_params and dp_result now START with `struct_size` (sizeof the
                          struct as the caller compiled it), so a caller built against another version of this header is DETECTED
                          (DP_ERR_INVALID with a message naming both sizes) instead of read past; fields appended later are read only
                          when struct_size covers them and take their defaults (0 / NULL) otherwise.  dp_result grew `status` (per-frame
                          health word) and `clock` (shader clock the launch ran at).  Use DP_PARAMS_INIT / DP_RESULT_INIT.
                          History: 0.4.0 dp_auto_kernel, Adam eps must be > 0; 0.3.0 dp_params grew `kernel` (the break that went
                          unstated), dp_optimize_sequence; 0.2 dp_temporal_*, max_trackers ignored */

#define DP_NUM_JOINTS 22
#define DP_LATENT 24
#define DP_POSE_CHANNELS 88 /* 22 joints x 4 quaternion channels */
#define DP_MAX_ITERS 1000000 /* a sanity bound, not a table size: the reference has no cap on max_iter.  Adam's per-iteration scalars come from a
                                256-entry table in the kernel arguments; a frame that iterates longer continues them in double on the device (0.5.0;
                                earlier versions refused n_iter > 256) */

typedef enum dp_status {
    DP_OK = 0,
    DP_ERR_INVALID = -1,     /* NULL pointer, bad size, unsupported topology */
    DP_ERR_DEVICE = -2,      /* no gfx950 device / HIP runtime failure */
    DP_ERR_UNSUPPORTED = -3, /* valid request this build does not implement */
    DP_ERR_LAUNCH = -4,      /* kernel launch failed */
    DP_ERR_TIMEOUT = -5      /* dp_temporal_predict (0.5.1): a team of workgroups of an EARLIER launch of the handle gave up waiting for a member;
                                see dp_temporal_status */
} dp_status;

typedef enum dp_weight_dtype {
    DP_WEIGHTS_FP32 = 0,
    DP_WEIGHTS_BF16 = 1 /* every decoder `weight` tensor rounded to bf16 (nearest-even) before folding;
                           activations and accumulation stay fp32 (BASELINE config 5) */
} dp_weight_dtype;

typedef struct dp_ctx dp_ctx;

/* Host pointers to the reference checkpoint's decoder tensors, fp32 row-major [out][in]
 * (state_dict keys under autoencoder.decoder.*), plus statistics and skeleton. */
typedef struct dp_model {
    const float* f_latent_w; /* [24][24]  f_latent.weight */
    const float* f_latent_b; /* [24]      f_latent.bias */
    const float* unpool_w[3]; /* layers.l.0.weight : [40][24], [60][40], [92][60] */
    const float* conv_w[3];   /* layers.l.1.weight : [40][40], [60][60], [92][92] (kernel size 1) */
    const float* conv_mask[3]; /* layers.l.1.mask  : same shapes */
    const float* conv_b[3];   /* layers.l.1.bias   : [40], [60], [92] */
    const float* mean_q; /* [88] first 4 of every 8 channels of means["dqs"] (drag_pose.py:27-29) */
    const float* std_q;  /* [88] same of stds["dqs"] */
    const float* mean_disp; /* [3] */
    const float* std_disp;  /* [3] */
    const int* parents;     /* [22], parents[0] = 0, parents[j] < j */
    const float* offsets;   /* [22][3], row 0 ignored (root offset is zero, train.py:340) */
    int weight_dtype;       /* dp_weight_dtype */
} dp_model;

/* Folded decoder (three dense layers, LeakyReLU(0.2) after the first two). */
typedef struct dp_folded {
    float A0[40 * 24], c0[40];
    float A1[60 * 40], b1[60];
    float A2[92 * 60], b2[92];
} dp_folded;

/* Per-frame inputs, DEVICE pointers.  Tracked joints are given densely per joint:
 * w[b][j] = (w_pos, w_rot) and tracked[b][j] != 0 where joint j carries a tracker in frame b
 * (the reference's mask_joints / weights_joints, drag_pose.py:116-124); targets of untracked
 * joints are ignored.  E_b = number of tracked joints sets the mean denominators 3 E_b, 9 E_b. */
typedef struct dp_batch {
    int n_frames;
    const float* z0;      /* [B][24] warm-start latent (self.latent) */
    const float* z_tgt;   /* [B][24] temporal prediction (target_latent, drag_pose.py:294) */
    const float* cur_rot; /* [B][4]  current_global_rot */
    const float* tgt_pos; /* [B][22][3] target_ee_pos scattered to joint slots */
    const float* tgt_rot; /* [B][22][9] target_ee_rot (row-major 3x3) scattered to joint slots.  Must be ROTATION matrices
                             (orthonormal, det +1) -- what the reference's callers pass (eval_drag.py:186-199 from FK,
                             run_drag.py:136 from quaternions): the kernel turns each into a quaternion once per launch and
                             evaluates |R - T|_F^2 as 8 |vec(conj(q_R) q_T)|^2, which equals the reference's element-wise
                             form only for rotations.  NOT validated here (device memory, no hidden sync): the Python
                             operator checks it on request (LatentOptimizer.optimize(validate_targets=True): one device
                             reduction and one host sync, raises ValueError beyond 1e-3) */
    const float* w;       /* [B][22][2] */
    const unsigned char* tracked; /* [B][22] */
} dp_batch;

typedef struct dp_params {
    unsigned struct_size; /* sizeof(dp_params) in the caller's translation unit (DP_PARAMS_INIT sets it).  The library refuses a
                             value below the 0.5.0 size and reads no field beyond it.  A 0.4 caller's first word is n_iter: below 56 it
                             fails this test; from 56 on the second word -- its `lr` -- is read as n_iter, which is beyond DP_MAX_ITERS
                             for every learning rate above 1.4e-39 and refused before anything else of the struct is touched */
    int n_iter;        /* max_iter; exactly n_iter iterations when early_stop == 0 (1 .. DP_MAX_ITERS) */
    float lr;          /* learning_rate */
    float beta1, beta2, eps; /* torch.optim.Adam defaults 0.9, 0.999, 1e-8.  eps must be > 0 (DP_ERR_INVALID otherwise): torch accepts 0,
                                where a gradient component that is exactly 0 gives 0/0 = NaN; the kernels also carry the latent's
                                padding lanes (dims 24..31, gradient 0) through the same update, so with eps = 0 EVERY frame would */
    float lambda_rot, lambda_tmp;
    int early_stop;    /* 1: per-frame while-condition of drag_pose.py:300-304 */
    float stop_eps_pos, stop_eps_rot, min_loss_incr;
    int max_trackers;  /* ignored (kept for ABI compatibility with version 1, where it selected a kernel variant): every
                          joint of a frame may carry a tracker, whatever this says */
    int kernel;        /* DP_KERNEL_AUTO / _W4 / _W16 (below).  Added in 0.3.0: see DP_VERSION for what that means to 0.2 callers.
                          dp_optimize_sequence ignores it (whole-sequence launches are DP_KERNEL_W4's) and implies early_stop = 1 */
} dp_params;
#define DP_PARAMS_INIT {(unsigned)sizeof(dp_params)} /* dp_params p = DP_PARAMS_INIT;  (every other field zero) */

/* Two kernels implement dp_optimize (same operator, same outputs, within the tolerance stated in DESIGN.md):
 *   DP_KERNEL_W4   4 frames per wavefront, fp32 MFMA (v_mfma_f32_4x4x1): every launch shape, forward-only, whole sequences.
 *   DP_KERNEL_W16  16 frames per wavefront, decoder on v_mfma_f32_16x16x32_bf16 in split precision (every fp32 operand the
 *                  exact sum of three bf16 terms, six term products per block accumulated in fp32), fixed iteration count or
 *                  early stop; the reference's 22-joint skeleton only (DP_ERR_UNSUPPORTED otherwise).  Pays when W4 needs a THIRD round
 *                  (W4 holds 16 frames per CU at a time: 4096 on an MI355X; it runs two rounds in 0.25 ms, W16 any batch up to 16 384
 *                  frames in 0.27-0.28 ms).
 *                  (Until 0.5.1 it took n_iter <= 256 only, the kernel-argument table of Adam scalars; it continues them on the device now, as W4 does.)
 *   DP_KERNEL_AUTO W16 for more than 32 frames per CU (> 8192 on an MI355X; > 4096 before 0.5.0), either weight type, with or without early stop (both
 *                  kernels compute in fp32-equivalent arithmetic and are held to the same reference runs,
 *                  tests/test_hip_parity.py::test_full_size_batch_properties, tests/test_hip_w16.py); W4 otherwise -- BASELINE's
 *                  1024- and 4096-frame batches, every sequence launch. */
#define DP_KERNEL_AUTO 0
#define DP_KERNEL_W4 1
#define DP_KERNEL_W16 2
/* AUTO is a function of the launch's OWN frame count and the device's CU count, and the two kernels differ in the last bits of
 * their arithmetic (on frames that sit on a LeakyReLU kink: by millimetres).  A caller that cuts ONE batch
 * into several launches (shards over GPUs, chunks over time) and wants every frame's result to be independent of the cut must
 * therefore pin the kernel: ask once with dp_auto_kernel for the size that decides (dragposer_amd/sharding.py: the largest shard)
 * and pass the answer as dp_params.kernel to every launch.  Returns DP_KERNEL_W4 or DP_KERNEL_W16, or a negative dp_status. */
int dp_auto_kernel(const dp_ctx* ctx, int n_frames);

/* Outputs, DEVICE pointers; any may be NULL.  All but z are those of the LAST forward pass
 * (the latent before the final Adam step), as the reference returns them (drag_pose.py:309-312). */
typedef struct dp_result {
    unsigned struct_size; /* sizeof(dp_result) in the caller's translation unit (DP_RESULT_INIT sets it) */
    unsigned reserved0;   /* must be 0 (a pre-0.5 dp_result starts with the pointer `z`: its upper half lands here and is refused) */
    float* z;          /* [B][24] latent after the last Adam step (next frame's warm start) */
    float* z_pre;      /* [B][24] latent of the last forward pass (current_latent) */
    float* pose;       /* [B][88] decoder output: normalised-space unit quaternions, root incremental */
    float* disp;       /* [B][3]  root-space displacement, de-normalised (metres) */
    float* world_disp; /* [B][3]  */
    float* world_rot;  /* [B][4]  */
    float* pos;        /* [B][22][3] joint positions w.r.t. the previous frame's root position */
    float* rot;        /* [B][22][9] global joint rotation matrices */
    float* loss;       /* [B][3]  loss_pos, lambda_rot*loss_rot, lambda_tmp*loss_tmp */
    int* iters;        /* [B]     iterations executed */
    int* status;       /* [B]     0 or DP_STATUS_* bits (new in 0.5.0): the per-frame counterpart of the failure detection the
                                  reference does not have -- there a NaN tracker sample silently poisons self.latent for good */
    unsigned long long* clock; /* [2] (new in 0.5.0) shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) workgroup 0 spent in
                                  its iteration loop: cycles / ticks x 0.1 = the shader clock in GHz the launch ran at (the chip lowers
                                  it under full load; bench.py reports it as roofline.sclk_ghz) */
} dp_result;
#define DP_RESULT_INIT {(unsigned)sizeof(dp_result)} /* dp_result r = DP_RESULT_INIT;  (every pointer NULL) */
/* dp_result.status bits.  A frame with a non-finite input cannot be optimised (the reference's loss is NaN on its first pass, its
 * while-condition then fails -- every comparison with NaN is false --, and Adam writes NaN into self.latent: drag_pose.py:300-304,
 * 342-344).  The kernels reproduce what the reference RETURNS for such a frame and say so here; frames that share a wavefront with
 * it are not affected (tests/test_hip_status.py). */
#define DP_STATUS_NONFINITE_RESULT 1 /* the returned latent z is not finite */
#define DP_STATUS_BAD_STATE 2        /* z0 or cur_rot not finite, or beyond DP_INPUT_LIMIT in magnitude: every result of the frame is NaN */
#define DP_STATUS_BAD_TARGETS 4      /* a tracked joint's target / weight, or z_tgt, not finite or beyond DP_INPUT_LIMIT: z and loss are
                                        NaN; with early_stop (or n_iter == 1) the pose results are those of the warm start z0 after ONE
                                        pass, as the reference returns them; without, NaN */
#define DP_STATUS_TARGET_NOT_ROTATION 8 /* (0.5.1) a tracked joint's tgt_rot is not a rotation matrix (rows orthonormal within DP_ROTATION_TOL,
                                        determinant positive).  The reference's rotation loss is an element-wise MSE on ANY 3 x 3 (drag_pose.py:
                                        121-124); the kernels use its quaternion form, which equals it for rotations only -- every reference
                                        caller passes rotations (eval_drag.py:199, run_drag.py:136).  The frame is computed as given (the matrix
                                        read as the rotation Shepperd's conversion makes of it): a report, not a refusal; costs the set-up six
                                        dot products per tracker, the loop nothing, the caller no synchronisation */
#define DP_ROTATION_TOL 1.0e-3f
#define DP_INPUT_LIMIT 1.0e4f        /* metres / weight units / latent units: keeps every intermediate of the loop finite in fp32 */

int dp_version(void);
const char* dp_last_error(const dp_ctx* ctx);

/* host-only: fold the raw decoder tensors (double accumulation, fp32 result). */
int dp_fold_decoder(const dp_model* model, dp_folded* out);

int dp_create(dp_ctx** out, const dp_model* model, int device);
int dp_destroy(dp_ctx* ctx);

int dp_optimize(dp_ctx* ctx, const dp_batch* in, const dp_params* params, const dp_result* out, void* hip_stream);

/* decode + FK only; `out` fields z, z_pre, loss, iters, clock are ignored (status: DP_STATUS_BAD_STATE where z or cur_rot was refused). */
int dp_forward(dp_ctx* ctx, int n_frames, const float* z, const float* cur_rot, const dp_result* out, void* hip_stream);

/* ---- per-frame epilogue of a sequence (reference: DragPose.run, drag_pose.py:369-402,414) -------------------------
 * After dp_optimize has solved frame t of S sequences advancing in lock-step (one row of every array per sequence),
 * dp_sequence_advance applies what the reference does between the optimise loop and `return`: global position /
 * rotation update, optional joint adjustment, the three history buffers shifted by one frame with the new entry
 * appended (the state of drag_pose.py:47-64), and the returned pose with its root channels replaced by the
 * normalised global rotation.  One launch, asynchronous on the stream, no host synchronisation. */
#define DP_MAX_HEIGHT_JOINTS 8
typedef struct dp_seq_state {   /* DEVICE pointers; S sequences */
    float* global_pos;  /* [S][3]      current_global_pos (in/out) */
    float* global_rot;  /* [S][4]      current_global_rot (out: world_rot of this frame) */
    float* latent_buf;  /* [S][H][24]  latent_buffer       (in/out, newest entry last) */
    float* disp_buf;    /* [S][H][3]   displacement_buffer (in/out) */
    float* heights_buf; /* [S][H][NH]  heights_buffer      (in/out) */
    int history;        /* H (the reference keeps 60 frames) */
    int n_heights;      /* NH <= DP_MAX_HEIGHT_JOINTS */
    int height_joints[DP_MAX_HEIGHT_JOINTS]; /* height_indices (drag_pose.py:393) */
} dp_seq_state;

typedef struct dp_seq_step {
    int adjust_joint;        /* joint_adjustment_indices[0], or -1: no joint adjustment (drag_pose.py:377-384) */
    int adjust_target_joint; /* the joint whose tracker row joint_adjustment_indices[1] selects (mask_joints[ee_index]) */
    float adjust_weight;     /* joint_adjustment_weight */
    const float* tgt_pos;    /* [S][22][3] this frame's dense targets (read for the adjustment only; may be NULL when off) */
    float* pose_ret;         /* [S][88] returned pose (drag_pose.py:399-402,414); may be NULL */
    float* pos_ret;          /* [S][3]  returned global position; may be NULL */
} dp_seq_step;

/* `res` = the dp_result dp_optimize filled for this frame: z_pre, pose, disp, world_disp, world_rot, pos are read. */
int dp_sequence_advance(dp_ctx* ctx, int n_sequences, const dp_result* res, const dp_seq_state* state, const dp_seq_step* step,
                        void* hip_stream);

/* ---- whole-sequence launches ------------------------------------------------------------------------------------------
 * n_steps consecutive frames of S sequences in ONE launch: per step the optimise loop with the reference's while-condition
 * (dp_params.early_stop is implied), then run()'s epilogue (what dp_sequence_advance does), the state carried from step to step
 * on the device.  The caller knows the targets of all steps up front (eval_drag.py:164-222 builds them from the ground-truth
 * file); what it cannot know is the running global position they are relative to -- hence tgt_root.  A second, tiny launch
 * appends the steps to the three history buffers.  The temporal predictor is not part of it: callers with a predictor cut the
 * sequence at the frames where it is due (every `window` frames) and call dp_temporal_predict in between. */
typedef struct dp_seq_frames {   /* DEVICE pointers; T = n_steps */
    int n_steps;
    const float* tgt_pos;  /* [T][S][22][3] position targets (see tgt_root) */
    const float* tgt_rot;  /* [T][S][22][9] rotation targets */
    const float* tgt_root; /* [T][S][3] or NULL.  Given: the position targets of step t are tgt_pos[t] + (tgt_root[t] - the
                              sequence's global position before step t), i.e. tgt_pos holds root-relative positions and tgt_root
                              the target root trajectory (eval_drag.py:186-199); NULL: tgt_pos is used as it is */
    const float* w;        /* [S][22][2] tracker weights, the same for every step */
    const unsigned char* tracked; /* [S][22] */
    const float* z_tgt;    /* temporal targets: element k of sequence s at step t = z_tgt[t * z_tgt_step + s * z_tgt_seq + k] */
    int z_tgt_step, z_tgt_seq; /* (one row held for all steps: z_tgt_step = 0; rows of a [S][window + 1][24] target buffer: 24, (window + 1) * 24) */
} dp_seq_frames;

typedef struct dp_seq_results {  /* DEVICE pointers, per step; any but hist_scratch may be NULL */
    unsigned struct_size; /* sizeof(dp_seq_results) as the caller compiled it (DP_SEQ_RESULTS_INIT), checked like dp_result's */
    unsigned reserved0;   /* must be 0 */
    float* pose_ret;     /* [T][S][88] what run() returns (root channels = the normalised world rotation) */
    float* pos_ret;      /* [T][S][3]  returned global position */
    float* world_rot;    /* [T][S][4]  global rotation after the step (= the state's global_rot then) */
    int* iters;          /* [T][S]     iterations executed */
    float* loss;         /* [T][S][3]  losses of the frame's last executed iteration */
    float* hist_scratch; /* [T][S][24 + 3 + NH] floats of caller-owned scratch (the steps' history rows before they are appended) */
    int* status;         /* [T][S]     DP_STATUS_* bits of every step (new in 0.5.0).  A sequence whose step t had a bad target returns that step's
                                       pose from its warm start (one pass) and is DP_STATUS_BAD_STATE from step t + 1 on -- the reference's latent
                                       is NaN from there (drag_pose.py:342-344); the other sequences of the launch are not affected */
} dp_seq_results;
#define DP_SEQ_RESULTS_INIT {(unsigned)sizeof(dp_seq_results)}

/* latent [S][24]: in = the warm start of the first step, out = the latent after the last.  state: global_pos / global_rot
 * in and out, the three history buffers advanced by n_steps.  adjust: the joint-adjustment fields of dp_seq_step (its pointers
 * are ignored).  Asynchronous on the stream, no allocation. */
int dp_optimize_sequence(dp_ctx* ctx, int n_sequences, float* latent, const dp_seq_frames* frames, const dp_params* params,
                         const dp_seq_state* state, const dp_seq_step* adjust, const dp_seq_results* out, void* hip_stream);

/* ---- temporal predictor (reference: Temporal, temporal_transformer.py:7-77, positional_encoding.py:6-32; its use in
 * DragPose.run, drag_pose.py:234-294) -----------------------------------------------------------------------------
 * The Transformer that predicts the latents of the next frames from the 60-frame history; its output is `z_tgt`, the
 * anchor of the lambda_temporal term of dp_optimize.  torch.nn.Transformer semantics (post-norm layers, ReLU, final
 * encoder / decoder LayerNorm, eps 1e-5, no masks, dropout off), fp32, one workgroup per sequence.
 * All pointers of dp_temporal_model are HOST pointers to the tensors of the reference's state_dict, row-major as
 * PyTorch stores them (Linear.weight = [out][in]). */
#define DP_TEMPORAL_MAX_LAYERS 8
#define DP_TEMPORAL_D_MODEL 48  /* features_transformer = 2 * latent_dim (train_temporal.py:26) */
#define DP_TEMPORAL_HEADS 4     /* train_temporal.py:27 */
#define DP_TEMPORAL_MAX_TOKENS 32 /* encoder tokens (ceil(H / sample_step) - 1) and decoder tokens (window / sample_step + 1) */
typedef struct dp_temporal_layer {
    const float *sa_in_w, *sa_in_b;   /* self_attn.in_proj_weight [144][48], in_proj_bias [144] */
    const float *sa_out_w, *sa_out_b; /* self_attn.out_proj.weight [48][48], bias [48] */
    const float *ca_in_w, *ca_in_b;   /* multihead_attn.* (decoder layers; NULL in encoder layers) */
    const float *ca_out_w, *ca_out_b;
    const float *lin1_w, *lin1_b;     /* linear1.weight [F][48], bias [F] */
    const float *lin2_w, *lin2_b;     /* linear2.weight [48][F], bias [48] */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [48] each */
    const float *norm3_w, *norm3_b;   /* decoder layers; NULL in encoder layers */
} dp_temporal_layer;
typedef struct dp_temporal_model {
    int n_heights;             /* heights per token (6: len(height_indices)) */
    int dim_feedforward;       /* F (train_temporal.py:30: 2048) */
    int n_encoder_layers, n_decoder_layers; /* <= DP_TEMPORAL_MAX_LAYERS */
    int max_len;               /* rows of pos_encoding */
    int sample_step;           /* train_temporal.py:15: 4 */
    const float *in_proj_encoder_w, *in_proj_encoder_b; /* [48][24 + 3 + n_heights], [48] */
    const float *in_proj_decoder_w, *in_proj_decoder_b; /* [48][24], [48] */
    const float *out_proj_w, *out_proj_b;               /* [24][48], [24] */
    const float* pos_encoding;                          /* [max_len][48] (positional_encoding.pos_encoding) */
    const float *enc_norm_w, *enc_norm_b, *dec_norm_w, *dec_norm_b; /* temporal.encoder.norm / decoder.norm */
    const float *means_latent, *stds_latent;            /* [24] (temporal.pt: means_latent / stds_latent) */
    const dp_temporal_layer* enc; /* [n_encoder_layers] */
    const dp_temporal_layer* dec; /* [n_decoder_layers] */
} dp_temporal_model;
typedef struct dp_temporal dp_temporal;
int dp_temporal_create(dp_temporal** out, const dp_temporal_model* model, int device);
int dp_temporal_destroy(dp_temporal* t);
const char* dp_temporal_last_error(const dp_temporal* t); /* NULL: last failure of dp_temporal_create on this thread */
/* The temporal target block of one frame step (drag_pose.py:248-292), for S sequences: tokens from the history buffers of
 * `state` (latent normalised, displacement accumulated over sample_step frames, heights), window / sample_step + 1
 * autoregressive calls of the Transformer, de-normalisation and the reference's step-hold "lerp".
 * target_buf (DEVICE) [S][window + 1][24]: row current_index of it is the frame's z_tgt.  Asynchronous on the stream.
 * One launch of a handle at a time (launches on one stream are; two streams sharing a handle must order themselves): up to half as many
 * sequences as the device has CUs, the workgroups that share a sequence exchange partial sums through memory the handle owns.  Such a launch
 * may be captured into a graph and replayed (the exchange keeps its counters on the device). */
int dp_temporal_predict(dp_temporal* t, int n_sequences, const dp_seq_state* state, int window, float* target_buf, void* hip_stream);
/* Health of a handle, read WITHOUT synchronising anything (0.5.1).  Launches of few sequences run TEAMS of workgroups per sequence whose members
 * wait for each other's partial sums, so every member must be resident on a CU.  The library sizes teams for that -- one workgroup per CU by the
 * runtime's occupancy query, all teams of a launch on at most half of the CUs the stream may use -- but cannot see what else the device is
 * running (another stream's long kernel, a second process, a CU mask set later).  If a member still waits after ~1 s it GIVES UP, and that is never
 * silent: (1) every target row of its sequence is NaN -- dp_optimize / dp_optimize_sequence then report DP_STATUS_BAD_TARGETS for the sequence
 * instead of pulling it towards garbage; (2) this word carries DP_TEMPORAL_TEAM_TIMEOUT from that moment on (the device writes it into page-locked
 * host memory; sticky); (3) team launches of the handle already queued or replayed from a graph write NaN and leave at once; (4) the NEXT
 * dp_temporal_predict launches nothing and returns DP_ERR_TIMEOUT (message: dp_temporal_last_error) -- from then on the handle runs one workgroup
 * per sequence, so calling again works. */
#define DP_TEMPORAL_TEAM_TIMEOUT 1
int dp_temporal_status(const dp_temporal* t); /* >= 0: DP_TEMPORAL_* bits; DP_ERR_INVALID for NULL */

/* Device-buffer helpers for callers that have no HIP binding of their own (the native Unity drop-in,
 * include/dragposer_unity.h).  Thin wrappers over hipMalloc / hipFree / hipMemcpyAsync / hipStreamSynchronize on the
 * context's device; PyTorch callers never need them. */
int dp_io_alloc(dp_ctx* ctx, unsigned long long bytes, void** dev_ptr);
int dp_io_free(dp_ctx* ctx, void* dev_ptr);
int dp_io_upload(dp_ctx* ctx, void* dev_dst, const void* host_src, unsigned long long bytes, void* hip_stream);
int dp_io_download(dp_ctx* ctx, void* host_dst, const void* dev_src, unsigned long long bytes, void* hip_stream);
int dp_stream_sync(dp_ctx* ctx, void* hip_stream);
/* page-locked HOST staging memory for dp_io_upload / dp_io_download (hipHostMalloc / hipHostFree): a copy from or into pageable
 * memory goes through the runtime's own staging buffer and blocks the caller; from pinned memory it is one asynchronous DMA.  New in
 * 0.4.0; the native Unity drop-in stages every frame's inputs and results through such buffers. */
int dp_io_alloc_host(dp_ctx* ctx, unsigned long long bytes, void** host_ptr);
int dp_io_free_host(dp_ctx* ctx, void* host_ptr);

/* introspection for the benchmark: frames per workgroup, workgroup size and LDS bytes of the kernel the context's
 * last dp_optimize / dp_forward launch used */
int dp_kernel_geometry(const dp_ctx* ctx, int* frames_per_block, int* threads_per_block, int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DRAGPOSER_H */

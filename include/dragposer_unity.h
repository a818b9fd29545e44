/* dragposer_unity.h -- native drop-in for the reference's Unity plugin ABI.
 *
 * The reference's DragPoserDLL (DragPoserDLL/exportFunc.h:61-70, exportFunc.cpp) embeds CPython and forwards
 * ten extern "C" functions to python/src/run_drag.py::RunDrag.  libDragPoserDLL.so exports the SAME ten
 * functions with the same argument lists and POD layouts (DragPoserDLL/utils.h:13-41; Unity side:
 * DragPoserUnity/Assets/Scripts/Core/DragPoserDLL.cs:10-29), but runs them natively: BVH skeleton parsing,
 * the encoder for the initial latent and the per-frame epilogue in C++, the optimise loop on the MI355X through
 * the dp_* ABI (include/dragposer.h).  No Python, no PyTorch in the process.
 *
 * Differences, all deliberate and reported through drag_poser_last_error():
 *   - load_models(modelPath) reads <modelPath>/dragposer_model.bin (flat export of generator.pt + data.pt,
 *     written by tools/export_model_bin.py) instead of un-pickling generator.pt / temporal.pt;
 *   - the temporal Transformer runs natively (dp_temporal_* of include/dragposer.h) when <modelPath>/temporal.bin is
 *     present (tools/export_temporal_bin.py writes it from a temporal.pt state_dict); without that file -- the
 *     reference's temporal.pt is not distributed with it -- a non-zero lambdaTemporal is accepted, the pull term stays
 *     off and drag_poser_last_error() says so;
 *   - functions that can fail keep the reference's void signatures; the message of the last failure is
 *     available from drag_poser_last_error(handle) (an addition, the reference only logs to a file).
 * Conventions are the reference's: quaternions (w,x,y,z); caller-owned buffers; result_pose holds
 * PARENT-LOCAL rotations (run_drag.py:161-176); one handle is not thread-safe.
 */
#ifndef DRAGPOSER_UNITY_H
#define DRAGPOSER_UNITY_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dp_quaternion { float w, x, y, z; } dp_quaternion; /* utils.h:13-22 `quaternion` */
typedef struct dp_float3 { float x, y, z; } dp_float3;            /* utils.h:24-32 `float3` */
typedef struct dp_float2 { float x, y; } dp_float2;               /* utils.h:34-41 `float2` */
typedef struct DragPoser DragPoser;

DragPoser* init_drag_poser(void);                                                    /* exportFunc.h:61 */
void set_reference_skeleton(DragPoser* dragPoser, char* bvhPath);                    /* :62 */
void load_models(DragPoser* dragPoser, char* modelPath);                             /* :63 */
void set_mask_and_weights(DragPoser* dragPoser, float* mask, dp_float2* weights);    /* :64 */
void init_drag_model(DragPoser* dragPoser, dp_float3 initialGlobalPos, dp_quaternion initialGlobalRot); /* :65 */
void set_optim_params(DragPoser* dragPoser, float stopEpsPos, float stopEpsRot, int maxIter, float lr); /* :66 */
void set_lambdas(DragPoser* dragPoser, float lambdaRot, float lambdaTemporal, int temporalFutureWindow); /* :67 */
void set_global_pos(DragPoser* dragPoser, dp_float3 globalPos);                      /* :68 */
void drag_pose(DragPoser* dragPoser, int nEndEffectors, dp_float3* targetEEPos, dp_quaternion* targetEERot,
               dp_quaternion* resultPose, dp_float3* resultGlobalPos);               /* :69 */
void destroy_drag_poser(DragPoser* dragPoser);                                       /* :70 */

/* additions */
const char* drag_poser_last_error(const DragPoser* dragPoser); /* "" when the last call succeeded */
int drag_poser_last_iterations(const DragPoser* dragPoser);    /* optimiser iterations of the last drag_pose */
void drag_poser_get_latent(const DragPoser* dragPoser, float* latent24); /* the warm-start latent (tests, checkpointing) */
void drag_poser_set_latent(DragPoser* dragPoser, const float* latent24); /* also refills the latent history with it (drag_pose.py:50-55) */
int drag_poser_has_temporal(const DragPoser* dragPoser); /* 1: <modelPath>/temporal.bin was loaded -- lambdaTemporal takes effect
                                                            (the Transformer of temporal_transformer.py on the GPU, dp_temporal_*) */

#ifdef __cplusplus
}
#endif
#endif

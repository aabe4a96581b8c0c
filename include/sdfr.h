/*
 * sdfr.h -- C ABI of libsdfr_hip.so: the MI355X (gfx950) implementation of
 * sdfest's differentiable-renderer hot path.
 *
 * This is the drop-in boundary.  It replaces the two entry points of the
 * reference's pybind module `sdf_renderer_cpp`
 *     forward (sdfest/differentiable_renderer/csrc/sdf_renderer.cpp:42-61)
 *     backward(sdfest/differentiable_renderer/csrc/sdf_renderer.cpp:63-86)
 * and, for the render-and-compare loop around them, the torch-op
 * implementations of
 *     losses.pc_loss          (sdfest/estimation/losses.py:32-135)
 *     SDFDecoder.forward      (sdfest/vae/sdf_vae.py:217-259)
 *
 * Rules of the boundary
 *   - plain C, no torch types.  Every pointer is a DEVICE pointer owned by the
 *     caller unless its name starts with `h_`.  All tensors are fp32,
 *     C-contiguous.
 *   - the library allocates nothing and never synchronises the host: all work
 *     is enqueued on `stream` (a hipStream_t; NULL = the default stream), so a
 *     call sequence can be captured into a hipGraph.  Scratch memory comes from
 *     the caller (`*_workspace_bytes`).
 *   - re-entrant and thread-safe: the reference's backward runs on a PyTorch
 *     autograd worker thread; every call does hipSetDevice(device) itself
 *     (reference: OptionalCUDAGuard, sdf_renderer.cpp:57-58, :82).
 *   - return 0 on success, a negative SDFR_E_* for argument errors, or a
 *     positive hipError_t.  sdfr_last_error() gives the thread's last message.
 *
 * Data layout
 *   sdf      [R][R][R]   index order sdf[x][y][z]
 *   pos      [B][3]      object position in the camera frame (OpenGL: -z forward)
 *   quat     [B][4]      object orientation, (x, y, z, w) scalar-last, unit norm
 *   inv_scale[B]         1 / half-width of the SDF volume
 *   depth    [B][H][W]   row 0 = top; 0 = no hit
 *   cx, cy               principal point in the pixel-centre-0.5 convention
 *                        (Camera.get_pinhole_camera_parameters(0.5),
 *                        sdf_renderer.py:116-133)
 *   B views share one SDF (sdf_view_stride = 0) or have one each
 *   (sdf_view_stride = R*R*R elements).
 *
 * STABILITY -- group 1 below is the boundary: its signatures and semantics are what a binding relies on and do not change
 * without SDFR_VERSION's major number changing.  Groups 2 - 4 are UNSTABLE: they exist for this repository's own host
 * code (the Python modules under sdfest_amd/), follow its needs from round to round (arguments were added in every round so far), and are
 * exported only because that host code is Python over ctypes; bind to them at your own risk, pinned to one SDFR_VERSION.
 *
 * CONTENTS -- four groups; a binding from another language needs group 1 only
 *   1. CORE: the reference boundary (what sdf_renderer_cpp, losses.pc_loss and SDFDecoder.forward are replaced by)
 *        sdfr_version, sdfr_last_error
 *        sdfr_render_forward[_workspace_bytes], sdfr_render_backward[_workspace_bytes]
 *        sdfr_pc_loss_forward, sdfr_pc_loss_backward[_workspace_bytes]
 *        sdfr_decoder_create / _destroy / _forward / _workspace_bytes / _tape_bytes,
 *        sdfr_decoder_backward_latent[_workspace_bytes], sdfr_decoder_set_option
 *   2. [unstable] BATCHED / STEP forms of the same arithmetic (fewer launches, fewer bytes; same results)
 *        sdfr_render_step_forward[_counted] / _step_backward / _step_workspace_bytes, sdfr_render_sync_offset,
 *        sdfr_render_partials_offset, sdfr_render_fixed_volume_offset, sdfr_fixed_to_float
 *        sdfr_render_forward_l1[_workspace_bytes], sdfr_render_backward_l1, sdfr_render_step_forward_l1,
 *        sdfr_render_step_backward_l1, sdfr_render_backward_l1_pc, sdfr_render_step_backward_l1_pc,
 *        sdfr_render_step_fused_l1_pc, sdfr_render_fused_view_count_offset, sdfr_render_fused_tile_loss_offset
 *        sdfr_pc_l1_backward[_accumulate]
 *   3. [unstable] LOOP: one render-and-compare iteration (SDFPipeline.__call__) as a fixed launch sequence
 *        sdfr_preprocess_depth, sdfr_depth_to_points_resident, sdfr_depth_count[_ordered|_centroid],
 *        sdfr_depth_to_points[_ordered|_shifted], sdfr_depth_points_workspace_bytes, sdfr_depth_centroid_workspace_bytes
 *        sdfr_pose_to_views[_objects], sdfr_views_to_pose_grad[_deferred], sdfr_decoder_backward_latent_deferred[_batch|_scaled], sdfr_decoder_forward_stage,
 *        sdfr_loop_tail, sdfr_loop_tail_fused, sdfr_loop_tail_objects, sdfr_adam_step, sdfr_point_constraint, sdfr_add_inplace
 *        sdfr_depth_l1_loss[_workspace_bytes], sdfr_pc_l1_loss, sdfr_inlier_ratio, sdfr_nn_loss_forward / _backward
 *        sharded over ranks: sdfr_loop_view_records, sdfr_loop_tail_records, sdfr_inlier_counts_record,
 *        sdfr_inlier_update_record
 *   4. [unstable] GENERATOR / INITIALISATION (the forward-only callers around the loop)
 *        sdfr_affine_mask; sdfr_pointnet_layer[_counted], sdfr_linear_vec, sdfr_orientation_posterior,
 *        sdfr_init_estimate
 * (Within the file the groups follow the order in which the reference's code runs; every declaration carries the
 * reference file:line it replaces.)
 */
#ifndef SDFR_H_
#define SDFR_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFR_VERSION 100 /* 0.1.0 */
#define SDFR_API __attribute__((visibility("default")))

#define SDFR_E_INVALID (-1)   /* bad shape / size / flag */
#define SDFR_E_NULL (-2)      /* required pointer is NULL */
#define SDFR_E_WORKSPACE (-3) /* workspace too small */

/* ray-march safety cap.  The reference loop (sdf_renderer_cuda.cu:283-293) is
 * unbounded and spins forever for threshold == 0 on an exactly-zero sample; a
 * ray that needs more evaluations than this is reported as a miss (depth 0). */
#define SDFR_MAX_MARCH_STEPS 4096

/* d depth / d sdf weight assignment */
#define SDFR_SDF_GRAD_EXACT 0       /* trilinear weights (simple_renderer.py:399-408) */
#define SDFR_SDF_GRAD_CUDA_COMPAT 1 /* the permutation in sdf_renderer_cuda.cu:373-388 */
/* Flag bit, OR-ed into either of the above: DETERMINISTIC d depth / d sdf.  The reference
 * sums the per-pixel contributions with float atomics (sdf_renderer_cuda.cu:373-388) and so does the default mode
 * here, one per touched voxel per tile: the result varies in the last bits from run to run.  With this flag every
 * contribution is rounded once, per pixel, to the fixed quantum 2^-SDFR_FIXED_QUANTUM_BITS and everything after
 * that is INTEGER addition into a 64-bit volume in the workspace, converted to float at the end: g_sdf is bitwise
 * identical from run to run and does not depend on tile shapes, batch composition or -- when the ranks of a
 * sharded batch add their 64-bit volumes (sdfr_render_fixed_volume_offset) and convert afterwards
 * (sdfr_fixed_to_float) -- on how the views are spread over GPUs.  For tests and reproducible runs: several
 * times slower than the default (64-bit LDS table, no dense box).  Shared gradient volume only
 * (g_sdf_view_stride = 0), R <= 128; every backward form takes it (the loss-fused ones and the sampler's blocks of
 * sdfr_render_backward_l1_pc add into the same 64-bit volume).  Representable range:
 * |sum| < 2^(63 - 40) = 8.4e6 per voxel; contributions that are not finite count as 0 (NaN) or saturate. */
#define SDFR_SDF_GRAD_DETERMINISTIC 0x100
#define SDFR_FIXED_QUANTUM_BITS 40
/* Flag bit, OR-ed into sdf_grad_mode: a PERFORMANCE HINT for batch launches of sdfr_render_backward /
 * sdfr_render_step_backward, "the views are close": (nearly) every object spans at least two pixels per voxel on
 * the screen,
 *     sqrt(|fx fy|) * (2 / (R - 1)) / (inv_scale * |pos|) >= 2.
 * The kernel picks each view's tile shape on the device (32 x 32 pixels for such views, 64 x 8 otherwise) and the
 * launch has to provide workgroups for the finer tiling although close views use half of them; with the hint the
 * grid has half the rows (backward of the benchmark 125 -> 112 us) and a view that is NOT close takes its tiles two
 * per workgroup, one after the other -- results are the same whether the hint is true or not, but such views are
 * slower with it (objects of ~1 pixel per voxel: 103 -> 174 us).  The poses live in device memory: a caller that
 * does not know them asks the forward to count the close views (sdfr_render_step_forward_counted) and sets the
 * hint from an earlier step's count.  Ignored for small calls, for the deterministic forms and where the sampler's
 * blocks share the launch (sdfr_render_*backward_l1_pc). */
#define SDFR_BWD_HALF_GRID 0x200
/* Flag bit, OR-ed into sdf_grad_mode: the backward always uses its 32 x 8 tiling, whatever the batch size.  A
 * view's pose gradients are fixed-order sums over ITS tiles, so with this flag they do not depend on how many
 * other views share the launch: a batch sharded over ranks (sdfest_amd.pipeline, the sharded render-and-compare
 * loop) then gets, together with SDFR_SDF_GRAD_DETERMINISTIC, bitwise the single-process gradients.  Slower for
 * large batches (more, smaller tiles).  Pass the same flag to sdfr_loop_view_records. */
#define SDFR_BWD_SMALL_TILES 0x400

SDFR_API int sdfr_version(void);
SDFR_API const char* sdfr_last_error(void);

/* ==== 1. CORE ================================================================================= */
/* ---- sphere-tracing depth render -------------------------------------------------------- */

/* Scratch for sdfr_render_forward (per-view set-up records + the re-packed grid, see
 * render.hip "cell records").
 * Every renderer workspace (forward, backward, step, depth-L1 forms) starts with the B view records and,
 * at byte sdfr_render_sync_offset(B), a small sync region that only the forward's prologue launch uses:
 *   word 0 (uint32)  epoch of the prologue's intra-launch hand-off of the grid's plane minima
 *   word 1 (uint32)  count of view set-ups that did NOT receive the plane minima in time and fell back to the
 *                    whole cube as their may-hit box (same depth, slower march); it only ever grows, so a caller
 *                    that zero-fills the workspace once can read "how often did that happen" at any time.
 *   words 6, 7       {SDFR_SYNC_POLLS_MAGIC, rounds}: the caller's bound of the prologue's wait for the plane minima,
 *                    in polling rounds, for the forwards on THIS workspace (default 65536; 0: every view set-up takes
 *                    the fall-back path -- tests reach it this way).  Any other word 6 -- a zero-filled or an
 *                    uninitialised workspace -- means the default.  The library only reads them.
 * No call writes into another call's part of a shared workspace's sync region. */
#define SDFR_SYNC_POLLS_MAGIC 0x504F4C4Cu
SDFR_API size_t sdfr_render_forward_workspace_bytes(int R, int B, int W, int H);
SDFR_API size_t sdfr_render_sync_offset(int B);

/* Deterministic mode: byte offset, in the workspace of sdfr_render_backward (step_layout = 0) or of
 * sdfr_render_step_backward (1), of the int64 [R][R][R] volume the last deterministic backward left:
 * g_sdf = volume * 2^-SDFR_FIXED_QUANTUM_BITS.  sdfr_fixed_to_float converts n such words (after, e.g., an
 * integer all-reduce over the ranks of a sharded batch). */
SDFR_API size_t sdfr_render_fixed_volume_offset(int R, int B, int W, int H, int step_layout);
SDFR_API int sdfr_fixed_to_float(const long long* fixed, size_t n, float* out, int device, void* stream);

/* Replaces sdf_renderer_cpp.forward (sdf_renderer.cpp:42-61 ->
 * sdf_renderer_cuda.cu:472-510, kernel :241-298), extended by a leading batch
 * dimension: view b of the output equals one reference call with pose b.
 * `depth` is fully overwritten (the reference zero-fills with torch::zeros). */
SDFR_API int sdfr_render_forward(const float* sdf, int R, long long sdf_view_stride,
                        const float* pos, const float* quat, const float* inv_scale, int B,
                        int W, int H, float cx, float cy, float fx, float fy, float threshold,
                        float* depth, void* workspace, size_t workspace_bytes, int device,
                        void* stream);

/* Scratch for sdfr_render_backward (set-up records + per-tile partial sums). */
SDFR_API size_t sdfr_render_backward_workspace_bytes(int R, int B, int W, int H);

/* Replaces sdf_renderer_cpp.backward (sdf_renderer.cpp:63-86 ->
 * sdf_renderer_cuda.cu:512-556, kernel :300-468), batched like the forward.
 *   depth        the forward's output for the SAME poses (pixels with depth != 0
 *                are differentiated, :334)
 *   g_sdf        overwritten.  g_sdf_view_stride = 0: [R][R][R], the sum over
 *                the B views (what autograd would accumulate for a shared SDF);
 *                R*R*R: one gradient volume per view.
 *   g_pos [B][3], g_quat [B][4] (x,y,z,w), g_inv_scale [B]: overwritten.
 * Pose gradients are reduced in a fixed order (bitwise reproducible); g_sdf uses
 * float atomics across workgroups (last-bit run-to-run variation, as in the
 * reference) unless sdf_grad_mode carries SDFR_SDF_GRAD_DETERMINISTIC. */
SDFR_API int sdfr_render_backward(const float* grad_depth, const float* depth, const float* sdf, int R,
                         long long sdf_view_stride, const float* pos, const float* quat,
                         const float* inv_scale, int B, int W, int H, float cx, float cy,
                         float fx, float fy, int sdf_grad_mode, float* g_sdf,
                         long long g_sdf_view_stride, float* g_pos, float* g_quat,
                         float* g_inv_scale, void* workspace, size_t workspace_bytes, int device,
                         void* stream);

/* ==== 2. BATCHED / STEP forms ================================================================= */
/* ---- one step = forward + backward of the SAME views ---------------------------------------- */

/* The reference's render-and-compare loop always runs the two halves as a pair:
 * SDFRendererFunctionGPU.forward saves (image, sdf, pos, quat, inv_scale) and .backward gets them back
 * (sdf_renderer.py:311-331, :334-357).  As two independent calls the backward repeats work the forward
 * has done (its own view set-up, with the whole cube's screen rectangle because it does not know the
 * threshold) and opens with a launch of its own (zero fill of g_sdf + set-up).  A step keeps the pair's
 * state in ONE workspace:
 *   sdfr_render_step_forward   = sdfr_render_forward; its prologue launch also zero-fills `g_sdf`
 *                                (g_sdf_view_stride as in sdfr_render_backward);
 *   sdfr_render_step_backward  = sdfr_render_backward for the views, camera and image size of the
 *                                preceding sdfr_render_step_forward on the same workspace: no prologue
 *                                launch, tiles culled with the forward's may-hit rectangles.  `depth` must
 *                                be that forward's output (it is 0 outside those rectangles), g_sdf the
 *                                volume it zero-filled.  pos / quat / inv_scale are not passed again.
 * Results equal the two stand-alone calls: depth bit for bit; pose gradients up to the rounding of their
 * fixed-order tile sums (the order follows the view's rectangle; still bitwise reproducible run to run),
 * g_sdf up to the order of its float atomics.  Nothing else may use the workspace between the two calls. */
SDFR_API size_t sdfr_render_step_workspace_bytes(int R, int B, int W, int H);
SDFR_API int sdfr_render_step_forward(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                             const float* quat, const float* inv_scale, int B, int W, int H, float cx,
                             float cy, float fx, float fy, float threshold, float* depth, float* g_sdf,
                             long long g_sdf_view_stride, void* workspace, size_t workspace_bytes,
                             int device, void* stream);
/* sdfr_render_step_forward that also REPORTS how many of its views are close -- the fact behind the
 * SDFR_BWD_HALF_GRID hint -- without anybody looking at the poses on the host: the forward counts, on the device,
 * the views whose set-up gave them 32 x 32 backward tiles and, when the last view has been counted, stores
 *     (forwards counted on this workspace << 32) | close views of this forward
 * into *close_views_word with one 64-bit store.  The word must be visible to the device AND readable by the host
 * without a synchronisation (pinned host memory: hipHostMalloc / a pinned torch tensor; 8-byte aligned; NULL: no
 * report).  The caller reads it whenever it likes -- it then holds the count of some earlier, complete forward --
 * and sets the hint of its next backward from it: stale by a step or more is fine for a hint that cannot change
 * results.  Counted by batch launches (the forward's macro-tile form: smaller calls have no batch backward to
 * hint) -- by the image kernel, one atomic per view, off the critical path; the workspace's sync region must have
 * been zero-filled once before its first use.
 * Sync header words 4 / 5 hold the same two numbers on the device. */
SDFR_API int sdfr_render_step_forward_counted(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                                     const float* quat, const float* inv_scale, int B, int W, int H, float cx,
                                     float cy, float fx, float fy, float threshold, float* depth, float* g_sdf,
                                     long long g_sdf_view_stride, void* workspace, size_t workspace_bytes,
                                     unsigned long long* close_views_word, int device, void* stream);
/* The same pair for the loss-fused forms (below): sdfr_render_step_forward_l1 = sdfr_render_forward_l1 that also
 * zero-fills g_sdf and leaves the view records; sdfr_render_step_backward_l1_pc = sdfr_render_backward_l1_pc without
 * its prologue launch.  For a few views of the plain grid the forward has no prologue launch either (every
 * workgroup derives its view's record): the captured loop's iteration loses two launches.  close_views_word
 * (nullable): as in sdfr_render_step_forward_counted -- the half-grid hint of sdfr_render_step_backward_l1. */
SDFR_API int sdfr_render_step_forward_l1(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                                const float* quat, const float* inv_scale, int B, int W, int H, float cx, float cy,
                                float fx, float fy, float threshold, const float* target, float* depth, float* loss,
                                float* loss_stats, float* g_sdf, long long g_sdf_view_stride, void* workspace,
                                size_t workspace_bytes, unsigned long long* close_views_word, int device,
                                void* stream);
SDFR_API int sdfr_render_step_backward_l1_pc(
    const float* loss_grad, float loss_weight, const float* loss_stats, const float* target, const float* depth,
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    int B, int W, int H, float cx, float cy, float fx, float fy, int sdf_grad_mode, float* g_sdf,
    long long g_sdf_view_stride, void* workspace, size_t workspace_bytes, float pc_weight, const float* points,
    const int* offsets, int max_view_points, const float* scale, void* pc_workspace, size_t pc_workspace_bytes,
    float* loss, float* loss_stats_out, int device, void* stream);
/* ONE LAUNCH for the whole loss-fused step of 1 .. 8 views of the plain grid (sdfr_render_step_forward_l1 with the
 * loss deferred + sdfr_render_step_backward_l1_pc): every image tile marches its rays and, while the depths are still
 * in registers, runs the backward of its hit pixels; the sampler's blocks run beside the tiles as before.  The depth
 * loss is a mean over a count that no tile knows before the launch ends, so the depth term is left UNSCALED, with
 * the bare sign of (estimate - observation) as its upstream gradient:
 *   g_depth  float [B][R^3]   d/dSDF of view b's depth term, unscaled, ADDED to (every view has its own k)
 *   g_sdf    float [R^3]      the point-cloud term of all views (its scale is known), ADDED to
 *     (both NULL: nobody wants d/dSDF -- the tiles and the sampler's blocks then skip it)
 *   workspace (sdfr_render_step_workspace_bytes; offsets by the functions named)
 *     sdfr_render_partials_offset(R, B, W, H, 1)       the tiles' pose sums, unscaled (32 x 8-pixel tiles)
 *     sdfr_render_fused_view_count_offset(B, H)        float [B]     the views' overlap counts, ADDED to (integers
 *                                                      below 2^24 held in floats: exact in any order)
 *     sdfr_render_fused_tile_loss_offset(R, B, W, H)   (sum |est - obs|, count) per tile, 8 floats apart
 * Nothing is zero-filled by this call: whoever consumes a sum clears it (sdfr_decoder_backward_latent_deferred_scaled
 * forms  g_sdf + sum_b k_b g_depth[b],  k_b = weight / count_b, and clears the volumes; sdfr_loop_tail_fused multiplies
 * the pose sums and resets the counts) -- the volumes and the counts must be zero before the FIRST call.
 * depth equals sdfr_render_step_forward_l1's bit for bit; the gradients equal the two launches' up to rounding (k
 * multiplies sums instead of terms).  A view whose observed point set is small (<= 6144 points, tuning.hpp) sends its
 * d/dSDF to the volumes by float atomics directly (no LDS pre-sum: what a small object costs is the depth of a tile's
 * chain of phases); larger ones, and the sampler's blocks of 3 and more views (they share one volume), pre-sum in their
 * LDS tables as the two-launch form does.  sdf_grad_mode: SDFR_SDF_GRAD_EXACT or _CUDA_COMPAT, no flags.  R <= 128.
 * pos / quat / inv_scale / scale: the views' poses (scale = 1 / inv_scale, what the sampler takes). */
SDFR_API size_t sdfr_render_fused_view_count_offset(int B, int H);
SDFR_API size_t sdfr_render_fused_tile_loss_offset(int R, int B, int W, int H);   /* the (sum, count) tile records */
SDFR_API int sdfr_render_step_fused_l1_pc(
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    const float* scale, int B, int W, int H, float cx, float cy, float fx, float fy, float threshold,
    const float* target, float* depth, int sdf_grad_mode, float* g_sdf, float* g_depth, void* workspace,
    size_t workspace_bytes, float pc_weight, const float* points, const int* offsets, int max_view_points,
    void* pc_workspace, size_t pc_workspace_bytes, int device, void* stream);
/* sdfr_render_step_backward_l1 = sdfr_render_backward_l1 as the second half of a step begun by
 * sdfr_render_step_forward_l1 (no prologue launch, the forward's view records and rectangles; pos / quat / inv_scale
 * are not passed again) -- the loss-fused form of sdfr_render_step_backward.
 * DEFERRED LOSS (both step backward forms): a step's forward called with loss = loss_stats = NULL leaves its per-tile
 * (sum, count) records unreduced -- no reduce launch between the two image kernels -- and the step's backward is then
 * given loss_stats = NULL and the two OUTPUTS loss [B], loss_stats_out [B][2]: every backward tile with a hit pixel
 * sums its view's counts itself (integers: exact in any order, so the upstream gradient is the same bit for bit) and
 * one workgroup per view writes loss / loss_stats_out in the reduce launch's fixed order (the same bits too), valid
 * after this call.  loss = loss_stats_out = NULL: loss_stats comes from the forward, as before. */
SDFR_API int sdfr_render_step_backward_l1(const float* loss_grad, float loss_weight, const float* loss_stats,
                                 const float* target, const float* depth, const float* sdf, int R,
                                 long long sdf_view_stride, int B, int W, int H, float cx, float cy, float fx,
                                 float fy, int sdf_grad_mode, float* g_sdf, long long g_sdf_view_stride, float* g_pos,
                                 float* g_quat, float* g_inv_scale, void* workspace, size_t workspace_bytes,
                                 float* loss, float* loss_stats_out, int device, void* stream);
/* byte offset of the backward's tile partials in its workspace (step_layout: of a step's workspace) */
SDFR_API size_t sdfr_render_partials_offset(int R, int B, int W, int H, int step_layout);
SDFR_API int sdfr_render_step_backward(const float* grad_depth, const float* depth, const float* sdf, int R,
                              long long sdf_view_stride, int B, int W, int H, float cx, float cy,
                              float fx, float fy, int sdf_grad_mode, float* g_sdf,
                              long long g_sdf_view_stride, float* g_pos, float* g_quat,
                              float* g_inv_scale, void* workspace, size_t workspace_bytes, int device,
                              void* stream);

/* ---- render + masked depth-L1 in one pass ------------------------------------ */

/* The depth term of SDFPipeline._compute_view_losses (sdfest/estimation/simple_setup.py:129-135)
 * folded into the renderer, so that neither the loss kernel nor the gradient image exists:
 *     overlap = (target > 0) & (depth > 0);   loss[b] = mean |depth - target| over overlap
 * sdfr_render_forward_l1 = sdfr_render_forward + loss[b] (NaN for an empty overlap, like
 * torch.mean of an empty selection) + loss_stats[b] = {sum |depth - target|, count} (2 floats per
 * view, consumed by the backward).  Sums are reduced in a fixed order (bitwise reproducible).
 *   target [B][H][W]  the observed depth images (read at hit pixels only). */
SDFR_API size_t sdfr_render_forward_l1_workspace_bytes(int R, int B, int W, int H);
SDFR_API int sdfr_render_forward_l1(const float* sdf, int R, long long sdf_view_stride,
                           const float* pos, const float* quat, const float* inv_scale, int B,
                           int W, int H, float cx, float cy, float fx, float fy, float threshold,
                           const float* target, float* depth, float* loss, float* loss_stats,
                           void* workspace, size_t workspace_bytes, int device, void* stream);

/* sdfr_render_backward with the upstream gradient image formed in the kernel:
 *     grad_depth = k_b * sign(depth - target) on the overlap, 0 elsewhere,
 *     k_b = loss_weight * (loss_grad ? loss_grad[b] : 1) / count_b      (0 for an empty overlap)
 * i.e. the gradient of sum_b loss_weight * loss_grad[b] * loss[b].  Results are identical to
 * sdfr_depth_l1_loss followed by sdfr_render_backward.  Workspace: sdfr_render_backward_workspace_bytes.
 *   loss_grad [B] DEVICE array or NULL;  loss_stats [B][2] from sdfr_render_forward_l1. */
SDFR_API int sdfr_render_backward_l1(const float* loss_grad, float loss_weight, const float* loss_stats,
                            const float* target, const float* depth, const float* sdf, int R,
                            long long sdf_view_stride, const float* pos, const float* quat,
                            const float* inv_scale, int B, int W, int H, float cx, float cy,
                            float fx, float fy, int sdf_grad_mode, float* g_sdf,
                            long long g_sdf_view_stride, float* g_pos, float* g_quat,
                            float* g_inv_scale, void* workspace, size_t workspace_bytes, int device,
                            void* stream);

/* ==== 1. CORE (continued) ===================================================================== */
/* ---- trilinear SDF sampler of the point-cloud loss --------------------------------------- */

/* Replaces losses.pc_loss (sdfest/estimation/losses.py:32-135), for all views of a step at once.
 *   points  [N][3]   camera-frame points; view v owns [offsets[v], offsets[v+1])
 *   offsets [B+1]    DEVICE int array; NULL is allowed for B == 1 and means [0, max_view_points)
 *   max_view_points  host-side upper bound of the longest segment (sizes the grid)
 *   pos [B][3], quat [B][4] (need not be normalised: the kernel normalises like the reference,
 *   :58), scale [B] (half-width, NOT its inverse)
 *   out     [N]      interpolated distance * scale, 0 for points outside the volume (:92-94,:134) */
SDFR_API int sdfr_pc_loss_forward(const float* points, const int* offsets, int B, int max_view_points,
                         const float* pos, const float* quat, const float* scale,
                         const float* sdf, int R, long long sdf_view_stride, float* out,
                         int device, void* stream);

SDFR_API size_t sdfr_pc_loss_backward_workspace_bytes(int B, int max_view_points);

/* VJP of the above for upstream grad_out[N]; equals torch autograd through losses.py:32-135
 * (the q normalisation is differentiated through; outside points get no gradient).
 * g_sdf is overwritten (stride 0: summed over the views; R^3: per view); g_pos [B][3],
 * g_quat [B][4], g_scale [B] are overwritten, reduced in a fixed order. */
SDFR_API int sdfr_pc_loss_backward(const float* grad_out, const float* points, const int* offsets, int B,
                          int max_view_points, const float* pos, const float* quat,
                          const float* scale, const float* sdf, int R, long long sdf_view_stride,
                          float* g_sdf, long long g_sdf_view_stride, float* g_pos, float* g_quat,
                          float* g_scale, void* workspace, size_t workspace_bytes, int device,
                          void* stream);

/* sdfr_pc_loss_backward for the loss the loop really uses (simple_setup.py:137-144),
 *     loss[v] = mean |pc_loss values of view v|,   objective = weight * sum_v loss[v]:
 * the upstream gradient +-weight / M_v is formed from the sign of each point's value inside the
 * kernel and loss[v] (NaN for a view without points, as torch.mean of nothing) is a by-product, so
 * neither sdfr_pc_loss_forward nor sdfr_pc_l1_loss has to run.  Gradients are identical to
 * forward -> sdfr_pc_l1_loss -> sdfr_pc_loss_backward.  Workspace: sdfr_pc_loss_backward_workspace_bytes. */
SDFR_API int sdfr_pc_l1_backward(float weight, float* loss, const float* points, const int* offsets, int B,
                        int max_view_points, const float* pos, const float* quat, const float* scale,
                        const float* sdf, int R, long long sdf_view_stride, float* g_sdf,
                        long long g_sdf_view_stride, float* g_pos, float* g_quat, float* g_scale,
                        void* workspace, size_t workspace_bytes, int device, void* stream);
/* the same, ADDING the d/dSDF contributions to g_sdf instead of overwriting it: in the loop the sampler runs
 * after the renderer's backward and sums into its volume (one zero fill and one addition launch less). */
SDFR_API int sdfr_pc_l1_backward_accumulate(float weight, float* loss, const float* points, const int* offsets, int B,
                        int max_view_points, const float* pos, const float* quat, const float* scale,
                        const float* sdf, int R, long long sdf_view_stride, float* g_sdf,
                        long long g_sdf_view_stride, float* g_pos, float* g_quat, float* g_scale,
                        void* workspace, size_t workspace_bytes, int device, void* stream);


/* ---- VAE decoder forward ------------------------------------------------------------------ */

/* Replaces SDFDecoder.forward / SDFVAE.decode (sdfest/vae/sdf_vae.py:217-259, :79-87).
 * The handle owns a device copy of the weights (re-laid-out for the kernels); creation is the
 * only call of this library that allocates device memory and synchronises.
 *   h_params   HOST pointer: the decoder's tensors flattened in state_dict order
 *              (decoder._fc_layers.{i}.weight [out][in], .bias, ..., decoder._conv_layers.{i}
 *              .weight [cout][cin][k][k][k], .bias)
 *   the five conv_* arrays and fc_out are the yaml's decoder.conv_layers / fc_layers entries
 *   volume     sdf_size (64); tsdf: truncation value or 0 for "False" */
typedef struct sdfr_decoder sdfr_decoder;
SDFR_API int sdfr_decoder_create(const float* h_params, size_t n_params, int latent, int n_fc,
                        const int* fc_out, int n_conv, const int* conv_in_size,
                        const int* conv_cin, const int* conv_cout, const int* conv_k,
                        const int* conv_relu, int volume, float tsdf, int device,
                        sdfr_decoder** out_handle);
SDFR_API void sdfr_decoder_destroy(sdfr_decoder* decoder);
SDFR_API size_t sdfr_decoder_workspace_bytes(const sdfr_decoder* decoder, int N);
/* z [N][latent] -> out [N][volume^3] (the reference returns (N,1,D,D,D)); on the decoder's device.
 * tape: NULL for inference; otherwise sdfr_decoder_tape_bytes(decoder, N) bytes that receive the
 * post-ReLU layer outputs a later sdfr_decoder_backward_latent needs. */
SDFR_API size_t sdfr_decoder_tape_bytes(const sdfr_decoder* decoder, int N);
SDFR_API int sdfr_decoder_forward(const sdfr_decoder* decoder, const float* z, int N, int enforce_tsdf,
                         float* out, float* tape, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Which of two equivalent kernel forms the calls on ONE decoder handle take.  The defaults are the measured-faster forms;
 * the results are the same bit for bit either way (the tests compare the forms through this call).  Per handle: nothing
 * is process-wide.  Returns the option's old value (>= 0) or SDFR_E_INVALID / SDFR_E_NULL.
 *   SDFR_DECODER_OPT_FUSED_RESIZE  how batched forwards treat an up-sampling resize in front of a 3x3x3 layer: 1 (default)
 *                                  inside that layer's patch load where that is faster (fine sizes up to 16), 2 wherever
 *                                  the folded form exists (fine sizes 16, 32, 64), 0 always as its own launch
 *   SDFR_DECODER_OPT_TILED_VJP     1 (default): the transposed resizes of the VJP in one launch each (an LDS-staged block
 *                                  per workgroup); 0: the three single-axis launches it replaces
 *   SDFR_DECODER_OPT_FC_ONE_WAVE   1 (default): the backward of a NARROW Linear stack (every layer input <= 64 wide, e.g.
 *                                  the mug decoder's 8 -> 20 -> 50) runs as one wave out of LDS -- in
 *                                  sdfr_decoder_backward_latent and inside sdfr_loop_tail[_records]; 0: the one-workgroup
 *                                  form wider stacks take
 *   SDFR_DECODER_OPT_FUSED_SINGLE  few latents (the render-and-compare loop decodes ONE): layer pairs as one launch each,
 *                                  the producer recomputed under the consumer's tile.  Bits (default 5 = the pairs that
 *                                  measured faster on MI355X, C5 0.1200 -> 0.1120 ms per iteration):
 *                                  1 an up-sampling resize + the 3x3x3 convolution behind it (sdf_vae.py:235-246),
 *                                  2 the Linear stack + the first convolution (:223-238),
 *                                  4 in the VJP the first transposed resize (+ ReLU mask, swapped 1x1x1 layer, padding)
 *                                    + the transposed convolution that reads it,
 *                                  8 the VJP's following stages likewise, chained through a z-pass epilogue */
#define SDFR_DECODER_OPT_FUSED_RESIZE 0
#define SDFR_DECODER_OPT_TILED_VJP 1
#define SDFR_DECODER_OPT_FC_ONE_WAVE 2
#define SDFR_DECODER_OPT_FUSED_SINGLE 3
SDFR_API int sdfr_decoder_set_option(sdfr_decoder* decoder, int option, int value);
/* Vector-Jacobian product of the decoder w.r.t. the latent, weights held constant: what
 * loss.backward() propagates to latent_shape in SDFPipeline.__call__
 * (sdfest/estimation/simple_setup.py:413-414, :456) through SDFDecoder.forward
 * (sdfest/vae/sdf_vae.py:217-259).  grad_out [N][volume^3] -> g_z [N][latent].
 * `tape` comes from the forward of the same z with enforce_tsdf = 0.  Deterministic (no atomics). */
SDFR_API size_t sdfr_decoder_backward_workspace_bytes(const sdfr_decoder* decoder, int N);
SDFR_API int sdfr_decoder_backward_latent(const sdfr_decoder* decoder, const float* z, const float* tape,
                                 const float* grad_out, int N, float* g_z, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* The same for ONE latent without its last launch (the backward of the small leading Linear layers, one workgroup):
 * *t_mid receives where, inside `workspace`, the gradient w.r.t. the wide layer's input lies; sdfr_loop_tail finishes
 * the product rule in its own launch.  Nothing else may use the workspace in between. */
SDFR_API int sdfr_decoder_backward_latent_deferred(const sdfr_decoder* decoder, const float* z, const float* tape,
                                          const float* grad_out, void* workspace, size_t workspace_bytes,
                                          void* stream, const float** t_mid);
/* sdfr_decoder_forward in two halves that meet in the tape (tape != NULL): stages = 1 the Linear stack only (z -> the
 * wide layer's output, in the tape's slot; out may be NULL), 2 the convolutional part from that slot (z is not read),
 * 3 both = sdfr_decoder_forward.  For the captured loop, whose tail launch leaves the NEXT iteration's Linear-stack
 * output in the tape (sdfr_loop_tail_fused, decoder_tape): its decode is then stage 2 alone, stage 1 runs once in
 * front of the first iteration.  Same kernels, same numbers. */
SDFR_API int sdfr_decoder_forward_stage(const sdfr_decoder* decoder, const float* z, int N, int enforce_tsdf,
                               float* out, float* tape, void* workspace, size_t workspace_bytes,
                               void* stream, int stages);
/* sdfr_decoder_backward_latent_deferred for an incoming gradient in 1 + n_scaled volumes, n_scaled of them not yet
 * normalised:
 *     grad = grad_out + sum_v k_v * grad_scaled[v],   k_v = count[v] > 0 ? weight / count[v] : 0   (count: device floats)
 * -- what sdfr_render_step_fused_l1_pc leaves: the point-cloud term, and every view's depth term before its division by
 * the view's overlap count.  With one scaled volume the first launch of the VJP forms the sum as it loads (decoders
 * whose first launch cannot, and n_scaled > 1, get it from one small launch in front), and ALL the volumes are
 * zero-filled once they have been read -- their producer adds into them.  volume^3 a multiple of 4, the volumes
 * 16-byte aligned, grad_scaled [n_scaled][volume^3], n_scaled <= 64. */
SDFR_API int sdfr_decoder_backward_latent_deferred_scaled(const sdfr_decoder* decoder, const float* z, const float* tape,
                                                 float* grad_out, float* grad_scaled, int n_scaled, const float* count,
                                                 float weight, void* workspace, size_t workspace_bytes, void* stream,
                                                 const float** t_mid);
/* ... and for N latents (the K objects of a frame): *t_mid is [N][width of the wide layer's input]; sdfr_loop_tail_objects
 * finishes object k's product rule from row k. */
SDFR_API int sdfr_decoder_backward_latent_deferred_batch(const sdfr_decoder* decoder, const float* z, const float* tape,
                                                const float* grad_out, int N, void* workspace, size_t workspace_bytes,
                                                void* stream, const float** t_mid);


/* ==== 3. LOOP ================================================================================= */
/* ---- glue of one render-and-compare iteration (SDFPipeline.__call__, simple_setup.py:408-470) --- */
/* Small kernels that replace the reference's per-iteration torch-op soup so that a whole
 * iteration is a fixed launch sequence (graph-capturable).  All pointers are device pointers. */

/* :411, :424-430 -- world-frame pose (position[3], orientation[4] un-normalised, scale[1]) to
 * the V camera frames: pos_c [V][3], quat_c [V][4], inv_scale [V] (= 1/scale), scale_v [V]. */
SDFR_API int sdfr_pose_to_views(const float* position, const float* orientation, const float* scale,
                       const float* cam_pos, const float* cam_quat, int V, float* pos_c,
                       float* quat_c, float* inv_scale, float* scale_v, int device, void* stream);

/* reverse of the above: per-view gradients from the renderer (ga_*: w.r.t. pos_c, quat_c,
 * inv_scale) and from the sampler (gb_*: w.r.t. pos_c, quat_c, scale), any of which may be NULL,
 * to g_position[3], g_orientation[4] (through the normalisation), g_scale[1]. */
SDFR_API int sdfr_views_to_pose_grad(const float* orientation, const float* scale, const float* cam_quat,
                            int V, const float* ga_pos, const float* ga_quat,
                            const float* ga_inv_scale, const float* gb_pos, const float* gb_quat,
                            const float* gb_scale, float* g_position, float* g_orientation,
                            float* g_scale, int device, void* stream);

/* The same chain with the two per-view reductions that precede it folded in: three launches of a
 * launch-bound loop in one.  Call sdfr_render_backward / _l1 with g_pos = g_quat = g_inv_scale = NULL and
 * sdfr_pc_loss_backward / sdfr_pc_l1_backward / _accumulate with g_pos = g_quat = g_scale = NULL ("deferred": they
 * then leave their per-tile / per-block partial sums in their workspaces and skip their reduce launch), then
 * pass those workspaces here, untouched in between, with the same V (= B), W, H, offsets, max_view_points and the
 * per-view quaternions quat_c the sampler was given.  Either workspace may be NULL (term absent).  pc_loss [V]:
 * the loss values of sdfr_pc_l1_backward* (which does not write them when deferred), NULL for the plain backward.
 * Same sums in the same order as the three separate launches (results agree to the last bit or two).  V <= 64. */
SDFR_API int sdfr_views_to_pose_grad_deferred(const float* orientation, const float* scale, const float* cam_quat,
                                     int V, const void* render_workspace, int W, int H,
                                     const void* pc_workspace, const int* offsets, int max_view_points,
                                     const float* quat_c, float* pc_loss, float* g_position,
                                     float* g_orientation, float* g_scale, int device, void* stream);

/* The tail of one iteration of the captured loop in ONE launch (one workgroup): sdfr_views_to_pose_grad_deferred (the
 * pose gradients go to grads[0..7]), sdfr_point_constraint (con_source NULL: none), sdfr_adam_step on params /
 * grads [position 3 | orientation 4 | scale 1 | latent n_params - 8] (grads[8..] must hold the latent gradient
 * already), and sdfr_pose_to_views for the NEXT iteration (pos_c / quat_c / inv_scale / scale_v are read by the
 * chain as this iteration's and then overwritten with the next one's).  Same arithmetic in the same order as the
 * four calls.  render_partials_offset: sdfr_render_partials_offset (the tile partials of a stand-alone or of a
 * step's backward).  decoder / decoder_t_mid (both or neither): the tail first runs the last stage of the decoder's
 * VJP that sdfr_decoder_backward_latent_deferred left out -- grads[8..] = d/d latent from decoder_t_mid, the latent
 * being params[8..] -- one launch less per iteration, same arithmetic. */
SDFR_API int sdfr_loop_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                   float lr_position, float lr_orientation, float lr_scale, float lr_latent, int update_latent,
                   const float* cam_pos, const float* cam_quat, int V, const void* render_workspace,
                   size_t render_partials_offset, int W, int H, const void* pc_workspace, const int* offsets,
                   int max_view_points, float* pos_c, float* quat_c, float* inv_scale, float* scale_v, float* pc_loss,
                   const float* con_source, const float* con_target, float con_weight, float* con_loss,
                   const sdfr_decoder* decoder, const float* decoder_t_mid, int device, void* stream);

/* sdfr_loop_tail behind sdfr_render_step_fused_l1_pc (the render pair as ONE launch): the tiles' pose sums of the
 * depth term come unscaled, beside the views' overlap counts -- the per-view reduction multiplies them by
 * k = depth_weight / count (the expression of the two-launch form's tiles), writes depth_loss [V] (nullable) =
 * sum |est - obs| / count from the tiles' records, and RESETS the counts for the next step.  view_count_offset /
 * tile_loss_offset: sdfr_render_fused_view_count_offset / _tile_loss_offset.  Everything else as sdfr_loop_tail. */
SDFR_API int sdfr_loop_tail_fused(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                         float lr_position, float lr_orientation, float lr_scale, float lr_latent, int update_latent,
                         const float* cam_pos, const float* cam_quat, int V, void* render_workspace,
                         size_t render_partials_offset, size_t view_count_offset, size_t tile_loss_offset,
                         float depth_weight, float* depth_loss, int W, int H, const void* pc_workspace,
                         const int* offsets, int max_view_points, float* pos_c, float* quat_c, float* inv_scale,
                         float* scale_v, float* pc_loss, const float* con_source, const float* con_target,
                         float con_weight, float* con_loss, const sdfr_decoder* decoder, const float* decoder_t_mid,
                         float* decoder_tape, unsigned* arrivals, int device, void* stream);
/* decoder_tape / arrivals (both or neither; with decoder, update_latent and a Linear stack whose leading layers are
 * at most 64 wide): the launch ALSO runs the decoder's Linear stack for the parameters it has just updated and leaves
 * the wide layer's output in the tape's slot -- the next iteration's decode is sdfr_decoder_forward_stage(stages = 2),
 * one launch less per iteration.  The launch then has one workgroup per 256 outputs of the wide layer: every one
 * repeats the latent's share of the tail (its gradient, its Adam step -- same inputs, same arithmetic, same numbers;
 * nothing is handed from workgroup to workgroup) and forms its slice; workgroup 0 is the tail proper and stores the
 * state.  arrivals: one device word, zero before a run's first iteration -- the other workgroups count themselves in
 * once they have read the state, workgroup 0 stores the new state only when all have (a bounded wait). */

/* sdfr_loop_tail for SEVERAL estimates at once -- the K detected objects of one frame, each with its own pose, scale,
 * latent and Adam state, optimised side by side in one launch sequence (the reference runs its pipeline once per object,
 * one after the other: simple_setup.py:213-225; a single estimate's iteration is a chain of dependent launches that
 * leaves most of the chip idle).  Workgroup k is object k:
 *   params / grads / exp_avg / exp_avg_sq  [K][n_params];  step [K]  (one counter per object)
 *   the launch's views are object-major: object k owns views k V .. k V + V - 1 of the render / sampler launches
 *   (set-up records, tile partials, point blocks, quat_c, pc_loss, and the pos_c / quat_c / inv_scale / scale_v written
 *   for the next iteration: all [K V ...]); cam_pos [V][3], cam_quat [V][4] are ONE camera list, the same for every object
 *   decoder == NULL: grads[k][8..] must hold d loss / d latent of object k (sdfr_decoder_backward_latent with N = K);
 *   decoder + decoder_t_mid (sdfr_decoder_backward_latent_deferred_batch with N = K, row k = object k): workgroup k runs
 *   the last stage of object k's VJP itself, as sdfr_loop_tail does for the single estimate (one launch less, and no
 *   copy of the latent gradients into `grads`)
 *   latents (may be NULL): [K][n_params - 8], receives the UPDATED latents packed for the next iteration's batched
 *   decode (sdfr_decoder_forward with N = K reads z [N][latent]; `params` holds them with stride n_params)
 *   No point constraint.  Same arithmetic per object, in the same order, as sdfr_loop_tail. */
/* sdfr_pose_to_views for the K rows of `params` ([K][n_params]: position 3 | orientation 4 | scale 1 | ...) and one camera
 * list: view k V + v of the outputs is object k seen from camera v; latents (may be NULL): as above, the CURRENT ones. */
SDFR_API int sdfr_pose_to_views_objects(const float* params, int n_params, int n_objects, const float* cam_pos,
                               const float* cam_quat, int V, float* pos_c, float* quat_c, float* inv_scale,
                               float* scale_v, float* latents, int device, void* stream);
SDFR_API int sdfr_loop_tail_objects(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                           int n_objects, float lr_position, float lr_orientation, float lr_scale, float lr_latent,
                           int update_latent, const float* cam_pos, const float* cam_quat, int V,
                           const void* render_workspace, size_t render_partials_offset, int W, int H,
                           const void* pc_workspace, const int* offsets, int max_view_points, float* pos_c,
                           float* quat_c, float* inv_scale, float* scale_v, float* pc_loss,
                           const sdfr_decoder* decoder, const float* decoder_t_mid, float* latents, int device,
                           void* stream);

/* ---- the loop sharded over ranks (one process per GPU) ------------------------------------------------
 * The reference's multi-view iteration is one Python loop over the views with ONE shared pose and ONE shared SDF
 * (simple_setup.py:420-446, update :456-462).  Sharded, every rank renders and samples a contiguous shard of the
 * views and the iteration has exactly one exchange, an all-reduce (sum) of one bucket
 *     [ d loss / d SDF, R^3 words ][ V_total view records of SDFR_VIEW_RECORD_FLOATS floats ]
 * after which every rank runs the same decoder VJP and the same Adam step on replicated state (no broadcast).
 * A view's record: [0..7] the renderer's pose sums (g_pos 3, g_quat 4, g_inv_scale), [8..15] the sampler's (g_pos 3,
 * g_quat 4 through the normalisation's Jacobian, g_scale), [16] its depth loss, [17] its point-cloud loss, rest 0.
 * Records instead of the 8 summed pose gradients: each is non-zero on exactly one rank, so the sum over ranks
 * reproduces it exactly, and the chain adds the views in one fixed order on every rank (thread t of the tail's
 * workgroup the views t, t + 256, ..., then a fixed tree: any number of views) -- the pose gradient does not
 * depend on how the views were spread (with SDFR_SDF_GRAD_DETERMINISTIC | SDFR_BWD_SMALL_TILES the whole iteration
 * is bitwise the single-process one; exchange the bucket as 64-bit integers then, the volume being the int64 one).
 *
 * sdfr_loop_view_records: one launch, one wave per view of the whole list: records[v] for this rank's views
 * [view_begin, view_begin + V_local) from the partials the deferred backward calls left (as
 * sdfr_views_to_pose_grad_deferred: render_workspace + sdfr_render_partials_offset, pc_workspace; either may be NULL;
 * with_pc_loss: the sampler ran in its L1 form and left loss partials), zeros for every other view.  loss_depth
 * [V_local] (or NULL): per-view depth losses to carry along.  sdf_grad_mode: the backward's (SDFR_BWD_SMALL_TILES). */
#define SDFR_VIEW_RECORD_FLOATS 20
SDFR_API int sdfr_loop_view_records(const void* render_workspace, size_t render_partials_offset, int W, int H,
                           int sdf_grad_mode, const void* pc_workspace, int with_pc_loss, const int* offsets,
                           int max_view_points, const float* quat_c, const float* loss_depth, int view_begin,
                           int V_local, int V_total, float* records, int device, void* stream);
/* sdfr_loop_tail working from the exchanged records of ALL views: (decoder VJP's last stage,) the chain over the
 * V_total records with cam_quat [V_total][4], the point constraint, Adam, and the next iteration's view poses for
 * this rank's shard (cam_pos / cam_quat are the whole lists; pos_c / quat_c / inv_scale / scale_v [V_local]). */
/* The inlier bookkeeping of sdfr_inlier_ratio split around the exchange: the rank that owns the LAST view counts
 * (sdfr_inlier_counts_record: the two counts, as floats -- exact below 2^24 pixels -- into words 18 and 19 of that
 * view's record, after sdfr_loop_view_records), every rank updates from the exchanged record after the tail
 * (sdfr_inlier_update_record: ratio, history, best-so-far state and parameters as sdfr_inlier_ratio). */
SDFR_API int sdfr_inlier_counts_record(const float* depth_input, const float* depth_estimate, int W, int H,
                              float relative_threshold, int* counts, float* record, int device, void* stream);
SDFR_API int sdfr_inlier_update_record(const float* record, const int* step, float* history, int max_history,
                              float* state, const float* params, int n_params, float* best_params, int device,
                              void* stream);
SDFR_API int sdfr_loop_tail_records(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                           float lr_position, float lr_orientation, float lr_scale, float lr_latent,
                           int update_latent, const float* cam_pos, const float* cam_quat, int V_total,
                           int view_begin, int V_local, const float* records, float* pos_c, float* quat_c,
                           float* inv_scale, float* scale_v, const float* con_source, const float* con_target,
                           float con_weight, float* con_loss, const sdfr_decoder* decoder,
                           const float* decoder_t_mid, int device, void* stream);

/* sdfr_render_backward_l1 and sdfr_pc_l1_backward_accumulate of one loop iteration in ONE launch (they are
 * independent and neither fills the chip for a handful of views): arguments as in those two calls -- the per-view
 * poses are pos / quat with inv_scale for the renderer and scale (= 1 / inv_scale) for the sampler -- and the same
 * results in g_sdf (zero-filled first, then both terms added).  The pose gradients are always left deferred in the
 * two workspaces: follow with sdfr_views_to_pose_grad_deferred, which also writes the point-cloud loss values. */
SDFR_API int sdfr_render_backward_l1_pc(
    const float* loss_grad, float loss_weight, const float* loss_stats, const float* target, const float* depth,
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    int B, int W, int H, float cx, float cy, float fx, float fy, int sdf_grad_mode, float* g_sdf,
    long long g_sdf_view_stride, void* workspace, size_t workspace_bytes, float pc_weight, const float* points,
    const int* offsets, int max_view_points, const float* scale, void* pc_workspace, size_t pc_workspace_bytes,
    int device, void* stream);

/* :125-131 -- loss[v] = mean |estimate - target| over (target > 0) & (estimate > 0) (NaN when the
 * overlap is empty, like torch.mean of an empty selection); grad_estimate = weight * d loss / d
 * estimate.  Deterministic two-pass reduction. */
SDFR_API size_t sdfr_depth_l1_workspace_bytes(int V, int W, int H);
SDFR_API int sdfr_depth_l1_loss(const float* estimate, const float* target, int V, int W, int H,
                       float weight, float* loss, float* grad_estimate, void* workspace,
                       size_t workspace_bytes, int device, void* stream);

/* :144 -- loss[v] = mean |values| over view v's points, grad_values = weight * sign / M_v */
SDFR_API int sdfr_pc_l1_loss(const float* values, const int* offsets, int V, int max_view_points,
                    float weight, float* loss, float* grad_values, int device, void* stream);

/* pointset_utils.depth_to_pointcloud (sdfest/initialization/pointset_utils.py:57-77, convention "opengl",
 * no mask, no normalisation) for V images at once: the non-zero pixels, view-major and row-major,
 *     x = (col - cx0) * z * rfx,  y = -(row - cy0) * z * rfy,  z = -depth     (pixel-centre-0 intrinsics).
 * rfx, rfy are the reciprocal focal lengths: torch evaluates `t / scalar` on the GPU as
 * `t * float(1.0 / scalar)` (reciprocal in double, rounded once), and the caller that holds the double
 * computes exactly that -- the points are then bit-identical to the reference expression's.
 * Two calls because the caller sizes the output: sdfr_depth_count writes counts[v] (DEVICE ints) and
 * the per-block counts into `workspace`; the caller reads the counts, builds the exclusive prefix
 * `offsets` [V] (DEVICE ints) and allocates points [sum][3]; sdfr_depth_to_points fills them from the
 * SAME depth and workspace. */
SDFR_API size_t sdfr_depth_points_workspace_bytes(int V, int W, int H);
SDFR_API int sdfr_depth_count(const float* depth, int V, int W, int H, int* counts, void* workspace,
                     size_t workspace_bytes, int device, void* stream);
SDFR_API int sdfr_depth_to_points(const float* depth, int V, int W, int H, float rfx, float rfy, float cx0,
                         float cy0, const int* offsets, const void* workspace, float* points,
                         int device, void* stream);
/* The same with the ORDER of a view's points chosen by the caller.  SDFR_POINT_ORDER_ROW_MAJOR is the reference's
 * (torch.nonzero).  SDFR_POINT_ORDER_TILED enumerates the image in tiles of 64 x 16 pixels, a tile in four 16 x 16
 * sub-tiles, a sub-tile row-major: the same points, permuted within their view, so that 256 consecutive points are a
 * compact patch of the surface.  For consumers that only sum over a view's points -- the point-cloud loss and its
 * gradients (losses.py:32-135 has no order): the sampler's backward pre-sums the d/dSDF of 256 consecutive points
 * in LDS before its global atomics, and compact patches share more voxels (global atomics per point of
 * back-projected depth images: 0.55 row-major, ~0.3 tiled).  Use the same order in both calls. */
#define SDFR_POINT_ORDER_ROW_MAJOR 0
#define SDFR_POINT_ORDER_TILED 1
SDFR_API int sdfr_depth_count_ordered(const float* depth, int V, int W, int H, int order, int* counts, void* workspace,
                             size_t workspace_bytes, int device, void* stream);
SDFR_API int sdfr_depth_to_points_ordered(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                 float cx0, float cy0, const int* offsets, const void* workspace, float* points,
                                 int device, void* stream);
/* The synthetic-view generator's point sets are normalised by their centroid (generated_dataset.py:318-326:
 * pointset -= mean(pointset), position -= mean; optionally + a noise vector).  Both passes over the images do that
 * on the way: sdfr_depth_count_centroid = sdfr_depth_count_ordered that also leaves centroid[v] = mean of view v's
 * back-projected points (fixed-order sums; 0 for an empty view) and, with `offsets` [V] (nullable), the exclusive
 * prefix of the counts that sdfr_depth_to_points_* takes -- the caller reads the counts (to size `points`) -- and
 * sdfr_depth_to_points_shifted writes (points - shift[v]) + noise[v]: two roundings, in the reference's order
 * (`pointset -= centroid`, then `pointset += noise`).  shift / noise [V][3], either may be NULL (that term is
 * skipped: shift = noise = NULL gives the plain points).  No pass over the packed points, no per-point owner index,
 * no atomics.
 * Workspace: sdfr_depth_centroid_workspace_bytes, 16-byte aligned, the same for both calls. */
SDFR_API size_t sdfr_depth_centroid_workspace_bytes(int V, int W, int H);
SDFR_API int sdfr_depth_count_centroid(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                              float cx0, float cy0, int* counts, int* offsets, float* centroid, void* workspace,
                              size_t workspace_bytes, int device, void* stream);
SDFR_API int sdfr_depth_to_points_shifted(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                 float cx0, float cy0, const int* offsets, const void* workspace, const float* shift,
                                 const float* noise, float* points, int device, void* stream);

/* The same pair of passes WITHOUT the host in between, for a caller that re-uses its buffers from observation to
 * observation (the captured render-and-compare loop, rebound to new depth images): `points` has room for every pixel
 * (V * W * H points), so nobody has to read the counts to size it.  One call: block counts -> counts [V] and the
 * exclusive prefix offsets [V + 1] (offsets[V] = all points; the format sdfr_pc_* take) -> compaction.  Consumers are
 * then launched for `max_view_points` = W * H (an upper bound sizes their grids; blocks beyond a view's real count leave
 * at once).  Workspace: sdfr_depth_points_workspace_bytes.  Same points, bit for bit, as the two-call form. */
SDFR_API int sdfr_depth_to_points_resident(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                  float cx0, float cy0, int* counts, int* offsets, void* workspace,
                                  size_t workspace_bytes, float* points, int device, void* stream);

/* SDFPipeline._preprocess_depth (sdfest/estimation/simple_setup.py:671-693), IN PLACE like the reference:
 *     depth[~mask] = 0;   if has_far_field: depth[depth > far_field] = 0
 * depth [V][H][W]; mask [V][H][W] bytes (a torch.bool tensor's storage: 0 = outside).  copy_to (nullable): the
 * preprocessed images are also written there (the loop's own target buffer: one pass instead of two). */
SDFR_API int sdfr_preprocess_depth(float* depth, const unsigned char* mask, int V, int W, int H, float far_field,
                          int has_far_field, float* copy_to, int device, void* stream);

/* a += b  (sums the renderer's and the sampler's d/dSDF) */
SDFR_API int sdfr_add_inplace(float* a, const float* b, size_t n, int device, void* stream);

/* :400-406, :458-462 -- one torch.optim.Adam step (default betas/eps) on params laid out
 * [position 3 | orientation 4 | scale 1 | latent ...], then orientation /= |orientation|.
 * step[0] (device int) is the number of steps taken so far and is incremented. */
SDFR_API int sdfr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   int* step, int n_params, float lr_position, float lr_orientation,
                   float lr_scale, float lr_latent, int update_latent, int device, void* stream);

/* :164-175 + losses.py:138-153 -- point constraint on the (un-normalised) orientation parameter:
 * loss[0] = weight * |q (source,0) conj(q) - target| (quaternion_apply, quaternion_utils.py:36-54);
 * its gradient w.r.t. q is ADDED to g_orientation[4].  loss / g_orientation may be NULL. */
SDFR_API int sdfr_point_constraint(const float* orientation, const float* source, const float* target,
                          float weight, float* loss, float* g_orientation, int device, void* stream);

/* :177-211 -- inlier ratio of one view (the reference passes the LAST view's loop variables,
 * :463-470) and the best-estimate bookkeeping: ratio = #(|in - est| / in < relative_threshold) /
 * #(in != 0); history[step[0] - 1] = ratio (if 1 <= step[0] <= max_history; step = sdfr_adam_step's
 * counter, call after it); a strictly larger ratio, or the first, copies params[n_params] to
 * best_params.  state[3] = {best ratio, its 1-based iteration, has_best}: zero it before a run.
 * counts[2] is integer scratch that must be zero on the first call (left zero by every call). */
SDFR_API int sdfr_inlier_ratio(const float* depth_input, const float* depth_estimate, int W, int H,
                      float relative_threshold, const int* step, int* counts, float* history,
                      int max_history, float* state, const float* params, int n_params,
                      float* best_params, int device, void* stream);

/* losses.py:8-29 -- nn_loss: dist[i] = min_j max(0, -2 a_i.b_j + |a_i|^2 + |b_j|^2), nearest[i] = the
 * first minimising j; points are (N,3) / (M,3).  The backward is the VJP autograd gives the
 * reference: through the selected pair, none for a clamped (zero) distance; g_to is overwritten. */
SDFR_API int sdfr_nn_loss_forward(const float* points_from, int N, const float* points_to, int M,
                         float* dist, int* nearest, int device, void* stream);
SDFR_API int sdfr_nn_loss_backward(const float* grad_dist, const float* points_from, int N,
                          const float* points_to, int M, const float* dist, const int* nearest,
                          float* g_from, float* g_to, int device, void* stream);

/* generated_dataset.py:234-245 -- perturbed mask of the view generator: mask[b][i][j] = 1 where the
 * input pixel nearest to M_b (j + 0.5 - W/2, i + 0.5 - H/2, 1) + (W/2 - 0.5, H/2 - 0.5) has depth != 0
 * (0 outside the image); matrices[B][6] = the row-major 2x3 INVERSE affine maps in pixels about the image
 * centre (what torchvision's RandomAffine hands to grid_sample(nearest, align_corners=False)). */
SDFR_API int sdfr_affine_mask(const float* depth, int B, int W, int H, const float* matrices,
                     unsigned char* mask, int device, void* stream);

/* ==== 4. GENERATOR / INITIALISATION (sdfr_affine_mask above; the network below) =============== */
/* ---- initialisation network forward: sdfest/initialization/pointnet.py:7-96,
 * sdf_pose_network.py:9-115, estimation/simple_setup.py:795-812 ------------------------------------ */

/* one per-point layer of VanillaPointNet on a set of M points, inference mode:
 *   Y = relu((X W[:, :cin]^T + cvec) * bn_scale + bn_shift);   y = resid ? resid + Y : Y  (y may be NULL);
 *   colmax[col] = max over the points of Y[:, col]  (written, cout floats).
 * x [M][ldx], w [cout][ldw] as torch stores nn.Linear.weight, y / resid [M][ldy].  bn_scale / bn_shift are
 * the folded BatchNorm1d (gamma / sqrt(var + eps), beta - mean * scale; 1 and 0 without batch norm).
 * A dense link's concatenated set maximum enters through cvec = bias + W[:, cin:] . max (sdfr_linear_vec). */
SDFR_API int sdfr_pointnet_layer(const float* x, int M, int cin, int ldx, const float* w, int ldw,
                        const float* cvec, const float* bn_scale, const float* bn_shift,
                        const float* resid, float* y, int ldy, int cout, float* colmax, int device,
                        void* stream);

/* The same with the number of points ON THE DEVICE (row_count[0], clamped to M_capacity = the rows x, y and resid have
 * room for): for a caller that never reads the count of a depth image's points back -- the initialisation network
 * inside a captured launch sequence (sdfest_amd.init_network.ResidentInit).  The grid strides over the real rows. */
SDFR_API int sdfr_pointnet_layer_counted(const float* x, const int* row_count, int M_capacity, int cin, int ldx,
                                const float* w, int ldw, const float* cvec, const float* bn_scale,
                                const float* bn_shift, const float* resid, float* y, int ldy, int cout,
                                float* colmax, int device, void* stream);

/* y[cout] = act((W[:, koff:koff+k] . x + bias) * bn_scale + bn_shift): the head's layers on the set feature
 * and the bias vectors of dense links.  bias, bn_scale / bn_shift may be NULL; relu = 0 / 1. */
SDFR_API int sdfr_linear_vec(const float* w, int ldw, int koff, const float* x, int k, const float* bias,
                    const float* bn_scale, const float* bn_shift, int relu, float* y, int cout,
                    int device, void* stream);

/* simple_setup.py:795-812, :978-1009 -- posterior[C] = softmax(logits), with a prior: posterior * prior /
 * train_prior (train_prior may be NULL), L1-normalised; out_index = argmax (first maximum), out_max = its
 * probability. */
SDFR_API int sdfr_orientation_posterior(const float* logits, int C, const float* prior, const float* train_prior,
                               float* posterior, int* out_index, float* out_max, int device, void* stream);

/* simple_setup.py:790-838 for one view, on the device: the head's output row -> the estimate in the WORLD frame, written
 * into params [position 3 | orientation 4 | scale 1 | latent] (the loop's parameter vector: sdfr_loop_tail).
 *   head          [latent + 4 + C] (discretised orientation: latent, position, scale, C logits) or [latent + 8]
 *   grid_quats    [C][4] with index = the argmax of sdfr_orientation_posterior (SO3Grid.index_to_quat of every cell,
 *                 uploaded once), or NULL: the head's quaternion, normalised (sdf_pose_network.py:97-101)
 *   centroid [3]  nullable: `position += centroid` (:793-794);  cam_pos [3], cam_quat [4]: camera -> world (:819-825)
 *   mean_shape    zero latent (:790-791)
 *   take_if_better 0: "first" -- always written; 1: "best" -- written only where posterior_max[0] > best[0], which
 *                 then takes its value (:829-838; the caller zeroes best[0] in front of the first view). */
SDFR_API int sdfr_init_estimate(const float* head, int latent, const float* grid_quats, const int* index,
                       const float* centroid, const float* cam_pos, const float* cam_quat, int mean_shape,
                       int take_if_better, const float* posterior_max, float* best, float* params, int device,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFR_H_ */

/*
 * instascene_rasterizer.h — C-ABI of the MI355X-native 2D-Gaussian ("surfel")
 * rasterizer.  This is the drop-in boundary for the reference's
 *
 *   CudaRasterizer::Rasterizer::forward      cuda_rasterizer/rasterizer.h:35-66
 *   CudaRasterizer::Rasterizer::backward     cuda_rasterizer/rasterizer.h:68-93
 *   CudaRasterizer::Rasterizer::markVisible  cuda_rasterizer/rasterizer.h:28-33
 *
 * (paths relative to submodules/diff-surfel-rasterization/ of zju3dv/InstaScene).
 *
 * Differences from the reference interface, all at the ABI level only:
 *   - the three std::function<char*(size_t)> resize callbacks are replaced by
 *     size queries (isr_*_bytes) + caller-provided workspaces; forward is split
 *     at the one point where the instance count R must be known to size the
 *     binning workspace (rasterizer_impl.cu:283-291);
 *   - every call takes the HIP stream to launch on (the reference uses the
 *     legacy default stream); pass NULL for the default stream;
 *   - optional inputs are NULL (shs xor colors_precomp; (scales,rotations) xor
 *     transMat_precomp; extra_attrs with ED = 0);
 *   - all pointers are DEVICE pointers to contiguous row-major fp32/int32 data
 *     unless stated; no torch types.
 *
 * Return value: 0 on success, a negative ISR_E* code otherwise;
 * isr_last_error() returns a human-readable message for the calling thread.
 */
#ifndef INSTASCENE_RASTERIZER_H
#define INSTASCENE_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISR_OK 0
#define ISR_EINVAL (-1)    /* bad argument combination (reference: AT_ERROR / std::runtime_error) */
#define ISR_EHIP (-2)      /* a HIP runtime call or kernel launch failed */
#define ISR_ECAPACITY (-3) /* binning workspace smaller than the instance count */

/* arithmetic mode of the blend kernels */
#define ISR_MODE_EXACT 0 /* op-for-op IEEE fp32, bit-identical to oracle/surfel_oracle.cpp */
#define ISR_MODE_FAST 1  /* explicit FMA contraction + hardware rcp/exp in the per-pixel loops */
#define ISR_MODE_PREBINNED 0x100 /* flag for isr_forward_render: isr_forward_bin already ran on these buffers */
#define ISR_MODE_FEATURE_ONLY 0x200 /* opt-in flag for isr_forward_render (with ISR_MODE_FAST, ED > 0): only out_extra and the
                                     * state the feature-only backward reads are produced; out_color / out_others (may be NULL)
                                     * and the tracer list are left untouched.  The reference always renders everything. */

/* which gradients isr_backward must produce (bit mask) */
#define ISR_GRAD_EXTRA 1u    /* dL_dextra only needs the blend weights */
#define ISR_GRAD_GEOMETRY 2u /* everything else (colors, opacity, means, scales, rotations, sh, transMat, means2D) */

const char* isr_last_error(void);
int isr_version(void);

/* Debug mode of the calling host thread - the reference's `debug` argument (rasterizer.h:60,90; CHECK_CUDA,
 * auxiliary.h:297-304): while on, every entry point of this library synchronises the stream after each kernel it
 * launches and fails with ISR_EHIP naming the kernel that faulted, instead of the error surfacing at some later call.
 * ISR_DEBUG_SYNC=1 in the environment turns it on for the whole process.  Returns the previous setting.
 * `fault_after` (testing aid): with debug on, the n-th checked launch from now (n >= 1) reports an injected fault; 0 = none. */
int isr_set_debug(int on, int fault_after);

/* ---- optional per-kernel timing (HIP events on the launch stream), used by bench.py.
 * isr_profile_enable(1) clears and starts recording every kernel, (2) only the forward blend kernel (two events per
 * step instead of ~25: the timed region of bench.py), (0) stops; isr_profile_summary() synchronises and writes one
 * "kernel_name launches total_ms" line per kernel into buf (returns bytes written). */
void isr_profile_enable(int on);
/* Work counters of the forward blend kernel (bench.py's `roofline.valu`): the NEXT isr_forward_render call of this host
 * thread in ISR_MODE_FAST also adds, into device_counters (u64[16], device memory, zeroed by the caller): [0] (wave, splat)
 * cull tests, [1] (wave, splat) pairs evaluated, [2] pairs with at least one blending lane, [3] blending (pixel, splat) pairs,
 * [4] consecutive evaluated pairs whose alpha >= 1/255 bounds share no pixel of the wave's block (greedy; tile-wide kernel),
 * [5] 4x4 sub-blocks with a blending pixel, [6] evaluations on the EXACT path, [7] pairs outside the guard bands deciding unlike
 * EXACT (must be 0), [8..15] the walk lengths of a per-8x4-half / per-4x4-quad decomposition (isr_forward_fast.hip).
 * A few extra atomics per wave; not meant for timed runs. */
void isr_forward_set_counters(unsigned long long* device_counters);
/* Work counters of the dense geometry backward (k_render_bwd_geo, the train.py step's dominant kernel): the NEXT isr_backward call (of any
 * host thread: autograd runs backward passes on threads of its own) that takes that kernel also adds, into device_counters (u64[16], zeroed by the caller): [0] chunks of 64 splat slots
 * walked by a wave, [1] slots holding a splat, [2] (chunk, pixel row) pairs, [3] those a splat of the chunk reaches, [4] (chunk, pixel)
 * iterations with a candidate lane, [5] candidate lanes, [6] iterations with a blending lane, [7] blending lanes, [8] chunks whose
 * always-EXACT splat was pre-evaluated, [9] partial rows stored.  Not meant for timed runs. */
void isr_backward_set_counters(unsigned long long* device_counters);
size_t isr_profile_summary(char* buf, size_t len);

/* ---- workspace sizes (bytes); the layouts are opaque forward->backward hand-offs
 *      (reference: GeometryState / ImageState / BinningState, rasterizer_impl.h:29-65). */
size_t isr_geom_bytes(int P);
size_t isr_image_bytes(int width, int height);
/* width / height: the image's (the binning workspace also holds one 64-bit hit mask per 64 list entries, tile and 8x8 block:
 * its size depends on the number of tiles); the same num_rendered must be passed to every call that takes this workspace */
size_t isr_binning_bytes(int64_t num_rendered, int width, int height);
/* scratch for the deterministic (atomic-free) gradient reduction in isr_backward */
size_t isr_backward_scratch_bytes(int64_t num_rendered, int ED, unsigned grad_mask);

/* ---- forward, part 1: per-Gaussian preprocess (K1), per-tile counting and scans.
 * Replaces rasterizer_impl.cu:233-287.  Writes radii[P].  If num_rendered_host is
 * non-NULL the call synchronises the stream and stores R there (the reference
 * always does this blocking read, rasterizer_impl.cu:287); pass NULL to stay
 * asynchronous and read R later with isr_read_num_rendered().
 * `prefiltered`: bit 0 is the reference's flag (unused there too); ISR_PREPARE_TIGHT_RECTS drops from every splat's
 * tile rectangle the tiles in which its alpha is certainly below 1/255 (the reference bins the square of the larger
 * 3-sigma extent) — same images and gradients, fewer tile instances; the tile lists then differ from the reference's,
 * so the bit-exact mode does not use it. */
#define ISR_PREPARE_TIGHT_RECTS 0x100
int isr_forward_prepare(int P, int D, int M, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* transMat_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                        float tan_fovx, float tan_fovy, int prefiltered,
                        int* radii, void* geom_buffer, void* image_buffer,
                        int64_t* num_rendered_host, void* stream);

int isr_read_num_rendered(const void* geom_buffer, int64_t* num_rendered_host, void* stream);

/* ---- forward, part 2a (optional): the binning alone — key scatter into the tile buckets and the per-tile sort.
 * Like part 1 it reads geometry only, so a caller may issue it ahead of time (e.g. while a gradient all-reduce is
 * in flight) and then call isr_forward_render with (mode | ISR_MODE_PREBINNED).
 * Exactly ONE binning per isr_forward_prepare on the same buffers (directly, or inside isr_forward_render without
 * ISR_MODE_PREBINNED): it consumes part 1's scatter cursors, and its last launch completes the per-Gaussian row offsets the
 * backward entries read (the reference's InclusiveSum over tiles_touched, rasterizer_impl.cu:283: not an input of the
 * binning itself, so it is kept off the path to the tile lists; ISR_SCAN_LATE=0 moves it back into part 1). */
int isr_forward_bin(int P, int width, int height, void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                    void* image_buffer, void* stream);

/* The same; scatter_done_event (a hipEvent_t, may be NULL) is recorded on `stream` right behind the key scatter - the one kernel of
 * the chain that suffers beside an HBM-saturating neighbour: a caller that runs the chain on a side stream can make that
 * neighbour's stream wait for it (hipStreamWaitEvent) and let the sorts run beside it. */
int isr_forward_bin_event(int P, int width, int height, void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                          void* image_buffer, void* scatter_done_event, void* stream);

/* ---- forward, part 2: binning (K4-K7 equivalent) and the per-tile blend (K8).
 * Replaces rasterizer_impl.cu:289-351.  binning_capacity is the R the binning
 * workspace was sized for.  out_extra may be NULL when ED == 0.  The tracer
 * ("gau_related_pixels", forward.cu:422-428) is optional: pass NULL to skip it;
 * otherwise tracer_pairs[tracer_capacity][2] receives (gaussian, pixel) pairs
 * with blend weight > 0.1 in unspecified order and *tracer_count (device int32)
 * the index of the last pair = their number - 1 (-1 when there is none) — the reference's
 * "gau_pixel_indices" (rasterize_points.cu:150); entries beyond capacity are counted but not stored. */
int isr_forward_render(int P, int ED, int width, int height, int mode,
                       const float* background, const float* colors_precomp,
                       const float* transMat_precomp, const float* extra_attrs,
                       void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                       void* image_buffer,
                       float* out_color, float* out_others, float* out_extra,
                       int32_t* tracer_pairs, int64_t tracer_capacity, int32_t* tracer_count,
                       void* stream);

/* The same with the feature rows handed over RAW plus their two row-normalisation factors (extension for a trainer that owns the
 * feature table; the reference's render() normalises the table itself, gaussian_renderer/__init__.py:57-62, after the model's getter
 * did, scene/gaussian_model.py:122-125): the blend stages (extra_attrs[g] * extra_row_scale[g][0]) * extra_row_scale[g][1] - what
 * isr_feature_rows_step would otherwise have written out as a normalised copy of the whole table every step.  ISR_MODE_FAST, the
 * per-block blend kernel only; extra_row_scale = NULL is isr_forward_render. */
int isr_forward_render_scaled(int P, int ED, int width, int height, int mode,
                              const float* background, const float* colors_precomp,
                              const float* transMat_precomp, const float* extra_attrs, const float* extra_row_scale /*[P,2]*/,
                              void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                              void* image_buffer,
                              float* out_color, float* out_others, float* out_extra,
                              int32_t* tracer_pairs, int64_t tracer_capacity, int32_t* tracer_count,
                              void* stream);

/* ---- backward (K9 + K10).  Replaces rasterizer_impl.cu:355-463.  Gradient
 * outputs are fully written by the call (no zero-initialisation needed); the
 * ones not selected by grad_mask may be NULL.  dL_dout_* may be NULL meaning
 * "all zeros" (autograd passes no gradient for an unused output). */
int isr_backward(int P, int D, int M, int64_t num_rendered, int ED, int width, int height, int mode,
                 unsigned grad_mask,
                 const float* background, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* transMat_precomp, const float* extra_attrs,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                 const float* dL_dout_color, const float* dL_dout_others, const float* dL_dout_extra,
                 float* dL_dmean2D /*[P,3]*/, float* dL_dnormal /*[P,3]*/, float* dL_dopacity /*[P]*/,
                 float* dL_dcolor /*[P,3]*/, float* dL_dmean3D /*[P,3]*/, float* dL_dtransMat /*[P,9]*/,
                 float* dL_dsh /*[P,M,3]*/, float* dL_dscale /*[P,2]*/, float* dL_drot /*[P,4]*/,
                 float* dL_dextra /*[P,ED]*/,
                 void* scratch, size_t scratch_bytes, void* stream);

/* ---- extension: the feature map is only read at n sampled pixels (train_semantic.py:118-129 picks 8192 pixels per
 * loss).  isr_sample_extra gathers sampled[i, :] = out_extra[:, pixels[i]] (pixels = y*W + x, int64, may repeat);
 * isr_backward_sampled turns dL/dsampled [n, ED] into dL_dextra [P, ED] (added to its content when accumulate != 0;
 * NULL = stop before the per-Gaussian reduction, see isr_feature_rows_step) without ever materialising the dense [ED, H, W] gradient: the samples are binned per tile and each tile's list is
 * walked once with a lane per splat.  Same forward state (geom / binning / image buffers) as isr_backward. */
size_t isr_backward_sampled_scratch_bytes(int64_t num_rendered, int ED, int n_samples, int width, int height);
int isr_sample_extra(int ED, int width, int height, int n_samples, const float* out_extra, const long long* pixels,
                     float* sampled, void* stream);
int isr_backward_sampled(int P, int64_t num_rendered, int ED, int width, int height, int mode, int n_samples,
                         const long long* pixels, const float* dL_dsampled, const float* transMat_precomp,
                         const void* geom_buffer, const void* binning_buffer, const void* image_buffer, float* dL_dextra,
                         int accumulate, void* scratch, size_t scratch_bytes, void* stream);

/* ---- extension: the per-Gaussian tail of a feature-training step in one pass over [P, ED] (ED % 4 == 0, ED <= 256).
 * isr_backward_sampled called with dL_dextra == NULL leaves the per-(tile, Gaussian) partial rows in its scratch
 * ("rows_scratch" here, NULL = no rows).  This entry then, for every Gaussian row in [row_begin, row_begin + row_count)
 * (all [P, ED] pointers are the tables' base addresses; a data-parallel caller walks the table in a few row ranges so that
 * the all-reduce of one range overlaps the kernels of the others):
 *   dL/dz  = sum of its flagged partial rows (+ gz_dense[P, ED] if not NULL)
 *   dL/dy  = gy[P, ED] (or NULL) + the sparse part (gy_slot[P], gy_merged) made by iso_rows_compact (or NULL, NULL);
 *            every slot entry the pass reads is reset to -1: once all rows have been walked the table is clean again and the
 *            next iso_rows_compact may be told so (slot_is_clean) instead of filling 4 P bytes
 *   dL/dx  = chain of dL/dz and dL/dy through  y = x/(|x|+eps1), z = y/(|y|+eps2)
 *            (scene/gaussian_model.py:122-125 and gaussian_renderer/__init__.py:61-62)
 *   grad_out != NULL:  grad_out = dL/dx, nothing else is written (a data-parallel caller all-reduces it);
 *   grad_out == NULL:  torch.optim.Adam step (lr, betas, eps, step counted from 1; scene/gaussian_model.py:249) on
 *                      x / exp_avg / exp_avg_sq in place, and y, z of the UPDATED rows are written for the next forward
 *                      (y may be NULL: not stored; iso_gather_rownorm recomputes the rows a caller needs). */
int isr_feature_rows_step(int P, int row_begin, int row_count, int64_t num_rendered, int ED, const void* geom_buffer,
                          const void* rows_scratch,
                          const float* gz_dense, const float* gy, int* gy_slot, const float* gy_merged, float eps1,
                          float eps2, float* x, float* grad_out, double lr, double beta1, double beta2, double eps,
                          long long step, float* exp_avg, float* exp_avg_sq, float* y, float* z, void* stream);

/* The same, writing - when z_scale != NULL - only the two row-normalisation factors (q1, q2) of every updated row into
 * z_scale [P, 2] instead of the normalised copy z [P, ED] of the table (z may then be NULL): z = (x q1) q2, which
 * isr_forward_render_scaled applies to the rows it stages.  One [P, ED] stream less per step.  isr_row_scales computes the
 * factors of a table no step has touched yet (the same expressions: the same bits). */
int isr_feature_rows_step_scaled(int P, int row_begin, int row_count, int64_t num_rendered, int ED, const void* geom_buffer,
                                 const void* rows_scratch,
                                 const float* gz_dense, const float* gy, int* gy_slot, const float* gy_merged, float eps1,
                                 float eps2, float* x, float* grad_out, double lr, double beta1, double beta2, double eps,
                                 long long step, float* exp_avg, float* exp_avg_sq, float* y, float* z, float* z_scale, void* stream);
int isr_row_scales(int P, int ED, float eps1, float eps2, const float* x, float* z_scale, void* stream);

/* ---- extension: everything of a train_semantic.py iteration behind the blend in ONE host call (reference
 * train_semantic.py:118-129 the two single-view losses on 2 B sampled pixels, :175-190 the 3-D loss on B sampled Gaussians,
 * loss.backward() and the optimiser step :203-204): iso_gather_rownorm -> iso_contrastive_forward_batch ->
 * iso_contrastive_backward_batch -> iso_rows_compact -> isr_backward_sampled (rows only) -> isr_feature_rows_step_scaled, the
 * launches the separate entry points would make, in that order, on `stream` - no torch, no autograd graph, no allocation.
 *   pixels [2 B] (y * W + x), sampled [2 B, ED] = isr_sample_extra of the forward's feature map at them; labels_a / labels_b [B]
 *   (int64): the two single-view label sets; pick3d / labels3d [B] (int64; NULL or w_3d == 0: no 3-D loss); class_feat [K, ED] or NULL;
 *   losses: w_a * L(sampled[:B], labels_a, cluster means) + w_b * L(sampled[B:], labels_b, class_feat) + w_3d * L(normalize(x)[pick3d],
 *   labels3d, class_feat); loss_parts [3] and loss_total [1] receive them; dL_dloss: device scalar (1.0).
 *   Workspaces (caller-owned, uninitialised): loss_state >= nb * iso_contrastive_scratch_bytes(B, ED, K); rows3d, merged [B, ED];
 *   grad_rows [3 B, ED]; chain [B]; slot [P] + slot_is_clean as for iso_rows_compact; bwd_scratch >=
 *   isr_backward_sampled_scratch_bytes(num_rendered, ED, 2 B, width, height).  x / exp_avg / exp_avg_sq / z / z_scale / lr ... as
 *   for isr_feature_rows_step_scaled. */
int isr_seg_step_tail(int P, int ED, int K, int B, int width, int height, int mode, int64_t num_rendered,
                      const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                      const long long* pixels, const float* sampled, const long long* labels_a, const long long* labels_b,
                      const long long* pick3d, const long long* labels3d, const float* class_feat,
                      float w_a, float w_b, float w_3d, float temp_lambda,
                      float* x, float* exp_avg, float* exp_avg_sq, float* z, float* z_scale, double lr, double beta1, double beta2,
                      double eps, long long step, float eps1, float eps2, int* slot, int slot_is_clean,
                      void* loss_state, size_t loss_state_bytes, float* rows3d, float* grad_rows, float* merged, int* chain,
                      void* bwd_scratch, size_t bwd_scratch_bytes, const float* dL_dloss, float* loss_parts, float* loss_total,
                      void* wait_before_rows /* hipEvent_t or NULL: `stream` waits for it in front of the per-Gaussian tail */,
                      void* stream);

/* ---- extension: everything of a train.py iteration behind render() in ONE host call (reference train.py:89-103 the loss - L1 + SSIM,
 * normal consistency, distortion -, :104 loss.backward(), :153-156 optimizer.step()): iso_train_loss_forward ->
 * iso_train_loss_backward -> iso_render_post_backward -> isr_backward (geometry) -> iso_gaussian_adam_step, the launches the separate
 * entry points would make, in that order, on `stream`.  image / allmap: the forward's out_color [3,H,W] / out_others [7,H,W];
 * rend_normal, surf_normal, rend_dist, surf_depth: render()'s derived maps (iso_render_post_forward); shs: the ACTIVATED [P,M,3]
 * tensor the forward consumed (likewise scales, rotations, and opacity inside the state buffers); params / exp_avg / exp_avg_sq /
 * lr: the six groups as for iso_gaussian_adam_step, a_*: the next forward's activations.  Everything from loss5 on is caller-owned
 * workspace / output: loss5 [5] (total, L1, SSIM, normal error, distortion), dmaps [3,3,H,W], loss_scratch
 * (iso_train_loss_scratch_bytes), d_image [3,H,W], d_rend_normal / d_surf_normal [3,H,W], d_rend_dist [H W], post_scratch [6,H,W],
 * d_allmap [7,H,W], the nine gradient tables of isr_backward, bwd_scratch (isr_backward_scratch_bytes(num_rendered, 0,
 * ISR_GRAD_GEOMETRY)); dL_dloss: device scalar (1.0). */
int isr_rgb_step_tail(int P, int D, int M, int width, int height, int mode, int64_t num_rendered,
                      const float* image, const float* gt, const float* allmap, const float* rend_normal, const float* surf_normal,
                      const float* rend_dist, const float* surf_depth, float lambda_dssim, float lambda_normal, float lambda_dist,
                      float depth_ratio, const float* rays_d, const float* rays_o,
                      const float* background, const float* means3D, const float* shs, const float* scales, float scale_modifier,
                      const float* rotations, const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, const int* radii, const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                      float* const params[6], float* const exp_avg[6], float* const exp_avg_sq[6], const double lr[6], double beta1,
                      double beta2, double eps, long long step, float* a_shs, float* a_opacity, float* a_scale, float* a_rotation,
                      float* loss5, float* dmaps, void* loss_scratch, size_t loss_scratch_bytes, float* d_image, float* d_rend_normal,
                      float* d_surf_normal, float* d_rend_dist, float* post_scratch, float* d_allmap, float* dL_dmean2D,
                      float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh,
                      float* dL_dscale, float* dL_drot, void* bwd_scratch, size_t bwd_scratch_bytes, const float* dL_dloss,
                      void* stream);

/* ---- rasterizer_impl.cu:141-153 */
int isr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* ---- introspection for the parity tests (device->host copies of integer state;
 * any output pointer may be NULL).  Host pointers. */
int isr_debug_state(int P, int width, int height, int64_t num_rendered,
                    const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                    uint32_t* tiles_touched /*[P]*/, uint32_t* point_list /*[R]*/,
                    uint32_t* ranges /*[tiles,2]*/, uint32_t* n_contrib /*[2,N]*/,
                    float* final_T /*[3,N]*/, float* splat_records /*[P,20]*/, void* stream);

/* Test infrastructure: checks k_pack_hits' per-(tile entry, block half) hit masks of a prepared view against the blend kernels' own
 * per-pixel evaluation.  Adds into device_counters (u64[8], zeroed by the caller): [0] halves whose bit is clear, [1] pixels of those
 * halves whose pair FAST's test passes, [2] pixels where EXACT's pair test passes (both must be 0: a clear bit never hides a pair a
 * kernel would blend), [3] halves whose bit is set, [4] of those, halves with no such pixel (what an ideal test would
 * also clear).  Reference: the per-pair skip tests of forward.cu:356-393. */
int isr_debug_check_hit_masks(int P, int width, int height, int64_t num_rendered, const void* geom_buffer, const void* binning_buffer,
                              const void* image_buffer, unsigned long long* device_counters, void* stream);

#ifdef __cplusplus
}
#endif
#endif

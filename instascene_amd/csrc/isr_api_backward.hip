// C-ABI entry points of libinstascene_hip.so, backward half: K9-K11 (declared in include/instascene_rasterizer.h).
// Host-side only: argument checking, workspace carving and kernel launches on the
// caller's stream.  No torch types, no allocation, no hidden synchronisation except
// where the header says so.
#include "isr_host.hpp"
#include "isr_backward.hip"

using namespace isr;

namespace isr { std::atomic<unsigned long long*> g_bwd_counters{nullptr}; }

extern "C" {

void isr_backward_set_counters(unsigned long long* device_counters) { isr::g_bwd_counters.store(device_counters); }

size_t isr_backward_scratch_bytes(int64_t num_rendered, int ED, unsigned grad_mask) {
    return backward_scratch_bytes(num_rendered, ED, grad_mask);
}

int isr_backward(int P, int D, int M, int64_t num_rendered, int ED, int width, int height, int mode, unsigned grad_mask,
                 const float* background, const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                 const float* extra_attrs, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                 float tan_fovx, float tan_fovy, const int* radii, const void* geom_buffer, const void* binning_buffer,
                 const void* image_buffer, const float* dL_dout_color, const float* dL_dout_others,
                 const float* dL_dout_extra, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 float* dL_dextra, void* scratch, size_t scratch_bytes, void* stream) {
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(ISR_EINVAL, "null state buffer");
    if (mode != ISR_MODE_EXACT && mode != ISR_MODE_FAST) return fail(ISR_EINVAL, "unknown mode %d", mode);
    if ((grad_mask & ~(ISR_GRAD_EXTRA | ISR_GRAD_GEOMETRY)) || grad_mask == 0) return fail(ISR_EINVAL, "bad grad_mask");
    if ((grad_mask & ISR_GRAD_EXTRA) && ED > 0 && !dL_dextra) return fail(ISR_EINVAL, "dL_dextra is NULL");
    if ((grad_mask & ISR_GRAD_GEOMETRY) &&
        (!dL_dmean2D || !dL_dnormal || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dtransMat || !dL_dscale ||
         !dL_drot || (shs && !dL_dsh)))
        return fail(ISR_EINVAL, "a geometry gradient output is NULL");
    if (scratch_bytes < backward_scratch_bytes(num_rendered, ED, grad_mask) || (!scratch && scratch_bytes))
        return fail(ISR_EINVAL, "backward scratch too small");
    const int rc = launch_backward(P, D, M, num_rendered, ED, width, height, mode, grad_mask, background, means3D, shs,
                           colors_precomp, scales, scale_modifier, rotations, transMat_precomp, extra_attrs, viewmatrix,
                           projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer,
                           dL_dout_color, dL_dout_others, dL_dout_extra, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor,
                           dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, dL_dextra, scratch, scratch_bytes,
                           (hipStream_t)stream);
    if (rc != 0) return ISR_EHIP;       // message set by the launcher
    return ISR_OK;
}

size_t isr_backward_sampled_scratch_bytes(int64_t num_rendered, int ED, int n_samples, int width, int height) {
    return backward_sampled_scratch_bytes(num_rendered, ED, n_samples, width, height);
}

int isr_sample_extra(int ED, int width, int height, int n_samples, const float* out_extra, const long long* pixels,
                     float* sampled, void* stream) {
    if (ED < 0 || width <= 0 || height <= 0 || n_samples < 0) return fail(ISR_EINVAL, "bad sample_extra sizes");
    if (n_samples > 0 && ED > 0 && (!out_extra || !pixels || !sampled)) return fail(ISR_EINVAL, "sample_extra: null pointer");
    if (launch_sample_gather(n_samples, ED, (long long)width * height, out_extra, pixels, sampled, (hipStream_t)stream) != 0)
        return ISR_EHIP;
    return ISR_OK;
}

int isr_backward_sampled(int P, int64_t num_rendered, int ED, int width, int height, int mode, int n_samples,
                         const long long* pixels, const float* dL_dsampled, const float* transMat_precomp,
                         const void* geom_buffer, const void* binning_buffer, const void* image_buffer, float* dL_dextra,
                         int accumulate, void* scratch, size_t scratch_bytes, void* stream) {
    if (P < 0 || ED <= 0 || width <= 0 || height <= 0 || n_samples < 0) return fail(ISR_EINVAL, "bad backward_sampled sizes");
    if (mode != ISR_MODE_EXACT && mode != ISR_MODE_FAST) return fail(ISR_EINVAL, "unknown mode %d", mode);
    if (!geom_buffer || !binning_buffer || !image_buffer || !scratch) return fail(ISR_EINVAL, "null buffer");
    if (n_samples > 0 && (!pixels || !dL_dsampled)) return fail(ISR_EINVAL, "backward_sampled: pixels / dL_dsampled required");
    if (scratch_bytes < backward_sampled_scratch_bytes(num_rendered, ED, n_samples, width, height))
        return fail(ISR_EINVAL, "backward_sampled scratch too small");
    const int rc = launch_backward_sampled(P, num_rendered, ED, width, height, mode, n_samples, pixels, dL_dsampled,
                                           transMat_precomp, geom_buffer, binning_buffer, image_buffer, dL_dextra, accumulate,
                                           scratch, (hipStream_t)stream);
    if (rc != 0) return ISR_EHIP;
    return ISR_OK;
}

int isr_feature_rows_step(int P, int row_begin, int row_count, int64_t num_rendered, int ED, const void* geom_buffer,
                          const void* rows_scratch,
                          const float* gz_dense, const float* gy, int* gy_slot, const float* gy_merged, float eps1,
                          float eps2, float* x, float* grad_out, double lr, double beta1, double beta2, double eps,
                          long long step, float* exp_avg, float* exp_avg_sq, float* y, float* z, void* stream) {
    return isr_feature_rows_step_scaled(P, row_begin, row_count, num_rendered, ED, geom_buffer, rows_scratch, gz_dense, gy, gy_slot,
                                        gy_merged, eps1, eps2, x, grad_out, lr, beta1, beta2, eps, step, exp_avg, exp_avg_sq, y, z,
                                        nullptr, stream);
}

int isr_row_scales(int P, int ED, float eps1, float eps2, const float* x, float* z_scale, void* stream) {
    if (P < 0 || ED <= 0 || (ED & 3) != 0 || ED > 256) return fail(ISR_EINVAL, "row_scales needs ED % 4 == 0 and ED <= 256");
    if (P > 0 && (!x || !z_scale)) return fail(ISR_EINVAL, "row_scales: null pointer");
    return launch_row_scales(P, ED, eps1, eps2, x, z_scale, (hipStream_t)stream) != 0 ? ISR_EHIP : ISR_OK;
}

int isr_feature_rows_step_scaled(int P, int row_begin, int row_count, int64_t num_rendered, int ED, const void* geom_buffer,
                                 const void* rows_scratch,
                                 const float* gz_dense, const float* gy, int* gy_slot, const float* gy_merged, float eps1,
                                 float eps2, float* x, float* grad_out, double lr, double beta1, double beta2, double eps,
                                 long long step, float* exp_avg, float* exp_avg_sq, float* y, float* z, float* z_scale, void* stream) {
    if (P < 0 || ED <= 0 || (ED & 3) != 0 || ED > 256) return fail(ISR_EINVAL, "feature_rows_step needs ED % 4 == 0 and ED <= 256");
    if (row_begin < 0 || row_count < 0 || row_begin + row_count > P) return fail(ISR_EINVAL, "feature_rows_step: bad row range");
    if (P == 0 || row_count == 0) return ISR_OK;
    if (!x || (rows_scratch && !geom_buffer) || ((gy_slot != nullptr) != (gy_merged != nullptr)))
        return fail(ISR_EINVAL, "feature_rows_step: null pointer");
    float lr_over_bc1 = 0.f, inv_sqrt_bc2 = 0.f;
    if (grad_out == nullptr) {
        if (!exp_avg || !exp_avg_sq || (!z && !z_scale)) return fail(ISR_EINVAL, "feature_rows_step: Adam state / outputs required");
        if (step < 1) return fail(ISR_EINVAL, "feature_rows_step: step counts from 1");
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        lr_over_bc1 = (float)(lr / bc1);
        inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    const int rc = launch_feature_rows_step(P, row_begin, row_count, num_rendered, ED, geom_buffer, rows_scratch, gz_dense, gy, gy_slot, gy_merged, eps1,
                                            eps2, x, grad_out, lr_over_bc1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                                            inv_sqrt_bc2, (float)eps, exp_avg, exp_avg_sq, y, z, z_scale, (hipStream_t)stream);
    if (rc != 0) return ISR_EHIP;
    return ISR_OK;
}

// Everything of a train_semantic.py iteration that follows the blend (reference train_semantic.py:118-129, 175-201 and the
// optimiser step :203-204), as ONE host call: the launches are those of the separate entry points, in the order the autograd
// graph of the Python trainer issues them - the same kernels with the same arguments, hence the same bits.
int isr_seg_step_tail(int P, int ED, int K, int B, int width, int height, int mode, int64_t num_rendered,
                      const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                      const long long* pixels, const float* sampled, const long long* labels_a, const long long* labels_b,
                      const long long* pick3d, const long long* labels3d, const float* class_feat,
                      float w_a, float w_b, float w_3d, float temp_lambda,
                      float* x, float* exp_avg, float* exp_avg_sq, float* z, float* z_scale, double lr, double beta1, double beta2,
                      double eps, long long step, float eps1, float eps2, int* slot, int slot_is_clean,
                      void* loss_state, size_t loss_state_bytes, float* rows3d, float* grad_rows, float* merged, int* chain,
                      void* bwd_scratch, size_t bwd_scratch_bytes, const float* dL_dloss, float* loss_parts, float* loss_total,
                      void* wait_before_rows, void* stream) {
    if (P <= 0 || ED <= 0 || (ED & 3) != 0 || ED > 256 || K <= 0 || B <= 0) return fail(ISR_EINVAL, "seg_step_tail: bad sizes");
    if (!geom_buffer || !binning_buffer || !image_buffer || !pixels || !sampled || !labels_a || !labels_b || !x || !exp_avg ||
        !exp_avg_sq || (!z && !z_scale) || !loss_state || !grad_rows || !bwd_scratch || !dL_dloss || !loss_parts || !loss_total)
        return fail(ISR_EINVAL, "seg_step_tail: null pointer");
    const bool has3d = pick3d != nullptr && labels3d != nullptr && w_3d != 0.0f;
    if (has3d && (!rows3d || !merged || !chain || !slot)) return fail(ISR_EINVAL, "seg_step_tail: the 3-D loss needs rows3d / merged / chain / slot");
    const int nb = has3d ? 3 : 2;
    if (loss_state_bytes < (size_t)nb * iso_contrastive_scratch_bytes(B, ED, K)) return fail(ISR_EINVAL, "seg_step_tail: loss_state too small");
    int rc;
    // the 3-D loss' rows: normalize(x)[pick3d] gathered from the raw parameter (:183-190)
    if (has3d && (rc = iso_gather_rownorm(B, ED, P, eps1, x, pick3d, rows3d, stream)) != 0) return rc;
    const size_t BF = (size_t)B * ED;
    const float* feats[3] = {sampled, sampled + BF, rows3d};
    const void* labs[3] = {labels_a, labels_b, labels3d};
    const float* pre[3] = {nullptr, class_feat, class_feat};
    const float w[3] = {w_a, w_b, w_3d};
    if ((rc = iso_contrastive_forward_batch(nb, B, ED, K, feats, labs, 1, pre, 0, 0, temp_lambda, w, loss_parts, loss_total, loss_state,
                                            loss_state_bytes, stream)) != 0) return rc;
    const int flags[3] = {0, class_feat != nullptr ? 1 : 0, class_feat != nullptr ? 1 : 0};
    float* outs[3] = {grad_rows, grad_rows + BF, grad_rows + 2 * BF};
    if ((rc = iso_contrastive_backward_batch(nb, B, ED, K, flags, dL_dloss, w, outs, loss_state, loss_state_bytes, stream)) != 0) return rc;
    // dL/d normalize(x) rows of the 3-D loss, repeats merged in index order (sparse: the tail takes slot + merged)
    if (has3d && (rc = iso_rows_compact(B, ED, P, pick3d, outs[2], slot, merged, chain, slot_is_clean, stream)) != 0) return rc;
    // the 2 B sampled pixels' gradient through the blend, left as per-(tile, Gaussian) rows
    if ((rc = isr_backward_sampled(P, num_rendered, ED, width, height, mode, 2 * B, pixels, grad_rows, nullptr, geom_buffer, binning_buffer,
                                   image_buffer, nullptr, 0, bwd_scratch, bwd_scratch_bytes, stream)) != 0) return rc;
    if (wait_before_rows != nullptr && hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_before_rows, 0) != hipSuccess)
        return fail(ISR_EHIP, "seg_step_tail: hipStreamWaitEvent failed");
    // row reduction + both normalisations' chain rule + Adam + the next forward's normalisation
    return isr_feature_rows_step_scaled(P, 0, P, num_rendered, ED, geom_buffer, bwd_scratch, nullptr, nullptr, has3d ? slot : nullptr,
                                        has3d ? merged : nullptr, eps1, eps2, x, nullptr, lr, beta1, beta2, eps, step, exp_avg, exp_avg_sq,
                                        nullptr, z, z_scale, stream);
}

// Everything of a train.py iteration behind render() (reference train.py:89-103 the loss, :104 loss.backward(), :153-156 the
// optimiser step) as ONE host call: the launches of the separate entry points in the order the autograd graph of the Python trainer
// issues them - the same kernels with the same arguments, hence the same bits.
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
                      void* stream) {
    if (P <= 0 || width <= 0 || height <= 0) return fail(ISR_EINVAL, "rgb_step_tail: bad sizes");
    if (!image || !gt || !allmap || !loss5 || !dmaps || !loss_scratch || !d_image || !d_allmap || !dL_dloss || !params || !exp_avg ||
        !exp_avg_sq || !lr)
        return fail(ISR_EINVAL, "rgb_step_tail: null pointer");
    const bool use_n = rend_normal != nullptr && surf_normal != nullptr && lambda_normal != 0.0f;
    const bool use_d = rend_dist != nullptr && lambda_dist != 0.0f;
    if (use_n && (!d_rend_normal || !d_surf_normal || !post_scratch || !surf_depth || !rays_d || !rays_o))
        return fail(ISR_EINVAL, "rgb_step_tail: the normal term needs d_rend_normal / d_surf_normal / post_scratch / surf_depth / rays");
    if (use_d && !d_rend_dist) return fail(ISR_EINVAL, "rgb_step_tail: the distortion term needs d_rend_dist");
    int rc;
    // the loss (photometric + both regularisers) and its gradient on the image and the derived maps
    if ((rc = iso_train_loss_forward(3, height, width, image, gt, lambda_dssim, use_n ? rend_normal : nullptr, use_n ? surf_normal : nullptr,
                                     use_n ? lambda_normal : 0.0f, use_d ? rend_dist : nullptr, use_d ? lambda_dist : 0.0f, loss5, dmaps,
                                     loss_scratch, loss_scratch_bytes, stream)) != 0) return rc;
    if ((rc = iso_train_loss_backward(3, height, width, image, gt, dmaps, lambda_dssim, use_n ? rend_normal : nullptr,
                                      use_n ? surf_normal : nullptr, use_n ? lambda_normal : 0.0f, use_d ? lambda_dist : 0.0f, dL_dloss,
                                      d_image, use_n ? d_rend_normal : nullptr, use_n ? d_surf_normal : nullptr,
                                      use_d ? d_rend_dist : nullptr, stream)) != 0) return rc;
    // render()'s derived maps back to the rasterizer's seven-channel map
    const float* dO = nullptr;
    if (use_n || use_d) {
        if ((rc = iso_render_post_backward(width, height, depth_ratio, allmap, viewmatrix, rays_d, rays_o, surf_depth, nullptr,
                                           use_n ? d_rend_normal : nullptr, use_d ? d_rend_dist : nullptr, nullptr,
                                           use_n ? d_surf_normal : nullptr, nullptr, nullptr, use_n ? post_scratch : nullptr, d_allmap,
                                           stream)) != 0) return rc;
        dO = d_allmap;
    }
    // the blend's and the per-Gaussian backward
    if ((rc = isr_backward(P, D, M, num_rendered, 0, width, height, mode, ISR_GRAD_GEOMETRY, background, means3D, shs, nullptr, scales,
                           scale_modifier, rotations, nullptr, nullptr, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii,
                           geom_buffer, binning_buffer, image_buffer, d_image, dO, nullptr, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor,
                           dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, nullptr, bwd_scratch, bwd_scratch_bytes, stream)) != 0)
        return rc;
    // chain rule of the getters + Adam on the six groups + the next forward's activations
    return iso_gaussian_adam_step(P, M, params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, dL_dmean3D, dL_dsh, dL_dopacity, dL_dscale,
                                  dL_drot, a_shs, a_opacity, a_scale, a_rotation, stream);
}

int isr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float*, uint8_t* present, void* stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(ISR_EINVAL, "null argument");
    if (P == 0) return ISR_OK;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, means3D, viewmatrix, present);
    ISR_LAUNCH_CHECK_S("k_mark_visible", (hipStream_t)stream);
    return ISR_OK;
}

}  // extern "C"

// C-ABI entry points of libinstascene_hip.so, the operators around the rasterizer (declared in include/instascene_ops.h).
// Host-side only: argument checking, workspace carving and kernel launches on the
// caller's stream.  No torch types, no allocation, no hidden synchronisation except
// where the header says so.
#include <algorithm>
#include <cstring>
#include "isr_host.hpp"
#include "iso_knn.hip"
#include "iso_contrastive.hip"
#include "iso_post.hip"
#include "iso_ssim.hip"
#include "iso_optim.hip"

using namespace isr;

extern "C" {

size_t iso_knn_scratch_bytes(int P) { return iso::knn_bytes(P); }

int iso_dist2_3nn(int P, const float* points, float* mean_dist2, void* scratch, size_t scratch_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!points || !mean_dist2 || !scratch))) return fail(ISR_EINVAL, "null argument");
    if (P == 0) return ISR_OK;
    if (scratch_bytes < iso::knn_bytes(P)) return fail(ISR_EINVAL, "knn scratch too small");
    iso::KnnView v = iso::knn_view(scratch, P);
    const int ccap = iso::knn_cell_cap(P);
    const int nblk = 256, pb = (P + 255) / 256;
    hipLaunchKernelGGL(iso::kk_minmax, dim3(nblk), dim3(256), 0, s, P, points, v.partial);
    hipLaunchKernelGGL(iso::kk_setup, dim3(1), dim3(64), 0, s, P, nblk, ccap, v.partial, v.grid);
    ISR_HIP(hipMemsetAsync(v.count, 0, sizeof(uint32_t) * ccap, s));
    ISR_HIP(hipMemsetAsync(v.cursor, 0, sizeof(uint32_t) * ccap, s));
    hipLaunchKernelGGL(iso::kk_count, dim3(pb), dim3(256), 0, s, P, points, v.grid, v.count);
    if (launch_scan_u32(ccap, v.count, v.offset, v.sums, s) != 0)      // the whole capacity: unused cells hold zero
        return fail(ISR_EHIP, "iso_dist2_3nn: scan launch failed");
    hipLaunchKernelGGL(iso::kk_scatter, dim3(pb), dim3(256), 0, s, P, points, v.grid, v.offset, v.cursor, v.sorted);
    // ISO_KNN_LDS=1: the LDS-bucketed query (built as north_star names it, measured 0.8-8x SLOWER than the ring walk from
    // global memory - profiles/r02_knn_timing.json - because the cell-sorted points are L2-resident anyway and a clustered
    // cloud overflows any fixed LDS budget; kept for A/B)
    static const bool lds_query = [] { const char* e = getenv("ISO_KNN_LDS"); return e && e[0] == '1'; }();
    if (lds_query)      // a fixed grid strides over the (segment, row) work items: the grid's dimensions live on the device
        hipLaunchKernelGGL(iso::kk_query_lds, dim3(16384), dim3(256), 0, s, P, v.grid, v.offset, v.count, v.sorted, mean_dist2);
    else
        hipLaunchKernelGGL(iso::kk_query, dim3(pb), dim3(256), 0, s, P, v.grid, v.offset, v.count, v.sorted, mean_dist2);
    ISR_LAUNCH_CHECK("iso_dist2_3nn");
    return ISR_OK;
}

size_t iso_contrastive_scratch_bytes(int N, int F, int K) { return iso::cstate_bytes(N < 1 ? 1 : N, F < 1 ? 1 : F, K < 1 ? 1 : K); }

static int contrastive_forward_impl(int nb, int N, int F, int K, const float* const* features, const void* const* labels,
                                    int labels_are_int64, const float* const* predef_u, int consider_negative,
                                    int min_pixnum, float temp_lambda, const float* weights, float* loss,
                                    float* loss_total, void* state, size_t state_bytes, hipStream_t s) {
    if (nb < 1 || nb > iso::CK_MAXB) return fail(ISR_EINVAL, "contrastive batch of %d (1..%d supported)", nb, iso::CK_MAXB);
    if (N <= 0 || F <= 0 || K <= 0 || !features || !labels || !loss || !state) return fail(ISR_EINVAL, "bad contrastive arguments");
    if (F > 1024) return fail(ISR_EINVAL, "feature dimension %d > 1024 unsupported", F);
    const size_t one = iso::cstate_bytes(N, F, K);
    if (state_bytes < one * (size_t)nb) return fail(ISR_EINVAL, "contrastive state too small");
    iso::CKBatch bt = {};
    bt.stride = (long long)one;
    for (int b = 0; b < nb; b++) {
        if (!features[b] || !labels[b]) return fail(ISR_EINVAL, "bad contrastive arguments (problem %d)", b);
        bt.x[b] = features[b];
        bt.labels[b] = labels[b];
        bt.predef[b] = predef_u ? predef_u[b] : nullptr;
        bt.w[b] = weights ? weights[b] : 1.0f;
    }
    iso::CState st = iso::cstate(state, N, F, K);
    const int shift = consider_negative ? 0 : 1;
    const int nblk = (N + 31) / 32, nt = (N + 255) / 256;
    int* ticket_phi = st.hist + K + 2;
    int* ticket_loss = st.hist + K + 3;
    int* ticket_total = st.hist + K + 4;          // problem 0's
    bool any_mean = false;
    for (int b = 0; b < nb; b++) any_mean = any_mean || bt.predef[b] == nullptr;
    hipLaunchKernelGGL(iso::ck_zero, dim3(1, nb), dim3(256), 0, s, K + 6, st.hist, bt);       // histogram + tickets
    hipLaunchKernelGGL(iso::ck_count, dim3(nt, nb), dim3(256), 0, s, N, K, shift, labels_are_int64, st.hist, bt);
    ISR_STAGE("ck_count", s);
    int lpr = 1;
    if ((F & 3) == 0) while (lpr < (F >> 2) && lpr < 64) lpr <<= 1;
    hipLaunchKernelGGL(iso::ck_normalize, dim3((unsigned)(((long long)N * lpr + 255) / 256), nb), dim3(256), 0, s, N, F, K, shift,
                       consider_negative, min_pixnum, labels_are_int64, st.hist, st.f, st.inv, st.col, lpr, bt);
    ISR_STAGE("ck_normalize", s);
    if (any_mean)
        hipLaunchKernelGGL(iso::ck_gemm_tn, dim3((K + 31) / 32, (F + 31) / 32, iso::CK_NSPLIT * nb), dim3(256), 0, s, N, F, K, 1,
                           st.col, (const float*)nullptr, st.f, st.split, bt);
    hipLaunchKernelGGL(iso::ck_finish_u, dim3((K * F + 255) / 256, nb), dim3(256), 0, s, F, K, min_pixnum, st.hist, st.split,
                       st.U, st.cnt, bt);
    ISR_STAGE("ck_gemm_tn/ck_finish_u", s);
    hipLaunchKernelGGL(iso::ck_phi, dim3(nt, nb), dim3(256), 0, s, N, F, K, st.f, st.col, st.U, st.cnt, temp_lambda, st.phi_part,
                       ticket_phi, st.phi, st.Us, bt);
    ISR_STAGE("ck_phi", s);
    if (F <= 64 && K <= 96) {
        static const int last_max = [] { const char* e = getenv("ISR_CK_LAST_MAX"); return e ? atoi(e) : 0; }();
        // (default: never - A/B on one box at the headline's 256 x 3 workgroups: 1.660 / 1.664 ms per step with the last-workgroup
        // form, 1.650 / 1.647 with ck_loss_reduce; at 1 024 x 3 the kernel takes 135 us against 32)
        const bool by_last = nblk <= last_max;       // (workgroups per problem)
#define ISO_SIM2(NT, LAST_, FS_)                                                                                                          \
    hipLaunchKernelGGL((iso::ck_similarity_small<NT, LAST_, FS_>), dim3(nblk, nb), dim3(64), 0, s, N, F, K, st.f, st.U, st.phi, st.cnt, \
                       st.col, st.G, st.part, ticket_loss, loss, loss_total, ticket_total, bt)
#define ISO_SIM(NT)                                                                                                                \
    do { if (F > 32) ISO_SIM2(NT, false, 32); else if (by_last) ISO_SIM2(NT, true, 16); else ISO_SIM2(NT, false, 16); } while (0)
        if (K <= 32) ISO_SIM(1); else if (K <= 64) ISO_SIM(2); else ISO_SIM(3);
#undef ISO_SIM
#undef ISO_SIM2
        if (!by_last || F > 32)
            hipLaunchKernelGGL(iso::ck_loss_reduce, dim3(1, nb), dim3(256), 0, s, nblk, st.part, loss, loss_total, ticket_total, bt);
    } else {
        hipLaunchKernelGGL(iso::ck_similarity, dim3(nblk, nb), dim3(64), 0, s, N, F, K, st.f, st.U, st.phi, st.cnt, st.col, st.G,
                           st.part, bt);
        hipLaunchKernelGGL(iso::ck_loss_reduce, dim3(1, nb), dim3(256), 0, s, nblk, st.part, loss, loss_total, ticket_total, bt);
    }
    ISR_STAGE("ck_similarity", s);
    ISR_LAUNCH_CHECK("iso_contrastive_forward");
    return ISR_OK;
}

static int contrastive_backward_impl(int nb, int N, int F, int K, const int* prototypes_predefined, const float* dL_dloss,
                                     const float* weights, float* const* dL_dfeatures, void* state, size_t state_bytes,
                                     hipStream_t s) {
    if (nb < 1 || nb > iso::CK_MAXB) return fail(ISR_EINVAL, "contrastive batch of %d (1..%d supported)", nb, iso::CK_MAXB);
    if (N <= 0 || F <= 0 || K <= 0 || !dL_dloss || !dL_dfeatures || !state || !prototypes_predefined)
        return fail(ISR_EINVAL, "bad contrastive arguments");
    const size_t one = iso::cstate_bytes(N, F, K);
    if (state_bytes < one * (size_t)nb) return fail(ISR_EINVAL, "contrastive state too small");
    iso::CKBatch bt = {};
    bt.stride = (long long)one;
    bool any_mean = false;
    for (int b = 0; b < nb; b++) {
        if (!dL_dfeatures[b]) return fail(ISR_EINVAL, "bad contrastive arguments (problem %d)", b);
        bt.dX[b] = dL_dfeatures[b];
        bt.predef[b] = prototypes_predefined[b] ? reinterpret_cast<const float*>(state) : nullptr;   // only its null-ness is read
        bt.w[b] = weights ? weights[b] : 1.0f;
        any_mean = any_mean || !prototypes_predefined[b];
    }
    iso::CState st = iso::cstate(state, N, F, K);
    if (any_mean) {
        hipLaunchKernelGGL(iso::ck_gemm_tn, dim3((K + 31) / 32, (F + 31) / 32, iso::CK_NSPLIT * nb), dim3(256), 0, s, N, F, K, 0,
                           st.col, st.G, st.f, st.split, bt);
        hipLaunchKernelGGL(iso::ck_finish_du, dim3((K * F + 255) / 256, nb), dim3(256), 0, s, F, K, st.split, st.phi, st.cnt,
                           st.dU, bt);
    }
    hipLaunchKernelGGL(iso::ck_grad_f, dim3((N + 31) / 32, nb), dim3(64), 0, s, N, F, K, st.G, st.Us, st.cnt, st.dU, st.col,
                       st.inv, dL_dloss, bt);
    ISR_LAUNCH_CHECK("iso_contrastive_backward");
    return ISR_OK;
}

int iso_contrastive_forward(int N, int F, int K, const float* features, const void* labels, int labels_are_int64,
                            const float* predef_u, int consider_negative, int min_pixnum, float temp_lambda, float* loss,
                            void* state, size_t state_bytes, void* stream) {
    return contrastive_forward_impl(1, N, F, K, &features, &labels, labels_are_int64, &predef_u, consider_negative, min_pixnum,
                                    temp_lambda, nullptr, loss, nullptr, state, state_bytes, (hipStream_t)stream);
}

int iso_contrastive_backward(int N, int F, int K, int prototypes_predefined, const float* dL_dloss, float* dL_dfeatures,
                             void* state, size_t state_bytes, void* stream) {
    return contrastive_backward_impl(1, N, F, K, &prototypes_predefined, dL_dloss, nullptr, &dL_dfeatures, state, state_bytes,
                                     (hipStream_t)stream);
}

int iso_contrastive_forward_batch(int nb, int N, int F, int K, const float* const* features, const void* const* labels,
                                  int labels_are_int64, const float* const* predef_u, int consider_negative, int min_pixnum,
                                  float temp_lambda, const float* weights, float* loss, float* loss_total, void* state,
                                  size_t state_bytes, void* stream) {
    return contrastive_forward_impl(nb, N, F, K, features, labels, labels_are_int64, predef_u, consider_negative, min_pixnum,
                                    temp_lambda, weights, loss, loss_total, state, state_bytes, (hipStream_t)stream);
}

int iso_contrastive_backward_batch(int nb, int N, int F, int K, const int* prototypes_predefined, const float* dL_dloss,
                                   const float* weights, float* const* dL_dfeatures, void* state, size_t state_bytes,
                                   void* stream) {
    return contrastive_backward_impl(nb, N, F, K, prototypes_predefined, dL_dloss, weights, dL_dfeatures, state, state_bytes,
                                     (hipStream_t)stream);
}

int iso_rownorm(long long N, int F, float eps, int backward, const float* x, const float* dy, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || F <= 0 || (N > 0 && (!x || !out || (backward && !dy)))) return fail(ISR_EINVAL, "bad rownorm arguments");
    if (N == 0) return ISR_OK;
    if ((F & 3) == 0 && F <= 1024) {
        int q = F >> 2, lpr = 1;
        while (lpr < q && lpr < 64) lpr <<= 1;
        const long long threads = N * lpr;
        const unsigned blocks = (unsigned)((threads + 255) / 256);
        if (backward) hipLaunchKernelGGL(iso::rn_kernel<true>, dim3(blocks), dim3(256), 0, s, N, F, eps, x, dy, out);
        else hipLaunchKernelGGL(iso::rn_kernel<false>, dim3(blocks), dim3(256), 0, s, N, F, eps, x, dy, out);
    } else {
        hipLaunchKernelGGL(iso::rn_scalar, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, F, eps, backward, x, dy, out);
    }
    ISR_LAUNCH_CHECK("iso_rownorm");
    return ISR_OK;
}

int iso_render_post_forward(int W, int H, float depth_ratio, const float* allmap, const float* viewmatrix,
                            const float* rays_d, const float* rays_o, float* rend_alpha, float* rend_normal,
                            float* rend_dist, float* surf_depth, float* surf_normal, float* rend_depth,
                            float* rend_median, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (W <= 0 || H <= 0) return fail(ISR_EINVAL, "bad image size %dx%d", W, H);
    if (!allmap || !viewmatrix || !rays_d || !rays_o || !rend_alpha || !rend_normal || !rend_dist || !surf_depth ||
        !surf_normal || !rend_depth || !rend_median)
        return fail(ISR_EINVAL, "render_post_forward: null pointer");
    const long long N = (long long)W * H;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    { ProfScope ps_("pp_maps", s);
    hipLaunchKernelGGL(iso::pp_maps, dim3(blocks), dim3(256), 0, s, N, depth_ratio, 1.0f - depth_ratio, allmap, viewmatrix,
                       rend_alpha, rend_normal, rend_dist, surf_depth, rend_depth, rend_median); }
    ISR_LAUNCH_CHECK("pp_maps");
    { ProfScope ps_("pp_surf_normal", s);
    hipLaunchKernelGGL(iso::pp_surf_normal, dim3(blocks), dim3(256), 0, s, W, H, surf_depth, rend_alpha, rays_d, rays_o,
                       surf_normal); }
    ISR_LAUNCH_CHECK("pp_surf_normal");
    return ISR_OK;
}

int iso_render_post_backward(int W, int H, float depth_ratio, const float* allmap, const float* viewmatrix,
                             const float* rays_d, const float* rays_o, const float* surf_depth, const float* g_alpha,
                             const float* g_normal, const float* g_dist, const float* g_surf_depth,
                             const float* g_surf_normal, const float* g_depth, const float* g_median, float* scratch,
                             float* dL_dallmap, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (W <= 0 || H <= 0) return fail(ISR_EINVAL, "bad image size %dx%d", W, H);
    if (!allmap || !viewmatrix || !rays_d || !rays_o || !surf_depth || !dL_dallmap)
        return fail(ISR_EINVAL, "render_post_backward: null pointer");
    if (g_surf_normal && !scratch) return fail(ISR_EINVAL, "render_post_backward: scratch [6,H,W] needed with g_surf_normal");
    const long long N = (long long)W * H;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (g_surf_normal) {
        hipLaunchKernelGGL(iso::pp_bwd_stencil, dim3(blocks), dim3(256), 0, s, W, H, surf_depth, allmap + N, rays_d, rays_o,
                           g_surf_normal, scratch);
        ISR_LAUNCH_CHECK("pp_bwd_stencil");
    }
    hipLaunchKernelGGL(iso::pp_bwd_maps, dim3(blocks), dim3(256), 0, s, W, H, depth_ratio, 1.0f - depth_ratio, allmap,
                       viewmatrix, rays_d, g_surf_normal ? scratch : (const float*)nullptr, g_alpha, g_normal, g_dist,
                       g_surf_depth, g_depth, g_median, dL_dallmap);
    ISR_LAUNCH_CHECK("pp_bwd_maps");
    return ISR_OK;
}

static iso::SsimTaps ssim_taps() {
    iso::SsimTaps t;
    double g[11], sum = 0.0;
    for (int i = 0; i < 11; i++) { g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < 11; i++) t.w[i] = (float)(g[i] / sum);
    return t;
}

size_t iso_ssim_scratch_bytes(int C, int H, int W) {
    const size_t nb = (size_t)((W + iso::SS_TW - 1) / iso::SS_TW) * ((H + iso::SS_TH - 1) / iso::SS_TH) * (size_t)(C > 0 ? C : 1);
    return 2 * nb * sizeof(float) + 256;        // SSIM-map and |a - b| block sums
}

int iso_photometric_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_mean, float* l1_mean,
                            float* dmaps, void* scratch, size_t scratch_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_mean || !scratch) return fail(ISR_EINVAL, "bad ssim arguments");
    if (scratch_bytes < iso_ssim_scratch_bytes(C, H, W)) return fail(ISR_EINVAL, "ssim scratch too small");
    const dim3 grid((W + iso::SS_TW - 1) / iso::SS_TW, (H + iso::SS_TH - 1) / iso::SS_TH, C);
    const int nb = (int)(grid.x * grid.y * grid.z);
    { ProfScope ps_("ssim_fwd", s);
    const iso::TrainReg no_reg = {nullptr, nullptr, nullptr};
    if (l1_mean) hipLaunchKernelGGL((iso::ssim_fwd<true, false>), grid, dim3(256), 0, s, C, H, W, ssim_taps(), img1, img2, (float*)scratch, dmaps, no_reg);
    else hipLaunchKernelGGL((iso::ssim_fwd<false, false>), grid, dim3(256), 0, s, C, H, W, ssim_taps(), img1, img2, (float*)scratch, dmaps, no_reg); }
    hipLaunchKernelGGL(iso::ssim_sum_parts, dim3(l1_mean ? 2 : 1), dim3(256), 0, s, nb, (const float*)scratch,
                       (float)(1.0 / ((double)C * H * W)), ssim_mean, l1_mean);
    ISR_LAUNCH_CHECK("iso_photometric_forward");
    return ISR_OK;
}

int iso_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_mean, float* dmaps,
                     void* scratch, size_t scratch_bytes, void* stream) {
    return iso_photometric_forward(C, H, W, img1, img2, ssim_mean, nullptr, dmaps, scratch, scratch_bytes, stream);
}

int iso_photometric_backward(int C, int H, int W, const float* img1, const float* img2, const float* dmaps,
                             const float* g_ssim, const float* g_l1, float* dL_dimg1, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dmaps || !g_ssim || !dL_dimg1) return fail(ISR_EINVAL, "bad ssim arguments");
    const dim3 grid((W + iso::SS_TW - 1) / iso::SS_TW, (H + iso::SS_TH - 1) / iso::SS_TH, C);
    { ProfScope ps_("ssim_bwd", s);
    const iso::TrainReg no_reg = {nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(iso::ssim_bwd, grid, dim3(256), 0, s, C, H, W, ssim_taps(), img1, img2, dmaps, g_ssim, g_l1,
                       (float)(1.0 / ((double)C * H * W)), dL_dimg1, 1.0f, 1.0f, no_reg, 0.0f, 0.0f, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr); }
    ISR_LAUNCH_CHECK("iso_photometric_backward");
    return ISR_OK;
}

size_t iso_train_loss_scratch_bytes(int C, int H, int W) {
    const size_t tiles = (size_t)((W + iso::SS_TW - 1) / iso::SS_TW) * ((H + iso::SS_TH - 1) / iso::SS_TH);
    return (2 * tiles * (size_t)(C > 0 ? C : 1) + 2 * tiles) * sizeof(float) + 256;
}

int iso_train_loss_forward(int C, int H, int W, const float* image, const float* gt, float lambda_dssim,
                           const float* rend_normal, const float* surf_normal, float lambda_normal, const float* rend_dist,
                           float lambda_dist, float* out5, float* dmaps, void* scratch, size_t scratch_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !image || !gt || !out5 || !scratch) return fail(ISR_EINVAL, "bad train_loss arguments");
    if ((rend_normal != nullptr) != (surf_normal != nullptr)) return fail(ISR_EINVAL, "train_loss: rend_normal and surf_normal come together");
    if (scratch_bytes < iso_train_loss_scratch_bytes(C, H, W)) return fail(ISR_EINVAL, "train_loss scratch too small");
    const dim3 grid((W + iso::SS_TW - 1) / iso::SS_TW, (H + iso::SS_TH - 1) / iso::SS_TH, C);
    const int tiles = (int)(grid.x * grid.y), nb = tiles * C;
    const iso::TrainReg reg = {rend_normal, surf_normal, rend_dist};
    { ProfScope ps_("ssim_fwd", s);
    hipLaunchKernelGGL((iso::ssim_fwd<true, true>), grid, dim3(256), 0, s, C, H, W, ssim_taps(), image, gt, (float*)scratch, dmaps, reg); }
    hipLaunchKernelGGL(iso::train_loss_sum, dim3(1), dim3(256), 0, s, nb, tiles, (const float*)scratch,
                       (float)(1.0 / ((double)C * H * W)), (float)(1.0 / ((double)H * W)), lambda_dssim,
                       rend_normal ? lambda_normal : 0.0f, rend_dist ? lambda_dist : 0.0f, out5);
    ISR_LAUNCH_CHECK("iso_train_loss_forward");
    return ISR_OK;
}

int iso_train_loss_backward(int C, int H, int W, const float* image, const float* gt, const float* dmaps, float lambda_dssim,
                            const float* rend_normal, const float* surf_normal, float lambda_normal, float lambda_dist,
                            const float* g_total, float* dL_dimage, float* dL_drend_normal, float* dL_dsurf_normal,
                            float* dL_drend_dist, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !image || !gt || !dmaps || !g_total || !dL_dimage) return fail(ISR_EINVAL, "bad train_loss arguments");
    if ((dL_drend_normal != nullptr) != (dL_dsurf_normal != nullptr)) return fail(ISR_EINVAL, "train_loss: both normal gradients or none");
    if (dL_drend_normal && (!rend_normal || !surf_normal)) return fail(ISR_EINVAL, "train_loss: normal maps required for their gradients");
    const dim3 grid((W + iso::SS_TW - 1) / iso::SS_TW, (H + iso::SS_TH - 1) / iso::SS_TH, C);
    const iso::TrainReg reg = {rend_normal, surf_normal, nullptr};
    const float inv_hw = (float)(1.0 / ((double)H * W));
    { ProfScope ps_("ssim_bwd", s);
    hipLaunchKernelGGL(iso::ssim_bwd, grid, dim3(256), 0, s, C, H, W, ssim_taps(), image, gt, dmaps, g_total, g_total,
                       (float)(1.0 / ((double)C * H * W)), dL_dimage, -lambda_dssim, 1.0f - lambda_dssim, reg,
                       lambda_normal * inv_hw, lambda_dist * inv_hw, dL_drend_normal, dL_dsurf_normal, dL_drend_dist); }
    ISR_LAUNCH_CHECK("iso_train_loss_backward");
    return ISR_OK;
}

int iso_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dmaps, const float* g_mean,
                      float* dL_dimg1, void* stream) {
    return iso_photometric_backward(C, H, W, img1, img2, dmaps, g_mean, nullptr, dL_dimg1, stream);
}

int iso_densify_stats(int P, int C, const float* viewspace_grad, const unsigned char* visible, const int* radii,
                      float* grad_accum, float* denom, float* max_radii, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || C <= 0) return fail(ISR_EINVAL, "bad densify_stats sizes");
    if (P == 0) return ISR_OK;
    if (!viewspace_grad || !visible || !radii || !grad_accum || !denom || !max_radii) return fail(ISR_EINVAL, "densify_stats: null pointer");
    hipLaunchKernelGGL(iso::densify_stats, dim3((P + 255) / 256), dim3(256), 0, s, P, C, viewspace_grad, visible, radii,
                       grad_accum, denom, max_radii);
    ISR_LAUNCH_CHECK("iso_densify_stats");
    return ISR_OK;
}

int iso_adam_rownorm2(long long N, int F, double lr, double beta1, double beta2, double eps, long long step, float eps1,
                      float eps2, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* y, float* z,
                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || F <= 0 || (F & 3) != 0 || F > 256) return fail(ISR_EINVAL, "adam_rownorm2 needs F % 4 == 0 and F <= 256");
    if (step < 1) return fail(ISR_EINVAL, "adam_rownorm2: step counts from 1");
    if (N > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !z)) return fail(ISR_EINVAL, "adam_rownorm2: null pointer");
    if (N == 0) return ISR_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    int q = F >> 2, lpr = 1;
    while (lpr < q) lpr <<= 1;
    const unsigned blocks = (unsigned)((N * lpr + 255) / 256);
    // hyper-parameters are doubles like torch's: 1 - 0.999f would be off by 5e-5 relative
    hipLaunchKernelGGL(iso::adam_rn2_kernel, dim3(blocks), dim3(256), 0, s, N, F, (float)(lr / bc1), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)(1.0 / sqrt(bc2)), (float)eps, eps1, eps2, param, grad,
                       exp_avg, exp_avg_sq, y, z);
    ISR_LAUNCH_CHECK("iso_adam_rownorm2");
    return ISR_OK;
}

int iso_gaussian_adam_step(int P, int M, float* const params[6], float* const exp_avg[6], float* const exp_avg_sq[6],
                           const double lr[6], double beta1, double beta2, double eps, long long step, const float* g_xyz,
                           const float* g_shs, const float* g_opacity, const float* g_scale, const float* g_rotation,
                           float* a_shs, float* a_opacity, float* a_scale, float* a_rotation, void* stream) {
    if (P < 0 || M < 1 || M > 64) return fail(ISR_EINVAL, "gaussian_adam_step: bad sizes P=%d M=%d", P, M);
    if (step < 1) return fail(ISR_EINVAL, "gaussian_adam_step: step counts from 1");
    if (P == 0) return ISR_OK;
    if (!params || !exp_avg || !exp_avg_sq || !lr) return fail(ISR_EINVAL, "gaussian_adam_step: null table");
    iso::GaussAdamArgs a = {};
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    for (int g = 0; g < 6; g++) {
        if (g == 2 && M == 1) continue;                     // no higher-order coefficients
        if (!params[g] || !exp_avg[g] || !exp_avg_sq[g]) return fail(ISR_EINVAL, "gaussian_adam_step: group %d has a null tensor", g);
        a.p[g] = params[g]; a.m[g] = exp_avg[g]; a.v[g] = exp_avg_sq[g];
        a.lr_over_bc1[g] = (float)(lr[g] / bc1);
    }
    a.g_xyz = g_xyz; a.g_shs = g_shs; a.g_opa = g_opacity; a.g_scale = g_scale; a.g_rot = g_rotation;
    a.a_shs = a_shs; a.a_opa = a_opacity; a.a_scale = a_scale; a.a_rot = a_rotation;
    a.om1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.om2 = (float)(1.0 - beta2);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.eps = (float)eps;
    a.P = P; a.M = M;
    const long long threads = (long long)P * (3 + 3LL * M + 1 + 2 + 1);
    hipLaunchKernelGGL(iso::gaussian_adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    ISR_LAUNCH_CHECK_S("iso_gaussian_adam_step", (hipStream_t)stream);
    return ISR_OK;
}

int iso_gather_rownorm(int n, int F, long long P, float eps, const float* x, const long long* idx, float* out, void* stream) {
    if (n < 0 || F <= 0 || (F & 3) != 0 || F > 256 || P < 0) return fail(ISR_EINVAL, "gather_rownorm needs F % 4 == 0 and F <= 256");
    if (n == 0) return ISR_OK;
    if (!x || !idx || !out) return fail(ISR_EINVAL, "gather_rownorm: null pointer");
    int q = F >> 2, lpr = 1;
    while (lpr < q) lpr <<= 1;
    hipLaunchKernelGGL(iso::gather_rownorm_kernel, dim3((unsigned)(((long long)n * lpr + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n, F, P, eps, x, idx, out);
    ISR_LAUNCH_CHECK_S("iso_gather_rownorm", (hipStream_t)stream);
    return ISR_OK;
}

int iso_sample_step(unsigned long long seed, unsigned long long step, int B, long long n_pool2d, const long long* pool2d,
                    const long long* segmap_a, const long long* segmap_b, long long n_pool3d, const long long* pool3d,
                    const long long* labels3d, long long* pix, long long* lab_a, long long* lab_b, long long* pick3d,
                    long long* lab3d, void* stream) {
    if (B < 0 || n_pool2d < 0 || n_pool3d < 0) return fail(ISR_EINVAL, "sample_step: bad sizes");
    if (B == 0) return ISR_OK;
    if (n_pool2d > 0 && (!pool2d || !segmap_a || !segmap_b || !pix || !lab_a || !lab_b)) return fail(ISR_EINVAL, "sample_step: null 2-D argument");
    if (n_pool3d > 0 && (!pool3d || !labels3d || !pick3d || !lab3d)) return fail(ISR_EINVAL, "sample_step: null 3-D argument");
    hipLaunchKernelGGL(iso::sample_step_kernel, dim3((3 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, step, B,
                       n_pool2d, pool2d, segmap_a, segmap_b, n_pool3d, pool3d, labels3d, pix, lab_a, lab_b, pick3d, lab3d);
    ISR_LAUNCH_CHECK_S("iso_sample_step", (hipStream_t)stream);
    return ISR_OK;
}

int iso_rows_compact(int n, int F, long long P, const long long* idx, const float* vals, int* slot, float* merged,
                     int* chain, int slot_is_clean, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || F <= 0 || P < 0) return fail(ISR_EINVAL, "rows_compact: bad sizes");
    if (P > 0 && !slot) return fail(ISR_EINVAL, "rows_compact: null slot table");
    if (n > 0 && (!idx || !vals || !merged || !chain)) return fail(ISR_EINVAL, "rows_compact: null pointer");
    if (P > 0 && !slot_is_clean && hipMemsetAsync(slot, 0xFF, sizeof(int) * (size_t)P, s) != hipSuccess) return fail(ISR_EHIP, "rows_compact: memset failed");
    if (n == 0 || P == 0) return ISR_OK;
    if (n > iso::ROWS_SPLIT_MAX) return fail(ISR_EINVAL, "rows_compact: at most %d rows", iso::ROWS_SPLIT_MAX);
    static const bool lds_form = [] { const char* e = getenv("ISR_ROWS_COMPACT"); return e && e[0] == 'l'; }();    // "lds": the one-launch form
    if (lds_form && n <= iso::ROWS_COMPACT_MAX) {
        hipLaunchKernelGGL(iso::rows_compact_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, n, F, P, idx, vals, slot, merged);
        ISR_LAUNCH_CHECK("iso_rows_compact");
        return ISR_OK;
    }
    unsigned* us = reinterpret_cast<unsigned*>(slot);
    const int pos_bits = iso::rows_pos_bits(n);
    hipLaunchKernelGGL(iso::rows_first_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, P, idx, us);
    ISR_LAUNCH_CHECK("rows_first_kernel");
    const long long quads = ((long long)n * F + 3) / 4;
    hipLaunchKernelGGL(iso::rows_copy_kernel, dim3((unsigned)((std::max<long long>(quads, n) + 255) / 256)), dim3(256), 0, s, n, F, P, idx, vals, us, merged, pos_bits, reinterpret_cast<unsigned*>(chain));
    ISR_LAUNCH_CHECK("rows_copy_kernel");
    hipLaunchKernelGGL(iso::rows_merge_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, F, P, idx, vals, us, merged, pos_bits, reinterpret_cast<const unsigned*>(chain));
    ISR_LAUNCH_CHECK("rows_merge_kernel");
    return ISR_OK;
}

int iso_rownorm2(long long N, int F, float eps1, float eps2, int backward, const float* x, const float* gy,
                 const float* gz, float* out1, float* out2, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || F <= 0 || (F & 3) != 0 || F > 256) return fail(ISR_EINVAL, "rownorm2 needs F % 4 == 0 and F <= 256");
    if (N > 0 && (!x || (backward && !out1) || (!backward && !out2))) return fail(ISR_EINVAL, "bad rownorm2 arguments");
    if (N == 0) return ISR_OK;
    int q = F >> 2, lpr = 1;
    while (lpr < q) lpr <<= 1;
    const unsigned blocks = (unsigned)((N * lpr + 255) / 256);
    if (backward) hipLaunchKernelGGL(iso::rn2_kernel<true>, dim3(blocks), dim3(256), 0, s, N, F, eps1, eps2, x, gy, gz, out1, out2);
    else hipLaunchKernelGGL(iso::rn2_kernel<false>, dim3(blocks), dim3(256), 0, s, N, F, eps1, eps2, x, gy, gz, out1, out2);
    ISR_LAUNCH_CHECK("iso_rownorm2");
    return ISR_OK;
}

int iso_peer_sum(int W, const float* const* sources, long long begin, long long count, float* dst, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (W < 1 || W > 16 || begin < 0 || count < 0) return fail(ISR_EINVAL, "peer_sum: 1..16 sources, non-negative range");
    if (count == 0) return ISR_OK;
    if (!sources || !dst) return fail(ISR_EINVAL, "peer_sum: null pointer");
    if (((uintptr_t)dst & 15) != 0) return fail(ISR_EINVAL, "peer_sum: dst must be 16-byte aligned");
    iso::PeerPtrs p = {};
    for (int w = 0; w < W; w++) {
        if (!sources[w] || ((uintptr_t)sources[w] & 15) != 0) return fail(ISR_EINVAL, "peer_sum: source %d is null or not 16-byte aligned", w);
        p.src[w] = sources[w];
    }
    hipLaunchKernelGGL(iso::peer_sum_kernel, dim3((unsigned)((count + 1023) / 1024)), dim3(256), 0, s, W, p, begin, count, dst);
    ISR_LAUNCH_CHECK("iso_peer_sum");
    return ISR_OK;
}


// ---- fine-grained, IPC-shareable device memory for the exchange's flags / headers ---------------------------------------------
int iso_ipc_alloc(size_t bytes, void** ptr, void* handle64) {
    if (!ptr || !handle64 || bytes == 0) return fail(ISR_EINVAL, "ipc_alloc: null pointer or zero size");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, bytes); }          // (no fine-grained pool: plain device memory)
    if (e != hipSuccess) return fail(ISR_EHIP, "ipc_alloc: %s", hipGetErrorString(e));
    if ((e = hipMemset(p, 0, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) { (void)hipFree(p); return fail(ISR_EHIP, "ipc_alloc: %s", hipGetErrorString(e)); }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    hipIpcMemHandle_t h;
    if ((e = hipIpcGetMemHandle(&h, p)) != hipSuccess) { (void)hipFree(p); return fail(ISR_EHIP, "ipc_alloc: hipIpcGetMemHandle: %s", hipGetErrorString(e)); }
    memcpy(handle64, &h, 64);
    *ptr = p;
    return ISR_OK;
}
int iso_ipc_open(const void* handle64, void** ptr) {
    if (!ptr || !handle64) return fail(ISR_EINVAL, "ipc_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(ISR_EHIP, "ipc_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return ISR_OK;
}
int iso_ipc_close(void* ptr, int owner) {
    const hipError_t e = owner ? hipFree(ptr) : hipIpcCloseMemHandle(ptr);
    if (e != hipSuccess) return fail(ISR_EHIP, "ipc_close: %s", hipGetErrorString(e));
    return ISR_OK;
}
int iso_enable_peer_access(int peer_device) {
    int dev = 0, can = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(ISR_EHIP, "enable_peer_access: no current device");
    if (peer_device == dev) return ISR_OK;
    if (hipDeviceCanAccessPeer(&can, dev, peer_device) != hipSuccess || !can)
        return fail(ISR_EHIP, "device %d cannot access device %d", dev, peer_device);
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(ISR_EHIP, "hipDeviceEnablePeerAccess(%d): %s", peer_device, hipGetErrorString(e));
    (void)hipGetLastError();
    return ISR_OK;
}
int iso_flag_set(void* flag, unsigned value, void* stream) {
    if (!flag) return fail(ISR_EINVAL, "flag_set: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(iso::flag_set_kernel, dim3(1), dim3(1), 0, s, (unsigned*)flag, value);
    ISR_LAUNCH_CHECK("iso_flag_set");
    return ISR_OK;
}
int iso_flag_wait(int W, const void* const* flags, int skip, unsigned value, void* status, int timeout_ms, void* stream) {
    if (W < 1 || W > 16 || !flags || !status) return fail(ISR_EINVAL, "flag_wait: 1..16 flags and a status word");
    iso::FlagPtrs p = {};
    for (int w = 0; w < W; w++) {
        if (!flags[w]) return fail(ISR_EINVAL, "flag_wait: flag %d is null", w);
        p.f[w] = (const unsigned*)flags[w];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(iso::flag_wait_kernel, dim3(1), dim3(64), 0, s, W, p, skip, value, (unsigned*)status,
                       (long long)timeout_ms * 100000ll);           // wall_clock64: 100 MHz
    ISR_LAUNCH_CHECK("iso_flag_wait");
    return ISR_OK;
}
int iso_rows_pack(int P, int F, const unsigned char* touched, const float* grad, int* idx, float* rows, int* count, void* stream) {
    if (P < 0 || F < 1) return fail(ISR_EINVAL, "rows_pack: P >= 0, F >= 1");
    if (!touched || !grad || !idx || !rows || !count) return fail(ISR_EINVAL, "rows_pack: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) return fail(ISR_EHIP, "rows_pack: memset");
    if (P > 0) hipLaunchKernelGGL(iso::rows_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, F, touched, grad, idx, rows, count);
    ISR_LAUNCH_CHECK("iso_rows_pack");
    return ISR_OK;
}
int iso_rows_scatter_add(int F, int P, long long max_rows, const int* count, const int* idx, const float* rows, float* dst, int assign, void* stream) {
    if (F < 1 || P < 0 || max_rows < 0) return fail(ISR_EINVAL, "rows_scatter_add: F >= 1, P >= 0");
    if (!count || !idx || !rows || !dst) return fail(ISR_EINVAL, "rows_scatter_add: null pointer");
    if (max_rows == 0) return ISR_OK;
    const long long work = max_rows * ((F + 3) / 4);
    const unsigned grid = (unsigned)(work + 255 < 256ll * 8192 ? (work + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(iso::rows_scatter_add_kernel, dim3(grid), dim3(256), 0, s, F, P, count, idx, rows, dst, assign);
    ISR_LAUNCH_CHECK("iso_rows_scatter_add");
    return ISR_OK;
}

}  // extern "C"

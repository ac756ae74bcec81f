/*
 * instascene_ops.h — C-ABI of the two companion ops on the InstaScene hot path.
 *
 *   iso_dist2_3nn            replaces  simple_knn._C.distCUDA2
 *                            (submodules/simple-knn/spatial.cu:15-26, simple_knn.cu:186-222)
 *   iso_contrastive_forward/ replace the arithmetic core of
 *   iso_contrastive_backward utils/contrastive_utils.py:41-71 (contrastive_loss) after the
 *                            label filtering / dense relabelling of :25-50, which stays on the
 *                            host side (instascene_amd/contrastive.py) exactly as in the reference.
 *
 * All pointers are DEVICE pointers to contiguous row-major data; `stream` is a
 * hipStream_t (NULL = default stream).  Return 0 on success, negative on error
 * (isr_last_error() in instascene_rasterizer.h carries the message).
 */
#ifndef INSTASCENE_OPS_H
#define INSTASCENE_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mean of the squared distances to the 3 nearest neighbours of every point (self excluded).
 * Exact (equal to brute force; the reference's Morton-box pruning is exact too).  Points
 * with fewer than 3 neighbours keep FLT_MAX terms like the reference (P < 4). */
size_t iso_knn_scratch_bytes(int P);
int iso_dist2_3nn(int P, const float* points /*[P,3]*/, float* mean_dist2 /*[P]*/, void* scratch,
                  size_t scratch_bytes, void* stream);

/* ProtoNCE loss on N samples:
 *   f_i   = x_i / (|x_i| + 1e-9)                       (norm detached, :41)
 *   u_k   = mean_{i in k} f_i   or  predef_u[k]         (:44-45,:54-58)
 *   phi_k = clip(10 * sum_{i in k}|f_i - u_k| / (n_k * log(n_k + temp_lambda)), 0.5, 1)   (detached, :60-66)
 *   loss  = - sum_i log( exp(f_i.u_{y_i}/phi_{y_i}) / (sum_k exp(f_i.u_k/phi_k) + 1e-9) )  (:68-71)
 * `state` (iso_contrastive_scratch_bytes) is the opaque forward->backward hand-off.  predef_u may be NULL.
 * No host synchronisation (the reference's three torch.unique calls per loss each force one).
 * The similarity  [N,F].[F,K]  and the backward products run on the matrix cores with
 * the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32). */
size_t iso_contrastive_scratch_bytes(int N, int F, int K);
/* labels: the RAW label of every sample (int64 if labels_are_int64 else int32), exactly what the reference's
 * `masks` argument holds.  Samples are dropped like the reference drops them (:25-40): label <= 0 unless
 * consider_negative, and labels with <= min_pixnum samples.  K bounds the column ids: column = label - 1
 * (label when consider_negative) must be < K to survive; predef_u (if given) has K rows indexed the same way. */
int iso_contrastive_forward(int N, int F, int K, const float* features /*[N,F]*/, const void* labels /*[N]*/,
                            int labels_are_int64, const float* predef_u /*[K,F] or NULL*/, int consider_negative,
                            int min_pixnum, float temp_lambda, float* loss /*[1]*/, void* state, size_t state_bytes,
                            void* stream);
/* dL_dfeatures[N,F] = dL/dloss * d loss / d features  (dL_dloss: device scalar); zero rows for dropped samples. */
int iso_contrastive_backward(int N, int F, int K, int prototypes_predefined, const float* dL_dloss /*[1]*/,
                             float* dL_dfeatures /*[N,F]*/, void* state, size_t state_bytes, void* stream);

/* nb (1..4) losses of the same shape (N, F, K, flags) in one sequence of launches — train_semantic.py:118-190 evaluates
 * two single-view losses and the 3-D loss per iteration, each a handful of microsecond-sized kernels.  features / labels /
 * predef_u / dL_dfeatures are HOST arrays of nb device pointers (predef_u may be NULL or hold NULL entries = cluster
 * means); state holds nb consecutive blocks of iso_contrastive_scratch_bytes(N, F, K).  loss[b] = weights[b] * loss_b;
 * loss_total (may be NULL) = their sum in problem order.  The backward multiplies dL_dloss by weights[b]. */
int iso_contrastive_forward_batch(int nb, int N, int F, int K, const float* const* features, const void* const* labels,
                                  int labels_are_int64, const float* const* predef_u, int consider_negative, int min_pixnum,
                                  float temp_lambda, const float* weights /*host [nb] or NULL = 1*/, float* loss /*[nb]*/,
                                  float* loss_total /*[1] or NULL*/, void* state, size_t state_bytes, void* stream);
int iso_contrastive_backward_batch(int nb, int N, int F, int K, const int* prototypes_predefined /*host [nb]*/,
                                   const float* dL_dloss /*[1]*/, const float* weights /*host [nb] or NULL*/,
                                   float* const* dL_dfeatures /*host [nb] of [N,F]*/, void* state, size_t state_bytes,
                                   void* stream);

/* Row normalisation y = x / (|x|_2 + eps) on [N,F] (scene/gaussian_model.py:122-125, gaussian_renderer/__init__.py:61-62):
 * backward == 0: out = y;  backward != 0: out = dL/dx given dy = dL/dy (x is the forward input). */
int iso_rownorm(long long N, int F, float eps, int backward, const float* x, const float* dy, float* out, void* stream);

/* The two chained normalisations the reference applies to the [P,F] feature every step (getter eps 1e-6, then
 * render() eps 1e-9) as ONE pass each way; F % 4 == 0, F <= 256.
 *   y = x/(|x|+eps1), z = y/(|y|+eps2)
 * backward == 0: out1 = y, out2 = z.   backward != 0: out1 = dL/dx from gy = dL/dy and gz = dL/dz (either may be NULL). */
int iso_rownorm2(long long N, int F, float eps1, float eps2, int backward, const float* x, const float* gy,
                 const float* gz, float* out1, float* out2, void* stream);

/* render()'s post-processing of the rasterizer's 7-channel allmap (gaussian_renderer/__init__.py:127-167 with
 * utils/point_utils.py:10-40) in two streaming kernels each way.  All maps are [C,H,W] row-major device arrays:
 *   rend_alpha[1] = allmap[1]          rend_normal[3] = R_view . allmap[2:5]     rend_dist[1] = allmap[6]
 *   rend_depth[1] = nan_to_num(allmap[0]/allmap[1])      rend_median[1] = nan_to_num(allmap[5])
 *   surf_depth[1] = rend_depth*(1-depth_ratio) + depth_ratio*rend_median
 *   surf_normal[3] = normalize(cross(P[y+1,x]-P[y-1,x], P[y,x+1]-P[y,x-1])) * alpha (alpha detached), 0 on the border,
 *                    P = surf_depth * rays_d + rays_o
 * viewmatrix: the row-major 4x4 world_view_transform of the reference camera; rays_d [H*W,3], rays_o [3]: the per-pixel
 * ray table of utils/point_utils.py:10-27 (static per camera).
 * backward: g_* are the upstream gradients of the seven maps (each may be NULL = zero); scratch: [6,H,W] floats, needed
 * when g_surf_normal is given; writes all of dL_dallmap[7,H,W].  Where allmap[0]/allmap[1] is not finite the gradient
 * is 0 (torch autograd gives NaN there). */
int iso_render_post_forward(int W, int H, float depth_ratio, const float* allmap, const float* viewmatrix,
                            const float* rays_d, const float* rays_o, float* rend_alpha, float* rend_normal,
                            float* rend_dist, float* surf_depth, float* surf_normal, float* rend_depth,
                            float* rend_median, void* stream);
int iso_render_post_backward(int W, int H, float depth_ratio, const float* allmap, const float* viewmatrix,
                             const float* rays_d, const float* rays_o, const float* surf_depth, const float* g_alpha,
                             const float* g_normal, const float* g_dist, const float* g_surf_depth,
                             const float* g_surf_normal, const float* g_depth, const float* g_median, float* scratch,
                             float* dL_dallmap, void* stream);

/* Mean structural similarity of two [C,H,W] images (utils/loss_utils.py:39-63: 11x11 Gaussian window, sigma 1.5, zero
 * padding, size_average=True) and its gradient with respect to img1.  The window is applied separably through LDS
 * (the reference issues five depthwise conv2d's and their transposes).  dmaps [3,C,H,W] (optional in forward, required
 * by backward): the three partial derivatives of the SSIM map that the backward blurs.  g_mean: device scalar
 * dL/d(ssim_mean).  scratch: iso_ssim_scratch_bytes. */
size_t iso_ssim_scratch_bytes(int C, int H, int W);
int iso_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_mean, float* dmaps,
                     void* scratch, size_t scratch_bytes, void* stream);
int iso_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dmaps, const float* g_mean,
                      float* dL_dimg1, void* stream);

/* The whole photometric term of train.py:91 in the same two kernels: additionally *l1_mean = mean |img1 - img2|
 * (utils/loss_utils.py:18-19) in the forward, and  g_l1 * sign(img1 - img2) / (C H W)  added to dL_dimg1 in the backward
 * (g_l1 = dL/d l1_mean, device scalar; NULL = SSIM only).  l1_mean NULL: exactly iso_ssim_forward. */
int iso_photometric_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_mean, float* l1_mean,
                            float* dmaps, void* scratch, size_t scratch_bytes, void* stream);
int iso_photometric_backward(int C, int H, int W, const float* img1, const float* img2, const float* dmaps,
                             const float* g_ssim, const float* g_l1, float* dL_dimg1, void* stream);

/* The whole loss of a train.py iteration (train.py:89-103) in the same two kernels + one single-workgroup sum:
 *   total = ((1 - l) L1(image, gt) + l (1 - SSIM(image, gt))) + lambda_dist mean(rend_dist)
 *           + lambda_normal mean(1 - sum_c rend_normal[c] surf_normal[c])
 * image, gt [C,H,W]; rend_normal, surf_normal [3,H,W] (both or neither); rend_dist [H,W] or NULL.
 * out5 = total, L1 mean, SSIM mean, normal-error mean, distortion mean.  dmaps [3,C,H,W] as for iso_ssim_forward.
 * backward: g_total = device scalar dL/dtotal; writes dL/dimage and (where the pointers are given) the regularisers'
 * gradients  -lambda_normal g surf_normal / (H W),  -lambda_normal g rend_normal / (H W),  lambda_dist g / (H W). */
size_t iso_train_loss_scratch_bytes(int C, int H, int W);
int iso_train_loss_forward(int C, int H, int W, const float* image, const float* gt, float lambda_dssim,
                           const float* rend_normal, const float* surf_normal, float lambda_normal, const float* rend_dist,
                           float lambda_dist, float* out5, float* dmaps, void* scratch, size_t scratch_bytes, void* stream);
int iso_train_loss_backward(int C, int H, int W, const float* image, const float* gt, const float* dmaps, float lambda_dssim,
                            const float* rend_normal, const float* surf_normal, float lambda_normal, float lambda_dist,
                            const float* g_total, float* dL_dimage, float* dL_drend_normal, float* dL_dsurf_normal,
                            float* dL_drend_dist, void* stream);

/* Densification statistics of one iteration (train.py:140-142; scene/gaussian_model.py:601-604): for every Gaussian i
 * with visible[i] != 0:  grad_accum[i] += |viewspace_grad[i, 0:C]|_2,  denom[i] += 1,
 * max_radii[i] = max(max_radii[i], radii[i]). */
int iso_densify_stats(int P, int C, const float* viewspace_grad, const unsigned char* visible, const int* radii,
                      float* grad_accum, float* denom, float* max_radii, void* stream);

/* Adam update of a [N,F] parameter (torch.optim.Adam arithmetic: no weight decay, no amsgrad; `step` counts from 1;
 * scene/gaussian_model.py:249 with the group's lr and eps) fused with the two chained row normalisations of the updated
 * rows (iso_rownorm2 forward): param / exp_avg / exp_avg_sq are updated in place, y and z receive the normalised rows the
 * next forward needs (y may be NULL: not stored, see iso_gather_rownorm).  F % 4 == 0, F <= 256. */
int iso_adam_rownorm2(long long N, int F, double lr, double beta1, double beta2, double eps, long long step, float eps1,
                      float eps2, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* y, float* z,
                      void* stream);

/* Gradient of a row gather  rows = table[idx]  (table [P,F]; train_semantic.py:183-190 samples the visible Gaussians'
 * features with replacement) kept sparse: slot[P] (int32) is set to -1, then for every distinct valid idx[i] the FIRST
 * position i becomes slot[idx[i]] = i and merged[i, :] = sum of vals[j, :] over all j with idx[j] == idx[i], ascending j
 * (the order of index_put_(accumulate=True)).  Rows of `merged` that are not a slot target are not written.
 * n <= 65536 (the reference's default sample_batchsize is 32 768).  Consumed by isr_feature_rows_step(gy_slot, gy_merged), which resets the entries it reads.
 * `chain`: n ints of scratch (the occurrences of a repeated row find each other through it).  slot_is_clean != 0
 * says the table is all -1 already (a persistent table whose last contents were consumed) and skips the fill. */
int iso_rows_compact(int n, int F, long long P, const long long* idx, const float* vals, int* slot /*[P]*/,
                     float* merged /*[n,F]*/, int* chain /*[n] scratch*/, int slot_is_clean, void* stream);

/* The optimiser step of the train.py loop (scene/gaussian_model.py:206-253: torch.optim.Adam(lr = 0, eps = 1e-15) over six
 * parameter groups; train.py:153-156) as ONE pass, with the chain rule in front of it and the next forward's activations
 * behind it.  Tables of six device pointers in the order xyz [P,3], f_dc [P,3] (the reference's [P,1,3]), f_rest
 * [P, 3(M-1)] (its [P,M-1,3]), opacity [P,1], scaling [P,2], rotation [P,4]; lr[6] likewise (per group, host doubles).
 * g_*: gradients with respect to the ACTIVATED tensors the rasterizer consumed - xyz, shs = cat(f_dc, f_rest) [P,M,3],
 * sigmoid(opacity), exp(scaling), normalize(rotation) (scene/gaussian_model.py:109-138) - a NULL one leaves its group(s)
 * untouched, like a parameter without .grad.  a_*: the same activations of the UPDATED parameters (any may be NULL).
 * Dense torch.optim.Adam arithmetic (no weight decay, no amsgrad; `step` counts from 1, shared by the groups). */
int iso_gaussian_adam_step(int P, int M, float* const params[6], float* const exp_avg[6], float* const exp_avg_sq[6],
                           const double lr[6], double beta1, double beta2, double eps, long long step, const float* g_xyz,
                           const float* g_shs, const float* g_opacity, const float* g_scale, const float* g_rotation,
                           float* a_shs, float* a_opacity, float* a_scale, float* a_rotation, void* stream);

/* out[i, :] = x[idx[i], :] / (|x[idx[i], :]|_2 + eps), i < n: rows of the normalised feature (scene/gaussian_model.py:122-125)
 * gathered straight from the raw parameter x[P, F] - bit-identical to gathering them from iso_rownorm2's / iso_adam_rownorm2's
 * `y`, which a trainer that only ever reads a few thousand rows of it (train_semantic.py:183-190) need not store.
 * F % 4 == 0, F <= 256; indices outside [0, P) give zero rows. */
int iso_gather_rownorm(int n, int F, long long P, float eps, const float* x, const long long* idx, float* out, void* stream);

/* The index sampling of one train_semantic.py iteration (:118-129, :163-168, :183-190) as one launch: pix[2B] = 2B uniform
 * draws with replacement from pool2d[n_pool2d] (flat indices of the view's labelled pixels), lab_a[B] = segmap_a[pix[:B]],
 * lab_b[B] = segmap_b[pix[B:]]; pick3d[B] = B draws from pool3d[n_pool3d] (visible labelled Gaussians), lab3d = labels3d[
 * pick3d].  Counter-based generator (splitmix64 of seed, step, draw): the same (seed, step) gives the same samples.  A
 * pool of size 0 leaves its outputs untouched. */
int iso_sample_step(unsigned long long seed, unsigned long long step, int B, long long n_pool2d, const long long* pool2d,
                    const long long* segmap_a, const long long* segmap_b, long long n_pool3d, const long long* pool3d,
                    const long long* labels3d, long long* pix, long long* lab_a, long long* lab_b, long long* pick3d,
                    long long* lab3d, void* stream);

/* dst[i] = sum_{w < W} sources[w][begin + i], i < count, added in the order w = 0 .. W-1: the reduce step of a direct
 * reduce-scatter over peer-mapped buffers (every rank pulls its shard from all peers at once over the
 * point-to-point xGMI links, instead of RCCL's ring).  `sources` is a HOST array of W <= 16 device pointers (the peers'
 * buffers as mapped into this process, e.g. hipIpcOpenMemHandle; the rank's own buffer for w = rank), 16-byte aligned like
 * `dst`.  Synchronisation between the ranks (buffers complete before they are read, reads complete before they are reused)
 * is the caller's (instascene_amd/peer_exchange.py). */
int iso_peer_sum(int W, const float* const* sources, long long begin, long long count, float* dst, void* stream);

/* ---- the direct exchange's control plane on the device (instascene_amd/peer_exchange.py) ------------
 * iso_ipc_alloc: `bytes` of zeroed fine-grained device memory (plain device memory where the driver has no fine-grained pool)
 * and its 64-byte hipIpcMemHandle, which another process turns into a mapping with iso_ipc_open; iso_ipc_close unmaps (owner = 0)
 * or frees (owner = 1).  iso_enable_peer_access: hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess from the current device.
 * iso_flag_set: after everything enqueued on `stream` so far, *flag = value (release, system scope).
 * iso_flag_wait: `stream` stalls until *flags[w] >= value (generation counters; signed distance) for every w < W except `skip`
 * (acquire, system scope; no host involvement); a flag that does not arrive within timeout_ms sets bit w of *status (device
 * memory, 32 bits) instead of hanging the device.
 * iso_rows_pack: the rows r with touched[r] != 0 of grad[P,F] -> idx[n], rows[n,F], *count = n (the buffers hold up to P rows).
 * iso_rows_scatter_add: dst[idx[e]] += rows[e] (or = with assign != 0) for e < *count; count / idx / rows may be a peer's memory;
 * max_rows bounds the launch (the list's capacity). */
int iso_ipc_alloc(size_t bytes, void** ptr, void* handle64);
int iso_ipc_open(const void* handle64, void** ptr);
int iso_ipc_close(void* ptr, int owner);
int iso_enable_peer_access(int peer_device);
int iso_flag_set(void* flag, unsigned value, void* stream);
int iso_flag_wait(int W, const void* const* flags, int skip, unsigned value, void* status, int timeout_ms, void* stream);
int iso_rows_pack(int P, int F, const unsigned char* touched, const float* grad, int* idx, float* rows, int* count, void* stream);
int iso_rows_scatter_add(int F, int P, long long max_rows, const int* count, const int* idx, const float* rows, float* dst, int assign,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif

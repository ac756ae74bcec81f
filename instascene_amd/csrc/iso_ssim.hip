// Structural similarity (reference utils/loss_utils.py:39-63: 11x11 Gaussian window, sigma 1.5, zero padding,
// C1 = 0.01^2, C2 = 0.03^2) for the train.py photometric loss, forward and backward with respect to the first image.
// The reference runs five depthwise conv2d's forward and their transposes backward (MIOpen: ~9 ms per 1080p step
// here); the window is separable, so one workgroup blurs a 32x8 output tile of all five moment maps through LDS
// (rows, then columns), evaluates the SSIM map and its three partial derivatives in registers, and reduces the map
// in fixed order.  Backward: the three derivative maps are blurred the same way (the zero-padded blur is symmetric).
#include "isr_common.hpp"

namespace iso {

constexpr int SS_R = 5;                      // window radius (11 taps)
constexpr int SS_TW = 32, SS_TH = 8;         // output tile: one pixel per thread of a 256-thread workgroup
constexpr int SS_IW = SS_TW + 2 * SS_R;      // 42 input columns
constexpr int SS_IH = SS_TH + 2 * SS_R;      // 18 input rows
struct SsimTaps { float w[2 * SS_R + 1]; };
static_assert(SS_TW * SS_TH == 256, "one output pixel per thread");

// Horizontal then vertical blur of NM maps held in LDS as in[m][SS_IH][SS_IW]; the thread (tx, ty) of the 32x8 tile
// receives the blurred values of its pixel in out[NM] (256 threads = 32 x 8 pixels).  hbuf[m][SS_IH][SS_TW] is scratch.
template <int NM>
__device__ __forceinline__ void tile_blur(const float* __restrict__ in, float* __restrict__ hbuf, const SsimTaps& tp,
                                          float (&out)[NM]) {
    const int tid = threadIdx.x;
    // rows: SS_IH x SS_TW outputs per map
    for (int e = tid; e < NM * SS_IH * SS_TW; e += 256) {
        const int m = e / (SS_IH * SS_TW), r = (e / SS_TW) % SS_IH, c = e % SS_TW;
        const float* src = in + (m * SS_IH + r) * SS_IW + c;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * SS_R; k++) acc = __builtin_fmaf(tp.w[k], src[k], acc);
        hbuf[e] = acc;
    }
    __syncthreads();
    const int tx = tid & (SS_TW - 1), ty = tid >> 5;
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const float* src = hbuf + (m * SS_IH + ty) * SS_TW + tx;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * SS_R; k++) acc = __builtin_fmaf(tp.w[k], src[k * SS_TW], acc);
        out[m] = acc;
    }
}

// grid (ceil(W/32), ceil(H/8), C).  map_part[block] = sum of the SSIM map over the block's pixels.
// dmaps (optional) [3][C][H][W]: d map / d mu1, d map / d E[a^2], d map / d E[ab].
// L1: also map_part[nblocks + block] = sum of |img1 - img2| over the block's pixels (the other half of train.py's
// photometric loss, utils/loss_utils.py:18-19: the images are in LDS already).
// REG (train.py:93-103, the geometric regularisers of the same step): the workgroups of channel 0 also sum
// 1 - <rend_normal, surf_normal> and rend_dist over their tile: map_part[2 nblocks + tile] and [2 nblocks + tiles + tile]
// (either pair of pointers may be NULL: that term is zero).
struct TrainReg { const float* rend_normal; const float* surf_normal; const float* rend_dist; };
template <bool L1, bool REG>
__global__ __launch_bounds__(256) void ssim_fwd(int C, int H, int W, SsimTaps tp, const float* __restrict__ img1,
                                                const float* __restrict__ img2, float* __restrict__ map_part,
                                                float* __restrict__ dmaps, TrainReg reg) {
    __shared__ float s_in[5 * SS_IH * SS_IW];      // a, b, a*a, b*b, a*b  (15.1 KB)
    __shared__ float s_h[5 * SS_IH * SS_TW];       // (11.5 KB)
    __shared__ float s_red[4];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_TW - SS_R, y0 = blockIdx.y * SS_TH - SS_R;
    const size_t plane = (size_t)H * W;
    for (int e = threadIdx.x; e < SS_IH * SS_IW; e += 256) {
        const int r = e / SS_IW, c = e - r * SS_IW;
        const int x = x0 + c, y = y0 + r;
        float a = 0.0f, b = 0.0f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            a = img1[ch * plane + (size_t)y * W + x];
            b = img2[ch * plane + (size_t)y * W + x];
        }
        s_in[e] = a; s_in[SS_IH * SS_IW + e] = b; s_in[2 * SS_IH * SS_IW + e] = a * a;
        s_in[3 * SS_IH * SS_IW + e] = b * b; s_in[4 * SS_IH * SS_IW + e] = a * b;
    }
    __syncthreads();
    float m[5];
    tile_blur<5>(s_in, s_h, tp, m);
    const int tx = threadIdx.x & (SS_TW - 1), ty = threadIdx.x >> 5;
    const int x = blockIdx.x * SS_TW + tx, y = blockIdx.y * SS_TH + ty;
    float val = 0.0f, l1v = 0.0f;
    if (x < W && y < H) {
        if (L1) {
            const int ce = (ty + SS_R) * SS_IW + tx + SS_R;
            l1v = fabsf(s_in[ce] - s_in[SS_IH * SS_IW + ce]);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1 = m[0], mu2 = m[1];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        const float n1 = 2.0f * mu12 + C1, n2 = 2.0f * s12 + C2;
        const float d1 = mu1_sq + mu2_sq + C1, d2 = s1 + s2 + C2;
        const float inv = 1.0f / (d1 * d2);
        val = n1 * n2 * inv;
        if (dmaps != nullptr) {
            // map = n1 n2 / (d1 d2) with  s1 = Eaa - mu1^2, s12 = Eab - mu1 mu2
            //   d/dEaa = -map / d2;   d/dEab = 2 n1 / (d1 d2)
            //   d/dmu1 = 2 mu2 (n2 - n1) / (d1 d2) - map * 2 mu1 (1/d1 - 1/d2)
            const size_t o = ch * plane + (size_t)y * W + x;
            const size_t cp = (size_t)C * plane;
            dmaps[o] = 2.0f * mu2 * (n2 - n1) * inv - val * 2.0f * mu1 * (1.0f / d1 - 1.0f / d2);
            dmaps[cp + o] = -val / d2;
            dmaps[2 * cp + o] = 2.0f * n1 * inv;
        }
    }
    // fixed-order block sum
    float v = val;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) map_part[blk] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    if (L1) {
        __syncthreads();
        float u = l1v;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = u;
        __syncthreads();
        if (threadIdx.x == 0) map_part[gridDim.x * gridDim.y * gridDim.z + blk] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
    if (REG && ch == 0) {
        float ne = 0.0f, dv = 0.0f;
        if (x < W && y < H) {
            const size_t o = (size_t)y * W + x;
            if (reg.rend_normal != nullptr) {
                const float dot = (reg.rend_normal[o] * reg.surf_normal[o] + reg.rend_normal[plane + o] * reg.surf_normal[plane + o]) +
                                  reg.rend_normal[2 * plane + o] * reg.surf_normal[2 * plane + o];
                ne = 1.0f - dot;
            }
            if (reg.rend_dist != nullptr) dv = reg.rend_dist[o];
        }
        const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
        float* rp = map_part + 2 * (size_t)tiles * gridDim.z;
#pragma unroll
        for (int which = 0; which < 2; which++) {
            __syncthreads();
            float u = which == 0 ? ne : dv;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o);
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = u;
            __syncthreads();
            if (threadIdx.x == 0) rp[which * tiles + tile] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}

// ONE workgroup: the four partial-sum arrays of ssim_fwd<true, true> -> out[0] = the loss of train.py:91-103,
//   ((1 - l) L1 + l (1 - SSIM)) + lambda_dist mean(rend_dist) + lambda_normal mean(1 - <n, n'>),
// out[1..4] = L1 mean, SSIM mean, normal-error mean, distortion mean.  Fixed summation order.
__global__ __launch_bounds__(256) void train_loss_sum(int nblk, int tiles, const float* __restrict__ part, float inv_chw,
                                                      float inv_hw, float lam, float ln, float ldist, float* __restrict__ out) {
    __shared__ float s_tot[4];
    // wave a sums array a (fixed order: 64 strided partial sums, then the butterfly)
    const int a = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* p = part + (a == 0 ? 0 : a == 1 ? nblk : a == 2 ? 2 * nblk : 2 * nblk + tiles);
    const int n = a < 2 ? nblk : tiles;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    int i = lane;
    for (; i + 192 < n; i += 256) { v0 += p[i]; v1 += p[i + 64]; v2 += p[i + 128]; v3 += p[i + 192]; }
    for (; i < n; i += 64) v0 += p[i];
    float v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_tot[a] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ssim = s_tot[0] * inv_chw, l1 = s_tot[1] * inv_chw, ne = s_tot[2] * inv_hw, dm = s_tot[3] * inv_hw;
        const float photo = (1.0f - lam) * l1 + lam * (1.0f - ssim);
        out[0] = (photo + ldist * dm) + ln * ne;
        out[1] = l1; out[2] = ssim; out[3] = ne; out[4] = dm;
    }
}

// block b sums part[b * n .. (b + 1) * n) into out_b (b = 0: SSIM map, b = 1: |a - b|)
__global__ __launch_bounds__(256) void ssim_sum_parts(int n, const float* __restrict__ part, float inv_count,
                                                      float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float s_red[4];
    part += (size_t)blockIdx.x * n;
    float* out = blockIdx.x == 0 ? out0 : out1;
    float a = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * inv_count;
}

// dL/dimg1 = g * ( blur(dm1) + 2 a blur(dm2) + b blur(dm3) ),  g = dL/d(mean) / (C H W)
//            [+ g_l1 / (C H W) * sign(a - b): the L1 term's gradient, when g_l1 is given]
__global__ __launch_bounds__(256) void ssim_bwd(int C, int H, int W, SsimTaps tp, const float* __restrict__ img1,
                                                const float* __restrict__ img2, const float* __restrict__ dmaps,
                                                const float* __restrict__ g_mean, const float* __restrict__ g_l1,
                                                float inv_count, float* __restrict__ dimg1, float k_ssim, float k_l1,
                                                TrainReg reg, float k_normal, float k_dist, float* __restrict__ d_rn,
                                                float* __restrict__ d_sn, float* __restrict__ d_dist) {
    // k_ssim / k_l1 scale the two upstream scalars (train loss: both point at dL/dtotal, k = -lambda and 1 - lambda).
    // Channel-0 workgroups also write the regularisers' gradients of their tile (when the outputs are given):
    //   d rend_normal = -k_normal g sn,  d surf_normal = -k_normal g rn,  d rend_dist = k_dist g      (k = lambda / (H W))
    __shared__ float s_in[3 * SS_IH * SS_IW];
    __shared__ float s_h[3 * SS_IH * SS_TW];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_TW - SS_R, y0 = blockIdx.y * SS_TH - SS_R;
    const size_t plane = (size_t)H * W, cp = (size_t)C * plane;
    for (int e = threadIdx.x; e < SS_IH * SS_IW; e += 256) {
        const int r = e / SS_IW, c = e - r * SS_IW;
        const int x = x0 + c, y = y0 + r;
        float d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = ch * plane + (size_t)y * W + x;
            d1 = dmaps[o]; d2 = dmaps[cp + o]; d3 = dmaps[2 * cp + o];
        }
        s_in[e] = d1; s_in[SS_IH * SS_IW + e] = d2; s_in[2 * SS_IH * SS_IW + e] = d3;
    }
    __syncthreads();
    float m[3];
    tile_blur<3>(s_in, s_h, tp, m);
    const int tx = threadIdx.x & (SS_TW - 1), ty = threadIdx.x >> 5;
    const int x = blockIdx.x * SS_TW + tx, y = blockIdx.y * SS_TH + ty;
    if (x < W && y < H) {
        const size_t o = ch * plane + (size_t)y * W + x;
        const float a = img1[o], b = img2[o];
        float v = (g_mean[0] * k_ssim * inv_count) * (m[0] + 2.0f * a * m[1] + b * m[2]);
        if (g_l1 != nullptr) {
            const float df = a - b;
            v += (g_l1[0] * k_l1 * inv_count) * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));     // torch: sign(0) = 0
        }
        dimg1[o] = v;
        if (ch == 0) {
            const size_t q = (size_t)y * W + x;
            const float g = g_mean[0];
            if (d_rn != nullptr) {
                const float k = -(k_normal * g);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    d_rn[c * plane + q] = k * reg.surf_normal[c * plane + q];
                    d_sn[c * plane + q] = k * reg.rend_normal[c * plane + q];
                }
            }
            if (d_dist != nullptr) d_dist[q] = k_dist * g;
        }
    }
}

}  // namespace iso

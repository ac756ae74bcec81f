// render() post-processing of the rasterizer's 7-channel `allmap` (reference gaussian_renderer/__init__.py:127-167
// and utils/point_utils.py:10-40): alpha, view->world normals, expected / median / surface depth, and the normals of
// the surface-depth map by central differences of the back-projected points.  The reference runs ~25 elementwise
// torch kernels forward and ~40 backward over 2 M pixels every step; here: two streaming kernels each way.
//
// allmap channels: 0 depth*w sum, 1 alpha, 2..4 normal (view space), 5 median depth, 6 distortion.
#include "isr_common.hpp"

namespace iso {

__device__ __forceinline__ float pp_nan_to_num(float v) {      // torch.nan_to_num(v, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest
    if (v != v) return 0.0f;
    if (v == __builtin_inff()) return 0.0f;
    if (v == -__builtin_inff()) return -3.40282346638528859812e+38f;
    return v;
}
__device__ __forceinline__ bool pp_finite(float v) { return (v == v) && v != __builtin_inff() && v != -__builtin_inff(); }

// per-pixel maps.  view: row-major 4x4 world_view_transform (only [:3,:3] is read)
__global__ __launch_bounds__(256) void pp_maps(long long N, float ratio, float one_minus_ratio,
                                               const float* __restrict__ allmap, const float* __restrict__ view,
                                               float* __restrict__ rend_alpha, float* __restrict__ rend_normal,
                                               float* __restrict__ rend_dist, float* __restrict__ surf_depth,
                                               float* __restrict__ rend_depth, float* __restrict__ rend_median) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float a0 = allmap[i], alpha = allmap[N + i];
    const float n0 = allmap[2 * N + i], n1 = allmap[3 * N + i], n2 = allmap[4 * N + i];
    const float med = pp_nan_to_num(allmap[5 * N + i]);
    const float expd = pp_nan_to_num(a0 / alpha);
    rend_alpha[i] = alpha;
    rend_dist[i] = allmap[6 * N + i];
#pragma unroll
    for (int c = 0; c < 3; c++)
        rend_normal[c * N + i] = (n0 * view[4 * c] + n1 * view[4 * c + 1]) + n2 * view[4 * c + 2];
    rend_depth[i] = expd;
    rend_median[i] = med;
    surf_depth[i] = expd * one_minus_ratio + ratio * med;
}

struct PP3 { float x, y, z; };
__device__ __forceinline__ PP3 pp_point(const float* __restrict__ depth, const float* __restrict__ rays_d, PP3 o, long long p) {
    const float d = depth[p];
    return {d * rays_d[3 * p] + o.x, d * rays_d[3 * p + 1] + o.y, d * rays_d[3 * p + 2] + o.z};
}
__device__ __forceinline__ PP3 pp_cross(PP3 a, PP3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// surf_normal = normalize(cross(P[y+1,x] - P[y-1,x], P[y,x+1] - P[y,x-1])) * alpha, zero on the image border
__global__ __launch_bounds__(256) void pp_surf_normal(int W, int H, const float* __restrict__ surf_depth,
                                                      const float* __restrict__ alpha, const float* __restrict__ rays_d,
                                                      const float* __restrict__ rays_o, float* __restrict__ out) {
    const long long N = (long long)W * H;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int x = (int)(i % W), y = (int)(i / W);
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        const PP3 o = {rays_o[0], rays_o[1], rays_o[2]};
        const PP3 pd = pp_point(surf_depth, rays_d, o, i + W), pu = pp_point(surf_depth, rays_d, o, i - W);
        const PP3 pr = pp_point(surf_depth, rays_d, o, i + 1), pl = pp_point(surf_depth, rays_d, o, i - 1);
        const PP3 dx = {pd.x - pu.x, pd.y - pu.y, pd.z - pu.z}, dy = {pr.x - pl.x, pr.y - pl.y, pr.z - pl.z};
        const PP3 c = pp_cross(dx, dy);
        const float nrm = __builtin_sqrtf((c.x * c.x + c.y * c.y) + c.z * c.z);
        const float den = fmaxf(nrm, 1e-12f);                     // F.normalize eps
        const float a = alpha[i];
        nx = (c.x / den) * a; ny = (c.y / den) * a; nz = (c.z / den) * a;
    }
    out[i] = nx; out[N + i] = ny; out[2 * N + i] = nz;
}

// backward, pass A: dL/d(dx), dL/d(dy) of every interior pixel's cross product -> scratch[6, N]
__global__ __launch_bounds__(256) void pp_bwd_stencil(int W, int H, const float* __restrict__ surf_depth,
                                                      const float* __restrict__ alpha, const float* __restrict__ rays_d,
                                                      const float* __restrict__ rays_o, const float* __restrict__ g_sn,
                                                      float* __restrict__ scratch) {
    const long long N = (long long)W * H;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int x = (int)(i % W), y = (int)(i / W);
    PP3 gdx = {0, 0, 0}, gdy = {0, 0, 0};
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        const float a = alpha[i];
        const PP3 g = {g_sn[i] * a, g_sn[N + i] * a, g_sn[2 * N + i] * a};      // alpha is detached (:160)
        if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f) {
            const PP3 o = {rays_o[0], rays_o[1], rays_o[2]};
            const PP3 pd = pp_point(surf_depth, rays_d, o, i + W), pu = pp_point(surf_depth, rays_d, o, i - W);
            const PP3 pr = pp_point(surf_depth, rays_d, o, i + 1), pl = pp_point(surf_depth, rays_d, o, i - 1);
            const PP3 dx = {pd.x - pu.x, pd.y - pu.y, pd.z - pu.z}, dy = {pr.x - pl.x, pr.y - pl.y, pr.z - pl.z};
            const PP3 c = pp_cross(dx, dy);
            const float nrm = __builtin_sqrtf((c.x * c.x + c.y * c.y) + c.z * c.z);
            PP3 gc;
            if (nrm > 1e-12f) {
                const float inv = 1.0f / nrm;
                const PP3 n = {c.x * inv, c.y * inv, c.z * inv};
                const float d = (n.x * g.x + n.y * g.y) + n.z * g.z;
                gc = {(g.x - n.x * d) * inv, (g.y - n.y * d) * inv, (g.z - n.z * d) * inv};
            } else {                                             // clamp_min branch of F.normalize: out = v / eps
                gc = {g.x * 1e12f, g.y * 1e12f, g.z * 1e12f};
            }
            gdx = pp_cross(dy, gc);        // d((dx x dy).g)/d dx = dy x g
            gdy = pp_cross(gc, dx);        // d((dx x dy).g)/d dy = g x dx
        }
    }
    scratch[i] = gdx.x; scratch[N + i] = gdx.y; scratch[2 * N + i] = gdx.z;
    scratch[3 * N + i] = gdy.x; scratch[4 * N + i] = gdy.y; scratch[5 * N + i] = gdy.z;
}

// backward, pass B: gather the stencil terms into dL/dsurf_depth and chain to dL/dallmap[7, N]
__global__ __launch_bounds__(256) void pp_bwd_maps(int W, int H, float ratio, float one_minus_ratio,
                                                   const float* __restrict__ allmap, const float* __restrict__ view,
                                                   const float* __restrict__ rays_d, const float* __restrict__ scratch,
                                                   const float* __restrict__ g_alpha, const float* __restrict__ g_normal,
                                                   const float* __restrict__ g_dist, const float* __restrict__ g_surf,
                                                   const float* __restrict__ g_depth, const float* __restrict__ g_median,
                                                   float* __restrict__ out) {
    const long long N = (long long)W * H;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int x = (int)(i % W), y = (int)(i / W);
    float g_sd = g_surf ? g_surf[i] : 0.0f;
    if (scratch != nullptr) {
        PP3 gp = {0, 0, 0};
        if (y > 0) { gp.x += scratch[i - W]; gp.y += scratch[N + i - W]; gp.z += scratch[2 * N + i - W]; }
        if (y < H - 1) { gp.x -= scratch[i + W]; gp.y -= scratch[N + i + W]; gp.z -= scratch[2 * N + i + W]; }
        if (x > 0) { gp.x += scratch[3 * N + i - 1]; gp.y += scratch[4 * N + i - 1]; gp.z += scratch[5 * N + i - 1]; }
        if (x < W - 1) { gp.x -= scratch[3 * N + i + 1]; gp.y -= scratch[4 * N + i + 1]; gp.z -= scratch[5 * N + i + 1]; }
        g_sd += (gp.x * rays_d[3 * i] + gp.y * rays_d[3 * i + 1]) + gp.z * rays_d[3 * i + 2];
    }
    const float g_e = (g_depth ? g_depth[i] : 0.0f) + one_minus_ratio * g_sd;
    const float g_m = (g_median ? g_median[i] : 0.0f) + ratio * g_sd;
    const float a0 = allmap[i], alpha = allmap[N + i];
    const float quo = a0 / alpha;
    float g_a0 = 0.0f, g_al = g_alpha ? g_alpha[i] : 0.0f;
    if (pp_finite(quo) && g_e != 0.0f) {      // where the quotient is not finite torch yields 0 * (1/0) = NaN; here: 0
        g_a0 = g_e / alpha;
        g_al -= g_e * quo / alpha;
    }
    out[i] = g_a0;
    out[N + i] = g_al;
    float gn0 = 0.0f, gn1 = 0.0f, gn2 = 0.0f;
    if (g_normal) {
        const float r0 = g_normal[i], r1 = g_normal[N + i], r2 = g_normal[2 * N + i];
        gn0 = (r0 * view[0] + r1 * view[4]) + r2 * view[8];
        gn1 = (r0 * view[1] + r1 * view[5]) + r2 * view[9];
        gn2 = (r0 * view[2] + r1 * view[6]) + r2 * view[10];
    }
    out[2 * N + i] = gn0; out[3 * N + i] = gn1; out[4 * N + i] = gn2;
    out[5 * N + i] = pp_finite(allmap[5 * N + i]) ? g_m : 0.0f;
    out[6 * N + i] = g_dist ? g_dist[i] : 0.0f;
}

// Densification statistics of one training iteration (train.py:140-142, scene/gaussian_model.py:601-604): for every visible
// Gaussian  accum += |dL/dmean2D|,  denom += 1,  max_radii = max(max_radii, radii)  — one pass over the P rows.
__global__ __launch_bounds__(256) void densify_stats(int P, int C, const float* __restrict__ grad,
                                                     const uint8_t* __restrict__ visible, const int* __restrict__ radii,
                                                     float* __restrict__ accum, float* __restrict__ denom,
                                                     float* __restrict__ max_radii) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || !visible[i]) return;
    float s = 0.0f;
    for (int c = 0; c < C; c++) { const float g = grad[(size_t)i * C + c]; s += g * g; }
    accum[i] += __builtin_sqrtf(s);
    denom[i] += 1.0f;
    max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
}

}  // namespace iso

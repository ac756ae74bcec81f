// Shared device helpers for the MI355X (gfx950) surfel rasterizer.
//
// The whole library is compiled with -ffp-contract=off: every fused multiply-add
// in device code is written explicitly (__builtin_fmaf), so the EXACT arithmetic
// mode is op-for-op IEEE fp32 (bit-identical to the CPU oracle) and the FAST mode
// contracts only where this file says so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isr {

constexpr int TILE = 16;                    // reference config.h:16-17
constexpr int TILE_PIX = TILE * TILE;
constexpr float NEAR_N = 0.2f;              // reference auxiliary.h:38-41
constexpr float FAR_N = 100.0f;
constexpr float FILTER_SIZE = 0.707106f;
constexpr float FILTER_INV_SQ = 2.0f;
constexpr int REC = 20;                     // floats per splat record (80 B)
// record layout: [0..2]=Tu [3..5]=Tv [6..8]=Tw [9..10]=centre [11..13]=normal [14]=opacity
//                [15..17]=rgb [18]=view depth [19]=unused
constexpr int MAX_FCHUNK = 32;              // feature channels handled per pass of the blend kernels
// Per-tile counters are hit by R atomics per view.  Atomics on one cache line serialise in L2, so every tile gets
// CNT_SUB sub-counters (picked by gaussian id) and every sub-counter its own 128-byte line.
constexpr int CNT_SUB = 4;
constexpr int CNT_STRIDE = 32;              // uint32 slots per counter (one 128-B line)

constexpr int CULL_STRIDE = 8;              // floats per Gaussian of GeomView::cull and of GeomView::ellipse
struct alignas(8) Rect16 { uint16_t x0, y0, x1, y1; };

struct GeomView {          // carved from the caller's geometry workspace
    int64_t* header;       // [0]=num_rendered, [1]=P, [2..7] reserved
    float* rec;            // [P, REC]
    uint32_t* tiles_touched;   // [P]
    uint32_t* point_offsets;   // [P] exclusive scan of tiles_touched
    Rect16* rect;          // [P]
    uint8_t* clamped;      // [P] bit mask (bit c set = channel c clamped)
    int* radii;            // [P] private copy (caller's radii may be freed before backward)
    uint32_t* scan_tmp;    // block sums for the scan
    float* cull;           // [P,8] per-Gaussian cull bounds in pixels: box (x_lo, x_hi, y_lo, y_hi), then the same along x+y, x-y
    float* ellipse;        // [P,8] the alpha >= 1/255 ellipse + low-pass disc (splat_conic below): the 32 bytes k_pack_hits gathers per
                           // tile entry - a dense array of its own, so that four neighbouring Gaussians (neighbours on screen:
                           // the trainers store them in Z-order) share one 128-byte line
    unsigned long long* row_mask;   // [P] sampled backward only: bit i = the Gaussian's i-th tile instance received a
                                    // partial row (bit 63 = an instance >= 63 did: consult the byte flags from there on)
};

struct ImageView {
    float* final_T;        // [3, N]  T, M1, M2
    uint32_t* n_contrib;   // [2, N]  last contributor, median contributor
    uint32_t* tile_count;  // [tiles * CNT_SUB * CNT_STRIDE] padded sub-counters
    uint32_t* tile_offset; // [tiles + 1] exclusive scan of the tile totals (ranges[t] = off[t], off[t+1])
    uint32_t* tile_cursor; // [tiles * CNT_SUB * CNT_STRIDE] padded scatter cursors
    uint32_t* sub_offset;  // [tiles * CNT_SUB] start of every sub-bucket
    uint8_t* tile_mode;    // [tiles] backward only: live pixels of the tile (0 none, 255 = dense K9 kernel)
    uint32_t* live_list;   // [tiles, 32, 2] backward only: (x | y << 8, last contributor) of the live pixels
    uint32_t* tile_order;  // [tiles] tile ids by descending instance count (k_tile_scan): launch order of k_render_bwd_geo
};

struct BinView {
    unsigned long long* keys;  // [R] (depth_bits << 32) | gaussian, bucketed by tile
    uint32_t* point_list;      // [R] sorted gaussian ids
    uint32_t* box4;            // [R] per-instance cull box, tile-relative int8 (x_lo, x_hi, y_lo, y_hi), written by k_pack_hits
    unsigned long long* hit_mask;   // [(R / 64 + tiles + 2), HM_WORDS] per 64-instance chunk of a tile's list: words 0..3 = the tile's four
                                    // 8x8 blocks, bit l = instance 64 c + l meets the block (k_pack_hits -> k_render_fwd_fast_w); words
                                    // 4..11 = the same per 8x4 HALF of a block (word 4 + 2 block + half: rows 0-3 / 4-7 of the block)
};
constexpr int HM_WORDS = 12;
// chunk c of tile t (list start r0) owns the HM_WORDS words from here: distinct for every (t, c), monotone in t
__host__ __device__ inline size_t hit_mask_word(int64_t r0, int tile, int chunk) { return ((size_t)(r0 >> 6) + (size_t)tile + (size_t)chunk) * HM_WORDS; }

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
__host__ __device__ inline T* carve(char*& p, size_t count) {
    p = (char*)align_up((size_t)p, 256);
    T* r = (T*)p;
    p += count * sizeof(T);
    return r;
}

__host__ __device__ inline int tiles_x(int W) { return (W + TILE - 1) / TILE; }
__host__ __device__ inline int tiles_y(int H) { return (H + TILE - 1) / TILE; }

inline GeomView geom_view(void* buf, int P) {
    char* p = (char*)buf;
    GeomView g;
    g.header = carve<int64_t>(p, 8);
    g.rec = carve<float>(p, (size_t)P * REC);
    g.tiles_touched = carve<uint32_t>(p, P);
    g.point_offsets = carve<uint32_t>(p, P);
    g.rect = carve<Rect16>(p, P);
    g.clamped = carve<uint8_t>(p, P);
    g.radii = carve<int>(p, P);
    g.scan_tmp = carve<uint32_t>(p, (size_t)(P / 256 + 2) * 2);       // per 256 Gaussians (a K1 workgroup): total, then (from +nb+1) maximum
    g.row_mask = carve<unsigned long long>(p, P);
    g.cull = carve<float>(p, (size_t)P * CULL_STRIDE);
    g.ellipse = carve<float>(p, (size_t)P * CULL_STRIDE);
    return g;
}
inline size_t geom_bytes(int P) {
    GeomView g = geom_view((void*)0, P);
    return (size_t)(g.ellipse + (size_t)P * CULL_STRIDE) + 256;
}
inline ImageView image_view(void* buf, int W, int H) {
    char* p = (char*)buf;
    size_t N = (size_t)W * H, T = (size_t)tiles_x(W) * tiles_y(H);
    ImageView v;
    v.final_T = carve<float>(p, 3 * N);
    v.n_contrib = carve<uint32_t>(p, 2 * N);
    v.tile_count = carve<uint32_t>(p, T * CNT_SUB * CNT_STRIDE);
    v.tile_offset = carve<uint32_t>(p, T + 1);
    v.tile_cursor = carve<uint32_t>(p, T * CNT_SUB * CNT_STRIDE);
    v.sub_offset = carve<uint32_t>(p, T * CNT_SUB);
    v.tile_mode = carve<uint8_t>(p, T);
    v.live_list = carve<uint32_t>(p, T * 64);
    v.tile_order = carve<uint32_t>(p, T);
    return v;
}
inline size_t image_bytes(int W, int H) {
    ImageView v = image_view((void*)0, W, H);
    return (size_t)(v.tile_order + (size_t)tiles_x(W) * tiles_y(H)) + 256;
}
inline BinView bin_view(void* buf, int64_t R) {
    char* p = (char*)buf;
    BinView b;
    b.keys = carve<unsigned long long>(p, (size_t)(R > 0 ? R : 1));
    b.point_list = carve<uint32_t>(p, (size_t)(R > 0 ? R : 1));
    b.box4 = carve<uint32_t>(p, (size_t)(R > 0 ? R : 1));
    b.hit_mask = carve<unsigned long long>(p, 0);           // to the end of the buffer (bin_bytes)
    return b;
}
inline size_t bin_bytes(int64_t R, int tiles) {
    BinView b = bin_view((void*)0, R);
    return (size_t)(b.hit_mask + ((size_t)((R > 0 ? R : 1) >> 6) + (size_t)tiles + 2) * HM_WORDS) + 256;
}

// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));     // MFMA 32x32 accumulator fragment
typedef float f32x4 __attribute__((ext_vector_type(4)));       // MFMA 16x16 accumulator fragment

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Saturating float -> int (v_cvt_i32_f32 saturates and maps NaN to 0; written
// out so the semantics do not depend on undefined behaviour).
__device__ __forceinline__ int sat_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// Tile rectangle of a disc (reference auxiliary.h:68-78), same fp32 op order.
__device__ __forceinline__ void tile_rect(float cx, float cy, int r, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1) {
    const float fr = (float)r;
    x0 = min(gx, max(0, sat_i32((cx - fr) / (float)TILE)));
    y0 = min(gy, max(0, sat_i32((cy - fr) / (float)TILE)));
    x1 = min(gx, max(0, sat_i32((cx + fr + (float)TILE - 1.0f) / (float)TILE)));
    y1 = min(gy, max(0, sat_i32((cy + fr + (float)TILE - 1.0f) / (float)TILE)));
}

// exp(x), x <= 0: fixed fma sequence shared with the oracle (oracle/surfel_oracle.cpp: exp_fixed).
__device__ __forceinline__ float exp_fixed(float x) {
    if (x < -87.0f) return 0.0f;
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int e = (int)n + 127;
    return y * __uint_as_float((unsigned)e << 23);
}

// Conservative screen-space box outside of which a splat certainly contributes nothing to a pixel
// (alpha < 1/255), i.e. BOTH rho3d > skip and rho2d > skip hold there:
//   * rho3d <= skip is the perspective image of the disc u^2+v^2 <= skip of the splat plane; when that disc
//     lies entirely in front of the camera plane (d < 0) its exact axis-aligned bound is the reference's own
//     AABB formula (forward.cu:119-145) evaluated with cutoff^2 = skip; otherwise the image is unbounded and
//     no culling is done;
//   * rho2d <= skip is the disc of radius sqrt(skip/2) around the low-pass centre.
// 1% + 0.5 px safety margins; any NaN yields an infinite box (never culls).  Returns (x_lo, x_hi, y_lo, y_hi).
__device__ __forceinline__ float4 splat_cull_box(F3 Tu, F3 Tv, F3 Tw, float cx, float cy, float skip) {
    const float inf = __builtin_inff();
    float4 box = make_float4(-inf, inf, -inf, inf);
    if (skip < inf) {
        const float d = skip * (Tw.x * Tw.x + Tw.y * Tw.y) - Tw.z * Tw.z;
        if (d < 0.0f) {
            const float fi = 1.0f / d;
            const float fa = skip * fi, fz = -fi;
            const float ex_c = fa * (Tu.x * Tw.x) + fa * (Tu.y * Tw.y) + fz * (Tu.z * Tw.z);
            const float ey_c = fa * (Tv.x * Tw.x) + fa * (Tv.y * Tw.y) + fz * (Tv.z * Tw.z);
            const float hx = ex_c * ex_c - (fa * (Tu.x * Tu.x) + fa * (Tu.y * Tu.y) + fz * (Tu.z * Tu.z));
            const float hy = ey_c * ey_c - (fa * (Tv.x * Tv.x) + fa * (Tv.y * Tv.y) + fz * (Tv.z * Tv.z));
            const float ex = __builtin_sqrtf(hx > 0.0f ? hx : 0.0f) * 1.01f + 0.5f;
            const float ey = __builtin_sqrtf(hy > 0.0f ? hy : 0.0f) * 1.01f + 0.5f;
            const float r2 = __builtin_sqrtf(0.5f * skip) * 1.01f + 0.5f;
            const float xl = fminf(ex_c - ex, cx - r2), xh = fmaxf(ex_c + ex, cx + r2);
            const float yl = fminf(ey_c - ey, cy - r2), yh = fmaxf(ey_c + ey, cy + r2);
            if (xl == xl && xh == xh && yl == yl && yh == yh && hx == hx && hy == hy) box = make_float4(xl, xh, yl, yh);
        }
    }
    return box;
}

// The same bound along the two diagonals (u1 = x + y, u2 = x - y): with the axis-aligned box this is the bounding
// octagon of the alpha >= 1/255 region, which is what an elongated splat at 45 degrees needs (its box is mostly empty).
// The extent of the rho3d <= skip conic along a direction  alpha x + beta y  is the box formula with  alpha Tu + beta Tv
// in place of Tu.  Returns (u1_lo, u1_hi, u2_lo, u2_hi); infinite whenever splat_cull_box would be.
__device__ __forceinline__ float4 splat_cull_diag(F3 Tu, F3 Tv, F3 Tw, float cx, float cy, float skip) {
    const float inf = __builtin_inff();
    float4 box = make_float4(-inf, inf, -inf, inf);
    if (skip < inf) {
        const float d = skip * (Tw.x * Tw.x + Tw.y * Tw.y) - Tw.z * Tw.z;
        if (d < 0.0f) {
            const float fi = 1.0f / d;
            const float fa = skip * fi, fz = -fi;
            const F3 P = {Tu.x + Tv.x, Tu.y + Tv.y, Tu.z + Tv.z}, M = {Tu.x - Tv.x, Tu.y - Tv.y, Tu.z - Tv.z};
            const float c1 = fa * (P.x * Tw.x) + fa * (P.y * Tw.y) + fz * (P.z * Tw.z);
            const float c2 = fa * (M.x * Tw.x) + fa * (M.y * Tw.y) + fz * (M.z * Tw.z);
            const float h1 = c1 * c1 - (fa * (P.x * P.x) + fa * (P.y * P.y) + fz * (P.z * P.z));
            const float h2 = c2 * c2 - (fa * (M.x * M.x) + fa * (M.y * M.y) + fz * (M.z * M.z));
            const float e1 = __builtin_sqrtf(h1 > 0.0f ? h1 : 0.0f) * 1.01f + 1.0f;       // 0.5 px on both axes
            const float e2 = __builtin_sqrtf(h2 > 0.0f ? h2 : 0.0f) * 1.01f + 1.0f;
            const float r2 = (__builtin_sqrtf(0.5f * skip) * 1.01f + 0.5f) * 1.41421357f + 0.01f;
            const float lo1 = fminf(c1 - e1, (cx + cy) - r2), hi1 = fmaxf(c1 + e1, (cx + cy) + r2);
            const float lo2 = fminf(c2 - e2, (cx - cy) - r2), hi2 = fmaxf(c2 + e2, (cx - cy) + r2);
            if (lo1 == lo1 && hi1 == hi1 && lo2 == lo2 && hi2 == hi2 && h1 == h1 && h2 == h2) box = make_float4(lo1, hi1, lo2, hi2);
        }
    }
    return box;
}

// The region rho3d <= h of the 3-D branch, exactly: p = k x l is affine in the pixel,
//   p(c + (dx, dy)) = Pc + dx A + dy B,   A = dp/dpx = Tw x l(c),  B = dp/dpy = k(c) x Tw,  Pc = k(c) x l(c)
// with k(c) = cx Tw - Tu, l(c) = cy Tw - Tv evaluated AT THE SPLAT'S CENTRE (both small there: none of the cancellation of the
// large Tu.z ~ cx Tw.z that the per-pixel evaluation carries), and rho3d = |p.xy|^2 / p.z^2 <= h  <=>
//   q(dx, dy) = |p.xy|^2 - h p.z^2 = qxx dx^2 + 2 qxy dx dy + qyy dy^2 + 2 qx dx + 2 qy dy + q0 <= 0,
// an ellipse; in centre form  (x - M)^T N (x - M) <= 1  with M = q's minimiser, N = Q / -q(M) = L^T L.  k_pack_hits minimises the form over
// the rectangle of a block half's pixel centres: the minimum of a convex quadratic over a box lies on the vertical line
// x = clamp(M.x) or on the horizontal line y = clamp(M.y) of the box (were it elsewhere, the direction towards M would be a feasible
// descent direction), so two clamped parabola vertices decide it - the bounding octagon above keeps ~1/3 more (half, splat) pairs
// than have a pixel inside the region.
// h = skip (K1's alpha < 1/255 bound: the threshold + 1 % + 0.05) + 0.2 % + 0.05: beyond it the pair's true rho is > threshold
// + 0.10, and FAST's and EXACT's evaluations (each within `exact_noise` <= 0.02 + band < 0.04 of it) are beyond band.hi.
// Returns (M.x, M.y, l11, l12), (l22, cx, cy, h / 2 = the low-pass disc rho2d <= h), absolute pixels.
// "Pass-all" form (L = 0, disc radius inf: every test passes, i.e. the octagon alone decides) whenever splat_cull_box gives no
// finite box (skip = inf, horizon in front of the camera), the splat is ill-conditioned (no finite guard band, exact_noise > 0.02),
// the form is not an ellipse numerically, or anything is NaN.
struct SplatConic { float4 a, b; };
__device__ __forceinline__ SplatConic splat_conic(F3 Tu, F3 Tv, F3 Tw, float cx, float cy, float skip, float4 cb, float exact_noise) {
    const float inf = __builtin_inff();
    SplatConic o = {make_float4(cx, cy, 0.f, 0.f), make_float4(0.f, cx, cy, inf)};
    if (skip < inf && exact_noise <= 0.02f && cb.x > -1e30f && cb.y < 1e30f && cb.z > -1e30f && cb.w < 1e30f) {
        const float h = __builtin_fmaf(skip, 1.002f, 0.05f);
        const F3 k = {__builtin_fmaf(cx, Tw.x, -Tu.x), __builtin_fmaf(cx, Tw.y, -Tu.y), __builtin_fmaf(cx, Tw.z, -Tu.z)};
        const F3 l = {__builtin_fmaf(cy, Tw.x, -Tv.x), __builtin_fmaf(cy, Tw.y, -Tv.y), __builtin_fmaf(cy, Tw.z, -Tv.z)};
        const F3 Pc = cross3(k, l), A = cross3(Tw, l), B = cross3(k, Tw);
        const float qxx = A.x * A.x + A.y * A.y - h * (A.z * A.z), qxy = A.x * B.x + A.y * B.y - h * (A.z * B.z);
        const float qyy = B.x * B.x + B.y * B.y - h * (B.z * B.z), qx = A.x * Pc.x + A.y * Pc.y - h * (A.z * Pc.z);
        const float qy = B.x * Pc.x + B.y * Pc.y - h * (B.z * Pc.z), q0 = Pc.x * Pc.x + Pc.y * Pc.y - h * (Pc.z * Pc.z);
        const float det = qxx * qyy - qxy * qxy;
        const float mx = (qy * qxy - qx * qyy) / det, my = (qx * qxy - qy * qxx) / det;
        const float kap = -(q0 + (qx * mx + qy * my));           // -q(M)
        const float ik = 1.0f / kap;
        const float nxx = qxx * ik, nxy = qxy * ik;
        // N = L^T L (Cholesky): e = (l11 dx + l12 dy)^2 + (l22 dy)^2.  Evaluated as the quadratic form, a thin needle seen thousands
        // of pixels from its centre cancels ~(distance / thickness)^2 ulps; as a sum of two squares, ~distance / thickness.
        const float l11 = __builtin_sqrtf(nxx), l12 = nxy / l11, l22 = __builtin_sqrtf(det * ik * ik / nxx);
        const float chk = ((l11 + l12) + l22) + (mx + my);
        // an ellipse, well away from degenerate (det of a needle at 45 degrees is itself a cancelling difference: beyond an aspect
        // of ~100 it is left to the octagon): positive definite with a positive level
        if (chk == chk && fabsf(chk) < inf && qxx > 0.0f && qyy > 0.0f && det > 1e-4f * (qxx * qyy) && kap > 0.0f && l11 > 0.0f && l22 > 0.0f) {
            // loosened by 0.2 % (the rounding of this conversion; k_pack_hits adds the rounding of its own evaluation)
            o.a = make_float4(cx + mx, cy + my, l11 * 0.998f, l12 * 0.998f);
            o.b = make_float4(l22 * 0.998f, cx, cy, 0.5f * h);
        }
    }
    return o;
}

// Tile-relative int8 packing of a cull box (rounded outward, saturated): the backward tests it without
// touching the splat record.
__device__ __forceinline__ uint32_t pack_box4(float4 box, float tile_x0, float tile_y0) {
    const int xl = (int)fmaxf(-128.0f, fminf(127.0f, __builtin_floorf(box.x - tile_x0)));
    const int xh = (int)fmaxf(-128.0f, fminf(127.0f, __builtin_ceilf(box.y - tile_x0)));
    const int yl = (int)fmaxf(-128.0f, fminf(127.0f, __builtin_floorf(box.z - tile_y0)));
    const int yh = (int)fmaxf(-128.0f, fminf(127.0f, __builtin_ceilf(box.w - tile_y0)));
    return (uint32_t)(xl & 255) | ((uint32_t)(xh & 255) << 8) | ((uint32_t)(yl & 255) << 16) | ((uint32_t)(yh & 255) << 24);
}

// ----------------------------------------------------------------------------
// Workgroup-level aggregation of per-tile counters.  Global atomics cost one memory transaction per distinct cache line
// per wave instruction (~26 G/s on MI355X whatever the scope, tools/micro/atomics.hip), and K1 + the key scatter issue
// 2R of them.  When neighbouring Gaussians in memory are neighbours on screen (scenes.spatially_sorted) the 256 Gaussians
// of a workgroup touch a few dozen tiles: their increments are first merged in a small LDS hash table (open addressing,
// integer LDS atomics) and each distinct tile costs ONE global atomic.  A tile that finds no slot within TH_PROBES steps
// falls back to the direct global atomic, consistently in every phase (slots never become free again).
constexpr int TH_SIZE = 2048, TH_BITS = 11, TH_PROBES = 32;
constexpr int BIG_RECT = 128;            // tiles: rectangles beyond this are walked by the workgroup, not by the splat's own lane
constexpr uint32_t TH_EMPTY = 0xffffffffu;
__device__ __forceinline__ int th_find_or_insert(uint32_t* keys, uint32_t tile) {
    uint32_t h = (tile * 2654435761u) >> (32 - TH_BITS);
    for (int p = 0; p < TH_PROBES; p++) {
        const uint32_t old = atomicCAS(&keys[h], TH_EMPTY, tile);
        if (old == TH_EMPTY || old == tile) return (int)h;
        h = (h + 1) & (TH_SIZE - 1);
    }
    return -1;
}
__device__ __forceinline__ int th_find(const uint32_t* keys, uint32_t tile) {
    uint32_t h = (tile * 2654435761u) >> (32 - TH_BITS);
    for (int p = 0; p < TH_PROBES; p++) {
        const uint32_t k = keys[h];
        if (k == tile) return (int)h;
        if (k == TH_EMPTY) return -1;
        h = (h + 1) & (TH_SIZE - 1);
    }
    return -1;
}
// every Gaussian of a workgroup uses the same sub-counter of a tile; concurrently running workgroups use different ones
__device__ __forceinline__ int counter_sub(int block) { return block & (CNT_SUB - 1); }


// Exclusive u32 scan over a workgroup of 1024 threads (s_warp: 32 words of LDS).
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* s_warp, uint32_t& total) {
    // blockDim.x == 1024: 16 waves
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_warp[wid] = x;
    __syncthreads();
    if (wid == 0) {
        uint32_t w = lane < 16 ? s_warp[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t y = __shfl_up(w, o);
            if (lane >= o) w += y;
        }
        if (lane < 16) s_warp[16 + lane] = w;   // inclusive
    }
    __syncthreads();
    const uint32_t base = wid == 0 ? 0u : s_warp[16 + wid - 1];
    total = s_warp[16 + 15];
    __syncthreads();
    return base + x - v;
}

// Arithmetic policies for the per-pixel loops.
struct ExactMath {
    static constexpr bool fast = false;
    __device__ static __forceinline__ float mad(float a, float b, float c) { return a * b + c; }   // two roundings
    __device__ static __forceinline__ float msub(float a, float b, float c) { return a * b - c; }
    __device__ static __forceinline__ float div(float a, float b) { return a / b; }                // IEEE
    __device__ static __forceinline__ float ex(float x) { return exp_fixed(x); }
};
struct FastMath {
    static constexpr bool fast = true;
    __device__ static __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    __device__ static __forceinline__ float msub(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }
    __device__ static __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    __device__ static __forceinline__ float ex(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
};

}  // namespace isr

// ---------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's
// `roofline` object).  Off by default: zero overhead.
#include <string>
#include <vector>
namespace isr {
struct ProfRec { const char* name; hipEvent_t a, b; };
struct Prof {
    bool on = false;
    bool dominant_only = false;     // mode 2: time only the forward blend kernel (least perturbation of the stream)
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};
inline Prof& prof() { static Prof p; return p; }
struct ProfScope {
    hipStream_t s; ProfRec r; bool on;
    ProfScope(const char* name, hipStream_t st)
        : s(st), on(prof().on && (!prof().dominant_only || __builtin_strcmp(name, "k_render_fwd") == 0)) {
        if (on) { r.name = name; r.a = prof().get(); r.b = prof().get(); (void)hipEventRecord(r.a, s); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, s); prof().recs.push_back(r); } }
};
}  // namespace isr

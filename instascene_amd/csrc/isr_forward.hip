// Forward path of the MI355X surfel rasterizer: per-Gaussian preprocess (K1),
// per-tile bucket binning (replaces the reference's duplicate + 64-bit global
// radix sort + range scan, rasterizer_impl.cu:70-138,283-323) and the per-tile
// front-to-back blend (K8, forward.cu:256-462).
//
// Binning design (MI355X-first, not a translation of the reference):
//   1. K1 also counts, per tile, how many splats touch it (non-returning atomics).
//   2. One small scan over tiles gives every tile its [start,end) range directly —
//      the reference derives ranges from the sorted keys.
//   3. A scatter pass drops (depth_bits<<32 | gaussian) into the tile's bucket.
//   4. One workgroup per tile sorts its bucket in LDS (bitonic, all-ascending
//      "flip" network with virtual +inf padding; in-place global fallback for
//      buckets larger than the LDS budget).
//   Because a Gaussian appears at most once per tile, ordering a bucket by
//   (depth_bits, gaussian) is exactly the order the reference's stable radix
//   sort on (tile, depth_bits) produces: point_list and ranges are bit-identical.
//   HBM traffic is ~20 B/instance instead of ~144 B/instance for six radix passes.
#include "isr_common.hpp"

namespace isr {

// ----------------------------------------------------------------------------
// K1: reference forward.cu:148-251 (+ auxiliary.h:186-236,286-293; forward.cu:20-145)
__device__ __forceinline__ void quat_to_cols(const float* q, F3& c0, F3& c1, F3& c2) {
    float s = 1.0f / __builtin_sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    c0 = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
    c1 = {2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
    c2 = {2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
}

__device__ __forceinline__ F3 sh_to_rgb(int deg, F3 pos, F3 cam, const float* shp, unsigned& clamp_mask) {
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    constexpr float C2a = 1.0925484305920792f, C2b = -1.0925484305920792f, C2c = 0.31539156525252005f,
                    C2d = -1.0925484305920792f, C2e = 0.5462742152960396f;
    constexpr float C3a = -0.5900435899266435f, C3b = 2.890611442640554f, C3c = -0.4570457994644658f,
                    C3d = 0.3731763325901154f, C3e = -0.4570457994644658f, C3f = 1.445305721320277f,
                    C3g = -0.5900435899266435f;
    F3 dir = pos - cam;
    float len = __builtin_sqrtf(dot3(dir, dir));
    dir = {dir.x / len, dir.y / len, dir.z / len};
    auto sh = [&](int k) { return F3{shp[3 * k], shp[3 * k + 1], shp[3 * k + 2]}; };
    F3 res = C0 * sh(0);
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        res = res - (C1 * y) * sh(1) + (C1 * z) * sh(2) - (C1 * x) * sh(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + (C2a * xy) * sh(4) + (C2b * yz) * sh(5) + (C2c * (2.0f * zz - xx - yy)) * sh(6) +
                  (C2d * xz) * sh(7) + (C2e * (xx - yy)) * sh(8);
            if (deg > 2) {
                res = res + (C3a * y * (3.0f * xx - yy)) * sh(9) + (C3b * xy * z) * sh(10) +
                      (C3c * y * (4.0f * zz - xx - yy)) * sh(11) +
                      (C3d * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh(12) +
                      (C3e * x * (4.0f * zz - xx - yy)) * sh(13) + (C3f * z * (xx - yy)) * sh(14) +
                      (C3g * x * (xx - 3.0f * yy)) * sh(15);
            }
        }
    }
    res = {res.x + 0.5f, res.y + 0.5f, res.z + 0.5f};
    clamp_mask = (res.x < 0 ? 1u : 0u) | (res.y < 0 ? 2u : 0u) | (res.z < 0 ? 4u : 0u);
    return {res.x < 0.0f ? 0.0f : res.x, res.y < 0.0f ? 0.0f : res.y, res.z < 0.0f ? 0.0f : res.z};
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

template <bool STAGE_SH>
__global__ __launch_bounds__(256) void k_preprocess(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rots, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ tm_pre, const float* __restrict__ col_pre, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, int W, int H, int gx, int gy,
    int* __restrict__ radii_out, GeomView g, uint32_t* __restrict__ tile_count, int tight_rects) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // The workgroup's 256 SH rows (192 B each) are one contiguous 48 KB piece of `shs`: it is read with fully coalesced
    // 16-byte loads and handed to the owning lanes through LDS.  (A lane reading its own row touches 64 different
    // cache lines per load instruction, and with ~250 KB of rows in flight per CU the 32 KB vector cache keeps none of
    // them between the twelve loads of a row: 0.65 GB of fetches for 0.47 GB of input.)
    constexpr int SH_STRIDE = 52;           // floats per staged row: 48 + 4 (16-byte aligned, spreads the LDS banks)
    constexpr int BIG_WORDS = 2 * 256 + 4;  // list of the workgroup's large rectangles (see the tile counting below)
    constexpr int LDS_WORDS = STAGE_SH ? 256 * SH_STRIDE : 2 * TH_SIZE + BIG_WORDS;     // the tile hash + that list reuse the SH staging area
    static_assert(LDS_WORDS >= 2 * TH_SIZE + BIG_WORDS, "tile hash does not fit");
    __shared__ __attribute__((aligned(16))) float s_sh[LDS_WORDS];
    uint32_t* th_key = reinterpret_cast<uint32_t*>(s_sh);
    uint32_t* th_cnt = th_key + TH_SIZE;
    if constexpr (STAGE_SH) {
        const int i0 = blockIdx.x * 256;
        const int nq = min(256, P - i0) * 12;                      // float4s to move
        const float4* src = reinterpret_cast<const float4*>(shs + (size_t)i0 * 48);
        float4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11;      // twelve independent loads in flight
#define ISR_SH_LD(u, v) { const int e = min((int)threadIdx.x + u * 256, nq - 1); v = src[e]; }
        ISR_SH_LD(0, v0) ISR_SH_LD(1, v1) ISR_SH_LD(2, v2) ISR_SH_LD(3, v3) ISR_SH_LD(4, v4) ISR_SH_LD(5, v5)
        ISR_SH_LD(6, v6) ISR_SH_LD(7, v7) ISR_SH_LD(8, v8) ISR_SH_LD(9, v9) ISR_SH_LD(10, v10) ISR_SH_LD(11, v11)
#undef ISR_SH_LD
#define ISR_SH_ST(u, v) { const int e = (int)threadIdx.x + u * 256;                                               \
                          if (e < nq) { const int r = e / 12, k = e - r * 12;                                       \
                                        *reinterpret_cast<float4*>(s_sh + r * SH_STRIDE + 4 * k) = v; } }
        ISR_SH_ST(0, v0) ISR_SH_ST(1, v1) ISR_SH_ST(2, v2) ISR_SH_ST(3, v3) ISR_SH_ST(4, v4) ISR_SH_ST(5, v5)
        ISR_SH_ST(6, v6) ISR_SH_ST(7, v7) ISR_SH_ST(8, v8) ISR_SH_ST(9, v9) ISR_SH_ST(10, v10) ISR_SH_ST(11, v11)
#undef ISR_SH_ST
        __syncthreads();
    }
    int radius_i = 0;
    uint32_t touched = 0;
    Rect16 rc = {0, 0, 0, 0};
    do {
        if (i >= P) break;
        const F3 p = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
        const F3 pv = {view[0] * p.x + view[4] * p.y + view[8] * p.z + view[12],
                       view[1] * p.x + view[5] * p.y + view[9] * p.z + view[13],
                       view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14]};
        if (pv.z <= 0.2f) break;   // near cull, auxiliary.h:201
        F3 Tu, Tv, Tw, normal;
        if (tm_pre == nullptr) {
            F3 c0, c1, c2;
            const float q[4] = {rots[4 * (size_t)i], rots[4 * (size_t)i + 1], rots[4 * (size_t)i + 2],
                                rots[4 * (size_t)i + 3]};
            quat_to_cols(q, c0, c1, c2);
            const float sx = mod * scales[2 * (size_t)i], sy = mod * scales[2 * (size_t)i + 1];
            const F3 L0 = c0 * sx, L1 = c1 * sy, L2 = c2;
            const float S[3][4] = {{L0.x, L0.y, L0.z, 0.f}, {L1.x, L1.y, L1.z, 0.f}, {p.x, p.y, p.z, 1.f}};
            float n[3][4];
            n[0][0] = (float)((double)(float)W / 2.0); n[0][1] = 0.f; n[0][2] = 0.f; n[0][3] = (float)((double)(float)(W - 1) / 2.0);
            n[1][0] = 0.f; n[1][1] = (float)((double)(float)H / 2.0); n[1][2] = 0.f; n[1][3] = (float)((double)(float)(H - 1) / 2.0);
            n[2][0] = 0.f; n[2][1] = 0.f; n[2][2] = 0.f; n[2][3] = 1.f;
            float A[3][4], T[3][3];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    A[r][j] = S[r][0] * proj[j] + S[r][1] * proj[4 + j] + S[r][2] * proj[8 + j] + S[r][3] * proj[12 + j];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 3; r++)
                    T[c][r] = A[r][0] * n[c][0] + A[r][1] * n[c][1] + A[r][2] * n[c][2] + A[r][3] * n[c][3];
            Tu = {T[0][0], T[0][1], T[0][2]};
            Tv = {T[1][0], T[1][1], T[1][2]};
            Tw = {T[2][0], T[2][1], T[2][2]};
            normal = {view[0] * L2.x + view[4] * L2.y + view[8] * L2.z, view[1] * L2.x + view[5] * L2.y + view[9] * L2.z,
                      view[2] * L2.x + view[6] * L2.y + view[10] * L2.z};
        } else {
            const float* t = tm_pre + 9 * (size_t)i;
            Tu = {t[0], t[1], t[2]}; Tv = {t[3], t[4], t[5]}; Tw = {t[6], t[7], t[8]};
            normal = {0.f, 0.f, 1.f};
        }
        const F3 pn = pv * normal;
        const float cosv = -(pn.x + pn.y + pn.z);
        if (cosv == 0.0f) break;
        normal = (cosv > 0 ? 1.0f : -1.0f) * normal;
        // 3-sigma screen-space box, forward.cu:119-145
        const F3 tt = {9.0f, 9.0f, -1.0f};
        const float d = dot3(tt, Tw * Tw);
        if (d == 0.0f) break;
        const F3 f = (1.0f / d) * tt;
        const float cx = dot3(f, Tu * Tw), cy = dot3(f, Tv * Tw);
        const float hx = cx * cx - dot3(f, Tu * Tu);
        const float hy = cy * cy - dot3(f, Tv * Tv);
        const float ex = __builtin_sqrtf(1e-4f < hx ? hx : 1e-4f);
        const float ey = __builtin_sqrtf(1e-4f < hy ? hy : 1e-4f);
        const float em = ex < ey ? ey : ex;
        const float fmin_r = 3.0f * FILTER_SIZE;
        const float radius = __builtin_ceilf(em < fmin_r ? fmin_r : em);
        int x0, y0, x1, y1;
        tile_rect(cx, cy, sat_i32(radius), gx, gy, x0, y0, x1, y1);
        if ((unsigned)(x1 - x0) * (unsigned)(y1 - y0) == 0u) break;
        // Bounds outside which alpha < 1/255 is certain, in pixels: a box and the same along the two diagonals.  They depend
        // on the Gaussian only, so they are computed here, once, and every (tile, Gaussian) instance the blend kernel stages
        // reads them (they used to be recomputed per instance and per feature pass: a division, three square roots and
        // ~200 instructions each, by two of a workgroup's four waves while the other two waited at the barrier).
        const float opa = opacities[i];
        float skip = __builtin_inff();
        if (opa <= 1.0f) {
            const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
            skip = 2.0f * l * 1.01f + 0.05f;
        }
        const float4 cb = splat_cull_box(Tu, Tv, Tw, cx, cy, skip);
        reinterpret_cast<float4*>(g.cull + (size_t)i * 8)[0] = cb;
        reinterpret_cast<float4*>(g.cull + (size_t)i * 8)[1] = splat_cull_diag(Tu, Tv, Tw, cx, cy, skip);
        if (tight_rects) {
            // The reference bins a splat into the SQUARE of its larger 3-sigma extent.  Outside the box above
            // alpha < 1/255 is certain (the blend loops skip such pairs anyway), so tiles the box does not reach are
            // dropped from the splat's rectangle: fewer instances to count, scatter, sort and stage.  `radii` is not
            // changed.  Not used in the EXACT mode, whose tile lists are the reference's bit for bit.
            if (cb.x > -1e30f && cb.y < 1e30f && cb.z > -1e30f && cb.w < 1e30f) {
                const int bx0 = (int)fmaxf(0.0f, __builtin_floorf(cb.x * (1.0f / TILE)));
                const int bx1 = (int)fminf((float)gx, __builtin_floorf(cb.y * (1.0f / TILE)) + 1.0f);
                const int by0 = (int)fmaxf(0.0f, __builtin_floorf(cb.z * (1.0f / TILE)));
                const int by1 = (int)fminf((float)gy, __builtin_floorf(cb.w * (1.0f / TILE)) + 1.0f);
                x0 = max(x0, bx0); x1 = min(x1, bx1); y0 = max(y0, by0); y1 = min(y1, by1);
                if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; }      // reaches no tile: zero instances, radius kept
            }
        }

        float* rec = g.rec + (size_t)i * REC;
        F3 rgb = {0.f, 0.f, 0.f};
        unsigned cm = 0;
        if (col_pre == nullptr) {
            if constexpr (STAGE_SH) rgb = sh_to_rgb(D, p, F3{campos[0], campos[1], campos[2]}, s_sh + threadIdx.x * SH_STRIDE, cm);
            else rgb = sh_to_rgb(D, p, F3{campos[0], campos[1], campos[2]}, shs + (size_t)i * M * 3, cm);
        } else {
            rgb = {col_pre[3 * (size_t)i], col_pre[3 * (size_t)i + 1], col_pre[3 * (size_t)i + 2]};
        }
        g.clamped[i] = (uint8_t)cm;
        float4* r4 = reinterpret_cast<float4*>(rec);
        r4[0] = make_float4(Tu.x, Tu.y, Tu.z, Tv.x);
        r4[1] = make_float4(Tv.y, Tv.z, Tw.x, Tw.y);
        r4[2] = make_float4(Tw.z, cx, cy, normal.x);
        r4[3] = make_float4(normal.y, normal.z, opacities[i], rgb.x);
        r4[4] = make_float4(rgb.y, rgb.z, pv.z, 0.0f);
        radius_i = sat_i32(radius);
        touched = (unsigned)(y1 - y0) * (unsigned)(x1 - x0);
        rc = {(uint16_t)x0, (uint16_t)y0, (uint16_t)x1, (uint16_t)y1};
    } while (false);
    if (i < P) {
        radii_out[i] = radius_i;
        g.radii[i] = radius_i;
        g.tiles_touched[i] = touched;
        g.rect[i] = rc;
    }
    // ---- tile counts: merged per workgroup in LDS, one global atomic per distinct tile ----
    __syncthreads();                        // every lane is done with its staged SH row: the area becomes the hash table
    uint32_t* s_bigx = th_cnt + TH_SIZE;    // (behind the hash table, inside the same LDS block: no occupancy lost)
    uint32_t* s_bigy = s_bigx + 256;
    int& s_nbig = *reinterpret_cast<int*>(s_bigy + 256);
    if (threadIdx.x == 0) s_nbig = 0;
    for (int e = threadIdx.x; e < TH_SIZE; e += 256) { th_key[e] = TH_EMPTY; th_cnt[e] = 0u; }
    __syncthreads();
    // A splat that reaches more than BIG_RECT tiles (a background surfel grown over the whole view: 1 617 tiles at 779x519)
    // is not walked by its own lane - one lane looping while its workgroup waits at the barrier; a few such splats made this
    // kernel 5x and k_scatter 30x slower late in a train.py run - but by the whole workgroup, a tile per thread.
    const bool big = (unsigned)(rc.x1 - rc.x0) * (unsigned)(rc.y1 - rc.y0) > (unsigned)BIG_RECT;
    if (big) {
        const int k = atomicAdd(&s_nbig, 1);
        s_bigx[k] = (uint32_t)rc.x0 | ((uint32_t)rc.x1 << 16);
        s_bigy[k] = (uint32_t)rc.y0 | ((uint32_t)rc.y1 << 16);
    }
    __syncthreads();
    const int sub = counter_sub(blockIdx.x);
    if (!big)
        for (int y = rc.y0; y < rc.y1; y++)
            for (int x = rc.x0; x < rc.x1; x++) {
                const uint32_t tile = (uint32_t)y * gx + x;
                const int slot = th_find_or_insert(th_key, tile);
                if (slot >= 0) atomicAdd(&th_cnt[slot], 1u);
                else atomicAdd(tile_count + ((size_t)tile * CNT_SUB + sub) * CNT_STRIDE, 1u);
            }
    for (int b = 0; b < s_nbig; b++) {
        const int bx0 = (int)(s_bigx[b] & 0xffffu), bx1 = (int)(s_bigx[b] >> 16), by0 = (int)(s_bigy[b] & 0xffffu), by1 = (int)(s_bigy[b] >> 16);
        const int w = bx1 - bx0, area = w * (by1 - by0);
        for (int e = threadIdx.x; e < area; e += 256) {
            const uint32_t tile = (uint32_t)(by0 + e / w) * gx + (uint32_t)(bx0 + e % w);
            atomicAdd(tile_count + ((size_t)tile * CNT_SUB + sub) * CNT_STRIDE, 1u);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TH_SIZE; e += 256)
        if (th_key[e] != TH_EMPTY) atomicAdd(tile_count + ((size_t)th_key[e] * CNT_SUB + sub) * CNT_STRIDE, th_cnt[e]);
}

// ----------------------------------------------------------------------------
// Scans (exact u32).  Small single-block scan over tiles; three-pass scan over Gaussians.
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

// dense copy of the padded sub-counters (and reset of the scatter cursors), then a single-workgroup scan
__global__ __launch_bounds__(256) void k_gather_counts(int E, const uint32_t* __restrict__ count, uint32_t* __restrict__ dense,
                                                       uint32_t* __restrict__ cursor) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    dense[i] = count[(size_t)i * CNT_STRIDE];
    cursor[(size_t)i * CNT_STRIDE] = 0u;
}

__global__ __launch_bounds__(1024) void k_tile_scan(int T, uint32_t* __restrict__ sub_offset /* in: counts, out: offsets */,
                                                    uint32_t* __restrict__ offset, int64_t* header,
                                                    uint32_t* __restrict__ tile_order, int order_classes) {
    // one workgroup; every thread owns a contiguous run of elements (serial prefix in registers, 16-byte accesses) and
    // the 1024 run totals are scanned once — instead of E / 1024 dependent workgroup scans
    __shared__ uint32_t s_warp[32];
    const int E = T * CNT_SUB;              // scan over (tile, sub-counter) in tile-major order
    constexpr int RUN = 8;                  // elements per thread and round
    uint32_t carry = 0;
    for (int base = 0; base < E; base += 1024 * RUN) {
        const int i0 = base + threadIdx.x * RUN;
        uint32_t v[RUN];
        if (i0 + RUN <= E) {
            const uint4 a = *reinterpret_cast<const uint4*>(sub_offset + i0), b = *reinterpret_cast<const uint4*>(sub_offset + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int u = 0; u < RUN; u++) v[u] = i0 + u < E ? sub_offset[i0 + u] : 0u;
        }
        uint32_t run = 0;
#pragma unroll
        for (int u = 0; u < RUN; u++) { const uint32_t c = v[u]; v[u] = run; run += c; }
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(run, s_warp, total);
#pragma unroll
        for (int u = 0; u < RUN; u++) {
            const int i = i0 + u;
            if (i < E) {
                const uint32_t o = carry + ex + v[u];
                sub_offset[i] = o;
                if ((i & (CNT_SUB - 1)) == 0) offset[i / CNT_SUB] = o;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        offset[T] = carry;
        header[0] = (int64_t)carry;
    }
    // Launch order for kernels whose grid is only a few waves per SIMD (k_render_bwd_geo at 779x519: 1.6): heaviest tiles
    // first, so that the long lists start at once and the short ones fill in behind them.  A STABLE counting sort into
    // `ncls` classes of the tile sizes relative to the largest one (tiles of a class keep their row-major order, i.e.
    // neighbours - which share splats - still run close in time): thread i owns tiles [8 i, 8 i + 8).
    __shared__ uint32_t s_cls[16 * 1024];                 // [class][thread] counts, then offsets
    if (T < 4096) {
        // small views (where this kernel sits in the step's chain): 256 classes of 16 instances, LDS atomics - which tile of a
        // class comes first is left to the atomics (it changes no result)
        if (threadIdx.x < 256) s_cls[threadIdx.x] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += 1024) {
            const uint32_t n = offset[t + 1] - offset[t];
            atomicAdd(&s_cls[255u - min(255u, n >> 4)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {                  // exclusive scan of the 256 class sizes: four per lane + a wave scan
            uint32_t c[4], run = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) { c[u] = s_cls[threadIdx.x * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t x = c[u]; c[u] = run; run += x; }
            uint32_t inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if ((int)threadIdx.x >= d) inc += o;
            }
            const uint32_t ex = inc - run;
#pragma unroll
            for (int u = 0; u < 4; u++) s_cls[threadIdx.x * 4 + u] = ex + c[u];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += 1024) {
            const uint32_t n = offset[t + 1] - offset[t];
            tile_order[atomicAdd(&s_cls[255u - min(255u, n >> 4)], 1u)] = (uint32_t)t;
        }
    } else {
        __shared__ uint32_t s_max[1];
        const int ncls = order_classes < 1 ? 1 : (order_classes > 16 ? 16 : order_classes);
        if (threadIdx.x == 0) s_max[0] = 1;
        for (int e = threadIdx.x; e < ncls * 1024; e += 1024) s_cls[e] = 0;
        __syncthreads();
        const int per = (T + 1023) / 1024;
        const int t0 = threadIdx.x * per, t1 = min(T, t0 + per);
        uint32_t mx = 0;
        for (int t = t0; t < t1; t++) mx = max(mx, offset[t + 1] - offset[t]);
        atomicMax(&s_max[0], mx);
        __syncthreads();
        const uint32_t top = s_max[0];
        auto cls_of = [&](uint32_t n) { return (int)min((uint32_t)(ncls - 1), (uint32_t)(((unsigned long long)(top - n) * ncls) / (top + 1u))); };
        for (int t = t0; t < t1; t++) s_cls[cls_of(offset[t + 1] - offset[t]) * 1024 + threadIdx.x]++;
        __syncthreads();
        // exclusive scan over (class, thread): ncls * 1024 counters, ncls per thread + one block scan
        uint32_t mine[16], run = 0;
        for (int k = 0; k < ncls; k++) { const int e = threadIdx.x * ncls + k; mine[k] = s_cls[e]; }
        for (int k = 0; k < ncls; k++) { const uint32_t x = mine[k]; mine[k] = run; run += x; }
        uint32_t total;
        __syncthreads();
        const uint32_t ex = block_exclusive_scan_1024(run, s_warp, total);
        for (int k = 0; k < ncls; k++) s_cls[threadIdx.x * ncls + k] = ex + mine[k];
        __syncthreads();
        uint32_t cur[16];
        for (int k = 0; k < ncls; k++) cur[k] = s_cls[k * 1024 + threadIdx.x];
        for (int t = t0; t < t1; t++) {
            const int c = cls_of(offset[t + 1] - offset[t]);
            uint32_t at = 0;
            for (int k = 0; k < ncls; k++) if (k == c) at = cur[k]++;
            tile_order[at] = (uint32_t)t;
        }
    }
}

// block_max (optional): the largest input of each block; k_scan_tops reduces them into header[1] = the largest number of
// tiles any splat of this view touches - k_scatter and k_preprocess_bwd skip their workgroup-cooperative paths (and the
// barrier those need) when no splat is large.
__global__ __launch_bounds__(1024) void k_scan_blocks(int n, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                      uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_max) {
    __shared__ uint32_t s_warp[32];
    __shared__ uint32_t s_mx[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_1024(v, s_warp, total);
    if (i < n) out[i] = ex;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
    if (block_max != nullptr) {
        uint32_t m = v;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t mm = 0;
            for (int w = 0; w < 16; w++) mm = max(mm, s_mx[w]);
            block_max[blockIdx.x] = mm;
        }
    }
}
__global__ __launch_bounds__(1024) void k_scan_tops(int nb, uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ block_max,
                                                    int64_t* __restrict__ header) {
    __shared__ uint32_t s_warp[32];
    if (block_max != nullptr && threadIdx.x < 64) {
        uint32_t m = 0;
        for (int b = threadIdx.x; b < nb; b += 64) m = max(m, block_max[b]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) header[1] = (int64_t)m;
    }
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_warp, total);
        if (i < nb) block_sums[i] = carry + ex;
        carry += total;
    }
}
__global__ __launch_bounds__(1024) void k_scan_add(int n, uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sums) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}

// ----------------------------------------------------------------------------
// Scatter: one (depth_bits, gaussian) key per touched tile into that tile's bucket.  The workgroup first counts its keys
// per tile in the LDS hash, reserves one contiguous range per distinct tile with ONE returning global atomic, and then
// hands out the positions inside the ranges with LDS atomics — so the keys of a workgroup that go to the same bucket are
// neighbours in memory.  (The order inside a bucket is irrelevant: the per-tile sort orders the keys completely.)
__global__ __launch_bounds__(256) void k_scatter(int P, int gx, GeomView g, const uint32_t* __restrict__ sub_offset,
                                                 uint32_t* __restrict__ tile_cursor, unsigned long long* __restrict__ keys,
                                                 int64_t capacity) {
    __shared__ uint32_t th_key[TH_SIZE], th_cnt[TH_SIZE], th_base[TH_SIZE];
    for (int e = threadIdx.x; e < TH_SIZE; e += 256) { th_key[e] = TH_EMPTY; th_cnt[e] = 0u; }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = counter_sub(blockIdx.x);
    Rect16 rc = {0, 0, 0, 0};
    unsigned long long key = 0ull;
    if (i < P && g.tiles_touched[i] != 0) {
        rc = g.rect[i];
        key = ((unsigned long long)__float_as_uint(g.rec[(size_t)i * REC + 18]) << 32) | (unsigned)i;
    }
    constexpr int BIG_MAX = 64;             // (1 KB of LDS: six workgroups per CU as before; a 65th large splat of a workgroup walks alone)
    __shared__ int s_nbig;
    __shared__ uint32_t s_bigx[BIG_MAX], s_bigy[BIG_MAX];
    __shared__ unsigned long long s_bigkey[BIG_MAX];
    const bool any_big = g.header[1] > (int64_t)BIG_RECT;       // (view-uniform: k_scan_tops)
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const bool big = any_big && (unsigned)(rc.x1 - rc.x0) * (unsigned)(rc.y1 - rc.y0) > (unsigned)BIG_RECT;      // see k_preprocess
    if (big) {
        const int k = atomicAdd(&s_nbig, 1);
        if (k < BIG_MAX) {
            s_bigx[k] = (uint32_t)rc.x0 | ((uint32_t)rc.x1 << 16);
            s_bigy[k] = (uint32_t)rc.y0 | ((uint32_t)rc.y1 << 16);
            s_bigkey[k] = key;
            rc = {0, 0, 0, 0};              // its own lane walks nothing
        }
    }
    for (int y = rc.y0; y < rc.y1; y++)
        for (int x = rc.x0; x < rc.x1; x++) {
            const int slot = th_find_or_insert(th_key, (uint32_t)y * gx + x);
            if (slot >= 0) atomicAdd(&th_cnt[slot], 1u);
        }
    __syncthreads();
    for (int b = 0; b < min(s_nbig, BIG_MAX); b++) {      // the workgroup's large rectangles: a tile per thread, positions straight from the cursors
        const int bx0 = (int)(s_bigx[b] & 0xffffu), bx1 = (int)(s_bigx[b] >> 16), by0 = (int)(s_bigy[b] & 0xffffu), by1 = (int)(s_bigy[b] >> 16);
        const int w = bx1 - bx0, area = w * (by1 - by0);
        const unsigned long long bkey = s_bigkey[b];
        for (int e = threadIdx.x; e < area; e += 256) {
            const uint32_t tile = (uint32_t)(by0 + e / w) * gx + (uint32_t)(bx0 + e % w);
            const size_t ee = (size_t)tile * CNT_SUB + sub;
            const uint32_t pos = atomicAdd(tile_cursor + ee * CNT_STRIDE, 1u);
            const int64_t at = (int64_t)sub_offset[ee] + pos;
            if (at < capacity) keys[at] = bkey;
        }
    }
    for (int e = threadIdx.x; e < TH_SIZE; e += 256)
        if (th_key[e] != TH_EMPTY) {
            th_base[e] = atomicAdd(tile_cursor + ((size_t)th_key[e] * CNT_SUB + sub) * CNT_STRIDE, th_cnt[e]);
            th_cnt[e] = 0u;                 // becomes the fill counter of the reserved range
        }
    __syncthreads();
    for (int y = rc.y0; y < rc.y1; y++)
        for (int x = rc.x0; x < rc.x1; x++) {
            const uint32_t tile = (uint32_t)y * gx + x;
            const size_t e = (size_t)tile * CNT_SUB + sub;
            const int slot = th_find(th_key, tile);
            uint32_t pos;
            if (slot >= 0) pos = th_base[slot] + atomicAdd(&th_cnt[slot], 1u);
            else pos = atomicAdd(tile_cursor + e * CNT_STRIDE, 1u);
            const int64_t at = (int64_t)sub_offset[e] + pos;
            if (at < capacity) keys[at] = key;
        }
}

// ----------------------------------------------------------------------------
// Per-tile bucket sort.  All comparators are ascending (min to the lower index), so
// indices >= n behave as +inf without being stored.
constexpr int SORT_LDS_KEYS = 4096;

// Two network stages per pass: every thread loads the 4 keys of a group that is closed under both stages, does the
// four compare-exchanges in registers and stores them back — half the LDS traffic and half the barriers of a
// stage-per-pass bitonic sort (30 passes instead of 55 for 1024 keys).
//   flip(k) + step(k/4): {b+o, b+o+k/4, b+k-1-o-k/4, b+k-1-o}, o < k/4
//   step(j) + step(j/2): {p, p+j/2, p+j, p+3j/2}
#define ISR_CMPX(x, y) { if ((y) < (x)) { const unsigned long long t_ = (x); (x) = (y); (y) = t_; } }
template <typename KeyPtr>
__device__ __forceinline__ void sort_group4(KeyPtr a, int n, int i0, int i1, int i2, int i3, bool flip_first) {
    if (i0 >= n) return;                 // i0 is the smallest index: nothing real in the group
    const unsigned long long INF = ~0ull;
    unsigned long long v0 = a[i0], v1 = i1 < n ? a[i1] : INF, v2 = i2 < n ? a[i2] : INF, v3 = i3 < n ? a[i3] : INF;
    if (flip_first) { ISR_CMPX(v0, v3); ISR_CMPX(v1, v2); ISR_CMPX(v0, v1); ISR_CMPX(v2, v3); }
    else { ISR_CMPX(v0, v2); ISR_CMPX(v1, v3); ISR_CMPX(v0, v1); ISR_CMPX(v2, v3); }
    a[i0] = v0;                          // +inf never moves below a real key, so slots >= n stay virtual
    if (i1 < n) a[i1] = v1;
    if (i2 < n) a[i2] = v2;
    if (i3 < n) a[i3] = v3;
}

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_flip_sort(KeyPtr a, int n) {
    int npad = 1;
    while (npad < n) npad <<= 1;
    const int half = npad >> 1, quarter = npad >> 2;
    for (int k = 2; k <= npad; k <<= 1) {
        int j;
        if (k == 2) {
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                const int lo = 2 * i, hi = lo + 1;
                if (hi < n) {
                    const unsigned long long x = a[lo], y = a[hi];
                    if (y < x) { a[lo] = y; a[hi] = x; }
                }
            }
            __syncthreads();
            continue;
        }
        {   // flip(k) + step(k/4)
            const int q = k >> 2;
            for (int i = threadIdx.x; i < quarter; i += blockDim.x) {
                const int blk = i / q, o = i - blk * q, base = blk * k;
                sort_group4(a, n, base + o, base + o + q, base + k - 1 - o - q, base + k - 1 - o, true);
            }
            __syncthreads();
            j = k >> 3;
        }
        for (; j >= 2; j >>= 2) {   // step(j) + step(j/2)
            const int h = j >> 1;
            for (int i = threadIdx.x; i < quarter; i += blockDim.x) {
                const int p = (i / h) * 2 * j + (i % h);
                sort_group4(a, n, p, p + h, p + j, p + j + h, false);
            }
            __syncthreads();
        }
        if (j == 1) {               // a single step(1) is left over
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                const int lo = 2 * i, hi = lo + 1;
                if (hi < n) {
                    const unsigned long long x = a[lo], y = a[hi];
                    if (y < x) { a[lo] = y; a[hi] = x; }
                }
            }
            __syncthreads();
        }
    }
}
// The same network for a bucket that lives in LDS - the common case, and instruction-bound: 30 passes of ~90 vector
// instructions over 8160 tiles x 4 waves were 0.13 ms at 1080p.  Here the slots [n, npad) really hold +inf (the array has
// room up to the next power of two), so no access is bounds-checked, and every index is shifts and masks of powers of two
// instead of divisions by run-time values.
__device__ __forceinline__ void sort_group4_lds(unsigned long long* a, int i0, int i1, int i2, int i3, bool flip_first) {
    unsigned long long v0 = a[i0], v1 = a[i1], v2 = a[i2], v3 = a[i3];
    if (flip_first) { ISR_CMPX(v0, v3); ISR_CMPX(v1, v2); ISR_CMPX(v0, v1); ISR_CMPX(v2, v3); }
    else { ISR_CMPX(v0, v2); ISR_CMPX(v1, v3); ISR_CMPX(v0, v1); ISR_CMPX(v2, v3); }
    a[i0] = v0; a[i1] = v1; a[i2] = v2; a[i3] = v3;
}
__device__ __forceinline__ void sort_pairs_lds(unsigned long long* a, int half) {
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const unsigned long long x = a[2 * i], y = a[2 * i + 1];
        if (y < x) { a[2 * i] = y; a[2 * i + 1] = x; }
    }
    __syncthreads();
}
__device__ __forceinline__ void bitonic_flip_sort_lds(unsigned long long* a, int n) {
    int lg = 0;
    while ((1 << lg) < n) lg++;
    const int npad = 1 << lg, half = npad >> 1, quarter = npad >> 2;
    for (int i = n + threadIdx.x; i < npad; i += blockDim.x) a[i] = ~0ull;
    __syncthreads();
    for (int lk = 1; lk <= lg; lk++) {          // k = 1 << lk
        if (lk == 1) { sort_pairs_lds(a, half); continue; }
        {   // flip(k) + step(k/4)
            const int lq = lk - 2, q = 1 << lq, k1 = (1 << lk) - 1;
            for (int i = threadIdx.x; i < quarter; i += blockDim.x) {
                const int o = i & (q - 1), base = (i >> lq) << lk;
                sort_group4_lds(a, base + o, base + o + q, base + k1 - o - q, base + k1 - o, true);
            }
            __syncthreads();
        }
        int lj = lk - 3;                        // j = k / 8 = 1 << lj
        for (; lj >= 1; lj -= 2) {              // step(j) + step(j/2)
            const int j = 1 << lj, h = j >> 1, lh = lj - 1;
            for (int i = threadIdx.x; i < quarter; i += blockDim.x) {
                const int p = ((i >> lh) << (lj + 1)) + (i & (h - 1));
                sort_group4_lds(a, p, p + h, p + j, p + j + h, false);
            }
            __syncthreads();
        }
        if (lj == 0) sort_pairs_lds(a, half);   // a single step(1) is left over
    }
}
#undef ISR_CMPX

// Buckets of SORT_LDS_KEYS < n <= SORT_BIG_KEYS keys (dense scenes: C5 averages 3 100 instances per tile) are sorted by
// k_tile_sort_big - 1024 threads, 128 KB of dynamic LDS, one workgroup per CU - instead of the in-place network in global
// memory (a barrier and a round trip to L2 per pass: 2.1 ms of C5's side stream).  big_follows: this launch leaves them alone.
constexpr int SORT_BIG_KEYS = 16384;
__global__ __launch_bounds__(1024) void k_tile_sort_big(const uint32_t* __restrict__ tile_offset, unsigned long long* keys,
                                                        uint32_t* __restrict__ point_list, int64_t capacity, int min_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_big[];
    const uint32_t t = blockIdx.x;
    const int64_t r0 = tile_offset[t];
    int64_t r1 = tile_offset[t + 1];
    if (r1 > capacity) r1 = capacity;
    const int n = (int)(r1 - r0);
    if (n <= min_n || n > SORT_BIG_KEYS) return;
    unsigned long long* seg = keys + r0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_big[i] = seg[i];
    bitonic_flip_sort_lds(s_big, n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) point_list[r0 + i] = (uint32_t)s_big[i];
}

// Buckets of up to SORT_WAVE_KEYS keys (all but the densest tiles of a 1080p view) are sorted by ONE WAVE in registers:
// lane l holds K consecutive elements of the (virtual) array, so the network's strides below K are compare-exchanges between
// a lane's own registers and the strides of K and more are exchanges with lane l ^ (stride / K) (two ds_bpermute per key) -
// no LDS allocation, no barrier, a third of the LDS network's instructions, 6-8 waves per SIMD.  The input order is
// irrelevant to a sort, so the bucket is read striped (coalesced) and only the sorted ids are written lane-contiguous.
constexpr int SORT_WAVE_KEYS = 2048;
struct __attribute__((packed, aligned(4))) Ids4 { uint32_t x, y, z, w; };

template <int K>
__device__ __forceinline__ void wave_cmpx(unsigned long long& a, unsigned long long& b, bool desc) {
    const bool sw = (a > b) != desc;
    const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}

// in-register half-cleaner cascade: strides J, J/2, .. 1 over the K registers of every lane, one direction per lane
template <int K, int J>
__device__ __forceinline__ void wave_merge_regs(unsigned long long (&v)[K], bool desc) {
    if constexpr (J >= 1) {
#pragma unroll
        for (int r = 0; r < K; r++)
            if ((r & J) == 0) wave_cmpx<K>(v[r], v[r | J], desc);
        wave_merge_regs<K, J / 2>(v, desc);
    }
}

// phases k = 2 .. K of the network (strides inside a lane): direction of element i = l K + r is bit k of i
template <int K, int KK>
__device__ __forceinline__ void wave_sort_regs(unsigned long long (&v)[K], int lane) {
    if constexpr (KK <= K) {
        if constexpr (KK > 2) wave_sort_regs<K, KK / 2>(v, lane);
        // stride KK/2 .. 1 with per-register directions (bit KK of r; for KK == K: bit 0 of the lane)
#pragma unroll
        for (int j = KK / 2; j >= 1; j >>= 1) {
#pragma unroll
            for (int r = 0; r < K; r++)
                if ((r & j) == 0) {
                    const bool desc = KK < K ? (r & KK) != 0 : (lane & 1) != 0;
                    wave_cmpx<K>(v[r], v[r | j], desc);
                }
        }
    }
}

template <int K>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&v)[K], int lane) {
    if constexpr (K > 1) wave_sort_regs<K, K>(v, lane);
    // phases k = 2 K .. 64 K: lane-level bitonic merge (m = stride / K) followed by the in-register cascade
    for (int kk = 2; kk <= 64; kk <<= 1) {
        const bool desc = (lane & kk) != 0;                  // kk == 64: ascending everywhere
        for (int m = kk >> 1; m >= 1; m >>= 1) {
            const bool keep_min = ((lane & m) == 0) != desc;
            const int src = (lane ^ m) << 2;
#pragma unroll
            for (int r = 0; r < K; r++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)v[r]);
                const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(v[r] >> 32));
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                v[r] = ((o < v[r]) == keep_min) ? o : v[r];
            }
        }
        if constexpr (K > 1) wave_merge_regs<K, K / 2>(v, desc);
    }
}

template <int K>
__device__ __forceinline__ void wave_sort_bucket(const unsigned long long* __restrict__ seg, uint32_t* __restrict__ out, int n,
                                                 int lane) {
    unsigned long long v[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        const int i = r * 64 + lane;
        v[r] = i < n ? seg[i] : ~0ull;
    }
    wave_bitonic_sort<K>(v, lane);
    const int base = lane * K;
    if constexpr (K >= 4) {
        if (base + K <= n) {        // (dword-aligned 16-byte stores: the bucket's start is any multiple of 4 bytes)
#pragma unroll
            for (int r = 0; r < K; r += 4) {
                Ids4 q = {(uint32_t)v[r], (uint32_t)v[r + 1], (uint32_t)v[r + 2], (uint32_t)v[r + 3]};
                *reinterpret_cast<Ids4*>(out + base + r) = q;
            }
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < K; r++)
        if (base + r < n) out[base + r] = (uint32_t)v[r];
}

template <int MAXK>      // 32: buckets of up to 2 048 keys; 64 (dense scenes): up to 4 096, at half the waves per SIMD
__global__ __launch_bounds__(64) void k_tile_sort_wave(const uint32_t* __restrict__ tile_offset,
                                                       const unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ point_list, int64_t capacity) {
    const uint32_t t = blockIdx.x;
    const int64_t r0 = tile_offset[t];
    int64_t r1 = tile_offset[t + 1];
    if (r1 > capacity) r1 = capacity;
    const int n = (int)(r1 - r0);
    if (n <= 0 || n > 64 * MAXK) return;
    const unsigned long long* seg = keys + r0;
    uint32_t* out = point_list + r0;
    const int lane = threadIdx.x;
    if (n <= 64) wave_sort_bucket<1>(seg, out, n, lane);
    else if (n <= 128) wave_sort_bucket<2>(seg, out, n, lane);
    else if (n <= 256) wave_sort_bucket<4>(seg, out, n, lane);
    else if (n <= 512) wave_sort_bucket<8>(seg, out, n, lane);
    else if (n <= 1024) wave_sort_bucket<16>(seg, out, n, lane);
    else if (MAXK == 32 || n <= 2048) wave_sort_bucket<32>(seg, out, n, lane);
    else if (MAXK == 64 || n <= 4096) wave_sort_bucket<64>(seg, out, n, lane);
    else wave_sort_bucket<128>(seg, out, n, lane);
}

__global__ __launch_bounds__(256) void k_tile_sort(const uint32_t* __restrict__ tile_offset, unsigned long long* keys,
                                                   uint32_t* __restrict__ point_list, int64_t capacity, int big_follows) {
    __shared__ unsigned long long s_keys[SORT_LDS_KEYS];
    const uint32_t t = blockIdx.x;
    const int64_t r0 = tile_offset[t];
    int64_t r1 = tile_offset[t + 1];
    if (r1 > capacity) r1 = capacity;
    const int n = (int)(r1 - r0);
    if (n <= 0) return;
    if ((big_follows & 1) && n > SORT_LDS_KEYS && n <= SORT_BIG_KEYS) return;
    if ((big_follows & 2) && n <= SORT_WAVE_KEYS) return;         // k_tile_sort_wave's
    if ((big_follows & 4) && n <= 2 * SORT_WAVE_KEYS) return;     // k_tile_sort_wave<64>'s
    if ((big_follows & 8) && n <= 4 * SORT_WAVE_KEYS) return;     // k_tile_sort_wave<128>'s
    unsigned long long* seg = keys + r0;
    if (n <= SORT_LDS_KEYS) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) s_keys[i] = seg[i];
        bitonic_flip_sort_lds(s_keys, n);         // (its first barrier also covers the loads above)
        for (int i = threadIdx.x; i < n; i += blockDim.x) point_list[r0 + i] = (uint32_t)s_keys[i];
    } else {
        // rare: bucket larger than the LDS budget — same network, in place in global memory
        // (one workgroup; __syncthreads() orders the workgroup's own global accesses).
        bitonic_flip_sort(seg, n);
        for (int i = threadIdx.x; i < n; i += blockDim.x) point_list[r0 + i] = (uint32_t)seg[i];
    }
}

// ----------------------------------------------------------------------------
// K8: per-tile front-to-back blend.  256 threads = 4 waves; wave w owns the 8x8
// pixel block (w&1, w>>1) of the tile so that a splat's footprint diverges as
// little as possible inside a 64-lane wavefront.
template <class Math, int FCH, int BATCH>
__global__ __launch_bounds__(256, 4) void k_render_fwd(
    int W, int H, int ED, int ch_base, int first_pass, int gx, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const float* __restrict__ rec, const float* __restrict__ cull,
    const float* __restrict__ col_pre, const float* __restrict__ tm_pre, const float* __restrict__ extras,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_others, float* __restrict__ out_extra, int32_t* __restrict__ tracer, long long tracer_cap,
    int32_t* __restrict__ tracer_count, uint32_t* __restrict__ box4, int64_t capacity) {
    constexpr int RS = 16;   // staged floats per instance: Tu Tv Tw cx cy nx ny nz opa skip = 16
    __shared__ __attribute__((aligned(16))) float s_rec[BATCH * RS];
    __shared__ __attribute__((aligned(16))) float s_rgb[BATCH * 4];
    __shared__ __attribute__((aligned(16))) float s_feat[(FCH > 0 ? BATCH * FCH : 4)];
    __shared__ int s_id[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_box[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_diag[BATCH];     // the same bound along x + y and x - y
    // tracer (gaussian, pixel) pairs: every wave stages its pairs in its OWN LDS region and keeps the fill count in a
    // wave-uniform register, so an append is a ballot + popcount (no LDS atomic, no workgroup barrier); a region that
    // is nearly full is flushed by its wave alone with ONE global atomic (the reference does a global atomic on a
    // single counter per hit, forward.cu:425).
    constexpr int WCAP = 128;               // pairs per wave region; flushed when fewer than 64 slots are left
    __shared__ int s_trace[4 * 2 * WCAP];
    int wcnt = 0;

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned px = tx * TILE + (wv & 1) * 8 + (lane & 7);
    const unsigned py = ty * TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < (unsigned)W && py < (unsigned)H;
    const size_t N = (size_t)W * H;
    const size_t pix = (size_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;

    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int nfeat = FCH > 0 ? min(FCH, ED - ch_base) : 0;

    // Lane predicates live in 64-bit masks on the scalar unit (every test is a v_cmp into a mask; the one predicated region
    // per splat takes its exec mask from the combined mask): arithmetic-neutral, half the scalar instructions of the
    // compiler's exec-mask stacks for per-lane bools (see isr_forward_fast.hip).
    unsigned long long m_done = __ballot(!inside);      // lanes that have stopped (or lie outside the image)
    float T = 1.0f;
    unsigned contributor = 0, last_contributor = 0, median_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    // FAST arithmetic, 32-channel chunk: the feature accumulation  E[pix][ch] += w(pix, splat) * feat[splat][ch]  runs on
    // the matrix cores, two contributing splats per v_mfma_f32_32x32x2_f32 pair (one for pixels 0..31 of the wave, one
    // for 32..63) — 16 packed FMAs + 8 LDS reads per splat leave the VALU, which is what bounds this kernel.
    constexpr bool MF = Math::fast && FCH == 32;
    float E[(FCH > 0 && !MF) ? FCH : 1];
#pragma unroll
    for (int c = 0; c < ((FCH > 0 && !MF) ? FCH : 1); c++) E[c] = 0.0f;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
    float w_pend = 0.0f, f_pend = 0.0f;      // weight / feature fragment of a contributing splat waiting for its partner
    bool pending = false;                    // wave-uniform
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    const float bx0 = (float)(tx * TILE + (wv & 1) * 8), bx1 = bx0 + 7.0f;
    const float by0 = (float)(ty * TILE + (wv >> 1) * 8), by1 = by0 + 7.0f;

    auto flush_trace = [&]() {      // one wave, its own region: no workgroup barrier
        const int n = wcnt;
        if (n > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the lanes' LDS writes before the lanes' reads
            __builtin_amdgcn_wave_barrier();
            int gb = 0;
            if (lane == 0) gb = atomicAdd(tracer_count, n) + 1;        // counter starts at -1
            gb = __builtin_amdgcn_readfirstlane(gb);
            const int* src = s_trace + wv * 2 * WCAP;
            for (int e = lane; e < n; e += 64)
                if (gb + e < tracer_cap)
                    *reinterpret_cast<int2*>(tracer + 2 * (size_t)(gb + e)) = make_int2(src[2 * e], src[2 * e + 1]);
            __builtin_amdgcn_wave_barrier();
            wcnt = 0;
        }
    };
    // Staging: the ids of the NEXT batch are prefetched into a register while the current batch is blended, so a
    // round is: records (gather, ids already known) | barrier | features (gather by float4 column) | barrier | blend.
    // (Merging the feature gather into the record phase costs 27 more VGPRs -> 3 instead of 4 waves/SIMD: slower.)
    constexpr int QF4 = FCH > 0 ? FCH / 4 : 1;          // float4s per feature row
    const int my_inst = threadIdx.x & (BATCH - 1);
    int nid = (threadIdx.x < BATCH && r0 + my_inst < r1) ? (int)point_list[r0 + my_inst] : 0;
    for (int64_t base = r0; base < r1; base += BATCH) {
        if (__syncthreads_and(m_done == ~0ull)) break;
        const int nb = (int)min((int64_t)BATCH, r1 - base);
        const int id = nid;
        if (threadIdx.x < BATCH && base + BATCH + my_inst < r1) nid = (int)point_list[base + BATCH + my_inst];
        if (threadIdx.x < nb) {
            const int t = threadIdx.x;
            s_id[t] = id;
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)id * REC);
            float4 a = r4[0], b = r4[1], c = r4[2], d = r4[3], e = r4[4];
            if (tm_pre != nullptr) {
                const float* tp = tm_pre + 9 * (size_t)id;
                a = make_float4(tp[0], tp[1], tp[2], tp[3]);
                b = make_float4(tp[4], tp[5], tp[6], tp[7]);
                c.x = tp[8];
            }
            if (col_pre != nullptr) {
                d.w = col_pre[3 * (size_t)id]; e.x = col_pre[3 * (size_t)id + 1]; e.y = col_pre[3 * (size_t)id + 2];
            }
            const float opa = d.z;
            // conservative skip bound: opa*exp(-rho/2) < 1/255 for every rho > skip (1% + 0.05 margin)
            float skip = __builtin_inff();
            if (opa <= 1.0f) {
                const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
                skip = 2.0f * l * 1.01f + 0.05f;
            }
            float4* s4 = reinterpret_cast<float4*>(s_rec + t * RS);
            s4[0] = a;                                      // Tu.xyz Tv.x
            s4[1] = b;                                      // Tv.yz Tw.xy
            s4[2] = make_float4(c.x, c.y, c.z, c.w);        // Tw.z cx cy nx
            s4[3] = make_float4(d.x, d.y, opa, skip);       // ny nz opa skip
            const float4 cb = reinterpret_cast<const float4*>(cull + (size_t)id * 8)[0];      // per-Gaussian bounds from K1
            s_box[t] = cb;
            s_diag[t] = reinterpret_cast<const float4*>(cull + (size_t)id * 8)[1];
            if (first_pass) box4[base + t] = pack_box4(cb, (float)(tx * TILE), (float)(ty * TILE));
            reinterpret_cast<float4*>(s_rgb)[t] = make_float4(d.w, e.x, e.y, 0.0f);
        }
        __syncthreads();
        if (FCH > 0) {
            if ((ED & 3) == 0 && nfeat == FCH) {
                for (int e = threadIdx.x; e < nb * QF4; e += 256) {
                    const int inst = e / QF4, part = e - inst * QF4;
                    reinterpret_cast<float4*>(s_feat)[e] =
                        *reinterpret_cast<const float4*>(extras + (size_t)s_id[inst] * ED + ch_base + part * 4);
                }
            } else {
                for (int e = threadIdx.x; e < nb * FCH; e += 256) {
                    const int inst = e / FCH, c = e - inst * FCH;
                    s_feat[e] = c < nfeat ? extras[(size_t)s_id[inst] * ED + ch_base + c] : 0.0f;
                }
            }
            __syncthreads();
        }
        // ---- walk the batch: each wave visits only the splats whose cull box meets its 8x8 pixel block
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int jj = c0 + lane;
            bool hit = false;
            if (jj < nb) {
                const float4 bb = s_box[jj], dg = s_diag[jj];
                hit = !(bb.x > bx1) && !(bb.y < bx0) && !(bb.z > by1) && !(bb.w < by0) &&
                      !(dg.x > bx1 + by1) && !(dg.y < bx0 + by0) && !(dg.z > bx1 - by0) && !(dg.w < bx0 - by1);
            }
            unsigned long long m = __ballot(hit);
            while (m != 0ull) {
                if (m_done == ~0ull) break;
                const int j = c0 + __builtin_ctzll(m);
                m &= m - 1ull;
                // Flat, predicated evaluation (few exec-mask regions: the scalar unit is a co-bottleneck of this loop).
                // A lane for which `cand`/`ok` is false computes garbage that is never used — identical results to the
                // reference's chain of `continue`s.
                const float4 a = reinterpret_cast<const float4*>(s_rec + j * RS)[0];
                const float4 b = reinterpret_cast<const float4*>(s_rec + j * RS)[1];
                const float4 c = reinterpret_cast<const float4*>(s_rec + j * RS)[2];
                const float4 d = reinterpret_cast<const float4*>(s_rec + j * RS)[3];
                const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y}, Tw = {b.z, b.w, c.x};
                const F3 kk = {Math::msub(pxf, Tw.x, Tu.x), Math::msub(pxf, Tw.y, Tu.y), Math::msub(pxf, Tw.z, Tu.z)};
                const F3 ll = {Math::msub(pyf, Tw.x, Tv.x), Math::msub(pyf, Tw.y, Tv.y), Math::msub(pyf, Tw.z, Tv.z)};
                const F3 p = {Math::msub(kk.y, ll.z, kk.z * ll.y), Math::msub(kk.z, ll.x, kk.x * ll.z),
                              Math::msub(kk.x, ll.y, kk.y * ll.x)};
                const float dx = c.y - pxf, dy = c.z - pyf;
                const float rho2d = FILTER_INV_SQ * Math::mad(dy, dy, dx * dx);
                // exact-preserving early out: both candidate rho's certainly beyond the alpha<1/255 cut
                const float skip = d.w;
                // Wave-uniform masks are built from ballots of the individual compares and combined on the scalar unit
                // (a ballot of an already combined predicate costs two more vector instructions).
                const bool far_a = rho2d > skip, far_b = Math::mad(p.y, p.y, p.x * p.x) > skip * (p.z * p.z) * 1.01f;
                const bool nz = p.z != 0.0f;
                const unsigned long long m_cand = ~m_done & ~(__ballot(far_a) & __ballot(far_b)) & __ballot(nz);
                if (m_cand == 0ull) continue;
                const float sx = Math::div(p.x, p.z), sy = Math::div(p.y, p.z);
                const float rho3d = Math::mad(sy, sy, sx * sx);
                const float rho = fminf(rho3d, rho2d);
                const float depth = (rho3d <= rho2d) ? Math::mad(sy, Tw.y, sx * Tw.x) + Tw.z : Tw.z;
                const float power = -0.5f * rho;
                const float alpha = fminf(0.99f, d.z * Math::ex(power));
                const float test_T = T * (1 - alpha);
                const bool t_near = !(depth < NEAR_N), t_pow = !(power > 0.0f), t_alpha = !(alpha < 1.0f / 255.0f);
                const bool t_stop = test_T < 0.0001f;
                const unsigned long long m_pass = m_cand & __ballot(t_near) & __ballot(t_pow) & __ballot(t_alpha);
                const unsigned long long m_stop = m_pass & __ballot(t_stop);
                m_done |= m_stop;
                const unsigned long long m_ok = m_pass & ~m_stop;
                if (m_ok == 0ull) continue;
                float w_lane = 0.0f;
                if (__builtin_amdgcn_inverse_ballot_w64(m_ok)) {
                    const float w = alpha * T;
                    w_lane = w;
                    contributor = (unsigned)(base - r0) + (unsigned)j + 1u;
                    if (first_pass) {
                        const float A = 1 - T;
                        const float m_ = mscale * (1 - Math::div(NEAR_N, depth));
                        distortion += (Math::mad(m_ * m_, A, M2) - 2 * m_ * M1) * w;
                        D = Math::mad(depth, w, D);
                        M1 = Math::mad(m_, w, M1);
                        M2 = Math::mad(m_ * m_, w, M2);
                        if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                        N0 = Math::mad(c.w, w, N0); N1 = Math::mad(d.x, w, N1); N2 = Math::mad(d.y, w, N2);
                        const float4 col = reinterpret_cast<const float4*>(s_rgb)[j];
                        C0 = Math::mad(col.x, w, C0); C1 = Math::mad(col.y, w, C1); C2 = Math::mad(col.z, w, C2);
                    }
                    if (FCH > 0 && !MF) {
                        const float4* fj = reinterpret_cast<const float4*>(s_feat + j * FCH);
#pragma unroll
                        for (int q4 = 0; q4 < QF4; q4++) {
                            const float4 v = fj[q4];
                            if (Math::fast) {
                                E[4 * q4 + 0] = __builtin_fmaf(v.x, w, E[4 * q4 + 0]);
                                E[4 * q4 + 1] = __builtin_fmaf(v.y, w, E[4 * q4 + 1]);
                                E[4 * q4 + 2] = __builtin_fmaf(v.z, w, E[4 * q4 + 2]);
                                E[4 * q4 + 3] = __builtin_fmaf(v.w, w, E[4 * q4 + 3]);
                            } else {       // reference forward.cu:415 order: (e * alpha) * T
                                E[4 * q4 + 0] += v.x * alpha * T;
                                E[4 * q4 + 1] += v.y * alpha * T;
                                E[4 * q4 + 2] += v.z * alpha * T;
                                E[4 * q4 + 3] += v.w * alpha * T;
                            }
                        }
                    }
                    T = test_T;
                    last_contributor = contributor;
                }
                if (tracer != nullptr && first_pass) {
                    const unsigned long long m_tr = __ballot(w_lane >= 0.1f);     // (double)w > 0.1  <=>  w >= 0.1f
                    if (m_tr != 0ull) {
                        if (w_lane >= 0.1f) {
                            const int slot = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_tr >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m_tr, 0u));
                            int* dst = s_trace + wv * 2 * WCAP + 2 * slot;
                            dst[0] = s_id[j];
                            dst[1] = (int)pix;
                        }
                        wcnt += __popcll(m_tr);
                        if (wcnt > WCAP - 64) flush_trace();
                    }
                }
                if constexpr (MF) {
                    // A[i = channel][k = splat]: lanes 0..31 carry the pending splat's channels, 32..63 this splat's;
                    // B[k = splat][j = pixel]: v_permlane32_swap puts the two splats' weights of one half of the pixels
                    // into the two halves of the wave.
                    const float f_lane = s_feat[j * FCH + (lane & 31)];
                    if (!pending) { w_pend = w_lane; f_pend = f_lane; pending = true; }
                    else {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w_pend), __float_as_uint(w_lane), false, false);
                        const float a = lane < 32 ? f_pend : f_lane;
                        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[0]), accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[1]), accB, 0, 0, 0);
                        pending = false;
                    }
                }
            }
        }
    }
    if constexpr (MF) {
        if (pending) {          // odd number of contributing splats: pair the last one with a zero
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w_pend), 0u, false, false);
            const float a = lane < 32 ? f_pend : 0.0f;
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[0]), accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[1]), accB, 0, 0, 0);
        }
    }
    if (tracer != nullptr && first_pass) flush_trace();
    if (inside) {
        if (first_pass) {
            final_T[pix] = T;
            final_T[pix + N] = M1;
            final_T[pix + 2 * N] = M2;
            n_contrib[pix] = last_contributor;
            n_contrib[pix + N] = median_contributor;
            out_color[pix] = C0 + T * bg[0];
            out_color[N + pix] = C1 + T * bg[1];
            out_color[2 * N + pix] = C2 + T * bg[2];
            out_others[pix] = D;
            out_others[N + pix] = 1 - T;
            out_others[2 * N + pix] = N0;
            out_others[3 * N + pix] = N1;
            out_others[4 * N + pix] = N2;
            out_others[5 * N + pix] = median_depth;
            out_others[6 * N + pix] = distortion;
        }
        if (FCH > 0 && !MF) {
#pragma unroll
            for (int q = 0; q < FCH; q++)
                if (q < nfeat) out_extra[(size_t)(ch_base + q) * N + pix] = E[q];
        }
    }
    if constexpr (MF) {
        // D[row = channel (r&3) + 8*(r>>2) + 4*(lane>>5)][col = pixel lane&31 of the group]
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int p = grp * 32 + (lane & 31);                       // pixel (wave lane numbering) held by this lane
            const unsigned qx = tx * TILE + (wv & 1) * 8 + (p & 7), qy = ty * TILE + (wv >> 1) * 8 + (p >> 3);
            if (qx < (unsigned)W && qy < (unsigned)H) {
                const size_t qp = (size_t)W * qy + qx;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (ch < nfeat) out_extra[(size_t)(ch_base + ch) * N + qp] = grp == 0 ? accA[r] : accB[r];
                }
            }
        }
    }
}

// explicit instantiations used by the host API (isr_api.hip)
#define ISR_INST_FWD(M, F, B)                                                                                          \
    template __global__ void k_render_fwd<M, F, B>(int, int, int, int, int, int, const uint32_t*, const uint32_t*,     \
                                                   const float*, const float*, const float*, const float*, const float*, \
                                                   const float*, float*, uint32_t*, float*, float*, float*, int32_t*, \
                                                   long long, int32_t*, uint32_t*, int64_t);
ISR_INST_FWD(ExactMath, 0, 256)
ISR_INST_FWD(ExactMath, 8, 256)
ISR_INST_FWD(ExactMath, 16, 256)
ISR_INST_FWD(ExactMath, 32, 128)

}  // namespace isr

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
#include "isr_fast_pair.hpp"

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

template <bool STAGE_SH>
__global__ __launch_bounds__(256) void k_preprocess(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rots, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ tm_pre, const float* __restrict__ col_pre, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, int W, int H, int gx, int gy,
    int* __restrict__ radii_out, GeomView g, uint32_t* __restrict__ tile_count, int tight_rects) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) g.header[1] = 0;            // (k_gather_counts reduces the view's largest rectangle into it with atomicMax)
    // The workgroup's 256 SH rows (192 B each) are one contiguous 48 KB piece of `shs`: it is read with fully coalesced
    // 16-byte loads and handed to the owning lanes through LDS.  (A lane reading its own row touches 64 different
    // cache lines per load instruction, and with ~250 KB of rows in flight per CU the 32 KB vector cache keeps none of
    // them between the twelve loads of a row: 0.65 GB of fetches for 0.47 GB of input.)
    // The rows pass through LDS in two halves of 128 (ISR_K1_HALF_STAGE): 26 KB per workgroup instead of 52, five or six
    // workgroups per CU instead of three - the kernel streams 0.5 GB and was short of loads in flight, not of bandwidth.
#ifndef ISR_K1_HALF_STAGE
#define ISR_K1_HALF_STAGE 1
#endif
    constexpr int SH_STRIDE = 52;           // floats per staged row: 48 + 4 (16-byte aligned, spreads the LDS banks)
    constexpr int SH_ROWS = ISR_K1_HALF_STAGE ? 128 : 256;
    constexpr int BIG_WORDS = 2 * 256 + 4;  // list of the workgroup's large rectangles (see the tile counting below)
    constexpr int LDS_WORDS = STAGE_SH ? SH_ROWS * SH_STRIDE : 2 * TH_SIZE + BIG_WORDS;     // the tile hash + that list reuse the SH staging area
    static_assert(LDS_WORDS >= 2 * TH_SIZE + BIG_WORDS, "tile hash does not fit");
    __shared__ __attribute__((aligned(16))) float s_sh[LDS_WORDS];
    uint32_t* th_key = reinterpret_cast<uint32_t*>(s_sh);
    uint32_t* th_cnt = th_key + TH_SIZE;
    // every per-splat input is requested here, with the SH rows: one exposed memory latency per workgroup, not three
    // (rows, then the centre, then - behind the near cull - rotation, scale and opacity)
    F3 p_in = {0.f, 0.f, 0.f};
    float q_in[4] = {0.f, 0.f, 0.f, 0.f}, s_in[2] = {0.f, 0.f}, opa_in = 0.f;
    if (i < P) {
        p_in = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
        opa_in = opacities[i];
        if (tm_pre == nullptr) {
            q_in[0] = rots[4 * (size_t)i]; q_in[1] = rots[4 * (size_t)i + 1]; q_in[2] = rots[4 * (size_t)i + 2]; q_in[3] = rots[4 * (size_t)i + 3];
            s_in[0] = scales[2 * (size_t)i]; s_in[1] = scales[2 * (size_t)i + 1];
        }
    }
    F3 rgb_sh = {0.f, 0.f, 0.f};
    unsigned cm_sh = 0;
    if constexpr (STAGE_SH) {
        const int i0 = blockIdx.x * 256;
        const int nq = min(256, P - i0) * 12;                      // float4s to move
        const float4* src = reinterpret_cast<const float4*>(shs + (size_t)i0 * 48);
        float4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11;      // twelve independent loads in flight
#define ISR_SH_LD(u, v) { const int e = min((int)threadIdx.x + u * 256, nq - 1); v = src[e]; }
        ISR_SH_LD(0, v0) ISR_SH_LD(1, v1) ISR_SH_LD(2, v2) ISR_SH_LD(3, v3) ISR_SH_LD(4, v4) ISR_SH_LD(5, v5)
        ISR_SH_LD(6, v6) ISR_SH_LD(7, v7) ISR_SH_LD(8, v8) ISR_SH_LD(9, v9) ISR_SH_LD(10, v10) ISR_SH_LD(11, v11)
#undef ISR_SH_LD
        // (float4 e of the piece belongs to row e / 12: loads 0..5 of a thread are rows 0..127, loads 6..11 rows 128..255)
#define ISR_SH_ST(u, v) { const int e = (int)threadIdx.x + u * 256;                                               \
                          if (e < nq) { const int r = e / 12, k = e - r * 12;                                       \
                                        *reinterpret_cast<float4*>(s_sh + (r % SH_ROWS) * SH_STRIDE + 4 * k) = v; } }
        ISR_SH_ST(0, v0) ISR_SH_ST(1, v1) ISR_SH_ST(2, v2) ISR_SH_ST(3, v3) ISR_SH_ST(4, v4) ISR_SH_ST(5, v5)
        if (ISR_K1_HALF_STAGE) {
            __syncthreads();
            if (threadIdx.x < 128 && i < P && col_pre == nullptr)
                rgb_sh = sh_to_rgb(D, p_in, F3{campos[0], campos[1], campos[2]}, s_sh + threadIdx.x * SH_STRIDE, cm_sh);
            __syncthreads();
        }
        ISR_SH_ST(6, v6) ISR_SH_ST(7, v7) ISR_SH_ST(8, v8) ISR_SH_ST(9, v9) ISR_SH_ST(10, v10) ISR_SH_ST(11, v11)
#undef ISR_SH_ST
        __syncthreads();
        if (ISR_K1_HALF_STAGE && threadIdx.x >= 128 && i < P && col_pre == nullptr)
            rgb_sh = sh_to_rgb(D, p_in, F3{campos[0], campos[1], campos[2]}, s_sh + (threadIdx.x - 128) * SH_STRIDE, cm_sh);
    }
    int radius_i = 0;
    uint32_t touched = 0;
    Rect16 rc = {0, 0, 0, 0};
    do {
        if (i >= P) break;
        const F3 p = p_in;
        const F3 pv = {view[0] * p.x + view[4] * p.y + view[8] * p.z + view[12],
                       view[1] * p.x + view[5] * p.y + view[9] * p.z + view[13],
                       view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14]};
        if (pv.z <= 0.2f) break;   // near cull, auxiliary.h:201
        F3 Tu, Tv, Tw, normal;
        if (tm_pre == nullptr) {
            F3 c0, c1, c2;
            const float q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
            quat_to_cols(q, c0, c1, c2);
            const float sx = mod * s_in[0], sy = mod * s_in[1];
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
        const float opa = opa_in;
        float skip = __builtin_inff();
        if (opa <= 1.0f) {
            const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
            skip = 2.0f * l * 1.01f + 0.05f;
        }
        const float4 cb = splat_cull_box(Tu, Tv, Tw, cx, cy, skip);
        reinterpret_cast<float4*>(g.cull + (size_t)i * CULL_STRIDE)[0] = cb;
        reinterpret_cast<float4*>(g.cull + (size_t)i * CULL_STRIDE)[1] = splat_cull_diag(Tu, Tv, Tw, cx, cy, skip);
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
            if constexpr (STAGE_SH && ISR_K1_HALF_STAGE) { rgb = rgb_sh; cm = cm_sh; }
            else if constexpr (STAGE_SH) rgb = sh_to_rgb(D, p, F3{campos[0], campos[1], campos[2]}, s_sh + threadIdx.x * SH_STRIDE, cm);
            else rgb = sh_to_rgb(D, p, F3{campos[0], campos[1], campos[2]}, shs + (size_t)i * M * 3, cm);
        } else {
            rgb = {col_pre[3 * (size_t)i], col_pre[3 * (size_t)i + 1], col_pre[3 * (size_t)i + 2]};
        }
        g.clamped[i] = (uint8_t)cm;
        float4* r4 = reinterpret_cast<float4*>(rec);
        r4[0] = make_float4(Tu.x, Tu.y, Tu.z, Tv.x);
        r4[1] = make_float4(Tv.y, Tv.z, Tw.x, Tw.y);
        r4[2] = make_float4(Tw.z, cx, cy, normal.x);
        r4[3] = make_float4(normal.y, normal.z, opa, rgb.x);
        // rec[19]: the noise bound of FAST's rho against EXACT's over this splat's footprint (isr_fast_pair.hpp: guard bands)
        float exact_noise;
        r4[4] = make_float4(rgb.y, rgb.z, pv.z, splat_band(Tu, Tv, Tw, cx, cy, opa, cb, W, H, &exact_noise));
        {   // k_pack_hits' row of the splat: its alpha >= 1/255 ellipse (isr_common.hpp: splat_conic)
            const SplatConic cn = splat_conic(Tu, Tv, Tw, cx, cy, skip, cb, exact_noise);
            float4* c4 = reinterpret_cast<float4*>(g.ellipse + (size_t)i * CULL_STRIDE);
            c4[0] = cn.a; c4[1] = cn.b;
        }
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
    // ---- first level of the Gaussian offset scan (tiles touched -> row offsets of the backward's partial rows): exclusive scan
    // inside the workgroup, its total and its largest rectangle into scan_tmp; k_scan_add_tops256 adds the totals of the
    // workgroups before it (one launch, k_scan_blocks, less on the binning chain)
    {
        __shared__ uint32_t s_wsum[4], s_wmax[4];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const uint32_t v = i < P ? touched : 0u;
        uint32_t inc = v, mx = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += t;
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        if (lane == 63) { s_wsum[wv] = inc; s_wmax[wv] = mx; }
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wv; w++) base += s_wsum[w];
        if (i < P) g.point_offsets[i] = base + inc - v;
        if (threadIdx.x == 0) {
            g.scan_tmp[blockIdx.x] = (s_wsum[0] + s_wsum[1]) + (s_wsum[2] + s_wsum[3]);
            g.scan_tmp[gridDim.x + 1 + blockIdx.x] = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
        }
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
            const float4 cb = reinterpret_cast<const float4*>(cull + (size_t)id * CULL_STRIDE)[0];      // per-Gaussian bounds from K1
            s_box[t] = cb;
            s_diag[t] = reinterpret_cast<const float4*>(cull + (size_t)id * CULL_STRIDE)[1];
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

// ---------------------------------------------------------------------------------------------------------------------
// K8 in EXACT arithmetic with the per-block decomposition of isr_forward_fast.hip (k_render_fwd_fast_w): one wave = one
// workgroup = one 8x8 pixel block that finds its hits in k_pack_hits' masks, stages them (lane = hit) and walks them.  The
// per-pixel arithmetic is k_render_fwd<ExactMath>'s, operation for operation (the reference's order): every output stays
// bit-identical to the CPU oracle.  What changes is what a wave does NOT do any more: no workgroup barrier (the tile-wide
// kernel's waves wait for each other three times per 128 instances), no staging of instances that miss the block, no
// hit-mask iteration.
template <int FCH>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(4, 4))) void k_render_fwd_w(
    int W, int H, int ED, int ch_base, int first_pass, int gx, int tiles, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const float* __restrict__ rec, const float* __restrict__ col_pre,
    const float* __restrict__ tm_pre, const float* __restrict__ extras, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_others,
    float* __restrict__ out_extra, int32_t* __restrict__ tracer, long long tracer_cap, int32_t* __restrict__ tracer_count,
    const unsigned long long* __restrict__ hit_mask, int64_t capacity, const uint32_t* __restrict__ tile_order) {
    typedef ExactMath Math;
    constexpr int RS = 20, NH = 32, RING = 128;     // staged floats per hit: Tu Tv Tw cx cy n opa skip | rgb position
    __shared__ __attribute__((aligned(16))) float s_rec[NH * RS];
    __shared__ __attribute__((aligned(16))) float s_feat[FCH > 0 ? NH * FCH : 4];
    __shared__ __attribute__((aligned(8))) int2 s_ring[RING];
    constexpr int WCAP = 256;          // tracer pairs buffered per wave (the list's one counter: isr_forward_fast.hip)
    __shared__ int s_trace[2 * WCAP];
    int wcnt = 0;

    const int v = (int)blockIdx.x, kk = v >> 3;
    const int slot = (kk >> 2) * 8 + (v & 7), sub = kk & 3;
    if (slot >= tiles) return;
    const int tile = tile_order != nullptr ? (int)tile_order[slot] : slot;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const unsigned px = tx * TILE + (sub & 1) * 8 + (lane & 7), py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < (unsigned)W && py < (unsigned)H;
    const size_t N = (size_t)W * H;
    const size_t pix = (size_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    const int nfeat = FCH > 0 ? min(FCH, ED - ch_base) : 0;
    const size_t mask0 = hit_mask_word(r0, tile, 0) + (size_t)sub;

    unsigned long long m_done = __ballot(!inside);
    float T = 1.0f;
    unsigned last_contributor = 0, median_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    float E[FCH > 0 ? FCH : 1];
#pragma unroll
    for (int c = 0; c < (FCH > 0 ? FCH : 1); c++) E[c] = 0.0f;
    const float mscale = FAR_N / (FAR_N - NEAR_N);

    auto flush_trace = [&]() {
        const int n = wcnt;
        if (n > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int gb = 0;
            if (lane == 0) gb = atomicAdd(tracer_count, n) + 1;        // counter starts at -1
            gb = __builtin_amdgcn_readfirstlane(gb);
            for (int e = lane; e < n; e += 64)
                if (gb + e < tracer_cap)
                    *reinterpret_cast<int2*>(tracer + 2 * (size_t)(gb + e)) = make_int2(s_trace[2 * e], s_trace[2 * e + 1]);
            __builtin_amdgcn_wave_barrier();
            wcnt = 0;
        }
    };

    constexpr int QF4 = FCH > 0 ? FCH / 4 : 1;
    int scan = 0, head = 0, pend = 0;
    int nid = lane < len ? (int)point_list[r0 + lane] : 0;
    while (true) {
#pragma clang loop unroll(disable)
        while (pend < NH && scan < len) {
            const unsigned long long m = hit_mask[mask0 + (size_t)(scan >> 6) * HM_WORDS];
            const int i = scan + lane;
            const int id = nid;
            if (i + 64 < len) nid = (int)point_list[r0 + i + 64];
            if ((m >> lane) & 1ull) {
                const int at = pend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                s_ring[(head + at) & (RING - 1)] = make_int2(id, i + 1);
            }
            pend += __popcll(m);
            scan += 64;
        }
        if (pend == 0) break;
        const int nh = min(pend, NH);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nh) {
            const int2 hp = s_ring[(head + lane) & (RING - 1)];
            const int id = hp.x;
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
            float skip = __builtin_inff();          // opa * exp(-rho / 2) < 1/255 for every rho > skip (1 % + 0.05 margin)
            if (opa <= 1.0f) {
                const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
                skip = 2.0f * l * 1.01f + 0.05f;
            }
            float4* s4 = reinterpret_cast<float4*>(s_rec + lane * RS);
            s4[0] = a;                                      // Tu.xyz Tv.x
            s4[1] = b;                                      // Tv.yz Tw.xy
            s4[2] = c;                                      // Tw.z cx cy nx
            s4[3] = make_float4(d.x, d.y, opa, skip);       // ny nz opa skip
            s4[4] = make_float4(d.w, e.x, e.y, __int_as_float(hp.y));      // rgb, position in the tile's list (1-based)
        }
        if (FCH > 0) {
            if ((ED & 3) == 0 && nfeat == FCH) {
                for (int e = lane; e < nh * QF4; e += 64) {
                    const int inst = e / QF4, part = e - inst * QF4;
                    const int id = s_ring[(head + inst) & (RING - 1)].x;
                    reinterpret_cast<float4*>(s_feat)[e] = *reinterpret_cast<const float4*>(extras + (size_t)id * ED + ch_base + part * 4);
                }
            } else {
                for (int e = lane; e < nh * FCH; e += 64) {
                    const int inst = e / FCH, c = e - inst * FCH;
                    const int id = s_ring[(head + inst) & (RING - 1)].x;
                    s_feat[e] = c < nfeat ? extras[(size_t)id * ED + ch_base + c] : 0.0f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma clang loop unroll(disable)
        for (int j = 0; j < nh && m_done != ~0ull; j++) {
            // Flat, predicated evaluation: a lane for which a test fails computes garbage that is never used - identical results
            // to the reference's chain of `continue`s (forward.cu:356-393).
            const float4 a = reinterpret_cast<const float4*>(s_rec + j * RS)[0];
            const float4 b = reinterpret_cast<const float4*>(s_rec + j * RS)[1];
            const float4 c = reinterpret_cast<const float4*>(s_rec + j * RS)[2];
            const float4 d = reinterpret_cast<const float4*>(s_rec + j * RS)[3];
            const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y}, Tw = {b.z, b.w, c.x};
            const F3 kq = {Math::msub(pxf, Tw.x, Tu.x), Math::msub(pxf, Tw.y, Tu.y), Math::msub(pxf, Tw.z, Tu.z)};
            const F3 lq = {Math::msub(pyf, Tw.x, Tv.x), Math::msub(pyf, Tw.y, Tv.y), Math::msub(pyf, Tw.z, Tv.z)};
            const F3 p = {Math::msub(kq.y, lq.z, kq.z * lq.y), Math::msub(kq.z, lq.x, kq.x * lq.z), Math::msub(kq.x, lq.y, kq.y * lq.x)};
            const float dx = c.y - pxf, dy = c.z - pyf;
            const float rho2d = FILTER_INV_SQ * Math::mad(dy, dy, dx * dx);
            const float skip = d.w;
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
            const float4 col = reinterpret_cast<const float4*>(s_rec + j * RS)[4];
            float w_lane = 0.0f;
            if (__builtin_amdgcn_inverse_ballot_w64(m_ok)) {
                const float w = alpha * T;
                w_lane = w;
                const unsigned contributor = __float_as_uint(col.w);
                if (first_pass) {
                    const float A = 1 - T;
                    const float m_ = mscale * (1 - Math::div(NEAR_N, depth));
                    distortion += (Math::mad(m_ * m_, A, M2) - 2 * m_ * M1) * w;
                    D = Math::mad(depth, w, D);
                    M1 = Math::mad(m_, w, M1);
                    M2 = Math::mad(m_ * m_, w, M2);
                    if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                    N0 = Math::mad(c.w, w, N0); N1 = Math::mad(d.x, w, N1); N2 = Math::mad(d.y, w, N2);
                    C0 = Math::mad(col.x, w, C0); C1 = Math::mad(col.y, w, C1); C2 = Math::mad(col.z, w, C2);
                }
                if (FCH > 0) {
                    const float4* fj = reinterpret_cast<const float4*>(s_feat + j * FCH);
#pragma unroll
                    for (int q4 = 0; q4 < QF4; q4++) {
                        const float4 vv = fj[q4];            // reference forward.cu:415 order: (e * alpha) * T
                        E[4 * q4 + 0] += vv.x * alpha * T;
                        E[4 * q4 + 1] += vv.y * alpha * T;
                        E[4 * q4 + 2] += vv.z * alpha * T;
                        E[4 * q4 + 3] += vv.w * alpha * T;
                    }
                }
                T = test_T;
                last_contributor = contributor;
            }
            if (tracer != nullptr && first_pass) {
                const unsigned long long m_tr = __ballot(w_lane >= 0.1f);     // (double)w > 0.1  <=>  w >= 0.1f
                if (m_tr != 0ull) {
                    if (w_lane >= 0.1f) {
                        const int at = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_tr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_tr, 0u));
                        s_trace[2 * at] = s_ring[(head + j) & (RING - 1)].x;
                        s_trace[2 * at + 1] = (int)pix;
                    }
                    wcnt += __popcll(m_tr);
                    if (wcnt > WCAP - 64) flush_trace();
                }
            }
        }
        if (m_done == ~0ull) break;
        head = (head + nh) & (RING - 1);
        pend -= nh;
        __builtin_amdgcn_wave_barrier();
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
        if (FCH > 0) {
#pragma unroll
            for (int q = 0; q < FCH; q++)
                if (q < nfeat) out_extra[(size_t)(ch_base + q) * N + pix] = E[q];
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

// Backward path of the MI355X surfel rasterizer (K9 reference backward.cu:143-466,
// K10 backward.cu:469-656, K11 rasterizer_impl.cu:54-66).
//
// Design (MI355X-first; the reference issues ~(16+F) float atomics per
// (pixel, Gaussian) pair):
//   * No global atomics at all.  Every (tile, Gaussian) instance owns one row of a
//     scratch matrix, addressed by  slot = point_offsets[g] + rank(tile in g's rect),
//     so a Gaussian's rows are contiguous and a second, coalesced pass sums them.
//     Gradients are therefore bit-reproducible run to run.
//   * Inside a tile the sum over the 64 pixels of a wavefront of  w(pix,g) * dL/dout(pix,ch)
//     is a GEMM  [32 splats x 64 pixels] . [64 pixels x channels]  — it runs on the
//     matrix cores with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 for the feature
//     channels, v_mfma_f32_16x16x4_f32 for colour+normal), fed from a per-wave LDS
//     transpose of the blend weights.  MFMA is a separate pipe from the VALU that
//     recomputes the alphas, so the reduction is nearly free.
//   * The 12 gradient terms that are not weight-linear (dL/dT, dL/dcentre, dL/dopacity)
//     are reduced across the wavefront with butterflies, only for (wave, splat) pairs
//     where some pixel contributes.
//   * The per-channel feature recurrences of the reference (2F registers) collapse to
//     one scalar recurrence on q = <feature_g, dL/dfeature(pix)> (same algebra).
#include "isr_host.hpp"
#include "isr_fast_pair.hpp"

namespace isr {

constexpr int GEOM_ROW = 20;   // [0..8] dL_dT  [9,10] dL_dcentre  [11..13] dL_dnormal  [14] dL_dopacity  [15..17] dL_dcolor
constexpr int BB = 32;         // instances per backward batch
constexpr int WPAD = 65;       // row pitch (floats) of the per-wave weight transpose

__host__ __device__ inline int feat_row(int ED) { return ED > 0 ? ((ED + 31) / 32) * 32 : 0; }
__host__ __device__ inline int row_floats(int ED, unsigned mask) {
    return ((mask & 2u) ? GEOM_ROW : 0) + ((mask & 1u) ? feat_row(ED) : 0);
}
__host__ __device__ inline int n_passes(int ED, unsigned mask) { return ((mask & 1u) && ED > 32) ? (ED + 31) / 32 : 1; }
// geometry-only backward without a feature channel: room for the splat-major kernel's row per (tile, Gaussian, 8x4 half of an 8x8
// block) (isr_backward_geo.hip; eight slots per instance, of which ~2 are written)
__host__ __device__ inline int rows_per_instance(int ED, unsigned mask) { return ((mask & 3u) == 2u && ED == 0) ? 8 : 1; }
inline size_t rows_bytes(int64_t R, int ED, unsigned mask) {
    return align_up((size_t)(R > 0 ? R : 1) * rows_per_instance(ED, mask) * row_floats(ED, mask) * sizeof(float), 256);
}
// rows[R][stride] followed by one validity byte per (pass, row): a row is written (and flagged) only when some
// wave actually evaluated that (tile, splat) instance — everything else is skipped by the reductions.
size_t backward_scratch_bytes(int64_t R, int ED, unsigned mask) {
    if (ED <= 0) mask &= ~1u;        // no feature channel: the launch lays the scratch out without the EXTRA bit (launch_backward_t)
    return rows_bytes(R, ED, mask) + align_up((size_t)(R > 0 ? R : 1) * rows_per_instance(ED, mask) * n_passes(ED, mask), 256) + 256;
}

__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ view, uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
    const float vz = view[2] * x + view[6] * y + view[10] * z + view[14];
    present[i] = vz > 0.2f ? 1 : 0;
}

// Sum over the 64 lanes of a wavefront with DPP adds inside each row of 16 lanes (no LDS crossbar traffic: a
// __shfl_xor butterfly is six dependent ds_bpermute round trips per value) and four v_readlane's across the rows.
// The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
    return v + __int_as_float(x);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror: every lane now holds the sum of its row of 16
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Sum TWELVE per-lane values over the wavefront at once ("packed butterfly").  A plain butterfly spends 6 exchange
// levels on every value; here each level also halves the number of live registers, because a half-swap lets the two
// halves of the wave finish DIFFERENT values:
//   level 32: v_permlane32_swap(a, b) -> a' = [a.lo | b.lo], b' = [a.hi | b.hi]; a' + b' holds a's pair sums in lanes
//             0..31 and b's in lanes 32..63                                     (12 values -> 6 registers)
//   level 16: v_permlane16_swap on the odd/even rows of 16 lanes, same idea     (6 -> 3 registers)
//   levels 8..1: DPP adds inside each row of 16                                 (3 registers, 12 instructions)
// 30 instructions instead of 12 x 11.  On return every lane of row r (= lane >> 4) holds in out[i] the wave total of
// g[4 * i + QMAP[r]] with QMAP = {0, 2, 1, 3}.
__device__ __forceinline__ float swap_add32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    return dpp_add<0x140>(v);  // row_mirror
}
__device__ __forceinline__ void wave_sum12(const float (&g)[12], float (&out)[3]) {
    float h[6];
#pragma unroll
    for (int i = 0; i < 6; i++) h[i] = swap_add32(g[2 * i], g[2 * i + 1]);   // lanes <32: g[2i], lanes >=32: g[2i+1]
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = row_sum16(swap_add16(h[2 * i], h[2 * i + 1]));
    // rows of swap_add16(h[2i], h[2i+1]): row0 g[4i], row1 g[4i+2], row2 g[4i+1], row3 g[4i+3]
}

// ----------------------------------------------------------------------------
// K9.  One workgroup (4 waves) per tile, wave w owns the 8x8 pixel block (w&1, w>>1).
// GEOM: produce the 18 geometry/appearance terms.  FEAT: produce dL/dextra for the
// 32-channel chunk starting at ch_base.  Rows are written for EVERY instance.
// QF: feature channels whose dL/dfeature(pixel) is kept in registers for the q-dot of the geometry pass
//     (channels beyond QF are read from global memory; QF = 0 when there is no geometry pass or no feature).
template <class Math, bool GEOM, bool FEAT, int QF>
__global__ __launch_bounds__(256, (GEOM && !FEAT && QF == 0) ? 3 : 2) void k_render_bwd(
    int W, int H, int ED, int ch_base, int gx, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ box4, const float* __restrict__ rec,
    const float* __restrict__ col_pre, const float* __restrict__ tm_pre, const float* __restrict__ extras,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dC, const float* __restrict__ dO, const float* __restrict__ dE,
    const uint32_t* __restrict__ point_offsets,
    const Rect16* __restrict__ rects, float* __restrict__ partial, uint8_t* __restrict__ row_flags,
    const uint8_t* __restrict__ tile_mode, int row_stride, int geom_off, int feat_off, int64_t capacity) {
    // staged record, EXACT: Tu Tv Tw | centre normal | opacity skip.  FAST: the operand pairs of fast_ray (Tu.xy Tv.xy | Tw.xy Tu.z Tv.z |
    // Tw.z det cx cy | opacity band.hi band.lo: isr_fast_pair.hpp - the pair is evaluated exactly as k_render_fwd_fast did), and
    // only for the geometry gradient also the normal
    constexpr int RS = (Math::fast && GEOM) ? 20 : 16;
    constexpr int SB = 128;                 // (id, cull box) pairs staged per barrier round: two 64-bit hit masks per wave
    constexpr int PART = (GEOM ? GEOM_ROW : 0) + (FEAT ? 32 : 0);   // floats per instance per wave in LDS
    __shared__ __attribute__((aligned(16))) float s_rec[BB * RS];      // records of the current sub-batch (hit instances only)
    __shared__ __attribute__((aligned(16))) float s_rgb[BB * 4];
    __shared__ __attribute__((aligned(16))) float s_feat[QF > 0 ? BB * QF : 4];
    __shared__ unsigned s_slot[BB];
    __shared__ int s_id[SB];
    __shared__ unsigned s_box4[SB];
    __shared__ float s_W[4 * BB * WPAD];
    __shared__ __attribute__((aligned(16))) float s_part[4 * BB * PART];
    __shared__ unsigned s_hit[4 * (SB / 32)];      // [wave][sub-batch] 32-bit hit masks of the round
    __shared__ unsigned s_last[4];

    const int tile = blockIdx.x;       // (an XCD-contiguous tile map was measured 5 % slower: it unbalances the XCDs)
    if (tile_mode != nullptr && tile_mode[tile] != 255) return;   // nothing to do, or done by k_render_bwd_sparse
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
    const int len = (int)(r1 - r0);
    if (len <= 0) return;

    float* Ww = s_W + wv * BB * WPAD;

    const unsigned last_contributor = inside ? n_contrib[pix] : 0u;
    // ---- MFMA B operands (constant for the whole tile) -------------------------
    // feature tile: B[k][j] with k = lane>>5 (pixel 2s+k of this wave), j = lane&31 (channel)
    float Bf[FEAT ? 32 : 1];
    if constexpr (FEAT) {
        const int ch = ch_base + (lane & 31);
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const int q = 2 * s + (lane >> 5);
            const unsigned qx = tx * TILE + (wv & 1) * 8 + (q & 7), qy = ty * TILE + (wv >> 1) * 8 + (q >> 3);
            Bf[s] = (dE != nullptr && ch < ED && qx < (unsigned)W && qy < (unsigned)H)
                        ? dE[(size_t)ch * N + (size_t)W * qy + qx] : 0.0f;
        }
    }
    // colour+normal tile (16x16x4): B[k][j], k = lane>>4 (pixel 4s+k), j = lane&15 (0..2 dC, 3..5 dN)
    float Bl[GEOM ? 16 : 1];
    if constexpr (GEOM) {
        const int j = lane & 15;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int q = 4 * s + (lane >> 4);
            const unsigned qx = tx * TILE + (wv & 1) * 8 + (q & 7), qy = ty * TILE + (wv >> 1) * 8 + (q >> 3);
            float v = 0.0f;
            if (qx < (unsigned)W && qy < (unsigned)H) {
                const size_t qp = (size_t)W * qy + qx;
                if (j < 3) v = dC ? dC[(size_t)j * N + qp] : 0.0f;
                else if (j < 6) v = dO ? dO[(size_t)(2 + j - 3) * N + qp] : 0.0f;
            }
            Bl[s] = v;
        }
    }

    // ---- per-pixel state (GEOM) -------------------------------------------------
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dL_ddepth = 0, dL_daccum = 0, dL_dreg = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmedian = 0;
    float T_final = 0, final_D = 0, final_D2 = 0;
    unsigned median_contributor = 0;
    float dEp[QF > 0 ? QF : 1];
    if (inside) {
        T_final = final_T[pix];
        if (GEOM) {
            final_D = final_T[pix + N];
            final_D2 = final_T[pix + 2 * N];
            median_contributor = n_contrib[pix + N];
            if (dC) { dpx0 = dC[pix]; dpx1 = dC[N + pix]; dpx2 = dC[2 * N + pix]; }
            if (dO) {
                dL_ddepth = dO[pix]; dL_daccum = dO[N + pix]; dn0 = dO[2 * N + pix]; dn1 = dO[3 * N + pix];
                dn2 = dO[4 * N + pix]; dL_dmedian = dO[5 * N + pix]; dL_dreg = dO[6 * N + pix];
            }
        }
    }
    if constexpr (QF > 0) {
#pragma unroll
        for (int c = 0; c < QF; c++) dEp[c] = (inside && dE != nullptr && c < ED) ? dE[(size_t)c * N + pix] : 0.0f;
    }
    // ---- which pixels of this wave carry any upstream gradient?  A pixel whose dL/dout is exactly zero
    // contributes exactly zero to every sum, so it is skipped, and splats are culled against the bounding
    // rectangle of the wave's live pixels (train_semantic samples ~16k of 2M pixels: most waves go idle).
    bool has_grad;
    if constexpr (GEOM) {
        has_grad = (dpx0 != 0.0f) || (dpx1 != 0.0f) || (dpx2 != 0.0f) || (dL_ddepth != 0.0f) || (dL_daccum != 0.0f) ||
                   (dL_dreg != 0.0f) || (dn0 != 0.0f) || (dn1 != 0.0f) || (dn2 != 0.0f) || (dL_dmedian != 0.0f);
        if constexpr (QF > 0) {
#pragma unroll
            for (int c = 0; c < QF; c++) has_grad = has_grad || (dEp[c] != 0.0f);
            if (ED > QF) has_grad = true;
        }
    } else {
        unsigned long long pm = 0ull;       // bit q: row q of the wave's dL/dfeature block is non-zero
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const unsigned long long bal = __ballot(Bf[s] != 0.0f);
            if (bal & 0xffffffffull) pm |= 1ull << (2 * s);
            if (bal >> 32) pm |= 1ull << (2 * s + 1);
        }
        has_grad = (pm >> lane) & 1ull;
    }
    const bool wave_reg = GEOM && __ballot(dL_dreg != 0.0f) != 0ull;
    const bool lane_live = inside && has_grad && last_contributor > 0u;
    const bool wave_live = __ballot(lane_live) != 0ull;
    float ax0 = lane_live ? pxf : 3.0e38f, ax1 = lane_live ? pxf : -3.0e38f;
    float ay0 = lane_live ? pyf : 3.0e38f, ay1 = lane_live ? pyf : -3.0e38f;
    unsigned wave_last = lane_live ? last_contributor : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        ax0 = fminf(ax0, __shfl_xor(ax0, o)); ax1 = fmaxf(ax1, __shfl_xor(ax1, o));
        ay0 = fminf(ay0, __shfl_xor(ay0, o)); ay1 = fmaxf(ay1, __shfl_xor(ay1, o));
        wave_last = max(wave_last, (unsigned)__shfl_xor((int)wave_last, o));
    }
    // Nothing behind the deepest last contributor of the tile's live pixels can matter: walk only that prefix of
    // the list (a saturated pixel stops after a few dozen of the several hundred splats of its tile), and leave
    // immediately when the tile has no live pixel at all.
    if (lane == 0) s_last[wv] = wave_last;
    __syncthreads();
    const unsigned tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    if (tile_last == 0u) return;
    const int len_eff = min(len, (int)tile_last);
    // Sparse-wave mode (feature-only kernels): with at most 4 live pixels the splats are culled against the
    // individual pixels and dL/dfeat[g][ch] = sum_k w(pix_k, g) * dL/dE(pix_k, ch) is four FMAs per channel
    // lane — no weight transpose, no MFMA.
    const unsigned long long live_mask = __ballot(lane_live);
    const bool sparse = !GEOM && FEAT && __popcll(live_mask) <= 4;
    int lk[4] = {-1, -1, -1, -1}, lrx[4] = {0, 0, 0, 0}, lry[4] = {0, 0, 0, 0};
    float dEk[4] = {0.f, 0.f, 0.f, 0.f};
    if (sparse) {
        unsigned long long mm = live_mask;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (mm != 0ull) {
                lk[k] = __builtin_ctzll(mm);
                mm &= mm - 1ull;
                const int qx = tx * TILE + (wv & 1) * 8 + (lk[k] & 7), qy = ty * TILE + (wv >> 1) * 8 + (lk[k] >> 3);
                lrx[k] = qx - tx * TILE;
                lry[k] = qy - ty * TILE;
                const int ch = ch_base + (lane & 31);
                dEk[k] = (dE != nullptr && ch < ED) ? dE[(size_t)ch * N + (size_t)W * qy + qx] : 0.0f;
            }
        }
    }
    const float final_A = 1 - T_final;
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    float T = GEOM ? T_final : 1.0f;
    float acc_r0 = 0, acc_r1 = 0, acc_r2 = 0, lc0 = 0, lc1 = 0, lc2 = 0;
    float accum_depth_rec = 0, last_depth = 0, accum_alpha_rec = 0, an0 = 0, an1 = 0, an2 = 0, ln0 = 0, ln1 = 0, ln2 = 0;
    float last_dL_dT = 0, last_alpha = 0, accum_q = 0, last_q = 0, accum_S = 0;
    const float bg_dot = GEOM ? (bg[0] * dpx0 + bg[1] * dpx1) + bg[2] * dpx2 : 0.0f;

    // GEOM walks back to front (reference order); features-only walks front to back.
    // Per round, SB (id, packed cull box) pairs are read coalesced; every wave tests them against the rectangle of
    // its live pixels; only instances hit by some wave are gathered (record, features, row slot) and evaluated, in
    // sub-batches of BB = 32 (the MFMA M dimension).
    const int tile_x0 = tx * TILE, tile_y0 = ty * TILE;
    const int rx0 = (int)ax0 - tile_x0, rx1 = (int)ax1 - tile_x0, ry0 = (int)ay0 - tile_y0, ry1 = (int)ay1 - tile_y0;
    const int nround = (len_eff + SB - 1) / SB;
    float* Pw = s_part + wv * BB * PART;
    for (int ri = 0; ri < nround; ri++) {
        int round_lo, nsb;
        if (GEOM) { const int hi = len_eff - ri * SB; round_lo = max(0, hi - SB); nsb = hi - round_lo; }
        else { round_lo = ri * SB; nsb = min(SB, len_eff - round_lo); }
        __syncthreads();   // previous round fully consumed
        if (threadIdx.x < nsb) {
            s_id[threadIdx.x] = (int)point_list[r0 + round_lo + threadIdx.x];
            s_box4[threadIdx.x] = box4[r0 + round_lo + threadIdx.x];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < SB / 64; h++) {
            const int t = h * 64 + lane;
            bool hit = false;
            if (wave_live && t < nsb && (unsigned)(round_lo + t) < wave_last) {
                const unsigned bx = s_box4[t];
                const int xl = (int)(signed char)(bx & 255u), xh = (int)(signed char)((bx >> 8) & 255u);
                const int yl = (int)(signed char)((bx >> 16) & 255u), yh = (int)(signed char)(bx >> 24);
                // saturated coordinates (+-127/128) mean "beyond": treat as unbounded
                if (sparse) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        hit = hit || (lk[k] >= 0 && xl <= lrx[k] && xh >= lrx[k] && yl <= lry[k] && yh >= lry[k]);
                } else {
                    hit = xl <= rx1 && xh >= rx0 && yl <= ry1 && yh >= ry0;
                }
            }
            const unsigned long long mm = __ballot(hit);
            if (lane == 0) { s_hit[wv * (SB / 32) + 2 * h] = (unsigned)mm; s_hit[wv * (SB / 32) + 2 * h + 1] = (unsigned)(mm >> 32); }
        }
        __syncthreads();
        const int nsub = (nsb + BB - 1) / BB;
        for (int si = 0; si < nsub; si++) {
            // sub-batch = staged indices [sub_lo, sub_lo + nb); aligned to 32 so that it is one s_hit word
            int word;
            if (GEOM) word = (nsb - 1) / BB - si; else word = si;
            const int sub_lo = word * BB, nb = min(BB, nsb - sub_lo);
            const int lo = round_lo + sub_lo;
            const unsigned h0 = s_hit[0 * (SB / 32) + word], h1 = s_hit[1 * (SB / 32) + word];
            const unsigned h2 = s_hit[2 * (SB / 32) + word], h3 = s_hit[3 * (SB / 32) + word];
            const unsigned hany = h0 | h1 | h2 | h3;
            if (hany == 0u) continue;                       // uniform over the workgroup
            if (threadIdx.x < nb && ((hany >> threadIdx.x) & 1u)) {
                const int t = threadIdx.x;
                const int id = s_id[sub_lo + t];
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
                float skip = __builtin_inff();          // EXACT's cull: opa * exp(-rho / 2) < 1/255 for every rho > skip (1 % + 0.05 margin)
                if (opa <= 1.0f) {
                    const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
                    skip = 2.0f * l * 1.01f + 0.05f;
                }
                float4* s4 = reinterpret_cast<float4*>(s_rec + t * RS);
                if constexpr (Math::fast) {
                    const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y}, Tw = {b.z, b.w, c.x};
                    const FastBand fb = fast_band(opa, e.w);          // rec[19]: K1's guard band (isr_fast_pair.hpp)
                    s4[0] = make_float4(Tu.x, Tu.y, Tv.x, Tv.y);
                    s4[1] = make_float4(Tw.x, Tw.y, Tu.z, Tv.z);
                    s4[2] = make_float4(Tw.z, fast_det(Tu, Tv, Tw, c.y, c.z), c.y, c.z);
                    s4[3] = make_float4(opa, fb.hi, fb.lo, 0.0f);
                    if constexpr (GEOM) s4[4] = make_float4(c.w, d.x, d.y, 0.0f);
                } else {
                    s4[0] = a; s4[1] = b; s4[2] = c; s4[3] = make_float4(d.x, d.y, opa, skip);
                }
                reinterpret_cast<float4*>(s_rgb)[t] = make_float4(d.w, e.x, e.y, 0.0f);
                const Rect16 rc = rects[id];
                s_slot[t] = point_offsets[id] + (unsigned)(ty - rc.y0) * (unsigned)(rc.x1 - rc.x0) + (unsigned)(tx - rc.x0);
            }
            if constexpr (QF > 0) {
                for (int e = threadIdx.x; e < nb * QF; e += 256) {
                    const int inst = e / QF, c = e - inst * QF;
                    if ((hany >> inst) & 1u) s_feat[e] = (c < ED) ? extras[(size_t)s_id[sub_lo + inst] * ED + c] : 0.0f;
                }
            }
            __syncthreads();
            unsigned long long m = (wv == 0 ? h0 : (wv == 1 ? h1 : (wv == 2 ? h2 : h3)));
            if (m != 0ull) {
                for (int e = lane; e < BB * PART; e += 64) Pw[e] = 0.0f;
                if (!sparse)
                    for (int e = lane; e < BB * WPAD; e += 64) Ww[e] = 0.0f;     // culled splats have zero weight
                // ---- phase A: every live lane evaluates its pixel against the surviving splats ----------
                while (m != 0ull) {
                    int j;
                    if (GEOM) { j = 63 - __builtin_clzll(m); m &= ~(1ull << j); }     // back to front
                    else { j = __builtin_ctzll(m); m &= m - 1ull; }                     // front to back
                    const unsigned contributor = (unsigned)(lo + j);     // 0-based index == reference's decremented counter
                    float w = 0.0f;
                    bool act = lane_live && contributor < last_contributor;
                    float G = 0, alpha = 0, sx = 0, sy = 0, c_d = 0, rho3d = 0, rho2d = 0, dx = 0, dy = 0;
                    F3 kk = {0, 0, 0}, ll = {0, 0, 0}, p = {0, 0, 1};
                    const float4* sj = reinterpret_cast<const float4*>(s_rec + j * RS);
                    float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;        // the EXACT record's four quads (d.z = opacity)
                    F3 Tw = {0, 0, 1};
                    float nx = 0, ny = 0, nz = 0;
                    if constexpr (Math::fast) {
                        // the forward's own evaluation of the pair (isr_fast_pair.hpp): same decisions, bit for bit
                        const float4 qa = sj[0], qb = sj[1], qc = sj[2], qd = sj[3];
                        d.z = qd.x;
                        Tw = {qb.x, qb.y, qc.x};
                        a = make_float4(qa.x, qa.y, qb.z, qa.z);          // Tu.xyz, Tv.x
                        b = make_float4(qa.w, qb.w, qb.x, qb.y);          // Tv.yz, Tw.xy
                        if constexpr (GEOM) {
                            const float4 qn = sj[4];
                            nx = qn.x; ny = qn.y; nz = qn.z;
                        }
                        FastRay fr = fast_ray((v2f){pxf, pyf}, (v2f){qa.x, qa.y}, (v2f){qa.z, qa.w}, (v2f){qb.x, qb.y}, (v2f){qb.z, qb.w}, qc.x,
                                              (v2f){qc.z, qc.w});
                        FastHit fh = fast_hit(fr, qc.y, qc.x, qd.x);
                        const bool near = act && fast_near(fr, qd.y);
                        const bool inb = near && fast_in_band(fr, fast_band_of(qd.y, qd.z));
                        act = near && fast_pass(fh);
                        if (__ballot(inb) != 0ull) {          // rare: the pair is re-evaluated as EXACT does (the forward did the same)
                            FastRay er; FastHit eh;
                            const bool ep = exact_pair_rec(pxf, pyf, rec, __builtin_amdgcn_readfirstlane(s_id[sub_lo + j]), er, eh);
                            fast_take(inb, er, eh, fr, fh);
                            act = inb ? ep : act;
                        }
                        p = {fr.p_x, fr.p_y, fr.p_z};
                        dx = fr.dx; dy = fr.dy; rho2d = fr.rho2d; rho3d = fr.rho3d; sx = fr.sx; sy = fr.sy;
                        c_d = fh.depth; G = fh.G; alpha = fh.alpha;
                        if constexpr (GEOM) {       // the adjoint of the intersection is written on k, l (values only)
                            const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y};
                            kk = {Math::msub(pxf, Tw.x, Tu.x), Math::msub(pxf, Tw.y, Tu.y), Math::msub(pxf, Tw.z, Tu.z)};
                            ll = {Math::msub(pyf, Tw.x, Tv.x), Math::msub(pyf, Tw.y, Tv.y), Math::msub(pyf, Tw.z, Tv.z)};
                        }
                    } else {
                    a = sj[0]; b = sj[1]; c = sj[2]; d = sj[3];
                    Tw = {b.z, b.w, c.x};
                    nx = c.w; ny = d.x; nz = d.y;
                    if (act) {
                        const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y};
                        kk = {Math::msub(pxf, Tw.x, Tu.x), Math::msub(pxf, Tw.y, Tu.y), Math::msub(pxf, Tw.z, Tu.z)};
                        ll = {Math::msub(pyf, Tw.x, Tv.x), Math::msub(pyf, Tw.y, Tv.y), Math::msub(pyf, Tw.z, Tv.z)};
                        p = {Math::msub(kk.y, ll.z, kk.z * ll.y), Math::msub(kk.z, ll.x, kk.x * ll.z),
                             Math::msub(kk.x, ll.y, kk.y * ll.x)};
                        dx = c.y - pxf; dy = c.z - pyf;
                        rho2d = FILTER_INV_SQ * Math::mad(dy, dy, dx * dx);
                        const float skip = d.w;
                        if (rho2d > skip && Math::mad(p.y, p.y, p.x * p.x) > skip * (p.z * p.z) * 1.01f) act = false;
                        else if (p.z == 0.0f) act = false;
                    }
                    if (act) {
                        sx = Math::div(p.x, p.z); sy = Math::div(p.y, p.z);
                        rho3d = Math::mad(sy, sy, sx * sx);
                        const float rho = fminf(rho3d, rho2d);
                        c_d = (rho3d <= rho2d) ? Math::mad(sy, Tw.y, sx * Tw.x) + Tw.z : Tw.z;
                        const float power = -0.5f * rho;
                        if (c_d < NEAR_N || power > 0.0f) act = false;
                        else {
                            G = Math::ex(power);
                            alpha = fminf(0.99f, d.z * G);
                            if (alpha < 1.0f / 255.0f) act = false;
                        }
                    }
                    }
                    if (act) {
                        if (GEOM) { T = Math::div(T, 1.f - alpha); w = alpha * T; }
                        else { w = alpha * T; T = T * (1 - alpha); }
                    }
                    if (sparse) {
                        if constexpr (FEAT && !GEOM) {
                            float v = 0.0f;
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (lk[k] >= 0) v = __builtin_fmaf(__shfl(w, lk[k]), dEk[k], v);
                            if (lane < 32) Pw[j * PART + lane] = v;
                        }
                    } else {
                        Ww[j * WPAD + lane] = w;
                    }
                    if constexpr (GEOM) {
                        float g[12];
    #pragma unroll
                        for (int q = 0; q < 12; q++) g[q] = 0.0f;
                        if (act) {
                            const float4 col = reinterpret_cast<const float4*>(s_rgb)[j];
                            float dL_dalpha = 0.0f;
                            float dL_dz = 0.0f;
                            if (contributor == median_contributor - 1u) dL_dz += dL_dmedian;
                            float dL_dweight = 0.0f;
                            if (wave_reg) {     // wave-uniform: no pixel of this wave has a distortion gradient (lambda_dist = 0)
                                                // -> every term below is an exact zero
                                const float m_d = mscale * (1 - Math::div(NEAR_N, c_d));
                                const float dmd_dd = Math::div(FAR_N * NEAR_N, (FAR_N - NEAR_N) * c_d * c_d);
                                dL_dweight = (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                                const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                                dL_dz += dL_dmd * dmd_dd;
                            }
                            float q = 0.0f;
                            if constexpr (QF > 0) {
                                const float* fj = s_feat + j * QF;
    #pragma unroll
                                for (int ch = 0; ch < QF; ch++) q = __builtin_fmaf(fj[ch], dEp[ch], q);
                                if (ED > QF && dE != nullptr) {     // rare: more feature channels than the register budget
                                    const float* fg = extras + (size_t)s_id[sub_lo + j] * ED;
                                    for (int ch = QF; ch < ED; ch++) q = __builtin_fmaf(fg[ch], dE[(size_t)ch * N + pix], q);
                                }
                            }
                            if constexpr (Math::fast) {
                                // Every "what lies behind this splat" term of the reference - colour, normal, depth, alpha,
                                // distortion weight, feature - follows the SAME recurrence
                                //     acc <- alpha_prev * x_prev + (1 - alpha_prev) * acc,    dL/dalpha += (x - acc) * g
                                // and is linear in x, so they collapse into ONE recurrence on the pixel's scalar
                                // S = sum_x x * g (nine registers of state and ~30 instructions per pair fewer).
                                float S = col.x * dpx0;
                                S = __builtin_fmaf(col.y, dpx1, S); S = __builtin_fmaf(col.z, dpx2, S);
                                S = __builtin_fmaf(c_d, dL_ddepth, S); S += dL_daccum;
                                S = __builtin_fmaf(nx, dn0, S); S = __builtin_fmaf(ny, dn1, S); S = __builtin_fmaf(nz, dn2, S);
                                S += dL_dweight;
                                if constexpr (QF > 0) S += q;
                                dL_dalpha = S - accum_S;
                                accum_S = __builtin_fmaf(alpha, dL_dalpha, accum_S);      // alpha * S + (1 - alpha) * accum_S
                            } else {
                                acc_r0 = last_alpha * lc0 + (1.f - last_alpha) * acc_r0; lc0 = col.x; dL_dalpha += (col.x - acc_r0) * dpx0;
                                acc_r1 = last_alpha * lc1 + (1.f - last_alpha) * acc_r1; lc1 = col.y; dL_dalpha += (col.y - acc_r1) * dpx1;
                                acc_r2 = last_alpha * lc2 + (1.f - last_alpha) * acc_r2; lc2 = col.z; dL_dalpha += (col.z - acc_r2) * dpx2;
                                dL_dalpha += dL_dweight - last_dL_dT;
                                last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                                last_depth = c_d;
                                dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                                accum_alpha_rec = last_alpha * 1.0f + (1.f - last_alpha) * accum_alpha_rec;
                                dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                                an0 = last_alpha * ln0 + (1.f - last_alpha) * an0; ln0 = nx; dL_dalpha += (nx - an0) * dn0;
                                an1 = last_alpha * ln1 + (1.f - last_alpha) * an1; ln1 = ny; dL_dalpha += (ny - an1) * dn1;
                                an2 = last_alpha * ln2 + (1.f - last_alpha) * an2; ln2 = nz; dL_dalpha += (nz - an2) * dn2;
                                if constexpr (QF > 0) {
                                    accum_q = last_alpha * last_q + (1.f - last_alpha) * accum_q;
                                    last_q = q;
                                    dL_dalpha += q - accum_q;
                                }
                            }
                            dL_dalpha *= T;
                            last_alpha = alpha;
                            dL_dalpha += Math::div(-T_final, 1.f - alpha) * bg_dot;
                            const float dL_dG = d.z * dL_dalpha;
                            dL_dz += alpha * T * dL_ddepth;
                            if (rho3d <= rho2d) {
                                const float dsx = dL_dG * -G * sx + dL_dz * Tw.x;
                                const float dsy = dL_dG * -G * sy + dL_dz * Tw.y;
                                const float dsx_pz = Math::div(dsx, p.z), dsy_pz = Math::div(dsy, p.z);
                                const F3 dL_dp = {dsx_pz, dsy_pz, -(dsx_pz * sx + dsy_pz * sy)};
                                const F3 dL_dk = cross3(ll, dL_dp);
                                const F3 dL_dl = cross3(dL_dp, kk);
                                g[0] = -dL_dk.x; g[1] = -dL_dk.y; g[2] = -dL_dk.z;
                                g[3] = -dL_dl.x; g[4] = -dL_dl.y; g[5] = -dL_dl.z;
                                g[6] = pxf * dL_dk.x + pyf * dL_dl.x + dL_dz * sx;
                                g[7] = pxf * dL_dk.y + pyf * dL_dl.y + dL_dz * sy;
                                g[8] = pxf * dL_dk.z + pyf * dL_dl.z + dL_dz * 1.0f;
                            } else {
                                g[9] = dL_dG * (-G * FILTER_INV_SQ * dx);
                                g[10] = dL_dG * (-G * FILTER_INV_SQ * dy);
                                g[8] = dL_dz;
                            }
                            g[11] = G * dL_dalpha;
                        }
                        if (__ballot(act) != 0ull) {
                            float tot[3];
                            wave_sum12(g, tot);
                            if ((lane & 15) == 0) {     // first lane of each row of 16 stores its three totals
                                const int r = lane >> 4;
                                const int q0 = (r == 0) ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
                                float* o = Pw + j * PART;
                                o[q0] = tot[0];
                                o[4 + q0] = tot[1];
                                o[q0 == 3 ? 14 : 8 + q0] = tot[2];      // g[11] (dL/dopacity) lives in column 14
                            }
                        }
                    }
                }
                // ---- phase M: matrix-core reduction over the wave's 64 pixels -----------
                if constexpr (FEAT) if (!sparse) {
                    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    #pragma unroll
                    for (int s = 0; s < 32; s++) {
                        const float av = Ww[(lane & 31) * WPAD + 2 * s + (lane >> 5)];   // A[i = splat][k = pixel]
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bf[s], acc, 0, 0, 0);
                    }
                    // D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
                    constexpr int FO = GEOM ? GEOM_ROW : 0;
    #pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Pw[row * PART + FO + (lane & 31)] = acc[r];
                    }
                }
                if constexpr (GEOM) {
    #pragma unroll
                    for (int mt = 0; mt < 2; mt++) {
                        f32x4 acc = {0, 0, 0, 0};
    #pragma unroll
                        for (int s = 0; s < 16; s++) {
                            const float av = Ww[(mt * 16 + (lane & 15)) * WPAD + 4 * s + (lane >> 4)];
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bl[s], acc, 0, 0, 0);
                        }
                        // D[row = 4*(lane>>4) + r][col = lane&15]; cols 0..2 -> dL_dcolor, 3..5 -> dL_dnormal
                        const int col = lane & 15;
                        if (col < 6) {
                            const int dst = col < 3 ? 15 + col : 11 + (col - 3);
    #pragma unroll
                            for (int r = 0; r < 4; r++) Pw[(mt * 16 + 4 * (lane >> 4) + r) * PART + dst] = acc[r];
                        }
                    }
                }
            }
            __syncthreads();
            // ---- combine the waves that touched an instance (fixed order) and emit its row + flag -------
            {
                constexpr int Q4 = PART / 4;
                const float* P0 = s_part + 0 * BB * PART;
                const float* P1 = s_part + 1 * BB * PART;
                const float* P2 = s_part + 2 * BB * PART;
                const float* P3 = s_part + 3 * BB * PART;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = threadIdx.x; e < nb * Q4; e += 256) {
                    const int inst = e / Q4, q = e - inst * Q4;
                    if (!((hany >> inst) & 1u)) continue;
                    // a wave that culled the whole sub-batch never zeroed its block: treat it as zero
                    // (`if`, not `h ? *p : z4`: on float4 the conditional operator picks an ADDRESS - LDS or a scratch copy of
                    // the zeros - and the read becomes a flat load; see k_feature_rows_step)
                    float4 v0 = z4, v1 = z4, v2 = z4, v3 = z4;
                    if (h0 != 0u) v0 = reinterpret_cast<const float4*>(P0 + inst * PART)[q];
                    if (h1 != 0u) v1 = reinterpret_cast<const float4*>(P1 + inst * PART)[q];
                    if (h2 != 0u) v2 = reinterpret_cast<const float4*>(P2 + inst * PART)[q];
                    if (h3 != 0u) v3 = reinterpret_cast<const float4*>(P3 + inst * PART)[q];
                    const float4 v = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y),
                                                 (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w));
                    int dst;
                    if (GEOM && q < GEOM_ROW / 4) dst = geom_off + 4 * q;
                    else dst = feat_off + 4 * (q - (GEOM ? GEOM_ROW / 4 : 0));
                    const unsigned slot = s_slot[inst];
                    *reinterpret_cast<float4*>(partial + (size_t)slot * row_stride + dst) = v;
                    if (q == 0) row_flags[slot] = 1;
                }
            }
            __syncthreads();     // s_rec / s_slot / s_part are reused by the next hit sub-batch
        }
    }
}

// ----------------------------------------------------------------------------
// K9, sparse upstream gradient (feature-only): "pixel-major" walk.
//
// train_semantic samples ~16k of 2M pixels per step, i.e. ~2 live pixels per tile.  For such tiles the
// pixel-per-lane mapping of k_render_bwd leaves 62 of 64 lanes idle and pays three workgroup barriers per
// 32 instances.  Here ONE wave walks the tile's depth-sorted list with a lane per SPLAT: 64 (id, cull box)
// pairs are read coalesced, lanes whose box holds a live pixel gather their record, and for every live pixel
// the transmittance in front of each lane's splat is a wave-wide multiplicative prefix scan of (1 - alpha)
// (DPP row shifts + row broadcasts; no LDS).  Each (tile, Gaussian) instance is owned by exactly one lane, which
// sums  w(pix_k) * dL/dE(pix_k, :)  over the live pixels in fixed order and stores the finished 32-channel row:
// no cross-lane reduction, no barrier in the walk, deterministic.
// Tiles with more than SPARSE_LMAX live pixels are flagged in tile_mode[] and left to k_render_bwd.
constexpr int SPARSE_LMAX = 32;
constexpr int SPARSE_ROW_PITCH = 36;     // floats between staged rows (32 + 4: keeps float4 alignment, spreads the banks)
constexpr int SPARSE_REC_PITCH4 = 5;     // float4s between staged records (4 + 1)
constexpr int SPARSE_ROUND = 256;       // k_render_bwd_sparse, pass 1: splats culled per memory round trip
constexpr int SPARSE_QUEUE = SPARSE_ROUND + 64;   // culled splats waiting for pass 2 (one round + up to 63 left over)

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_fetch(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
// inclusive product scan over the 64 lanes of a wavefront
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= dpp_fetch<0x111, 0xF>(v, 1.0f);     // row_shr:1
    v *= dpp_fetch<0x112, 0xF>(v, 1.0f);     // row_shr:2
    v *= dpp_fetch<0x114, 0xF>(v, 1.0f);     // row_shr:4
    v *= dpp_fetch<0x118, 0xF>(v, 1.0f);     // row_shr:8   -> inclusive inside each row of 16
    v *= dpp_fetch<0x142, 0xA>(v, 1.0f);     // row_bcast:15 into rows 1 and 3
    v *= dpp_fetch<0x143, 0xC>(v, 1.0f);     // row_bcast:31 into rows 2 and 3
    return v;
}

// alpha of one splat at one pixel, 0 when the forward `continue`d on the pair (forward.cu:352-386)
template <class Math>
__device__ __forceinline__ float splat_alpha(const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy, float opa,
                                             float skip, float pxf, float pyf) {
    const F3 kk = {Math::msub(pxf, Tw.x, Tu.x), Math::msub(pxf, Tw.y, Tu.y), Math::msub(pxf, Tw.z, Tu.z)};
    const F3 ll = {Math::msub(pyf, Tw.x, Tv.x), Math::msub(pyf, Tw.y, Tv.y), Math::msub(pyf, Tw.z, Tv.z)};
    const F3 p = {Math::msub(kk.y, ll.z, kk.z * ll.y), Math::msub(kk.z, ll.x, kk.x * ll.z),
                  Math::msub(kk.x, ll.y, kk.y * ll.x)};
    const float dx = cx - pxf, dy = cy - pyf;
    const float rho2d = FILTER_INV_SQ * Math::mad(dy, dy, dx * dx);
    if (rho2d > skip && Math::mad(p.y, p.y, p.x * p.x) > skip * (p.z * p.z) * 1.01f) return 0.0f;
    if (p.z == 0.0f) return 0.0f;
    const float sx = Math::div(p.x, p.z), sy = Math::div(p.y, p.z);
    const float rho3d = Math::mad(sy, sy, sx * sx);
    const float rho = fminf(rho3d, rho2d);
    const float c_d = (rho3d <= rho2d) ? Math::mad(sy, Tw.y, sx * Tw.x) + Tw.z : Tw.z;
    const float power = -0.5f * rho;
    if (c_d < NEAR_N || power > 0.0f) return 0.0f;
    const float alpha = fminf(0.99f, opa * Math::ex(power));
    return alpha < 1.0f / 255.0f ? 0.0f : alpha;
}

// the same in FAST arithmetic: the forward's own evaluation of the pair (isr_fast_pair.hpp: EXACT inside the guard bands),
// tile-relative pixel (lx, ly) = absolute pixel (pxf, pyf)
__device__ __forceinline__ float splat_alpha_fast(const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy, float opa, float det,
                                                  const FastBand& fb, float pxf, float pyf) {
    FastRay fr; FastHit fh;
    return fast_pair_lane(Tu, Tv, Tw, cx, cy, opa, det, fb, pxf, pyf, fr, fh) ? fh.alpha : 0.0f;
}

// Step 1 — which pixels carry an upstream gradient?  A streaming pass over dL/dE: one workgroup per strip of four
// tiles (64 x 16 pixels), a thread per 4 consecutive pixels (float4 loads: 256 contiguous bytes per row and wave;
// the pixel-per-lane prologue of k_render_bwd reads 32-byte pieces).  Emits, per tile, the live pixels in raster
// order: (x | y << 8, last contributor), and tile_mode = their number (0 = nothing to do, 255 = more than
// SPARSE_LMAX: left to the dense kernel).
constexpr int TILE_DENSE = 255;
__global__ __launch_bounds__(256) void k_bwd_live_pixels(int W, int H, int ED, int ch_base, int gx,
                                                         const uint32_t* __restrict__ n_contrib,
                                                         const float* __restrict__ dE, uint32_t* __restrict__ live_list,
                                                         uint8_t* __restrict__ tile_mode) {
    __shared__ unsigned char s_c[4][64];
    const int x4 = threadIdx.x & 15, y = threadIdx.x >> 4;
    const int tl = x4 >> 2;                               // tile of the strip
    const int tx = blockIdx.x * 4 + tl, ty = blockIdx.y;
    const int px0 = blockIdx.x * 64 + x4 * 4, py = ty * TILE + y;
    const size_t N = (size_t)W * H;
    const int nch = dE != nullptr ? min(32, ED - ch_base) : 0;
    unsigned last[4] = {0u, 0u, 0u, 0u};
    bool nz[4] = {false, false, false, false};
    if (py < H && px0 < W) {
        const size_t p0 = (size_t)W * py + px0;
        if ((W & 3) == 0) {              // px0 + 3 < W and 16-byte alignment follow
            const uint4 lc = *reinterpret_cast<const uint4*>(n_contrib + p0);
            last[0] = lc.x; last[1] = lc.y; last[2] = lc.z; last[3] = lc.w;
#pragma unroll
            for (int cb = 0; cb < 32; cb += 8) {          // eight 16-byte loads in flight per round
                float4 v[8];
#pragma unroll
                for (int c = 0; c < 8; c++)
                    v[c] = cb + c < nch ? *reinterpret_cast<const float4*>(dE + (size_t)(ch_base + cb + c) * N + p0)
                                        : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    nz[0] = nz[0] || (v[c].x != 0.0f); nz[1] = nz[1] || (v[c].y != 0.0f);
                    nz[2] = nz[2] || (v[c].z != 0.0f); nz[3] = nz[3] || (v[c].w != 0.0f);
                }
            }
        } else {
            for (int j = 0; j < 4; j++) {
                if (px0 + j >= W) break;
                last[j] = n_contrib[p0 + j];
                for (int c = 0; c < nch; c++) nz[j] = nz[j] || (dE[(size_t)(ch_base + c) * N + p0 + j] != 0.0f);
            }
        }
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { nz[j] = nz[j] && last[j] > 0u; cnt += nz[j] ? 1 : 0; }
    const int o = y * 4 + (x4 & 3);                       // raster order of this thread inside its tile
    s_c[tl][o] = (unsigned char)cnt;
    __syncthreads();
    if (tx >= gx) return;
    const int tile = ty * gx + tx;
    if (o == 63) {
        int tot = 0;
        for (int q = 0; q < 64; q++) tot += s_c[tl][q];
        tile_mode[tile] = (uint8_t)(tot > SPARSE_LMAX ? TILE_DENSE : tot);
    }
    if (cnt == 0) return;
    int idx = 0;
    for (int q = 0; q < o; q++) idx += s_c[tl][q];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!nz[j]) continue;
        if (idx < SPARSE_LMAX) {
            const int lx = (x4 & 3) * 4 + j;
            live_list[((size_t)tile * SPARSE_LMAX + idx) * 2] = (unsigned)(lx | (y << 8));
            live_list[((size_t)tile * SPARSE_LMAX + idx) * 2 + 1] = last[j];
        }
        idx++;
    }
}

// Step 2 — one wave per tile with 1..SPARSE_LMAX live pixels.
// SAMPLED = false: live pixels come from k_bwd_live_pixels (dense dL/dE map, at most SPARSE_LMAX per tile).
// SAMPLED = true : the upstream gradient is given for n SAMPLES, dL/dE(pix[i], :) = sample_rows[i, :] (the map was
//                  only read at those pixels: render(..., sample_pixels=)); seg_off / seg_idx list the samples of every
//                  tile as records (sample, tile-relative pixel, last contributor) written by k_sample_fill.  A tile's samples are walked in groups of SPARSE_LMAX; from the second group on the lane adds to
//                  the row it wrote before (always the same lane of the same wave: no race).  A pixel sampled twice is
//                  two list entries — no merging needed.  Summation order is fixed (sample index) as long as a tile
//                  has at most 512 samples; beyond that only the order between blocks of 512 is the fill kernel's.
// A workgroup of k_render_bwd_sparse is ONE wave: its LDS accesses execute in program order, so between a lane's LDS
// write and another lane's read only the compiler has to be held back.  __syncthreads() would also drain the wave's
// outstanding global loads and stores (workgroup-scope release), i.e. put a memory round trip where none is needed.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <class Math, bool SAMPLED>
#ifndef ISR_SPARSE_WAVES
#define ISR_SPARSE_WAVES 3
#endif
__global__ __launch_bounds__(64, ISR_SPARSE_WAVES) void k_render_bwd_sparse(
    int W, int H, int ED, int ch_base, int gx, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ box4, const float* __restrict__ rec,
    const float* __restrict__ tm_pre, const float* __restrict__ dE, const uint32_t* __restrict__ point_offsets,
    const Rect16* __restrict__ rects, float* __restrict__ partial, uint8_t* __restrict__ row_flags,
    const uint32_t* __restrict__ live_list, const uint8_t* __restrict__ tile_mode, int row_stride, int feat_off,
    int64_t capacity, const uint32_t* __restrict__ n_contrib, const long long* __restrict__ sample_pix,
    const float* __restrict__ sample_rows, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_idx,
    unsigned long long* __restrict__ row_mask) {
    __shared__ int s_lxy[SPARSE_LMAX];                 // tile-relative x | y << 8 of the live pixels
    __shared__ unsigned s_llast[SPARSE_LMAX];          // their last contributor
    __shared__ unsigned s_sample[SPARSE_LMAX];
    __shared__ __attribute__((aligned(16))) float s_ldE[SPARSE_LMAX * 32];
    constexpr int SEG_SORT = 512;                      // samples of a tile put in index order (fixed summation order)
    // staging area: a chunk's records on their way in (64 x 80 B), then half a wave's finished rows on their way out
    __shared__ __attribute__((aligned(16))) float s_rows[64 * 4 * SPARSE_REC_PITCH4];
    static_assert(64 * 4 * SPARSE_REC_PITCH4 >= 32 * SPARSE_ROW_PITCH, "rows and records share the staging area");
    __shared__ unsigned s_q[2 * SPARSE_QUEUE];         // pass-1 survivors of the tile's list: Gaussian id ...
    int* const s_qid = reinterpret_cast<int*>(s_q);
    unsigned* const s_qhk = s_q + SPARSE_QUEUE;        // ... and which live pixels it may touch
    unsigned* const s_seg_in = s_q;                    // (the queue is empty whenever the samples are ranked)
    static_assert(2 * SPARSE_QUEUE >= SEG_SORT, "the sample ranking borrows the queue");

    const int tile = blockIdx.x;       // (longest lists first - as k_render_bwd_geo - changes nothing here: measured)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const size_t N = (size_t)W * H;
    int ntotal, seg0 = 0;
    if constexpr (SAMPLED) { seg0 = (int)seg_off[tile]; ntotal = (int)seg_off[tile + 1] - seg0; }
    else { ntotal = tile_mode[tile]; if (ntotal == TILE_DENSE) return; }
    if (ntotal == 0) return;
    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    if (len <= 0) return;
  for (int g0 = 0; g0 < ntotal; g0 += SPARSE_LMAX) {
    const int nlive = min(SPARSE_LMAX, ntotal - g0);
    unsigned mylast = 0u;
    if constexpr (SAMPLED) {
        __syncthreads();                                   // the previous group is done with the LDS lists
        {
            // the fill kernel placed the tile's samples in the order of its atomics: rank the (up to) 512 of this block
            // by sample index and take the 32 whose ranks are this group's, so that the groups and the order inside
            // them do not depend on that race.  (Ranked again for every group: a tile with more than 32 samples is rare.)
            const int blk0 = g0 - (g0 % SEG_SORT);
            const int nseg = min(SEG_SORT, ntotal - blk0);
            const uint32_t* recs = seg_idx + 3 * (size_t)(seg0 + blk0);
            unsigned xy0 = 0u, last0 = 0u;                 // the rest of this lane's first record, fetched alongside
            for (int e = lane; e < nseg; e += 64) s_seg_in[e] = recs[3 * e];
            if (lane < nseg) { xy0 = recs[3 * lane + 1]; last0 = recs[3 * lane + 2]; }
            __syncthreads();
            for (int e = lane; e < nseg; e += 64) {
                const unsigned mine = s_seg_in[e];
                int rank = 0;
                for (int j = 0; j < nseg; j++) rank += (s_seg_in[j] < mine) ? 1 : 0;
                rank -= g0 - blk0;
                if (rank >= 0 && rank < nlive) {
                    const unsigned xy = e == lane ? xy0 : recs[3 * e + 1];
                    const unsigned last = e == lane ? last0 : recs[3 * e + 2];
                    s_sample[rank] = mine;
                    s_lxy[rank] = (int)xy;
                    s_llast[rank] = last;
                    mylast = max(mylast, last);
                }
            }
        }
    } else {
        if (lane < nlive) {
            s_lxy[lane] = (int)live_list[((size_t)tile * SPARSE_LMAX + lane) * 2];
            mylast = live_list[((size_t)tile * SPARSE_LMAX + lane) * 2 + 1];
            s_llast[lane] = mylast;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mylast = max(mylast, (unsigned)__shfl_xor((int)mylast, o));
    __syncthreads();
    for (int k = lane >> 5; k < nlive; k += 2) {
        const int c = lane & 31, ch = ch_base + c;
        if constexpr (SAMPLED) {
            s_ldE[k * 32 + c] = ch < ED ? sample_rows[(size_t)s_sample[k] * ED + ch] : 0.0f;
        } else {
            const int xy = s_lxy[k];
            const size_t q = (size_t)W * (ty * TILE + (xy >> 8)) + (tx * TILE + (xy & 255));
            s_ldE[k * 32 + c] = ch < ED ? dE[(size_t)ch * N + q] : 0.0f;
        }
    }
    __syncthreads();
    const int len_eff = min(len, (int)mylast);
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);
    float Tvec = 1.0f;                 // lane k: running transmittance of live pixel k
    // The list is walked in two alternating passes.  Pass 1 only culls: (id, cull box) of 256 splats per round, coalesced
    // and independent of anything computed, against the live pixels; the splats that may touch one are appended, in list
    // order, to an LDS queue (id, hit mask).  Pass 2 takes 64 queued splats at a time - every lane has work - and does
    // what needs the dependent loads (record, rectangle, row offset) and the product scan.  The tile's walk is the critical
    // path of this kernel (one wave per tile), and with a handful of live pixels per tile 4 of 5 splats never reach pass 2.
    int n_c = 0;                       // queued splats (uniform)
    for (int base = 0; base < len_eff || n_c > 0;) {
        while (base < len_eff && n_c <= SPARSE_QUEUE - SPARSE_ROUND) {
            int idv[SPARSE_ROUND / 64];
            unsigned bxv[SPARSE_ROUND / 64];
#pragma unroll
            for (int u = 0; u < SPARSE_ROUND / 64; u++) {
                const int i = base + 64 * u + lane;
                idv[u] = 0; bxv[u] = 0u;
                if (i < len_eff) { idv[u] = (int)point_list[r0 + i]; bxv[u] = box4[r0 + i]; }
            }
#pragma unroll
            for (int u = 0; u < SPARSE_ROUND / 64; u++) {
                const int i = base + 64 * u + lane;
                unsigned hk = 0u;      // bit k: this lane's splat may touch live pixel k
                if (i < len_eff) {
                    const unsigned bx = bxv[u];
                    const int xl = (int)(signed char)(bx & 255u), xh = (int)(signed char)((bx >> 8) & 255u);
                    const int yl = (int)(signed char)((bx >> 16) & 255u), yh = (int)(signed char)(bx >> 24);
                    for (int k = 0; k < nlive; k++) {
                        const int xy = s_lxy[k];
                        const int x = xy & 255, y = xy >> 8;
                        if (xl <= x && xh >= x && yl <= y && yh >= y && (unsigned)i < s_llast[k]) hk |= 1u << k;
                    }
                }
                const unsigned long long b = __ballot(hk != 0u);
                if (hk != 0u) {
                    const int pos = n_c + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
                    s_qid[pos] = idv[u];
                    s_qhk[pos] = hk;
                }
                n_c += __popcll(b);
            }
            base += SPARSE_ROUND;
        }
        wave_lds_sync();
        const bool drained = base >= len_eff;
        int done = 0;
        // one chunk of the queue is in flight while the previous one is evaluated
        int id_n = 0;
        unsigned hk_n = 0u, po_n = 0u;
        // The 64-byte records are fetched four lanes to a record (piece lane & 3 of the records of queue entries
        // 16 j + lane / 4): an instruction touches 16 cache lines instead of 64.  They are turned round through LDS when
        // the chunk's turn comes.
        float4 piece_n[4];
#pragma unroll
        for (int j = 0; j < 4; j++) piece_n[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        Rect16 rc_n = {0, 0, 0, 0};
        float band_n = 0.0f;
        auto fetch = [&](int at) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = at + 16 * j + (lane >> 2);
                if (e < n_c)
                    piece_n[j] = reinterpret_cast<const float4*>(rec + (size_t)s_qid[e] * REC)[lane & 3];
            }
            const int e = at + lane;
            hk_n = 0u; id_n = 0;
            if (e < n_c) { id_n = s_qid[e]; hk_n = s_qhk[e]; }
            if (hk_n != 0u) {
                rc_n = rects[id_n];
                po_n = point_offsets[id_n];
                if constexpr (Math::fast) band_n = rec[(size_t)id_n * REC + 19];     // K1's guard band (isr_fast_pair.hpp)
            }
        };
        auto ready = [&](int at) { return n_c - at >= 64 || (drained && at < n_c); };
        if (ready(0)) fetch(0);
        while (ready(done)) {
            const int id = id_n;
            const unsigned hk = hk_n, po = po_n;
            const Rect16 rc = rc_n;
            const float band = band_n;
            wave_lds_sync();                                   // the staging area is free (rows of the previous chunk are out)
#pragma unroll
            for (int j = 0; j < 4; j++)
                reinterpret_cast<float4*>(s_rows)[(16 * j + (lane >> 2)) * SPARSE_REC_PITCH4 + (lane & 3)] = piece_n[j];
            wave_lds_sync();
            float4 a, b, c, d;
            {
                const float4* mine = reinterpret_cast<const float4*>(s_rows) + lane * SPARSE_REC_PITCH4;
                a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
            }
            if (tm_pre != nullptr && hk != 0u) {               // precomputed transforms replace the first nine entries
                const float* tp = tm_pre + 9 * (size_t)id;
                a = make_float4(tp[0], tp[1], tp[2], tp[3]);
                b = make_float4(tp[4], tp[5], tp[6], tp[7]);
                c.x = tp[8];
            }
            wave_lds_sync();
            if (ready(done + 64)) fetch(done + 64);
            done += 64;
            F3 Tu = {0, 0, 0}, Tv = {0, 0, 0}, Tw = {0, 0, 1};
            float cx = 0, cy = 0, opa = 0, skip = 0;
            unsigned slot = 0, ordinal = 0;
            if (hk != 0u) {
                ordinal = (unsigned)(ty - rc.y0) * (unsigned)(rc.x1 - rc.x0) + (unsigned)(tx - rc.x0);
                slot = po + ordinal;
                Tu = {a.x, a.y, a.z}; Tv = {a.w, b.x, b.y}; Tw = {b.z, b.w, c.x};
                cx = c.y; cy = c.z; opa = d.z;
                if (opa <= 1.0f) {       // EXACT's cull: opa * exp(-rho / 2) < 1/255 for every rho > skip (1 % + 0.05 margin)
                    const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
                    skip = 2.0f * l * 1.01f + 0.05f;
                } else skip = __builtin_inff();
            }
            float det = 0.0f;
            FastBand fb = {0.0f, 0.0f, 0.0f};
            if constexpr (Math::fast) { det = fast_det(Tu, Tv, Tw, cx, cy); fb = fast_band(opa, band); }
            float acc[32];
#pragma unroll
            for (int c2 = 0; c2 < 32; c2++) acc[c2] = 0.0f;
            bool wrote = false;
            for (int k = 0; k < nlive; k++) {
                if (__ballot((hk >> k) & 1u) == 0ull) continue;
                const int xy = s_lxy[k];
                float alpha = 0.0f;
                if ((hk >> k) & 1u) {
                    if constexpr (Math::fast)
                        alpha = splat_alpha_fast(Tu, Tv, Tw, cx, cy, opa, det, fb, tile_x0 + (float)(xy & 255), tile_y0 + (float)(xy >> 8));
                    else
                        alpha = splat_alpha<Math>(Tu, Tv, Tw, cx, cy, opa, skip, tile_x0 + (float)(xy & 255), tile_y0 + (float)(xy >> 8));
                }
                if (__ballot(alpha != 0.0f) == 0ull) continue;
                const float incl = wave_scan_mul(1.0f - alpha);
                const float excl = dpp_fetch<0x138, 0xF>(incl, 1.0f);        // wave_shr:1
                const float Tk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tvec), k));
                const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
                if (lane == k) Tvec = Tk * total;
                const float w = alpha * (Tk * excl);
                if (w != 0.0f) {
                    wrote = true;
                    const float4* e4 = reinterpret_cast<const float4*>(s_ldE + k * 32);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const float4 v = e4[q];
                        acc[4 * q + 0] = __builtin_fmaf(w, v.x, acc[4 * q + 0]);
                        acc[4 * q + 1] = __builtin_fmaf(w, v.y, acc[4 * q + 1]);
                        acc[4 * q + 2] = __builtin_fmaf(w, v.z, acc[4 * q + 2]);
                        acc[4 * q + 3] = __builtin_fmaf(w, v.w, acc[4 * q + 3]);
                    }
                }
            }
            const unsigned long long wmask = __ballot(wrote);
            if (wmask != 0ull) {
                if (SAMPLED && g0 > 0 && wrote && row_flags[slot]) {   // an earlier group of this tile's samples reached the splat too
                    const float4* o4 = reinterpret_cast<const float4*>(partial + (size_t)slot * row_stride + feat_off);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const float4 o = o4[q];
                        acc[4 * q] += o.x; acc[4 * q + 1] += o.y; acc[4 * q + 2] += o.z; acc[4 * q + 3] += o.w;
                    }
                }
                // The rows leave through LDS, half a wave at a time: a lane storing its own 128-byte row makes every store
                // instruction touch 64 cache lines with 16 bytes each; turned round, 8 lanes write one row and an
                // instruction touches 8 whole lines.  (Measured: the row stores were a quarter of this kernel.)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (((wmask >> (32 * h)) & 0xffffffffull) == 0ull) continue;
                    wave_lds_sync();
                    if ((lane >> 5) == h && wrote) {
                        float4* t4 = reinterpret_cast<float4*>(s_rows + (lane & 31) * SPARSE_ROW_PITCH);
#pragma unroll
                        for (int q = 0; q < 8; q++) t4[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int row = 8 * j + (lane >> 3), src = 32 * h + row;
                        const unsigned to = (unsigned)__shfl((int)slot, src);
                        if ((wmask >> src) & 1ull) {
                            const float4 v = *reinterpret_cast<const float4*>(s_rows + row * SPARSE_ROW_PITCH + 4 * (lane & 7));
                            *reinterpret_cast<float4*>(partial + (size_t)to * row_stride + feat_off + 4 * (lane & 7)) = v;
                        }
                    }
                }
                if (wrote) {
                    row_flags[slot] = 1;
                    // per-Gaussian summary of the flags (an OR: order-independent), so that the per-Gaussian pass finds the
                    // rows of a Gaussian with one load instead of one per tile instance
                    if (row_mask != nullptr) atomicOr(row_mask + id, 1ull << (ordinal < 63u ? ordinal : 63u));
                }
            }
        }
        // what is left (less than a chunk, unless the list is drained) moves to the front of the queue
        const int rem = n_c - done;
        int id_keep = 0;
        unsigned hk_keep = 0u;
        if (lane < rem) { id_keep = s_qid[done + lane]; hk_keep = s_qhk[done + lane]; }
        wave_lds_sync();
        if (lane < rem) { s_qid[lane] = id_keep; s_qhk[lane] = hk_keep; }
        n_c = rem > 0 ? rem : 0;
        wave_lds_sync();
    }
  }
}

// Samples -> per-tile segments: count, (single-workgroup) scan, fill.  The order inside a segment is the order of the
// cursor atomics; k_render_bwd_sparse re-orders every group of SPARSE_LMAX by sample index.
__global__ __launch_bounds__(256) void k_sample_count(int n, int W, int H, int gx, const long long* __restrict__ pix,
                                                      uint32_t* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long q = pix[i];
    if (q < 0 || q >= (long long)W * H) return;
    atomicAdd(cnt + ((int)(q / W) / TILE) * gx + ((int)(q % W) / TILE), 1u);
}
__global__ __launch_bounds__(1024) void k_sample_scan(int T, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off,
                                                      uint32_t* __restrict__ cursor) {
    // a thread owns `per` CONSECUTIVE tiles (its counts read up front, all loads in flight together), one workgroup scan of
    // the 1024 partial sums: one barrier-separated scan instead of one per 1024 tiles (8 of them at 1080p: 33 us -> 8 us)
    __shared__ uint32_t s_warp[32];
    constexpr int MAXPER = 16;
    const int per = (T + 1023) / 1024;
    if (per <= MAXPER) {
        const int t0 = (int)threadIdx.x * per;
        uint32_t c[MAXPER];
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            c[k] = (k < per && t0 + k < T) ? cnt[t0 + k] : 0u;
            mine += c[k];
        }
        uint32_t total;
        uint32_t run = block_exclusive_scan_1024(mine, s_warp, total);
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            if (k < per && t0 + k < T) { off[t0 + k] = run; cursor[t0 + k] = 0u; run += c[k]; }
        }
        if (threadIdx.x == 0) off[T] = total;
        return;
    }
    uint32_t carry = 0;
    for (int base = 0; base < T; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < T ? cnt[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_warp, total);
        if (i < T) { off[i] = carry + ex; cursor[i] = 0u; }
        carry += total;
    }
    if (threadIdx.x == 0) off[T] = carry;
}
__global__ __launch_bounds__(256) void k_sample_fill(int n, int W, int H, int gx, const long long* __restrict__ pix,
                                                     const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
                                                     const uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ seg_rec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long q = pix[i];
    if (q < 0 || q >= (long long)W * H) return;
    const int qx = (int)(q % W), qy = (int)(q / W);
    const int t = (qy / TILE) * gx + (qx / TILE);
    // (sample, tile-relative pixel, last contributor): everything the tile's wave needs about the sample, so that it
    // gets it with one load instead of three dependent ones
    uint32_t* r = seg_rec + 3 * (size_t)(off[t] + atomicAdd(cursor + t, 1u));
    r[0] = (uint32_t)i;
    r[1] = (uint32_t)((qx % TILE) | ((qy % TILE) << 8));
    r[2] = n_contrib[q];
}
// out[i, :] = map[:, pix[i]]  (the forward half of the sampled path)
__global__ __launch_bounds__(256) void k_sample_gather(int n, int F, long long N, const float* __restrict__ map,
                                                       const long long* __restrict__ pix, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n * F) return;
    const int i = (int)(e / F), c = (int)(e - (long long)i * F);
    const long long q = pix[i];
    out[e] = (q >= 0 && q < N) ? map[(size_t)c * N + q] : 0.0f;
}

// ----------------------------------------------------------------------------
// Row reduction: out[g, c] = sum over the Gaussian's tiles of partial[slot, src_off + c].
__global__ __launch_bounds__(256) void k_reduce_rows(int P, int ncol, const uint32_t* __restrict__ point_offsets,
                                                     const uint32_t* __restrict__ tiles_touched,
                                                     const float* __restrict__ partial, const uint8_t* __restrict__ row_flags,
                                                     int64_t R, int row_stride, int src_off, float* __restrict__ out,
                                                     int out_stride, int accumulate) {
    // one thread per (Gaussian, group of 4 channels); the row bytes of a Gaussian are contiguous
    const int q4 = (ncol + 3) >> 2;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)P * q4) return;
    const int g = (int)(e / q4), q = (int)(e - (size_t)g * q4);
    const int c = 4 * q;
    const uint32_t n = tiles_touched[g];
    const size_t base = point_offsets[g];
    const uint8_t* fl = row_flags + (size_t)(c >> 5) * R + base;     // pass (c / 32) wrote feature chunk (c / 32)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    // eight rows per round: flags, then the flagged rows, are independent loads in flight together (the common
    // Gaussian touches < 8 tiles, so the whole reduction is two memory round trips); summed in row order
    for (uint32_t r = 0; r < n; r += 8) {
        bool f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { f[u] = (r + u < n) && fl[r + u] != 0; any |= f[u]; }
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            v[u] = f[u] ? *reinterpret_cast<const float4*>(partial + (base + r + u) * row_stride + src_off + c)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (f[u]) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    float* o = out + (size_t)g * out_stride + c;
    if (accumulate && !any) return;         // nothing to add: the row is neither read nor written (a view's samples reach few rows)
    if (accumulate) {
        s.x += o[0];
        if (c + 1 < ncol) s.y += o[1];
        if (c + 2 < ncol) s.z += o[2];
        if (c + 3 < ncol) s.w += o[3];
    }
    if (c + 3 < ncol && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = s;
    else {
        o[0] = s.x;
        if (c + 1 < ncol) o[1] = s.y;
        if (c + 2 < ncol) o[2] = s.z;
        if (c + 3 < ncol) o[3] = s.w;
    }
}

// k_reduce_rows behind a sampled backward, which also leaves a 64-bit word per Gaussian (bit i = its i-th tile instance holds a
// row: k_render_bwd_sparse): one coalesced 8-byte read decides for the ~99 % of Gaussians a view's few thousand samples
// never reach - they are left alone when the sums are added to an existing gradient (accumulate), zero-filled otherwise -
// instead of the instance count, the offset and the byte flags of every instance (0.10 ms per view at P = 1.5 M however
// few rows there are).  Flagged rows are summed in ascending instance order, like k_reduce_rows.
__global__ __launch_bounds__(256) void k_reduce_rows_masked(int P, int ncol, const uint32_t* __restrict__ point_offsets,
                                                            const uint32_t* __restrict__ tiles_touched,
                                                            const unsigned long long* __restrict__ row_mask,
                                                            const float* __restrict__ partial, const uint8_t* __restrict__ row_flags,
                                                            int64_t R, int row_stride, float* __restrict__ out, int out_stride,
                                                            int accumulate) {
    const int q4 = (ncol + 3) >> 2;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)P * q4) return;
    const int g = (int)(e / q4), q = (int)(e - (size_t)g * q4);
    const int c = 4 * q;
    const unsigned long long mk = row_mask[g];
    if (mk == 0ull && accumulate) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mk != 0ull) {
        const size_t base = point_offsets[g];
        unsigned long long bits = mk & ~(1ull << 63);
        while (bits != 0ull) {
            int idx[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                idx[u] = bits != 0ull ? __builtin_ctzll(bits) : -1;
                bits &= bits - 1ull;
            }
            float4 pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx[u] >= 0) pv[u] = *reinterpret_cast<const float4*>(partial + (base + idx[u]) * row_stride + c);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (idx[u] >= 0) { s.x += pv[u].x; s.y += pv[u].y; s.z += pv[u].z; s.w += pv[u].w; }
        }
        if (mk >> 63) {                                  // more than 63 tile instances: byte flags from there on
            const uint32_t n = tiles_touched[g];
            const uint8_t* fl = row_flags + (size_t)(c >> 5) * R + base;
            for (uint32_t r = 63; r < n; r++)
                if (fl[r] != 0) {
                    const float4 pv = *reinterpret_cast<const float4*>(partial + (base + r) * row_stride + c);
                    s.x += pv.x; s.y += pv.y; s.z += pv.z; s.w += pv.w;
                }
        }
    }
    float* o = out + (size_t)g * out_stride + c;
    if (accumulate) {
        s.x += o[0];
        if (c + 1 < ncol) s.y += o[1];
        if (c + 2 < ncol) s.z += o[2];
        if (c + 3 < ncol) s.w += o[3];
    }
    if (c + 3 < ncol && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = s;
    else {
        o[0] = s.x;
        if (c + 1 < ncol) o[1] = s.y;
        if (c + 2 < ncol) o[2] = s.z;
        if (c + 3 < ncol) o[3] = s.w;
    }
}

// ----------------------------------------------------------------------------
// Feature training: the whole per-Gaussian tail of a step in ONE pass over the [P,F] rows.
//
// After the sampled backward the step still has to (1) sum each Gaussian's per-tile partial rows (k_reduce_rows),
// (2) chain that gradient through the two row normalisations y = x/(|x|+eps1), z = y/(|y|+eps2) together with the 3-D
// loss' gradient on y (rn2_kernel<true>), (3) apply Adam and (4) emit the next forward's y and z (adam_rn2_kernel):
// three streaming kernels with 2 + 4 + 9 passes over [P,F].  A row needs nothing from any other row, so here the lanes
// that own a row (one float4 each) do all of it with the row in registers: in  x, m, v, gy (+ the flagged partial rows),
// out  x, m, v, y, z.  ADAM = false stops after (2) and writes dL/dx (the multi-GPU trainer all-reduces it first).
// Same expressions as the three kernels it replaces (bit-identical results).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const nt_f4 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
    const nt_f4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<nt_f4*>(p));
}

template <bool ADAM>
__global__ __launch_bounds__(256) void k_feature_rows_step(
    int row0, int P, int F, const uint32_t* __restrict__ point_offsets, const uint32_t* __restrict__ tiles_touched,
    const unsigned long long* __restrict__ row_mask, const float* __restrict__ partial,
    const uint8_t* __restrict__ row_flags, int64_t R, int row_stride,
    const float* __restrict__ gz_dense, const float* __restrict__ gy, int* __restrict__ gy_slot,
    const float* __restrict__ gy_merged, float eps1, float eps2, float* __restrict__ x,
    float* __restrict__ grad_out, float lr_over_bc1, float om1, float beta2, float om2, float inv_sqrt_bc2, float eps,
    float* __restrict__ m, float* __restrict__ v, float* __restrict__ y, float* __restrict__ z, float* __restrict__ zscale) {
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q) lpr <<= 1;
    const int sub = (threadIdx.x & 63) & (lpr - 1);
    const long long row = row0 + ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;      // rows [row0, P) of the table
    const bool ok = row < P && sub < q;
    const int c = 4 * sub;
    const size_t off = (size_t)(row < P ? row : 0) * F + c;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the row's streams are requested first: they are in flight while the partial rows are chased (flag, then row).
    // (Written as `if`, not `ok ? *p : z4`: on a class type such as float4 the conditional operator selects between two
    // LVALUES - the compiler parks the zeros in scratch memory, picks an address and reads it with four flat dword loads;
    // every lane then also WRITES 16 bytes of scratch, a whole extra [P,F] stream, which is how it was found: WRITE_SIZE
    // showed five written streams for a kernel that stores four.)
    float4 xv = z4, a = z4;
    if (ok) xv = *reinterpret_cast<const float4*>(x + off);
    if (ok && gy != nullptr) a = *reinterpret_cast<const float4*>(gy + off);
    const int gslot = (ok && gy_slot != nullptr) ? gy_slot[row] : -1;       // dL/dy as (row -> merged entry), iso_rows_compact
    if (gslot >= 0 && sub == 0) gy_slot[row] = -1;      // consumed: the table is clean again after the pass (no 4 P-byte fill per step)
    float4 m4 = z4, v4 = z4;
    if (ADAM && ok) { m4 = nt_load4(m + off); v4 = nt_load4(v + off); }      // (moments: read once per step)
    // (1) dL/dz row: flagged per-tile partial rows in row order (+ a dense contribution, if any)
    float4 bd = z4;
    if (ok && gz_dense != nullptr) bd = *reinterpret_cast<const float4*>(gz_dense + off);
    float4 b = z4;
    if (ok && partial != nullptr) {
        // which of the Gaussian's tile instances hold a row: one 64-bit word (k_render_bwd_sparse) — two dependent memory
        // round trips (mask, rows) instead of three (instance count, byte flags, rows), and nothing at all for the ~70 % of
        // Gaussians no sampled pixel reached
        const unsigned long long mk = row_mask[row];
        if (mk != 0ull) {
            const size_t base = point_offsets[row];
            unsigned long long bits = mk & ~(1ull << 63);
            while (bits != 0ull) {                       // ascending instance order = the order of k_reduce_rows
                int idx[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    idx[u] = bits != 0ull ? __builtin_ctzll(bits) : -1;
                    bits &= bits - 1ull;
                }
                float4 pv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    pv[u] = z4;
                    if (idx[u] >= 0) pv[u] = *reinterpret_cast<const float4*>(partial + (base + idx[u]) * row_stride + c);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (idx[u] >= 0) { b.x += pv[u].x; b.y += pv[u].y; b.z += pv[u].z; b.w += pv[u].w; }
            }
            if (mk >> 63) {                              // a Gaussian spread over more than 63 tiles: byte flags from there on
                const uint32_t n = tiles_touched[row];
                const uint8_t* fl = row_flags + (size_t)(c >> 5) * R + base;
                for (uint32_t r = 63; r < n; r++)
                    if (fl[r] != 0) {
                        const float4 pv = *reinterpret_cast<const float4*>(partial + (base + r) * row_stride + c);
                        b.x += pv.x; b.y += pv.y; b.z += pv.z; b.w += pv.w;
                    }
            }
        }
    }
    if (gz_dense != nullptr) { b.x += bd.x; b.y += bd.y; b.z += bd.z; b.w += bd.w; }
    if (gslot >= 0) {
        const float4 e = *reinterpret_cast<const float4*>(gy_merged + (size_t)gslot * F + c);
        a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
    }
    // (2) through z = y/(|y|+eps2), y = x/(|x|+eps1)   (rn2_kernel<true>)
    float ss = xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
    float sa = xv.x * a.x + xv.y * a.y + xv.z * a.z + xv.w * a.w;
    float sb = xv.x * b.x + xv.y * b.y + xv.z * b.z + xv.w * b.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) {
        ss += __shfl_xor(ss, o); sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o);
    }
    const float nx = __builtin_sqrtf(ss), r1 = 1.0f / (nx + eps1);
    const float ny = nx * r1, r2 = 1.0f / (ny + eps2);
    const float k2 = ny > 0.0f ? r2 * r2 * (r1 * sb) / ny : 0.0f;
    const float sxu = sa + r2 * sb - k2 * r1 * ss;
    const float k1 = nx > 0.0f ? r1 * r1 * sxu / nx : 0.0f;
    const float cy = k2 * r1;
    float4 g4;
    g4.x = r1 * (a.x + r2 * b.x - cy * xv.x) - k1 * xv.x;
    g4.y = r1 * (a.y + r2 * b.y - cy * xv.y) - k1 * xv.y;
    g4.z = r1 * (a.z + r2 * b.z - cy * xv.z) - k1 * xv.z;
    g4.w = r1 * (a.w + r2 * b.w - cy * xv.w) - k1 * xv.w;
    if constexpr (!ADAM) {
        if (ok) *reinterpret_cast<float4*>(grad_out + off) = g4;
    } else {
        // (3) torch.optim.Adam arithmetic (adam_rn2_kernel)
        float4 np4 = z4;
        if (ok) {
#define ISR_ADAM1(e)                                                                    \
            m4.e = m4.e + om1 * (g4.e - m4.e);                                          \
            v4.e = beta2 * v4.e + om2 * (g4.e * g4.e);                                  \
            np4.e = xv.e - lr_over_bc1 * (m4.e / (__builtin_sqrtf(v4.e) * inv_sqrt_bc2 + eps));
            ISR_ADAM1(x) ISR_ADAM1(y) ISR_ADAM1(z) ISR_ADAM1(w)
#undef ISR_ADAM1
            nt_store4(m + off, m4);           // streamed once per step: do not displace the forward's records in L2
            nt_store4(v + off, v4);
            nt_store4(x + off, np4);
        }
        // (4) the next forward's normalisations of the updated row
        float s1 = np4.x * np4.x + np4.y * np4.y + np4.z * np4.z + np4.w * np4.w;
        for (int o = lpr >> 1; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o);
        const float q1 = 1.0f / (__builtin_sqrtf(s1) + eps1);
        const float4 y4 = make_float4(np4.x * q1, np4.y * q1, np4.z * q1, np4.w * q1);
        float s2 = y4.x * y4.x + y4.y * y4.y + y4.z * y4.z + y4.w * y4.w;
        for (int o = lpr >> 1; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
        const float q2 = 1.0f / (__builtin_sqrtf(s2) + eps2);
        if (ok) {
            if (y != nullptr) *reinterpret_cast<float4*>(y + off) = y4;      // optional: iso_gather_rownorm recomputes rows
            // z = (x q1) q2, the table the next forward blends: written out ([P,F], a seventh stream) - or only its two factors
            // per row ([P,2]), which the blend applies to the rows it stages (isr_forward_render_scaled: the same two multiplies)
            if (zscale != nullptr) { if (sub == 0) *reinterpret_cast<float2*>(zscale + 2 * (size_t)row) = make_float2(q1, q2); }
            else nt_store4(z + off, make_float4(y4.x * q2, y4.y * q2, y4.z * q2, y4.w * q2));
        }
    }
}

// The two factors alone, of rows that no step has touched yet (the first forward of a run): q1 = 1 / (|x| + eps1),
// q2 = 1 / (|x q1| + eps2) with step (4)'s own expressions and summation order above - the same bits.
__global__ __launch_bounds__(256) void k_row_scales(int P, int F, float eps1, float eps2, const float* __restrict__ x,
                                                    float* __restrict__ zscale) {
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q) lpr <<= 1;
    const int sub = (threadIdx.x & 63) & (lpr - 1);
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool ok = row < P && sub < q;
    float4 np4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) np4 = *reinterpret_cast<const float4*>(x + (size_t)row * F + 4 * sub);
    float s1 = np4.x * np4.x + np4.y * np4.y + np4.z * np4.z + np4.w * np4.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o);
    const float q1 = 1.0f / (__builtin_sqrtf(s1) + eps1);
    const float4 y4 = make_float4(np4.x * q1, np4.y * q1, np4.z * q1, np4.w * q1);
    float s2 = y4.x * y4.x + y4.y * y4.y + y4.z * y4.z + y4.w * y4.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
    const float q2 = 1.0f / (__builtin_sqrtf(s2) + eps2);
    if (ok && sub == 0) *reinterpret_cast<float2*>(zscale + 2 * (size_t)row) = make_float2(q1, q2);
}

// ----------------------------------------------------------------------------
// K10 (reference backward.cu:469-656) fused with the geometry row reduction.
__device__ __forceinline__ void quat_cols_b(const float* q, F3& c0, F3& c1, F3& c2, float& w, float& x, float& y, float& z) {
    float s = 1.0f / __builtin_sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    w = q[0] * s; x = q[1] * s; y = q[2] * s; z = q[3] * s;
    c0 = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
    c1 = {2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
    c2 = {2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
}
__device__ __forceinline__ F3 dnorm_dv(F3 v, F3 dv) {
    float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float inv = 1.0f / __builtin_sqrtf(s2 * s2 * s2);
    F3 r;
    r.x = ((+s2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * inv;
    r.y = (-v.x * v.y * dv.x + (s2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * inv;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (s2 - v.z * v.z) * dv.z) * inv;
    return r;
}

// STAGE_SH (degree-3 colours, M = 16): the workgroup's 256 SH rows and its 256 rows of dL/dSH are contiguous 48 KB pieces of
// their tensors; a lane walking its own 192-byte row touches 64 cache lines per instruction (and the gradient row was
// written twice: zeros, then values).  Both now go through LDS - coalesced 16-byte loads in, the lane's coefficients copied
// to registers, the gradient row assembled in the same LDS row, coalesced 16-byte stores out - as K1 does for its input.
template <bool STAGE_SH>
__global__ __launch_bounds__(256) void k_preprocess_bwd(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ tm_pre,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos, int Wd, int Hd,
    GeomView g, const float* __restrict__ partial, const uint8_t* __restrict__ row_flags, int row_stride, int geom_off,
    int rpi /* partial rows per tile instance: 1, or 8 (one per 8x4 block half: k_render_bwd_geo) */,
    float* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dnormal, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
    float* __restrict__ dL_dmean3D, float* __restrict__ dL_dtransMat, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
    constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int SH_STRIDE = 52;           // floats per staged row: 48 + 4 (16-byte aligned, spreads the LDS banks)
    __shared__ __attribute__((aligned(16))) float s_sh[STAGE_SH ? 256 * SH_STRIDE : 4];
    float creg[STAGE_SH ? 48 : 1];
    if constexpr (STAGE_SH) {
        const int i0 = blockIdx.x * 256;
        const int nq = min(256, P - i0) * 12;                      // float4s to move
        const float4* src = reinterpret_cast<const float4*>(shs + (size_t)i0 * 48);
        float4 v[12];
#pragma unroll
        for (int u = 0; u < 12; u++) v[u] = src[min((int)threadIdx.x + u * 256, nq - 1)];
#pragma unroll
        for (int u = 0; u < 12; u++) {
            const int e = (int)threadIdx.x + u * 256;
            if (e < nq) { const int r = e / 12, k = e - r * 12; *reinterpret_cast<float4*>(s_sh + r * SH_STRIDE + 4 * k) = v[u]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 q = *reinterpret_cast<const float4*>(s_sh + threadIdx.x * SH_STRIDE + 4 * k);
            creg[4 * k] = q.x; creg[4 * k + 1] = q.y; creg[4 * k + 2] = q.z; creg[4 * k + 3] = q.w;
        }
    }
    const size_t I = (size_t)i;
    // every output row is fully written (no zero-initialisation by the caller)
    float gs[18];
#pragma unroll
    for (int q = 0; q < 18; q++) gs[q] = 0.0f;
    const uint32_t nt = i < P ? g.tiles_touched[i] * (uint32_t)rpi : 0u;
    // A splat spread over very many tiles (a background surfel grown over the whole view: 6 468 candidate rows at 779x519) has
    // its rows summed by the WORKGROUP - a row per thread, a fixed reduction tree - not by its own lane while 255 others wait.
    constexpr uint32_t K10_BIG_ROWS = 512;
    // (its LDS lives in the four padding floats of the staged SH rows - three workgroups of 52 KB must still fit a CU - or, in
    // the build without staging, in a small block of its own)
    __shared__ float s_coop[STAGE_SH ? 1 : 3 * 256];
    float* const coop = STAGE_SH ? s_sh + 48 : s_coop;             // slot (k, j), j < 3: coop[k * CS + j]
    constexpr int CS = STAGE_SH ? SH_STRIDE : 3;
    int& s_nbig = *reinterpret_cast<int*>(coop);                                       // slot (0, 0)
    auto owner_at = [&](int k) -> int& { return *reinterpret_cast<int*>(coop + k * CS + 1); };       // slots (k, 1)
    auto sum_at = [&](int w, int q) -> float& { return coop[(w * 18 + q) * CS + 2]; };               // slots (18 w + q, 2)
    const bool big_rows = nt > K10_BIG_ROWS;
    // (view-uniform, from k_scan_tops: no barrier at all unless the view holds such a splat)
    const bool any_big = g.header[1] * (int64_t)rpi > (int64_t)K10_BIG_ROWS;
    if (any_big) {
        if (threadIdx.x == 0) s_nbig = 0;
        __syncthreads();
        if (big_rows) owner_at(atomicAdd(&s_nbig, 1)) = (int)threadIdx.x;
    }
    if (nt > 0 && !big_rows && (rpi & 3) == 0) {
        // eight rows per tile instance (k_render_bwd_geo): four flags are one aligned word, and the flagged rows of a word are
        // requested together - a quarter of the dependent memory round trips of the row-by-row walk (the kernel is bound by that
        // chain, not by bytes).  Same summation order: instance by instance, row 0..7.
        const float* src = partial + (size_t)g.point_offsets[i] * rpi * row_stride + geom_off;
        const uint32_t* fw = reinterpret_cast<const uint32_t*>(row_flags + (size_t)g.point_offsets[i] * rpi);
        const uint32_t ni = nt >> 2;
        uint32_t f_next = fw[0];
        for (uint32_t t = 0; t < ni; t++) {
            const uint32_t f4 = f_next;
            if (t + 1 < ni) f_next = fw[t + 1];
            if (f4 == 0u) continue;
            float4 rw[4][5];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const bool on = ((f4 >> (8 * b)) & 0xffu) != 0u;
                const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)(4 * t + b) * row_stride);
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    rw[b][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on) rw[b][k] = s4[k];
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (((f4 >> (8 * b)) & 0xffu) == 0u) continue;      // (an unflagged row adds nothing, not even +0)
                const float4 a = rw[b][0], bq = rw[b][1], c = rw[b][2], d = rw[b][3], e = rw[b][4];
                gs[0] += a.x; gs[1] += a.y; gs[2] += a.z; gs[3] += a.w; gs[4] += bq.x; gs[5] += bq.y; gs[6] += bq.z; gs[7] += bq.w;
                gs[8] += c.x; gs[9] += c.y; gs[10] += c.z; gs[11] += c.w; gs[12] += d.x; gs[13] += d.y; gs[14] += d.z;
                gs[15] += d.w; gs[16] += e.x; gs[17] += e.y;
            }
        }
    } else if (nt > 0 && !big_rows) {
        const float* src = partial + (size_t)g.point_offsets[i] * rpi * row_stride + geom_off;
        const uint8_t* fl = row_flags + (size_t)g.point_offsets[i] * rpi;
        for (uint32_t r = 0; r < nt; r++) {
            if (!fl[r]) continue;
            const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)r * row_stride);
            const float4 a = s4[0], b = s4[1], c = s4[2], d = s4[3], e = s4[4];
            gs[0] += a.x; gs[1] += a.y; gs[2] += a.z; gs[3] += a.w; gs[4] += b.x; gs[5] += b.y; gs[6] += b.z; gs[7] += b.w;
            gs[8] += c.x; gs[9] += c.y; gs[10] += c.z; gs[11] += c.w; gs[12] += d.x; gs[13] += d.y; gs[14] += d.z;
            gs[15] += d.w; gs[16] += e.x; gs[17] += e.y;
        }
    }
    if (any_big) {
        __syncthreads();
        for (int bb = 0; bb < s_nbig; bb++) {
            const int owner = owner_at(bb);
            const int oi = (int)(blockIdx.x * blockDim.x) + owner;
            const uint32_t ont = g.tiles_touched[oi] * (uint32_t)rpi;
            const float* src = partial + (size_t)g.point_offsets[oi] * rpi * row_stride + geom_off;
            const uint8_t* fl = row_flags + (size_t)g.point_offsets[oi] * rpi;
            float ps[18];
    #pragma unroll
            for (int q = 0; q < 18; q++) ps[q] = 0.0f;
            for (uint32_t r = threadIdx.x; r < ont; r += 256) {
                if (!fl[r]) continue;
                const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)r * row_stride);
                const float4 a = s4[0], b = s4[1], c = s4[2], d = s4[3], e = s4[4];
                ps[0] += a.x; ps[1] += a.y; ps[2] += a.z; ps[3] += a.w; ps[4] += b.x; ps[5] += b.y; ps[6] += b.z; ps[7] += b.w;
                ps[8] += c.x; ps[9] += c.y; ps[10] += c.z; ps[11] += c.w; ps[12] += d.x; ps[13] += d.y; ps[14] += d.z;
                ps[15] += d.w; ps[16] += e.x; ps[17] += e.y;
            }
    #pragma unroll
            for (int q = 0; q < 18; q++) {
    #pragma unroll
                for (int o = 32; o >= 1; o >>= 1) ps[q] += __shfl_xor(ps[q], o);
            }
            if ((threadIdx.x & 63) == 0) {
    #pragma unroll
                for (int q = 0; q < 18; q++) sum_at(threadIdx.x >> 6, q) = ps[q];
            }
            __syncthreads();
            if ((int)threadIdx.x == owner) {
    #pragma unroll
                for (int q = 0; q < 18; q++) gs[q] = (sum_at(0, q) + sum_at(1, q)) + (sum_at(2, q) + sum_at(3, q));
            }
            __syncthreads();
        }
    }
    if (i < P) {
    F3 g0 = {gs[0], gs[1], gs[2]}, g1 = {gs[3], gs[4], gs[5]}, g2 = {gs[6], gs[7], gs[8]};
    const float dmx = gs[9], dmy = gs[10];
    dL_dnormal[3 * I] = gs[11]; dL_dnormal[3 * I + 1] = gs[12]; dL_dnormal[3 * I + 2] = gs[13];
    dL_dopacity[I] = gs[14];
    dL_dcolor[3 * I] = gs[15]; dL_dcolor[3 * I + 1] = gs[16]; dL_dcolor[3 * I + 2] = gs[17];
    float m3x = 0, m3y = 0, m3z = 0, dsc0 = 0, dsc1 = 0, dq[4] = {0, 0, 0, 0};
    float m2x = 0, m2y = 0;
    const bool vis = g.radii[i] > 0;
    const bool precomp = (tm_pre != nullptr);
    float* const sh_out = STAGE_SH ? s_sh + threadIdx.x * SH_STRIDE : (dL_dsh != nullptr ? dL_dsh + I * M * 3 : nullptr);
    if (sh_out != nullptr)
        for (int k = 0; k < M * 3; k++) sh_out[k] = 0.0f;
    if (vis) {
        const float* rec = g.rec + I * REC;
        F3 Tu, Tv, Tw, normal = {0, 0, 0}, R0 = {0, 0, 0}, R1 = {0, 0, 0}, R2 = {0, 0, 0};
        float Pm[3][4];
        F3 p = {0, 0, 0};
        float qw = 0, qx = 0, qy = 0, qz = 0;
        if (precomp) {
            const float* t = tm_pre + 9 * I;
            Tu = {t[0], t[1], t[2]}; Tv = {t[3], t[4], t[5]}; Tw = {t[6], t[7], t[8]};
        } else {
            p = {means3D[3 * I], means3D[3 * I + 1], means3D[3 * I + 2]};
            const float q[4] = {rots[4 * I], rots[4 * I + 1], rots[4 * I + 2], rots[4 * I + 3]};
            quat_cols_b(q, R0, R1, R2, qw, qx, qy, qz);
            const float sx = 1.0f * scales[2 * I], sy = 1.0f * scales[2 * I + 1];   // modifier ignored, backward.cu:507
            const F3 L0 = R0 * sx, L1 = R1 * sy, L2 = R2;
            const float S[3][4] = {{L0.x, L0.y, L0.z, 0.f}, {L1.x, L1.y, L1.z, 0.f}, {p.x, p.y, p.z, 1.f}};
            float n[3][4];
            n[0][0] = (float)((double)(float)Wd / 2.0); n[0][1] = 0.f; n[0][2] = 0.f; n[0][3] = (float)((double)(float)(Wd - 1) / 2.0);
            n[1][0] = 0.f; n[1][1] = (float)((double)(float)Hd / 2.0); n[1][2] = 0.f; n[1][3] = (float)((double)(float)(Hd - 1) / 2.0);
            n[2][0] = 0.f; n[2][1] = 0.f; n[2][2] = 0.f; n[2][3] = 1.f;
            float T[3][3];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    Pm[c][k] = proj[4 * k + 0] * n[c][0] + proj[4 * k + 1] * n[c][1] + proj[4 * k + 2] * n[c][2] + proj[4 * k + 3] * n[c][3];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 3; r++)
                    T[c][r] = S[r][0] * Pm[c][0] + S[r][1] * Pm[c][1] + S[r][2] * Pm[c][2] + S[r][3] * Pm[c][3];
            Tu = {T[0][0], T[0][1], T[0][2]}; Tv = {T[1][0], T[1][1], T[1][2]}; Tw = {T[2][0], T[2][1], T[2][2]};
            normal = {view[0] * L2.x + view[4] * L2.y + view[8] * L2.z, view[1] * L2.x + view[5] * L2.y + view[9] * L2.z,
                      view[2] * L2.x + view[6] * L2.y + view[10] * L2.z};
        }
        const float raw_gT2 = g0.z, raw_gT5 = g1.z;
        bool early = false;
        if (dmx != 0 || dmy != 0) {
            const F3 tv = {9.0f, 9.0f, -1.0f};
            const float d = dot3(tv, Tw * Tw);
            const F3 f = tv * (1.0f / d);
            const F3 a0 = (dmx * f) * Tw;
            const F3 a1 = (dmy * f) * Tw;
            F3 a3 = (dmx * f) * Tu + (dmy * f) * Tv;
            const F3 dL_df = (dmx * Tu) * Tw + (dmy * Tv) * Tw;
            const float dL_dd = (float)((double)dot3(dL_df, f) * (-1.0 / (double)d));
            const F3 dd_dT3 = (tv * Tw) * 2.0f;
            a3 = a3 + dL_dd * dd_dT3;
            g0 = g0 + a0; g1 = g1 + a1; g2 = g2 + a3;
            if (precomp) early = true;
        }
        float hack2 = raw_gT2, hack5 = raw_gT5;
        if (precomp) {
            if (early) { hack2 = g0.z; hack5 = g1.z; }
            else { g0 = {gs[0], gs[1], gs[2]}; g1 = {gs[3], gs[4], gs[5]}; g2 = {gs[6], gs[7], gs[8]}; }
        }
        if (!precomp) {
            auto gT = [&](int c, int r) { const F3& v = (c == 0 ? g0 : (c == 1 ? g1 : g2)); return r == 0 ? v.x : (r == 1 ? v.y : v.z); };
            float dM[3][4];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) dM[r][k] = Pm[0][k] * gT(0, r) + Pm[1][k] * gT(1, r) + Pm[2][k] * gT(2, r);
            const F3 dn = {gs[11], gs[12], gs[13]};
            F3 dL_dtn = {view[0] * dn.x + view[1] * dn.y + view[2] * dn.z, view[4] * dn.x + view[5] * dn.y + view[6] * dn.z,
                         view[8] * dn.x + view[9] * dn.y + view[10] * dn.z};
            const F3 pv = {view[0] * p.x + view[4] * p.y + view[8] * p.z + view[12],
                           view[1] * p.x + view[5] * p.y + view[9] * p.z + view[13],
                           view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14]};
            const F3 pn = pv * normal;
            const float cosv = -(pn.x + pn.y + pn.z);
            dL_dtn = (cosv > 0 ? 1.0f : -1.0f) * dL_dtn;
            const F3 rs0 = {dM[0][0], dM[0][1], dM[0][2]}, rs1 = {dM[1][0], dM[1][1], dM[1][2]}, rs2 = dL_dtn;
            const float sx = scales[2 * I], sy = scales[2 * I + 1];
            const F3 r0 = rs0 * F3{sx, sx, sx}, r1 = rs1 * F3{sy, sy, sy}, r2 = rs2;
            // v_R(col,row)
            auto G = [&](int col, int row) { const F3& v = (col == 0 ? r0 : (col == 1 ? r1 : r2)); return row == 0 ? v.x : (row == 1 ? v.y : v.z); };
            const float w = qw, x = qx, y = qy, z = qz;
            dq[0] = 2.f * (x * (G(1, 2) - G(2, 1)) + y * (G(2, 0) - G(0, 2)) + z * (G(0, 1) - G(1, 0)));
            dq[1] = 2.f * (-2.f * x * (G(1, 1) + G(2, 2)) + y * (G(0, 1) + G(1, 0)) + z * (G(0, 2) + G(2, 0)) + w * (G(1, 2) - G(2, 1)));
            dq[2] = 2.f * (x * (G(0, 1) + G(1, 0)) - 2.f * y * (G(0, 0) + G(2, 2)) + z * (G(1, 2) + G(2, 1)) + w * (G(2, 0) - G(0, 2)));
            dq[3] = 2.f * (x * (G(0, 2) + G(2, 0)) + y * (G(1, 2) + G(2, 1)) - 2.f * z * (G(0, 0) + G(1, 1)) + w * (G(0, 1) - G(1, 0)));
            dsc0 = dot3(rs0, R0);
            dsc1 = dot3(rs1, R1);
            m3x = dM[2][0]; m3y = dM[2][1]; m3z = dM[2][2];
            // the reference does not write the AABB-augmented dL_dT back in this branch
            g0 = {gs[0], gs[1], gs[2]}; g1 = {gs[3], gs[4], gs[5]}; g2 = {gs[6], gs[7], gs[8]};
        }
        if (shs != nullptr) {
            const F3 pos = {means3D[3 * I], means3D[3 * I + 1], means3D[3 * I + 2]};
            const F3 dir_orig = pos - F3{campos[0], campos[1], campos[2]};
            const float len = __builtin_sqrtf(dot3(dir_orig, dir_orig));
            const F3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
            const float* shp = STAGE_SH ? creg : shs + I * M * 3;
            auto sh = [&](int k) { return F3{shp[3 * k], shp[3 * k + 1], shp[3 * k + 2]}; };
            const unsigned cm = g.clamped[i];
            F3 dRGB = {gs[15], gs[16], gs[17]};
            dRGB.x *= (cm & 1u) ? 0.f : 1.f; dRGB.y *= (cm & 2u) ? 0.f : 1.f; dRGB.z *= (cm & 4u) ? 0.f : 1.f;
            F3 ddx = {0, 0, 0}, ddy = {0, 0, 0}, ddz = {0, 0, 0};
            const float x = dir.x, y = dir.y, z = dir.z;
            float* out = sh_out;
            auto put = [&](int k, float c) { out[3 * k] = c * dRGB.x; out[3 * k + 1] = c * dRGB.y; out[3 * k + 2] = c * dRGB.z; };
            put(0, C0);
            if (D > 0) {
                put(1, -C1 * y); put(2, C1 * z); put(3, -C1 * x);
                ddx = -C1 * sh(3); ddy = -C1 * sh(1); ddz = C1 * sh(2);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    put(4, C2[0] * xy); put(5, C2[1] * yz); put(6, C2[2] * (2.f * zz - xx - yy));
                    put(7, C2[3] * xz); put(8, C2[4] * (xx - yy));
                    ddx = ddx + (C2[0] * y) * sh(4) + (C2[2] * 2.f * -x) * sh(6) + (C2[3] * z) * sh(7) + (C2[4] * 2.f * x) * sh(8);
                    ddy = ddy + (C2[0] * x) * sh(4) + (C2[1] * z) * sh(5) + (C2[2] * 2.f * -y) * sh(6) + (C2[4] * 2.f * -y) * sh(8);
                    ddz = ddz + (C2[1] * y) * sh(5) + (C2[2] * 2.f * 2.f * z) * sh(6) + (C2[3] * x) * sh(7);
                    if (D > 2) {
                        put(9, C3[0] * y * (3.f * xx - yy)); put(10, C3[1] * xy * z);
                        put(11, C3[2] * y * (4.f * zz - xx - yy));
                        put(12, C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                        put(13, C3[4] * x * (4.f * zz - xx - yy)); put(14, C3[5] * z * (xx - yy));
                        put(15, C3[6] * x * (xx - 3.f * yy));
                        ddx = ddx + ((C3[0] * sh(9)) * (3.f * 2.f * xy) + (C3[1] * sh(10)) * yz + (C3[2] * sh(11)) * (-2.f * xy) +
                                     (C3[3] * sh(12)) * (-3.f * 2.f * xz) + (C3[4] * sh(13)) * (-3.f * xx + 4.f * zz - yy) +
                                     (C3[5] * sh(14)) * (2.f * xz) + (C3[6] * sh(15)) * (3.f * (xx - yy)));
                        ddy = ddy + ((C3[0] * sh(9)) * (3.f * (xx - yy)) + (C3[1] * sh(10)) * xz +
                                     (C3[2] * sh(11)) * (-3.f * yy + 4.f * zz - xx) + (C3[3] * sh(12)) * (-3.f * 2.f * yz) +
                                     (C3[4] * sh(13)) * (-2.f * xy) + (C3[5] * sh(14)) * (-2.f * yz) +
                                     (C3[6] * sh(15)) * (-3.f * 2.f * xy));
                        ddz = ddz + ((C3[1] * sh(10)) * xy + (C3[2] * sh(11)) * (4.f * 2.f * yz) +
                                     (C3[3] * sh(12)) * (3.f * (2.f * zz - xx - yy)) + (C3[4] * sh(13)) * (4.f * 2.f * xz) +
                                     (C3[5] * sh(14)) * (xx - yy));
                    }
                }
            }
            const F3 dL_ddir = {dot3(ddx, dRGB), dot3(ddy, dRGB), dot3(ddz, dRGB)};
            const F3 dmean = dnorm_dv(dir_orig, dL_ddir);
            m3x += dmean.x; m3y += dmean.y; m3z += dmean.z;
        }
        const float depth = precomp ? tm_pre[9 * I + 8] : rec[8];
        m2x = (float)((double)hack2 * depth * 0.5 * (double)(float)Wd);
        m2y = (float)((double)hack5 * depth * 0.5 * (double)(float)Hd);
    } else {
        m2x = dmx; m2y = dmy;   // untouched accumulators (zero: an invisible Gaussian has no instances)
    }
    dL_dtransMat[9 * I + 0] = g0.x; dL_dtransMat[9 * I + 1] = g0.y; dL_dtransMat[9 * I + 2] = g0.z;
    dL_dtransMat[9 * I + 3] = g1.x; dL_dtransMat[9 * I + 4] = g1.y; dL_dtransMat[9 * I + 5] = g1.z;
    dL_dtransMat[9 * I + 6] = g2.x; dL_dtransMat[9 * I + 7] = g2.y; dL_dtransMat[9 * I + 8] = g2.z;
    dL_dmean2D[3 * I] = m2x; dL_dmean2D[3 * I + 1] = m2y; dL_dmean2D[3 * I + 2] = 0.0f;
    dL_dmean3D[3 * I] = m3x; dL_dmean3D[3 * I + 1] = m3y; dL_dmean3D[3 * I + 2] = m3z;
    dL_dscale[2 * I] = dsc0; dL_dscale[2 * I + 1] = dsc1;
    dL_drot[4 * I] = dq[0]; dL_drot[4 * I + 1] = dq[1]; dL_drot[4 * I + 2] = dq[2]; dL_drot[4 * I + 3] = dq[3];
    }
    if constexpr (STAGE_SH) {
        __syncthreads();
        const int i0 = blockIdx.x * 256;
        const int nq = min(256, P - i0) * 12;
        float4* dst = reinterpret_cast<float4*>(dL_dsh + (size_t)i0 * 48);
#pragma unroll
        for (int u = 0; u < 12; u++) {
            const int e = (int)threadIdx.x + u * 256;
            if (e < nq) { const int r = e / 12, k = e - r * 12; dst[e] = *reinterpret_cast<const float4*>(s_sh + r * SH_STRIDE + 4 * k); }
        }
    }
}

// ----------------------------------------------------------------------------
}  // namespace isr
#include "isr_backward_geo.hip"
namespace isr {

// (the launchers below return -2 with the message already in isr_last_error())
#define ISR_CHECK_LAUNCH_B(name)                                                                            \
    do {                                                                                                    \
        hipError_t e_ = hipGetLastError();                                                                  \
        if (e_ != hipSuccess) { fail(ISR_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); return -2; } \
        const char* m_ = debug_check(s);                                                                    \
        if (m_ != nullptr) { fail(ISR_EHIP, "[debug] kernel %s failed: %s", name, m_); return -2; }         \
    } while (0)

// ISR_SPARSE_BWD=0 disables the pixel-major kernel (A/B measurements)
static bool sparse_path_enabled() {
    static const bool on = [] { const char* e = getenv("ISR_SPARSE_BWD"); return !(e && e[0] == '0'); }();
    return on;
}

// ISR_GEO_SPLAT=0 keeps the pixel-major kernel for the geometry-only FAST backward (A/B measurements)
static bool geo_splat_enabled() {
    static const bool on = [] { const char* e = getenv("ISR_GEO_SPLAT"); return !(e && e[0] == '0'); }();
    return on;
}

static bool geo_heavy_first() {
    static const bool on = [] { const char* e = getenv("ISR_GEO_ORDER"); return !(e && e[0] == '0'); }();
    return on;
}

template <class Math>
static int launch_backward_t(int P, int D, int M, int64_t R, int ED, int W, int H, unsigned mask, const float* bg,
                             const float* means3D, const float* shs, const float* col_pre, const float* scales,
                             const float* rots, const float* tm_pre, const float* extras, const float* view,
                             const float* proj, const float* campos, float tan_fovx, float tan_fovy, const void* geom,
                             const void* binning, const void* image, const float* dC, const float* dO, const float* dE,
                             float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                             float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot,
                             float* dL_dextra, void* scratch, hipStream_t s) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    GeomView g = geom_view(const_cast<void*>(geom), P < 1 ? 1 : P);
    ImageView iv = image_view(const_cast<void*>(image), W, H);
    BinView bv = bin_view(const_cast<void*>(binning), R);
    const bool geomg = (mask & 2u) != 0, featg = (mask & 1u) != 0 && ED > 0;
    const int stride = row_floats(ED, (geomg ? 2u : 0u) | (featg ? 1u : 0u));
    const int geom_off = 0, feat_base = geomg ? GEOM_ROW : 0;
    float* partial = (float*)scratch;
    const unsigned eff_mask = (geomg ? 2u : 0u) | (featg ? 1u : 0u);
    uint8_t* flags = (uint8_t*)scratch + rows_bytes(R, ED, eff_mask);
    const int npass = n_passes(ED, eff_mask);
    if (P == 0) return 0;
    // FAST arithmetic, geometry only, no feature channel (the train.py step): splat-major kernel, a row per 8x8 block
    const bool geo_splat = Math::fast && geomg && !featg && ED == 0 && geo_splat_enabled();
    const int rpi = geo_splat ? rows_per_instance(ED, eff_mask) : 1;
    if (R > 0 && geo_splat) {
        if (hipMemsetAsync(flags, 0, (size_t)R * rpi, s) != hipSuccess) { fail(ISR_EHIP, "hipMemsetAsync failed in the backward"); return -2; }
        ProfScope ps_("k_render_bwd", s);
        // small grids (fewer blocks than ~2 rounds of the chip's wave slots): two waves per 8x8 block - the kernel's time there is
        // the longest list's, and half the pixels per wave halves it; large grids are throughput-bound: one wave per block
        static const int geo_two = [] { const char* e = getenv("ISR_GEO_TWO_WAVES_BELOW"); return e ? atoi(e) : 12000; }();
        unsigned long long* const counters = g_bwd_counters.exchange(nullptr);     // isr_backward_set_counters: this launch adds its work counters
#define ISR_GEO(NW_, ST_, HF_) hipLaunchKernelGGL((k_render_bwd_geo<NW_, ST_, HF_>), dim3(T * (HF_ ? 8 : 4)), dim3(64 * NW_), 0, s, W, H, gx, iv.tile_offset, \
                               bv.point_list, bv.box4, g.rec, col_pre, tm_pre, bg, iv.final_T, iv.n_contrib, dC, dO, g.point_offsets, g.rect, \
                               partial, flags, stride, geom_off, R, geo_heavy_first() ? iv.tile_order : (const uint32_t*)nullptr, bv.hit_mask, counters)
        // ISR_GEO_HALVES (default 1): a wave per 8x4 half of a block with the half's own culled list; 0: a wave (two on small grids) per block
        static const bool geo_halves = [] { const char* e = getenv("ISR_GEO_HALVES"); return !(e && e[0] == '0'); }();
        if (geo_halves) { if (counters) ISR_GEO(1, true, true); else ISR_GEO(1, false, true); }
        else if (T * 4 < geo_two) { if (counters) ISR_GEO(2, true, false); else ISR_GEO(2, false, false); }
        else { if (counters) ISR_GEO(1, true, false); else ISR_GEO(1, false, false); }
#undef ISR_GEO
        ISR_CHECK_LAUNCH_B("k_render_bwd_geo");
    } else if (R > 0) {
        if (hipMemsetAsync(flags, 0, (size_t)R * npass, s) != hipSuccess) { fail(ISR_EHIP, "hipMemsetAsync failed in the backward"); return -2; }
        int pass = 0;
        bool first = true;
        int ch = 0;
        do {
            ProfScope ps_("k_render_bwd", s);
            const bool do_geom = geomg && first, do_feat = featg;
            // feature-only pass: tiles with few live pixels are finished by the pixel-major kernel, the rest
            // (flagged in tile_mode) by the dense one
            const uint8_t* tmode = nullptr;
            if (!do_geom && do_feat && sparse_path_enabled()) {
                hipLaunchKernelGGL(k_bwd_live_pixels, dim3((gx + 3) / 4, gy), dim3(256), 0, s, W, H, ED, ch, gx, iv.n_contrib, dE,
                                   iv.live_list, iv.tile_mode);
                ISR_CHECK_LAUNCH_B("k_bwd_live_pixels");
                hipLaunchKernelGGL((k_render_bwd_sparse<Math, false>), dim3(T), dim3(64), 0, s, W, H, ED, ch, gx, iv.tile_offset,
                                   bv.point_list, bv.box4, g.rec, tm_pre, dE, g.point_offsets, g.rect, partial,
                                   flags + (size_t)pass * R, iv.live_list, iv.tile_mode, stride, feat_base + ch, R,
                                   (const uint32_t*)nullptr, (const long long*)nullptr, (const float*)nullptr,
                                   (const uint32_t*)nullptr, (const uint32_t*)nullptr, (unsigned long long*)nullptr);
                ISR_CHECK_LAUNCH_B("k_render_bwd_sparse");
                tmode = iv.tile_mode;
            }
#define ISR_GOB(GM, FT, Q)                                                                                           \
    hipLaunchKernelGGL((k_render_bwd<Math, GM, FT, Q>), dim3(T), dim3(256), 0, s, W, H, ED, ch, gx, iv.tile_offset,   \
                       bv.point_list, bv.box4, g.rec, col_pre, tm_pre, extras, bg, iv.final_T, iv.n_contrib, dC, dO, dE,      \
                       g.point_offsets, g.rect, partial, flags + (size_t)pass * R, tmode, stride, geom_off, feat_base + ch, R)
            // the geometry pass needs <feature_g, dL/dfeature(pix)> over ALL channels (dL/dalpha), whatever
            // chunk of dL/dextra it emits itself
            if (do_geom && ED > 32) { if (do_feat) ISR_GOB(true, true, 64); else ISR_GOB(true, false, 64); }
            else if (do_geom && ED > 0) { if (do_feat) ISR_GOB(true, true, 32); else ISR_GOB(true, false, 32); }
            else if (do_geom) ISR_GOB(true, false, 0);
            else if (do_feat) ISR_GOB(false, true, 0);
#undef ISR_GOB
            ISR_CHECK_LAUNCH_B("k_render_bwd");
            first = false;
            ch += 32;
            pass++;
        } while (featg && ch < ED);
    }
    if (featg) {
        const size_t total = (size_t)P * ((ED + 3) / 4);
        ProfScope ps_("k_reduce_rows", s);
        hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, ED, g.point_offsets,
                           g.tiles_touched, partial, flags, R, stride, feat_base, dL_dextra, ED, 0);
        ISR_CHECK_LAUNCH_B("k_reduce_rows");
    }
    if (geomg) {
        const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
        const int Wd = (int)(focal_x * tan_fovx * 2), Hd = (int)(focal_y * tan_fovy * 2);   // backward.cu:633-634
        ProfScope ps_("k_preprocess_bwd", s);
        if (shs != nullptr && dL_dsh != nullptr && M == 16)
            hipLaunchKernelGGL(k_preprocess_bwd<true>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, shs, scales, rots,
                               tm_pre, view, proj, campos, Wd, Hd, g, partial, flags, stride, geom_off, rpi, dL_dmean2D, dL_dnormal,
                               dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot);
        else
            hipLaunchKernelGGL(k_preprocess_bwd<false>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, shs, scales, rots,
                               tm_pre, view, proj, campos, Wd, Hd, g, partial, flags, stride, geom_off, rpi, dL_dmean2D, dL_dnormal,
                               dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot);
        ISR_CHECK_LAUNCH_B("k_preprocess_bwd");
    }
    return 0;
}

// Feature gradient from SAMPLED pixels: scratch = rows + flags (as backward_scratch_bytes(R, ED, GRAD_EXTRA)) followed by
// cnt[T], off[T+1], cursor[T], seg_rec[3 n] (u32).
size_t backward_sampled_scratch_bytes(int64_t R, int ED, int n, int W, int H) {
    const size_t T = (size_t)tiles_x(W) * tiles_y(H);
    return backward_scratch_bytes(R, ED, 1u) + align_up((3 * T + 1 + 3 * (size_t)(n > 0 ? n : 1)) * sizeof(uint32_t), 256) + 256;
}

// Two zero fills in one ordinary dispatch (16-byte stores; both regions start 16-byte aligned, their sizes are rounded up by the
// caller's layout): hipMemsetAsync is a blit kernel behind a barrier packet per call - two of them sat on the step's main chain.
__global__ __launch_bounds__(256) void k_zero2(uint4* __restrict__ a, size_t na16, uint4* __restrict__ b, size_t nb16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t k = i; k < na16; k += stride) a[k] = z;
    for (size_t k = i; k < nb16; k += stride) b[k] = z;
}

template <class Math>
static int launch_backward_sampled_t(int P, int64_t R, int ED, int W, int H, int n, const long long* pix,
                                     const float* rows_in, const float* tm_pre, const void* geom, const void* binning,
                                     const void* image, float* dL_dextra, int accumulate, void* scratch, hipStream_t s) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    GeomView g = geom_view(const_cast<void*>(geom), P < 1 ? 1 : P);
    ImageView iv = image_view(const_cast<void*>(image), W, H);
    BinView bv = bin_view(const_cast<void*>(binning), R);
    const int stride = row_floats(ED, 1u), npass = n_passes(ED, 1u);
    float* partial = (float*)scratch;
    uint8_t* flags = (uint8_t*)scratch + rows_bytes(R, ED, 1u);
    uint32_t* cnt = (uint32_t*)((char*)scratch + align_up(backward_scratch_bytes(R, ED, 1u), 256));
    uint32_t* off = cnt + T;
    uint32_t* cursor = off + T + 1;
    uint32_t* seg_idx = cursor + T;
    if (P == 0) return 0;
    // a per-Gaussian mask of flagged instances: for the caller that finishes the rows itself (launch_feature_rows_step) and
    // for the reduction below
    static const bool masked_reduce = [] { const char* e = getenv("ISR_MASKED_REDUCE"); return !(e && e[0] == '0'); }();
    unsigned long long* row_mask = (dL_dextra == nullptr || masked_reduce) ? g.row_mask : nullptr;
    static const bool own_fill = [] { const char* e = getenv("ISR_OWN_FILL"); return !(e && e[0] == '0'); }();
    const size_t flag_bytes = (size_t)((char*)(cnt + T) - (char*)flags);       // row flags and the per-tile sample counters are neighbours
    const bool fused_fill = own_fill && R > 0 && n > 0 && ((size_t)flags & 15) == 0 && (row_mask == nullptr || ((size_t)row_mask & 15) == 0);
    if (fused_fill) {
        // (the mask's P words and the flags' region both end inside 256-byte aligned carvings: rounding up to 16 bytes stays inside)
        const size_t na = row_mask != nullptr ? (sizeof(unsigned long long) * (size_t)P + 15) / 16 : 0, nb = (flag_bytes + 15) / 16;
        const unsigned blocks = (unsigned)std::min<size_t>(2048, (std::max(na, nb) + 255) / 256);
        hipLaunchKernelGGL(k_zero2, dim3(blocks ? blocks : 1), dim3(256), 0, s, reinterpret_cast<uint4*>(row_mask), na,
                           reinterpret_cast<uint4*>(flags), nb);
        ISR_CHECK_LAUNCH_B("k_zero2");
    } else if (row_mask != nullptr && hipMemsetAsync(row_mask, 0, sizeof(unsigned long long) * (size_t)P, s) != hipSuccess) { fail(ISR_EHIP, "hipMemsetAsync failed in the backward"); return -2; }
    if (R > 0 && n > 0) {
        if (!fused_fill && hipMemsetAsync(flags, 0, flag_bytes, s) != hipSuccess) { fail(ISR_EHIP, "hipMemsetAsync failed in the backward"); return -2; }
        ProfScope ps_("k_render_bwd", s);
        hipLaunchKernelGGL(k_sample_count, dim3((n + 255) / 256), dim3(256), 0, s, n, W, H, gx, pix, cnt);
        ISR_CHECK_LAUNCH_B("k_sample_count");
        hipLaunchKernelGGL(k_sample_scan, dim3(1), dim3(1024), 0, s, T, cnt, off, cursor);
        ISR_CHECK_LAUNCH_B("k_sample_scan");
        hipLaunchKernelGGL(k_sample_fill, dim3((n + 255) / 256), dim3(256), 0, s, n, W, H, gx, pix, off, cursor, iv.n_contrib, seg_idx);
        ISR_CHECK_LAUNCH_B("k_sample_fill");
        for (int pass = 0, ch = 0; ch < ED; pass++, ch += 32)
            hipLaunchKernelGGL((k_render_bwd_sparse<Math, true>), dim3(T), dim3(64), 0, s, W, H, ED, ch, gx, iv.tile_offset,
                               bv.point_list, bv.box4, g.rec, tm_pre, (const float*)nullptr, g.point_offsets, g.rect, partial,
                               flags + (size_t)pass * R, (const uint32_t*)nullptr, (const uint8_t*)nullptr, stride, ch, R,
                               iv.n_contrib, pix, rows_in, off, seg_idx, pass == 0 ? row_mask : (unsigned long long*)nullptr);
        ISR_CHECK_LAUNCH_B("k_render_bwd_sparse");
    } else if (R > 0) {
        if (hipMemsetAsync(flags, 0, align_up((size_t)R * npass, 256), s) != hipSuccess) { fail(ISR_EHIP, "hipMemsetAsync failed in the backward"); return -2; }
    }
    if (dL_dextra == nullptr) return 0;
    const size_t total = (size_t)P * ((ED + 3) / 4);
    ProfScope ps_("k_reduce_rows", s);
    if (row_mask != nullptr && R > 0 && n > 0)
        hipLaunchKernelGGL(k_reduce_rows_masked, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, ED, g.point_offsets,
                           g.tiles_touched, row_mask, partial, flags, R, stride, dL_dextra, ED, accumulate);
    else
        hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, ED, g.point_offsets,
                           g.tiles_touched, partial, flags, R, stride, 0, dL_dextra, ED, accumulate);
    ISR_CHECK_LAUNCH_B("k_reduce_rows");
    return 0;
}

int launch_feature_rows_step(int P, int row_begin, int row_count, int64_t R, int F, const void* geom, const void* rows_scratch, const float* gz_dense,
                             const float* gy, int* gy_slot, const float* gy_merged, float eps1, float eps2, float* x, float* grad_out, float lr_over_bc1,
                             float om1, float beta2, float om2, float inv_sqrt_bc2, float eps, float* m, float* v, float* y,
                             float* z, float* zscale, hipStream_t s) {
    if (P <= 0 || row_count <= 0) return 0;
    GeomView g = geom_view(const_cast<void*>(geom), P);
    const float* partial = (const float*)rows_scratch;
    const uint8_t* flags = rows_scratch ? (const uint8_t*)rows_scratch + rows_bytes(R, F, 1u) : nullptr;
    const int stride = row_floats(F, 1u);
    int q = F >> 2, lpr = 1;
    while (lpr < q) lpr <<= 1;
    const unsigned blocks = (unsigned)(((long long)row_count * lpr + 255) / 256);
    const int row_end = row_begin + row_count;
    ProfScope ps_("k_feature_rows_step", s);
    if (grad_out != nullptr)
        hipLaunchKernelGGL(k_feature_rows_step<false>, dim3(blocks), dim3(256), 0, s, row_begin, row_end, F, g.point_offsets,
                           g.tiles_touched, g.row_mask, partial, flags, R, stride, gz_dense, gy, gy_slot, gy_merged, eps1, eps2, x, grad_out, lr_over_bc1, om1, beta2, om2,
                           inv_sqrt_bc2, eps, m, v, y, z, zscale);
    else
        hipLaunchKernelGGL(k_feature_rows_step<true>, dim3(blocks), dim3(256), 0, s, row_begin, row_end, F, g.point_offsets,
                           g.tiles_touched, g.row_mask, partial, flags, R, stride, gz_dense, gy, gy_slot, gy_merged, eps1, eps2, x, grad_out, lr_over_bc1, om1, beta2, om2,
                           inv_sqrt_bc2, eps, m, v, y, z, zscale);
    ISR_CHECK_LAUNCH_B("k_feature_rows_step");
    return 0;
}

int launch_row_scales(int P, int F, float eps1, float eps2, const float* x, float* zscale, hipStream_t s) {
    if (P <= 0) return 0;
    int q = F >> 2, lpr = 1;
    while (lpr < q) lpr <<= 1;
    hipLaunchKernelGGL(k_row_scales, dim3((unsigned)(((long long)P * lpr + 255) / 256)), dim3(256), 0, s, P, F, eps1, eps2, x, zscale);
    ISR_CHECK_LAUNCH_B("k_row_scales");
    return 0;
}

int launch_backward_sampled(int P, int64_t R, int ED, int W, int H, int mode, int n, const long long* pix,
                            const float* rows_in, const float* tm_pre, const void* geom, const void* binning,
                            const void* image, float* dL_dextra, int accumulate, void* scratch, hipStream_t s) {
    if (mode == 0)
        return launch_backward_sampled_t<ExactMath>(P, R, ED, W, H, n, pix, rows_in, tm_pre, geom, binning, image, dL_dextra,
                                                    accumulate, scratch, s);
    return launch_backward_sampled_t<FastMath>(P, R, ED, W, H, n, pix, rows_in, tm_pre, geom, binning, image, dL_dextra,
                                               accumulate, scratch, s);
}

int launch_sample_gather(int n, int F, long long N, const float* map, const long long* pix, float* out, hipStream_t s) {
    if (n <= 0 || F <= 0) return 0;
    hipLaunchKernelGGL(k_sample_gather, dim3((unsigned)(((long long)n * F + 255) / 256)), dim3(256), 0, s, n, F, N, map, pix, out);
    ISR_CHECK_LAUNCH_B("k_sample_gather");
    return 0;
}

int launch_backward(int P, int D, int M, int64_t R, int ED, int W, int H, int mode, unsigned grad_mask, const float* bg,
                    const float* means3D, const float* shs, const float* col_pre, const float* scales, float,
                    const float* rots, const float* tm_pre, const float* extras, const float* view, const float* proj,
                    const float* campos, float tan_fovx, float tan_fovy, const int*, const void* geom,
                    const void* binning, const void* image, const float* dC, const float* dO, const float* dE,
                    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                    float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dextra,
                    void* scratch, size_t, hipStream_t stream) {
    if (mode == 0)
        return launch_backward_t<ExactMath>(P, D, M, R, ED, W, H, grad_mask, bg, means3D, shs, col_pre, scales, rots,
                                            tm_pre, extras, view, proj, campos, tan_fovx, tan_fovy, geom, binning, image,
                                            dC, dO, dE, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D,
                                            dL_dtransMat, dL_dsh, dL_dscale, dL_drot, dL_dextra, scratch, stream);
    return launch_backward_t<FastMath>(P, D, M, R, ED, W, H, grad_mask, bg, means3D, shs, col_pre, scales, rots, tm_pre,
                                       extras, view, proj, campos, tan_fovx, tan_fovy, geom, binning, image, dC, dO, dE,
                                       dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh,
                                       dL_dscale, dL_drot, dL_dextra, scratch, stream);
}

}  // namespace isr

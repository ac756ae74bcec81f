// K8 in FAST arithmetic (reference forward.cu:256-462): the per-tile front-to-back blend re-derived for the CDNA4
// issue ports instead of the reference's operation order.  The EXACT kernel (isr_forward.hip: k_render_fwd<ExactMath>)
// keeps the reference's order op for op; this one keeps its DECISIONS (which splats a pixel blends, in which order,
// where it stops) and its results to 1e-4, and spends as few vector and scalar instructions per (wave, splat) pair as
// the algebra allows - the kernel is issue-bound, not memory-bound:
//
//   * The ray-splat intersection  p = (px Tw - Tu) x (py Tw - Tv)  is AFFINE in the pixel: the px*py term is Tw x Tw = 0,
//     so  p = px (Tv x Tw) + py (Tw x Tu) + Tu x Tv.  The two cross products and p at the tile's origin are computed
//     once per staged (tile, splat) instance; a pixel needs 6 FMAs with its tile-relative coordinates (0..15: no large
//     pixel coordinate enters the per-pixel arithmetic) instead of 12 multiply-subtracts.
//   * The depth of the intersection,  s.x Tw.x + s.y Tw.y + Tw.z  with  s = p.xy / p.z,  is  <p, Tw> / p.z,  and
//     <p, Tw> = det(Tu, Tv, Tw) for every pixel (both cross products above are orthogonal to Tw): one multiply with
//     the reciprocal that s needs anyway.  Its reciprocal (for the distortion term's  near / depth)  is  p.z / det:
//     no second v_rcp_f32.
//   * Lane predicates never live in vector registers or in compiler-managed exec-mask stacks: every test is a v_cmp
//     that writes a 64-bit lane mask, masks are combined on the scalar unit, and the single predicated region per
//     splat takes its exec mask from the combined mask (inverse ballot).
//   * The feature channels accumulate on the matrix cores exactly as before (two splats per pair of
//     v_mfma_f32_32x32x2_f32); chunks narrower than 32 channels are zero-padded in LDS instead of taking a vector path.
#include "isr_common.hpp"
#include "isr_fast_pair.hpp"

namespace isr {

constexpr int FF_BATCH = 128;       // instances staged per round
constexpr int FF_RS = 24;           // staged floats per instance (six float4)
// staged record (the pairs fast_ray's packed instructions take, each an aligned register pair of a ds_read_b128):
//                 q0 = Tu.x, Tu.y, Tv.x, Tv.y
//                 q1 = Tw.x, Tw.y, Tu.z, Tv.z
//                 q2 = Tw.z, band.hi, cx, cy      alpha < 1/255 is certain for rho > band.hi
//                 q3 = det, opacity, band.lo, band.bw   ... and alpha >= 1/255 for rho <= band.lo (isr_fast_pair.hpp: guard bands)
//                 q4 = normal.xyz, -
//                 q5 = rgb, unused

// AUX = false ("feature-only forward", opt-in ISR_MODE_FEATURE_ONLY): colour, the seven auxiliary maps, the median
// contributor, the distortion moments and the tracer are not produced - only the feature map and the state the
// feature-only backward reads (final T, last contributor).
template <bool FEAT, bool STATS, bool AUX, bool ORDER>
__global__ __launch_bounds__(256, 4) void k_render_fwd_fast(
    int W, int H, int ED, int ch_base, int first_pass, int gx, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const float* __restrict__ rec, const float* __restrict__ cull,
    const float* __restrict__ col_pre, const float* __restrict__ tm_pre, const float* __restrict__ extras,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_others, float* __restrict__ out_extra, int32_t* __restrict__ tracer, long long tracer_cap,
    int32_t* __restrict__ tracer_count, uint32_t* __restrict__ box4, int64_t capacity, unsigned long long* __restrict__ stats,
    const uint32_t* __restrict__ tile_order) {
    constexpr int BATCH = FF_BATCH, RS = FF_RS, FCH = 32;
    __shared__ __attribute__((aligned(16))) float s_rec[BATCH * RS];
    __shared__ __attribute__((aligned(16))) float s_feat[FEAT ? BATCH * FCH : 4];
    __shared__ int s_id[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_box[BATCH];
    __shared__ __attribute__((aligned(16))) float4 s_diag[BATCH];
    constexpr int WCAP = 128;               // tracer pairs per wave region (see k_render_fwd)
    __shared__ int s_trace[4 * 2 * WCAP];
    int wcnt = 0;

    // small grids (a 779x519 view is 1.6 workgroups per slot of the chip): longest lists first; large ones keep the row-major
    // order (heaviest-first costs 4 % at 1080p: measured)
    const int tile = ORDER ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;      // (a template flag: the large-grid build is unchanged)
    const int tx = tile % gx, ty = tile / gx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lxi = (wv & 1) * 8 + (lane & 7), lyi = (wv >> 1) * 8 + (lane >> 3);       // tile-relative pixel
    const unsigned px = tx * TILE + lxi, py = ty * TILE + lyi;
    const bool inside = px < (unsigned)W && py < (unsigned)H;
    const size_t N = (size_t)W * H;
    const size_t pix = (size_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const v2f pq = {pxf, pyf};
    const float X0 = (float)(tx * TILE), Y0 = (float)(ty * TILE);

    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int nfeat = FEAT ? min(FCH, ED - ch_base) : 0;

    unsigned long long m_done = __ballot(!inside);      // lanes that have stopped (or lie outside the image)
    float T = 1.0f;
    unsigned last_contributor = 0, median_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
    float w_pend = 0.0f, f_pend = 0.0f;
    bool pending = false;                    // wave-uniform
    unsigned st_cull = 0, st_eval = 0, st_blend = 0, st_lanes = 0, st_merge = 0, st_sub = 0;      // STATS only
    bool st_skip_next = false;
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    // The distortion map  sum_i w_i (m_i^2 A_i - 2 m_i M1_i + M2_i)  (A, M1, M2: the sums over the splats in front of i) is
    // 1/2 sum_ij w_i w_j (m_i - m_j)^2 = A M2 - M1^2 of the FINAL sums: the loop keeps only the two moments, of  m - m_ref
    // with a workgroup-uniform m_ref near the tile's depth (the identity is shift-invariant; the shift keeps the final
    // subtraction from cancelling), and the epilogue forms the map and un-shifts the moments the backward reads.
    float m_ref = 0.9f, mshift = mscale - 0.9f;       // set from the first staged splat's depth (scalar registers)
    const float bx0 = X0 + (float)((wv & 1) * 8), bx1 = bx0 + 7.0f;
    const float by0 = Y0 + (float)((wv >> 1) * 8), by1 = by0 + 7.0f;

    auto flush_trace = [&]() {
        const int n = wcnt;
        if (n > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
            const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y}, Tw = {b.z, b.w, c.x};
            const float opa = d.z;
            const FastBand fb = fast_band(opa, e.w);
            float4* s4 = reinterpret_cast<float4*>(s_rec + t * RS);
            s4[0] = make_float4(Tu.x, Tu.y, Tv.x, Tv.y);
            s4[1] = make_float4(Tw.x, Tw.y, Tu.z, Tv.z);
            s4[2] = make_float4(Tw.z, fb.hi, c.y, c.z);
            s4[3] = make_float4(fast_det(Tu, Tv, Tw, c.y, c.z), opa, fb.lo, fb.bw);
            s4[4] = make_float4(c.w, d.x, d.y, 0.0f);
            s4[5] = make_float4(d.w, e.x, e.y, 0.0f);
            const float4 cb = reinterpret_cast<const float4*>(cull + (size_t)id * CULL_STRIDE)[0];      // per-Gaussian bounds from K1
            s_box[t] = cb;
            s_diag[t] = reinterpret_cast<const float4*>(cull + (size_t)id * CULL_STRIDE)[1];
            if (first_pass) box4[base + t] = pack_box4(cb, X0, Y0);
        }
        __syncthreads();
        if (FEAT) {
            if ((ED & 3) == 0 && nfeat == FCH) {
                for (int e = threadIdx.x; e < nb * (FCH / 4); e += 256) {
                    const int inst = e / (FCH / 4), part = e - inst * (FCH / 4);
                    reinterpret_cast<float4*>(s_feat)[e] =
                        *reinterpret_cast<const float4*>(extras + (size_t)s_id[inst] * ED + ch_base + part * 4);
                }
            } else {                    // narrow or ragged chunk: zero-padded to 32 channels
                for (int e = threadIdx.x; e < nb * FCH; e += 256) {
                    const int inst = e / FCH, c = e - inst * FCH;
                    s_feat[e] = c < nfeat ? extras[(size_t)s_id[inst] * ED + ch_base + c] : 0.0f;
                }
            }
            __syncthreads();
        }
        if (AUX && base == r0) {
            const float z0 = s_rec[8];                   // Tw.z of the tile's nearest splat
            const float mr = fminf(1.0f, fmaxf(0.0f, mscale - mscale * NEAR_N * __builtin_amdgcn_rcpf(z0)));
            m_ref = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mr)));
            mshift = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mscale - mr)));
        }
        const unsigned cbase = (unsigned)(base - r0) + 1u;
        // ---- walk the batch: each wave visits only the splats whose cull bounds meet its 8x8 pixel block
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int jj = c0 + lane;
            bool hit = false;
            if (jj < nb) {
                const float4 bb = s_box[jj], dg = s_diag[jj];
                hit = !(bb.x > bx1) && !(bb.y < bx0) && !(bb.z > by1) && !(bb.w < by0) &&
                      !(dg.x > bx1 + by1) && !(dg.y < bx0 + by0) && !(dg.z > bx1 - by0) && !(dg.w < bx0 - by1);
            }
            unsigned long long m = __ballot(hit);
            if (STATS) st_cull += (unsigned)min(64, nb - c0);
            while (m != 0ull) {
                if (m_done == ~0ull) break;
                const int j = c0 + __builtin_ctzll(m);
                m &= m - 1ull;
                if (STATS) {
                    st_eval++;
                    // how often could this splat and the next one of the wave's list share ONE evaluation (their alpha >= 1/255
                    // bounds touch disjoint pixels of the block)?  greedy pairs
                    if (st_skip_next) st_skip_next = false;
                    else if (m != 0ull) {
                        const int j2 = c0 + __builtin_ctzll(m);
                        const float fx = pxf, fy = pyf;
                        const float4 b1 = s_box[j], d1 = s_diag[j], b2 = s_box[j2], d2 = s_diag[j2];
                        const bool in1 = fx >= b1.x && fx <= b1.y && fy >= b1.z && fy <= b1.w && fx + fy >= d1.x && fx + fy <= d1.y && fx - fy >= d1.z && fx - fy <= d1.w;
                        const bool in2 = fx >= b2.x && fx <= b2.y && fy >= b2.z && fy <= b2.w && fx + fy >= d2.x && fx + fy <= d2.y && fx - fy >= d2.z && fx - fy <= d2.w;
                        if (__ballot(in1 && in2) == 0ull) { st_merge++; st_skip_next = true; }
                    }
                }
                const float4* q = reinterpret_cast<const float4*>(s_rec + j * RS);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                // the pair's arithmetic is isr_fast_pair.hpp's, shared with the FAST backward kernels: both passes take the
                // same decisions on the same pair, bit for bit
                FastRay fr = fast_ray(pq, (v2f){q0.x, q0.y}, (v2f){q0.z, q0.w}, (v2f){q1.x, q1.y}, (v2f){q1.z, q1.w}, q2.x, (v2f){q2.z, q2.w});
                // beyond band.hi alpha < 1/255 is certain: when that holds for the whole wave nothing else is needed
                const unsigned long long m_near = __ballot(fr.rho <= q2.y) & ~m_done;
                if (m_near == 0ull) continue;
                const float4 q3 = q[3], q4 = q[4];
                FastHit fh = fast_hit(fr, q3.x, q2.x, q3.y);
                // the same decisions as k_render_fwd_fast_w below (guard bands: isr_fast_pair.hpp)
                const unsigned long long m_cand = m_near;          // (depth >= near_n is certain outside the bands: fast_pass)
                const unsigned long long m_band = (m_near & __ballot(fr.rho > q3.z)) | (m_cand & __ballot(fabsf(fr.rho3d - fr.rho2d) <= q3.w));
                unsigned long long m_pass = m_cand & ~m_band;
                if (m_band != 0ull) {
                    FastRay er; FastHit eh;
                    // (the staged record holds the splat's raw rows: no second trip to memory)
                    const bool ep = exact_pair(pxf, pyf, {q0.x, q0.y, q1.z}, {q0.z, q0.w, q1.w}, {q1.x, q1.y, q2.x}, q2.z, q2.w, q3.y, er, eh);
                    fast_take((m_band >> lane) & 1ull, er, eh, fr, fh);
                    m_pass |= m_band & __ballot(ep);
                }
                const float depth = fh.depth, alpha = fh.alpha;
                const float test_T = __builtin_fmaf(-T, alpha, T);
                const unsigned long long m_stop = m_pass & __ballot(test_T < 0.0001f);
                m_done |= m_stop;
                const unsigned long long m_ok = m_pass & ~m_stop;
                if (m_ok == 0ull) continue;
                if (STATS) {
                    st_blend++;
                    st_lanes += (unsigned)__popcll(m_ok);
                    // how many of the block's four 4x4 sub-blocks hold a blending pixel (what a finer decomposition would visit)
                    const int sub = ((lyi >> 2) & 1) * 2 + ((lxi >> 2) & 1);
                    const bool okl = (m_ok >> lane) & 1ull;
                    for (int k = 0; k < 4; k++) st_sub += __ballot(okl && sub == k) != 0ull ? 1u : 0u;
                }
                float w_lane = 0.0f;
                if (__builtin_amdgcn_inverse_ballot_w64(m_ok)) {
                    const float w = alpha * T;
                    w_lane = w;
                    const unsigned contributor = cbase + (unsigned)j;
                    if (AUX && first_pass) {
                        const float4 q5 = q[5];
                        const float inv_depth = __builtin_amdgcn_rcpf(depth);
                        const float m_ = __builtin_fmaf(-(mscale * NEAR_N), inv_depth, mshift);      // m - m_ref
                        const float mw = m_ * w;
                        D = __builtin_fmaf(depth, w, D);
                        M1 += mw;
                        M2 = __builtin_fmaf(m_, mw, M2);
                        if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                        N0 = __builtin_fmaf(q4.x, w, N0); N1 = __builtin_fmaf(q4.y, w, N1); N2 = __builtin_fmaf(q4.z, w, N2);
                        C0 = __builtin_fmaf(q5.x, w, C0); C1 = __builtin_fmaf(q5.y, w, C1); C2 = __builtin_fmaf(q5.z, w, C2);
                    }
                    T = test_T;
                    last_contributor = contributor;
                }
                if (AUX && tracer != nullptr && first_pass) {
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
                if constexpr (FEAT) {
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
    if constexpr (FEAT) {
        if (pending) {          // odd number of contributing splats: pair the last one with a zero
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w_pend), 0u, false, false);
            const float a = lane < 32 ? f_pend : 0.0f;
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[0]), accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[1]), accB, 0, 0, 0);
        }
    }
    if (AUX && tracer != nullptr && first_pass) flush_trace();
    if (STATS) {
        if (lane == 0 && first_pass) {
            atomicAdd(stats + 0, (unsigned long long)st_cull);
            atomicAdd(stats + 1, (unsigned long long)st_eval);
            atomicAdd(stats + 2, (unsigned long long)st_blend);
            atomicAdd(stats + 3, (unsigned long long)st_lanes);
            atomicAdd(stats + 4, (unsigned long long)st_merge);
            atomicAdd(stats + 5, (unsigned long long)st_sub);
        }
    }
    if (!AUX && inside && first_pass) {
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
    }
    if (AUX && inside && first_pass) {
        const float A_ = 1.0f - T;
        distortion = __builtin_fmaf(A_, M2, -(M1 * M1));
        final_T[pix] = T;
        final_T[pix + N] = __builtin_fmaf(m_ref, A_, M1);
        final_T[pix + 2 * N] = __builtin_fmaf(m_ref, __builtin_fmaf(m_ref, A_, M1 + M1), M2);
        n_contrib[pix] = last_contributor;
        n_contrib[pix + N] = median_contributor;
        out_color[pix] = __builtin_fmaf(T, bg[0], C0);
        out_color[N + pix] = __builtin_fmaf(T, bg[1], C1);
        out_color[2 * N + pix] = __builtin_fmaf(T, bg[2], C2);
        out_others[pix] = D;
        out_others[N + pix] = 1 - T;
        out_others[2 * N + pix] = N0;
        out_others[3 * N + pix] = N1;
        out_others[4 * N + pix] = N2;
        out_others[5 * N + pix] = median_depth;
        out_others[6 * N + pix] = distortion;
    }
    if constexpr (FEAT) {
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
// The same blend with the four 8x8 blocks of a tile DECOUPLED: one wave = one workgroup = one 8x8 pixel block that culls,
// stages and walks its own hit list.  In the tile-wide kernel above a wave waits in workgroup barriers (its block's hit count
// differs from its neighbours' in every 128-instance round) and in the staging phase between them for about as long as it walks
// hits (cycle counters, docs/history/DESIGN_rounds_1-3.md 9.11); here there is no barrier at all:
//   scan:   64 entries of the tile's list per step: k_pack_hits' 64-bit word for (chunk, block) says which of them meet the
//           block (one scalar load; ids by one coalesced load); hits are appended (id, position in the tile's list) to a ring
//           in LDS until 32 are pending or the list ends;
//   stage:  lane = hit: record gather + the per-(tile, splat) precompute for HITS only (30 % of the instances);
//   walk:   the staged hits in order - no hit-mask iteration, every visited splat meets the block.
// A block stops scanning when its own 64 pixels are done, not when the tile's 256 are.  Workgroup v lands on XCD v % 8: the
// four blocks of tile t are slots 4 (t / 8) .. + 3 of XCD t % 8, i.e. they share one L2 and are dispatched together.
#ifndef ISR_FW_HITS
#define ISR_FW_HITS 32
#endif
#ifndef ISR_FW_WAVES
#define ISR_FW_WAVES 4
#endif
#ifndef ISR_FW_WAVES_NOFEAT
#define ISR_FW_WAVES_NOFEAT 5
#endif
#ifndef ISR_FW_HITS_NOFEAT
#define ISR_FW_HITS_NOFEAT 16
#endif
constexpr int FW_HITS = ISR_FW_HITS;         // hits staged per round (A/B builds: tools/build_variant.sh ... -DISR_FW_HITS=16 -DISR_FW_WAVES=5)
constexpr int FW_RING = 128;        // pending (id, position) pairs

// NC = 32-channel chunks of the feature blended per pass: 1 (four waves per SIMD), or 2 for F >= 64 - two more accumulator
// tiles and 8 KB of feature rows per wave cost the fourth wave (+27 % per pass), one pass instead of two is still 0.7x.
// CN ("colour and normal on the matrix cores"; FEAT, AUX, NC == 1, a single pass of at most 24 channels in whole float4s - the
// reference's default seg_feat_dim = 16, arguments/__init__.py:65): rows 24..29 of the zero-padded A operand carry the staged
// splat's rgb and normal, so the two MFMAs that blend the feature blend them too and the six v_fma per blending evaluation (and the
// registers of their sums) are gone; the epilogue reads the six sums out of the accumulator tiles.  The MFMA rounds as the
// VALU chain does not (its k slots are summed in order, unfused): colour and normal then differ from the CN-less kernel in their
// last bits - within FAST's 1e-4 like everything else, and the same bits in every FAST mode.
template <bool FEAT, bool STATS, bool AUX, bool ORDER, int NC, bool CN = false>
// Waves per SIMD: 4 with a feature channel (two 16-register accumulator tiles: 128 registers; at 96 they spill and the kernel is 5x
// slower - measured), 5 without one (the train.py step: 81 registers, 16 hits per round: -3 % at C2), 3 for the wide pass.
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(NC == 1 ? (FEAT ? ISR_FW_WAVES : ISR_FW_WAVES_NOFEAT) : 3,
                                                                                    NC == 1 ? (FEAT ? ISR_FW_WAVES : ISR_FW_WAVES_NOFEAT) : 3))) void k_render_fwd_fast_w(
    int W, int H, int ED, int ch_base, int first_pass, int gx, int tiles, const uint32_t* __restrict__ tile_offset,
    const uint32_t* __restrict__ point_list, const float* __restrict__ rec, const float* __restrict__ cull,
    const float* __restrict__ col_pre, const float* __restrict__ tm_pre, const float* __restrict__ extras,
    const float* __restrict__ xscale /* [P,2] or null: the staged feature row is (extras[id] * xscale[id][0]) * xscale[id][1] */,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_others, float* __restrict__ out_extra, int32_t* __restrict__ tracer, long long tracer_cap,
    int32_t* __restrict__ tracer_count, const unsigned long long* __restrict__ hit_mask, int64_t capacity,
    unsigned long long* __restrict__ stats, const uint32_t* __restrict__ tile_order) {
#ifndef ISR_FW_HITS2
#define ISR_FW_HITS2 24
#endif
    static_assert(!CN || (FEAT && AUX && NC == 1), "CN rides on the feature MFMAs of a single 32-channel pass with the aux outputs");
    constexpr int RS = FF_RS, FCH = 32 * NC, NH = NC == 1 ? (FEAT ? FW_HITS : ISR_FW_HITS_NOFEAT) : ISR_FW_HITS2;      // (64 channels: 24 hits per round = 11.25 KB of LDS per wave; 16 / 24 / 32: 1.10 / 1.05 / 1.21 ms at C5)
    __shared__ __attribute__((aligned(16))) float s_rec[NH * RS];
    __shared__ __attribute__((aligned(16))) float s_feat[FEAT ? NH * FCH : 4];
    __shared__ __attribute__((aligned(8))) int2 s_ring[FW_RING];
#ifndef ISR_WCAP
#define ISR_WCAP 512
#endif
    // tracer pairs buffered per wave, ONE atomic on the list's counter per flush (that counter bounded the kernel while a wave
    // flushed every 64 pairs: docs/history/DESIGN_rounds_1-3.md 9.11).  A pair is packed into 32 bits - pixel of the block << 26 | gaussian (the launcher
    // sends scenes of more than 2^26 Gaussians to the tile-wide kernel) - so 2 KB hold 512: a block has ~214, its wave flushes
    // once, at its end, and that flush's atomic is issued BEFORE the output maps are stored and consumed after them.
    constexpr int WCAP = ISR_WCAP;
    __shared__ unsigned s_trace[WCAP];
    int wcnt = 0;

    const int v = (int)blockIdx.x, kk = v >> 3;
    const int slot = (kk >> 2) * 8 + (v & 7), sub = kk & 3;
    if (slot >= tiles) return;
    const int tile = ORDER ? (int)tile_order[slot] : slot;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int lxi = (sub & 1) * 8 + (lane & 7), lyi = (sub >> 1) * 8 + (lane >> 3);       // tile-relative pixel
    const unsigned px = tx * TILE + lxi, py = ty * TILE + lyi;
    const bool inside = px < (unsigned)W && py < (unsigned)H;
    const size_t N = (size_t)W * H;
    const size_t pix = (size_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const v2f pq = {pxf, pyf};


    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    const int nfeat = FEAT ? min(FCH, ED - ch_base) : 0;
    if (FEAT && nfeat < FCH && (ED & 3) == 0 && (nfeat & 3) == 0) {      // narrow chunk: the channels the staging never writes
        for (int e = lane; e < NH * FCH / 4; e += 64) reinterpret_cast<float4*>(s_feat)[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }

    unsigned long long m_done = __ballot(!inside);
    float T = 1.0f;
    unsigned last_contributor = 0, median_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA, accC = accA, accD = accA;     // C, D: NC == 2
    float w_pend = 0.0f, f_pend = 0.0f, f_pend2 = 0.0f;
    bool pending = false;
    unsigned st_cull = 0, st_eval = 0, st_blend = 0, st_lanes = 0, st_sub = 0, st_slow = 0, st_viol = 0;
    // (STATS) what a finer decomposition would walk: near / blending evaluations per 8x4 half and per 4x4 quad of the block
    unsigned st_hn[2] = {0, 0}, st_qn[4] = {0, 0, 0, 0}, st_hb[2] = {0, 0}, st_qb[4] = {0, 0, 0, 0};
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    // the distortion moments are kept relative to m_ref, the mapped depth of the TILE's nearest splat (as in the tile-wide kernel:
    // every block of a tile uses the same shift, and the two kernels produce the same bits)
    float m_ref = 0.9f, mshift = mscale - 0.9f;
    if (AUX && len > 0) {
        const int id0 = (int)point_list[r0];
        const float z0 = tm_pre != nullptr ? tm_pre[9 * (size_t)id0 + 8] : rec[(size_t)id0 * REC + 8];
        const float mr = fminf(1.0f, fmaxf(0.0f, mscale - mscale * NEAR_N * __builtin_amdgcn_rcpf(z0)));
        m_ref = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mr)));
        mshift = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mscale - mr)));
    }
    const size_t mask0 = hit_mask_word(r0, tile, 0) + (size_t)sub;

    const int blk_x0 = tx * TILE + (sub & 1) * 8, blk_y0 = ty * TILE + (sub >> 1) * 8;
    auto store_trace = [&](int gb, int n) {
        for (int e = lane; e < n; e += 64)
            if (gb + e < tracer_cap) {
                const unsigned pk = s_trace[e];
                const int p6 = (int)(pk >> 26);
                *reinterpret_cast<int2*>(tracer + 2 * (size_t)(gb + e)) =
                    make_int2((int)(pk & 0x3ffffffu), W * (blk_y0 + (p6 >> 3)) + blk_x0 + (p6 & 7));
            }
    };
    auto flush_trace = [&]() {
        const int n = wcnt;
        if (n > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int gb = 0;
            if (lane == 0) gb = atomicAdd(tracer_count, n) + 1;        // counter starts at -1
            gb = __builtin_amdgcn_readfirstlane(gb);
            store_trace(gb, n);
            __builtin_amdgcn_wave_barrier();
            wcnt = 0;
        }
    };

    int scan = 0, head = 0, pend = 0;                 // wave-uniform
    int nid = lane < len ? (int)point_list[r0 + lane] : 0;
    while (true) {
        // ---- scan: append this block's hits among the next instances of the tile's list (k_pack_hits' word per chunk and block)
#pragma clang loop unroll(disable)
        while (pend < NH && scan < len) {
            const unsigned long long m = hit_mask[mask0 + (size_t)(scan >> 6) * HM_WORDS];
            const int i = scan + lane;
            const int id = nid;
            if (i + 64 < len) nid = (int)point_list[r0 + i + 64];
            if ((m >> lane) & 1ull) {
                const int at = pend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                s_ring[(head + at) & (FW_RING - 1)] = make_int2(id, i + 1);
            }
            pend += __popcll(m);
            if (STATS) st_cull += (unsigned)min(64, len - scan);
            scan += 64;
        }
        if (pend == 0) break;
        const int nh = min(pend, NH);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- stage: lane = hit
        if (lane < nh) {
            const int2 hp = s_ring[(head + lane) & (FW_RING - 1)];
            const int id = hp.x;
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            const v4f_* r4 = reinterpret_cast<const v4f_*>(rec + (size_t)id * REC);
            v4f_ ra = r4[0], rb = r4[1], rc_ = r4[2], rd = r4[3], re = r4[4];
            // five whole 16-byte requests (left alone the compiler trims the unused words and issues six: +2 %.  Three requests
            // per lane with the record split over lanes l and l + 32 and v_permlane32_swap measured +1.5 % again.)
            asm volatile("" : "+v"(ra), "+v"(rb), "+v"(rc_), "+v"(rd), "+v"(re));
            float4 a = make_float4(ra.x, ra.y, ra.z, ra.w), b = make_float4(rb.x, rb.y, rb.z, rb.w), c = make_float4(rc_.x, rc_.y, rc_.z, rc_.w);
            float4 d = make_float4(rd.x, rd.y, rd.z, rd.w), e = make_float4(re.x, re.y, re.z, re.w);
            if (tm_pre != nullptr) {
                const float* tp = tm_pre + 9 * (size_t)id;
                a = make_float4(tp[0], tp[1], tp[2], tp[3]);
                b = make_float4(tp[4], tp[5], tp[6], tp[7]);
                c.x = tp[8];
            }
            if (col_pre != nullptr) {
                d.w = col_pre[3 * (size_t)id]; e.x = col_pre[3 * (size_t)id + 1]; e.y = col_pre[3 * (size_t)id + 2];
            }
            const F3 Tu = {a.x, a.y, a.z}, Tv = {a.w, b.x, b.y}, Tw = {b.z, b.w, c.x};
            const float opa = d.z;
            const FastBand fb = fast_band(opa, e.w);
            float4* s4 = reinterpret_cast<float4*>(s_rec + lane * RS);
            s4[0] = make_float4(Tu.x, Tu.y, Tv.x, Tv.y);
            s4[1] = make_float4(Tw.x, Tw.y, Tu.z, Tv.z);
            s4[2] = make_float4(Tw.z, fb.hi, c.y, c.z);
            s4[3] = make_float4(fast_det(Tu, Tv, Tw, c.y, c.z), opa, fb.lo, fb.bw);
            s4[4] = make_float4(c.w, d.x, d.y, 0.0f);
            s4[5] = make_float4(d.w, e.x, e.y, __int_as_float(hp.y));          // .w: position in the tile's list (1-based)
            if (CN) {            // channels 24..29 of the hit's feature row: rgb, normal (the narrow staging below writes [0, nfeat <= 24) only)
                float* fr_ = s_feat + lane * FCH + 24;
                *reinterpret_cast<float4*>(fr_) = make_float4(d.w, e.x, e.y, c.w);
                *reinterpret_cast<float2*>(fr_ + 4) = make_float2(d.x, d.y);
            }
        }
        if (FEAT) {
            if ((ED & 3) == 0 && nfeat == FCH) {
                // NH * 8 float4 over 64 lanes: four requests in flight per lane, then the four LDS writes
                float4 fv[NH * (FCH / 4) / 64];
#pragma unroll
                for (int k = 0; k < NH * (FCH / 4) / 64; k++) {
                    const int e = lane + 64 * k;
                    const int inst = e / (FCH / 4), part = e - inst * (FCH / 4);
                    const int id = s_ring[(head + inst) & (FW_RING - 1)].x;           // (a stale ring entry beyond nh: a valid id, unused)
                    fv[k] = e < nh * (FCH / 4) ? *reinterpret_cast<const float4*>(extras + (size_t)id * ED + ch_base + part * 4)
                                               : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (xscale != nullptr && e < nh * (FCH / 4)) {          // (beyond nh the ring entry is stale: no load through it)
                        // the trainer hands over the RAW feature and its two row-normalisation factors instead of a normalised copy
                        // of the table (one [P,F] stream less per step): the same two multiplies, in the same order, as the
                        // kernel that would have written that copy (k_feature_rows_step: y = x q1, z = y q2) - the same bits
                        const float2 sc = *reinterpret_cast<const float2*>(xscale + 2 * (size_t)id);
                        fv[k] = make_float4((fv[k].x * sc.x) * sc.y, (fv[k].y * sc.x) * sc.y, (fv[k].z * sc.x) * sc.y, (fv[k].w * sc.x) * sc.y);
                    }
                }
#pragma unroll
                for (int k = 0; k < NH * (FCH / 4) / 64; k++) {
                    const int e = lane + 64 * k;
                    if (e < nh * (FCH / 4)) reinterpret_cast<float4*>(s_feat)[e] = fv[k];
                }
            } else if ((ED & 3) == 0 && (nfeat & 3) == 0) {
                // a narrow chunk of whole float4s (the reference's default seg_feat_dim = 16, arguments/__init__.py:65): 16-byte
                // requests for the channels that exist; the rows' upper channels were zeroed once, before the first round
                const int q4 = nfeat >> 2;
                for (int e = lane; e < nh * q4; e += 64) {
                    const int inst = e / q4, part = e - inst * q4;
                    const int id = s_ring[(head + inst) & (FW_RING - 1)].x;
                    float4 f4 = *reinterpret_cast<const float4*>(extras + (size_t)id * ED + ch_base + part * 4);
                    if (xscale != nullptr) {
                        const float2 sc = *reinterpret_cast<const float2*>(xscale + 2 * (size_t)id);
                        f4 = make_float4((f4.x * sc.x) * sc.y, (f4.y * sc.x) * sc.y, (f4.z * sc.x) * sc.y, (f4.w * sc.x) * sc.y);
                    }
                    reinterpret_cast<float4*>(s_feat)[inst * (FCH / 4) + part] = f4;
                }
            } else {                    // ragged chunk: zero-padded to 32 channels
                for (int e = lane; e < nh * FCH; e += 64) {
                    const int inst = e / FCH, c = e - inst * FCH;
                    const int id = s_ring[(head + inst) & (FW_RING - 1)].x;
                    float fvs = c < nfeat ? extras[(size_t)id * ED + ch_base + c] : 0.0f;
                    if (xscale != nullptr) fvs = (fvs * xscale[2 * (size_t)id]) * xscale[2 * (size_t)id + 1];
                    s_feat[e] = fvs;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- walk
#pragma clang loop unroll(disable)
        for (int j = 0; j < nh && m_done != ~0ull; j++) {
            if (STATS) st_eval++;
            typedef float v4f __attribute__((ext_vector_type(4)));
            const float4* q = reinterpret_cast<const float4*>(s_rec + j * RS);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2];
            v4f q3v = reinterpret_cast<const v4f*>(q)[3];
            FastRay fr = fast_ray(pq, (v2f){q0.x, q0.y}, (v2f){q0.z, q0.w}, (v2f){q1.x, q1.y}, (v2f){q1.z, q1.w}, q2.x, (v2f){q2.z, q2.w});
            asm volatile("" : "+v"(q3v), "+v"(fr.rho));          // q3 is requested with q0..q2, not after the first branch
            const unsigned long long m_near = __ballot(fr.rho <= q2.y) & ~m_done;
            if (!STATS && m_near == 0ull) continue;
            const float4 q3 = make_float4(q3v.x, q3v.y, q3v.z, q3v.w);
            v4f q4e = reinterpret_cast<const v4f*>(q)[4], q5e = reinterpret_cast<const v4f*>(q)[5];
            float f_early = FEAT ? s_feat[j * FCH + (lane & 31)] : 0.0f;
            float f_early2 = (FEAT && NC == 2) ? s_feat[j * FCH + 32 + (lane & 31)] : 0.0f;
            FastHit fh = fast_hit(fr, q3.x, q2.x, q3.y);
            if (NC == 2) asm volatile("" : "+v"(f_early2));
            asm volatile("" : "+v"(q4e), "+v"(q5e), "+v"(f_early), "+v"(fh.alpha));      // ... and the blend's operands before ITS branch
            // decisions (isr_fast_pair.hpp): a near pair outside the guard bands certainly has alpha >= 1/255 and FAST's branch
            // and near-plane test are EXACT's; a pair inside them is re-evaluated with EXACT's instruction sequence
            const unsigned long long m_cand = m_near;          // (depth >= near_n is certain outside the bands: fast_pass)
            const unsigned long long m_band = (m_near & __ballot(fr.rho > q3.z)) | (m_cand & __ballot(fabsf(fr.rho3d - fr.rho2d) <= q3.w));
            unsigned long long m_pass = m_cand & ~m_band;
            if (STATS || m_band != 0ull) {
                FastRay er; FastHit eh;          // (the staged record holds the splat's raw rows: no second trip to memory)
                const bool ep = exact_pair(pxf, pyf, {q0.x, q0.y, q1.z}, {q0.z, q0.w, q1.w}, {q1.x, q1.y, q2.x}, q2.z, q2.w, q3.y, er, eh);
                if (STATS) {
                    if (m_band != 0ull) st_slow++;
                    // outside the band FAST's decisions must be EXACT's (near pairs: pass and branch; far pairs: skipped)
                    const bool fpass = (m_pass >> lane) & 1ull, inb = (m_band >> lane) & 1ull, alive = !((m_done >> lane) & 1ull);
                    const bool bad = alive && !inb && (fpass != ep || (ep && fh.use3d != eh.use3d));
                    st_viol += (unsigned)__popcll(__ballot(bad));
                }
                fast_take((m_band >> lane) & 1ull, er, eh, fr, fh);
                m_pass |= m_band & __ballot(ep);
                if (STATS) {
                    constexpr unsigned long long QM[4] = {0x0f0f0f0full, 0xf0f0f0f0ull, 0x0f0f0f0f00000000ull, 0xf0f0f0f000000000ull};
                    for (int k = 0; k < 2; k++) st_hn[k] += (m_near & (0xffffffffull << (32 * k))) != 0ull;
                    for (int k = 0; k < 4; k++) st_qn[k] += (m_near & QM[k]) != 0ull;
                }
                if (STATS && m_near == 0ull) continue;
            }
            const bool use3d = fh.use3d;
            const float depth = fh.depth, alpha = fh.alpha;
            const float test_T = __builtin_fmaf(-T, alpha, T);
            const unsigned long long m_stop = m_pass & __ballot(test_T < 0.0001f);
            m_done |= m_stop;
            const unsigned long long m_ok = m_pass & ~m_stop;
            if (m_ok == 0ull) continue;
            if (STATS) {
                st_blend++;
                st_lanes += (unsigned)__popcll(m_ok);
                const int sb = ((lyi >> 2) & 1) * 2 + ((lxi >> 2) & 1);
                const bool okl = (m_ok >> lane) & 1ull;
                for (int k = 0; k < 4; k++) st_sub += __ballot(okl && sb == k) != 0ull ? 1u : 0u;
                constexpr unsigned long long QM[4] = {0x0f0f0f0full, 0xf0f0f0f0ull, 0x0f0f0f0f00000000ull, 0xf0f0f0f000000000ull};
                for (int k = 0; k < 2; k++) st_hb[k] += (m_ok & (0xffffffffull << (32 * k))) != 0ull;
                for (int k = 0; k < 4; k++) st_qb[k] += (m_ok & QM[k]) != 0ull;
            }
            float w_lane = 0.0f;
            if (__builtin_amdgcn_inverse_ballot_w64(m_ok)) {
                const float w = alpha * T;
                w_lane = w;
                const unsigned contributor = __float_as_uint(q5e.w);
                if (AUX && first_pass) {
                    const float inv_depth = __builtin_amdgcn_rcpf(depth);
                    const float m_ = __builtin_fmaf(-(mscale * NEAR_N), inv_depth, mshift);      // m - m_ref
                    const float mw = m_ * w;
                    D = __builtin_fmaf(depth, w, D);
                    M1 += mw;
                    M2 = __builtin_fmaf(m_, mw, M2);
                    if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                    if (!CN) {
                        N0 = __builtin_fmaf(q4e.x, w, N0); N1 = __builtin_fmaf(q4e.y, w, N1); N2 = __builtin_fmaf(q4e.z, w, N2);
                        C0 = __builtin_fmaf(q5e.x, w, C0); C1 = __builtin_fmaf(q5e.y, w, C1); C2 = __builtin_fmaf(q5e.z, w, C2);
                    }
                }
                T = test_T;
                last_contributor = contributor;
            }
            (void)use3d;
            if (AUX && tracer != nullptr && first_pass) {
                const unsigned long long m_tr = __ballot(w_lane >= 0.1f);     // (double)w > 0.1  <=>  w >= 0.1f
                if (m_tr != 0ull) {
                    if (w_lane >= 0.1f) {
                        const int slot_ = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_tr >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((unsigned)m_tr, 0u));
                        s_trace[slot_] = ((unsigned)lane << 26) | (unsigned)s_ring[(head + j) & (FW_RING - 1)].x;
                    }
                    wcnt += __popcll(m_tr);
                    if (wcnt > WCAP - 64) flush_trace();
                }
            }
            if constexpr (FEAT) {
                const float f_lane = f_early;
                if (!pending) { w_pend = w_lane; f_pend = f_lane; f_pend2 = f_early2; pending = true; }
                else {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w_pend), __float_as_uint(w_lane), false, false);
                    const float a = lane < 32 ? f_pend : f_lane;
                    accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[0]), accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[1]), accB, 0, 0, 0);
                    if (NC == 2) {
                        const float a2 = lane < 32 ? f_pend2 : f_early2;
                        accC = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, __uint_as_float(sw[0]), accC, 0, 0, 0);
                        accD = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, __uint_as_float(sw[1]), accD, 0, 0, 0);
                    }
                    pending = false;
                }
            }
        }
        if (m_done == ~0ull) break;
        head = (head + nh) & (FW_RING - 1);
        pend -= nh;
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (FEAT) {
        if (pending) {          // odd number of contributing splats: pair the last one with a zero
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w_pend), 0u, false, false);
            const float a = lane < 32 ? f_pend : 0.0f;
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[0]), accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(sw[1]), accB, 0, 0, 0);
            if (NC == 2) {
                const float a2 = lane < 32 ? f_pend2 : 0.0f;
                accC = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, __uint_as_float(sw[0]), accC, 0, 0, 0);
                accD = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, __uint_as_float(sw[1]), accD, 0, 0, 0);
            }
        }
    }
    // the last flush: its atomic goes out here, the output maps are stored while it is under way, the pairs follow at the end
    int gb_last = 0;
    const int n_last = (AUX && tracer != nullptr && first_pass) ? wcnt : 0;
    if (n_last > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) gb_last = atomicAdd(tracer_count, n_last) + 1;
    }
    if (STATS) {
        if (lane == 0 && first_pass) {
            atomicAdd(stats + 0, (unsigned long long)st_cull);
            atomicAdd(stats + 1, (unsigned long long)st_eval);
            atomicAdd(stats + 2, (unsigned long long)st_blend);
            atomicAdd(stats + 3, (unsigned long long)st_lanes);
            atomicAdd(stats + 5, (unsigned long long)st_sub);
            atomicAdd(stats + 6, (unsigned long long)st_slow);      // (wave, splat) evaluations that took the EXACT path
            atomicAdd(stats + 7, (unsigned long long)st_viol);      // pairs outside the guard bands whose decision differs from EXACT's
            // 8..15: iterations a wave would need if each 8x4 half / 4x4 quad walked its own sub-list (the longest sub-list) and the
            // sub-lists' total length, counting the evaluations with a lane inside band.hi (8..11) or with a blending lane (12..15)
            atomicAdd(stats + 8, (unsigned long long)max(st_hn[0], st_hn[1]));
            atomicAdd(stats + 9, (unsigned long long)max(max(st_qn[0], st_qn[1]), max(st_qn[2], st_qn[3])));
            atomicAdd(stats + 10, (unsigned long long)(st_hn[0] + st_hn[1]));
            atomicAdd(stats + 11, (unsigned long long)(st_qn[0] + st_qn[1] + st_qn[2] + st_qn[3]));
            atomicAdd(stats + 12, (unsigned long long)max(st_hb[0], st_hb[1]));
            atomicAdd(stats + 13, (unsigned long long)max(max(st_qb[0], st_qb[1]), max(st_qb[2], st_qb[3])));
            atomicAdd(stats + 14, (unsigned long long)(st_hb[0] + st_hb[1]));
            atomicAdd(stats + 15, (unsigned long long)(st_qb[0] + st_qb[1] + st_qb[2] + st_qb[3]));
        }
    }
    if (!AUX && inside && first_pass) {
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
    }
    if (AUX && inside && first_pass) {
        const float A_ = 1.0f - T;
        distortion = __builtin_fmaf(A_, M2, -(M1 * M1));
        final_T[pix] = T;
        final_T[pix + N] = __builtin_fmaf(m_ref, A_, M1);
        final_T[pix + 2 * N] = __builtin_fmaf(m_ref, __builtin_fmaf(m_ref, A_, M1 + M1), M2);
        n_contrib[pix] = last_contributor;
        n_contrib[pix + N] = median_contributor;
        if (!CN) {
            out_color[pix] = __builtin_fmaf(T, bg[0], C0);
            out_color[N + pix] = __builtin_fmaf(T, bg[1], C1);
            out_color[2 * N + pix] = __builtin_fmaf(T, bg[2], C2);
            out_others[2 * N + pix] = N0;
            out_others[3 * N + pix] = N1;
            out_others[4 * N + pix] = N2;
        }
        out_others[pix] = D;
        out_others[N + pix] = 1 - T;
        out_others[5 * N + pix] = median_depth;
        out_others[6 * N + pix] = distortion;
    }
    if constexpr (CN) {
        // rows 24..29 of the accumulator tiles: channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) - lanes 0..31 hold rgb and normal.x of
        // pixel grp * 32 + lane in elements 12..15, lanes 32..63 normal.y, normal.z of pixel grp * 32 + (lane - 32) in elements 12, 13.
        // The background term needs the pixel's own T: lane p holds pixel p's.
        const float T_other = __shfl_xor(T, 32);
        if (first_pass) {
#pragma unroll
            for (int grp = 0; grp < 2; grp++) {
                const int p = grp * 32 + (lane & 31);
                const unsigned qx = tx * TILE + (sub & 1) * 8 + (p & 7), qy = ty * TILE + (sub >> 1) * 8 + (p >> 3);
                if (qx < (unsigned)W && qy < (unsigned)H) {
                    const size_t qp = (size_t)W * qy + qx;
                    const float a8 = grp == 0 ? accA[12] : accB[12], a9 = grp == 0 ? accA[13] : accB[13];
                    const float a10 = grp == 0 ? accA[14] : accB[14], a11 = grp == 0 ? accA[15] : accB[15];
                    if (lane < 32) {
                        const float Tp = grp == 0 ? T : T_other;
                        out_color[qp] = __builtin_fmaf(Tp, bg[0], a8);
                        out_color[N + qp] = __builtin_fmaf(Tp, bg[1], a9);
                        out_color[2 * N + qp] = __builtin_fmaf(Tp, bg[2], a10);
                        out_others[2 * N + qp] = a11;
                    } else {
                        out_others[3 * N + qp] = a8;
                        out_others[4 * N + qp] = a9;
                    }
                }
            }
        }
    }
    if constexpr (FEAT) {
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int p = grp * 32 + (lane & 31);                       // pixel (wave lane numbering) held by this lane
            const unsigned qx = tx * TILE + (sub & 1) * 8 + (p & 7), qy = ty * TILE + (sub >> 1) * 8 + (p >> 3);
            if (qx < (unsigned)W && qy < (unsigned)H) {
                const size_t qp = (size_t)W * qy + qx;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (ch < nfeat) out_extra[(size_t)(ch_base + ch) * N + qp] = grp == 0 ? accA[r] : accB[r];
                    if (NC == 2 && ch + 32 < nfeat) out_extra[(size_t)(ch_base + 32 + ch) * N + qp] = grp == 0 ? accC[r] : accD[r];
                }
            }
        }
    }
    if (n_last > 0) store_trace(__builtin_amdgcn_readfirstlane(gb_last), n_last);
}

}  // namespace isr

// K9 for the train.py step (reference backward.cu:143-466) in FAST arithmetic, geometry only (no feature channel), dense
// upstream gradient - the kernel BASELINE config 2 lives on.  "Splat-major": a lane owns a SPLAT, not a pixel.
//
// The pixel-major kernel (isr_backward.hip: k_render_bwd<GEOM>) evaluates a splat on the 64 pixels of a wave and then has
// to sum twelve non-linear terms and the blend weights over those pixels - a packed butterfly, LDS partials of four
// waves, two MFMA phases, three workgroup barriers per 32 instances; ~220 vector instructions per (wave, splat) pair at
// 3 waves/SIMD.  Here one wave owns an 8x8 pixel block of a tile and walks the block's culled splat list BACK TO FRONT, 64
// splats at a time, a lane per splat, pixel after pixel:
//   * what the reference carries along the list for one pixel - the transmittance in front of a splat and the
//     "accumulated behind it" recurrences - are PREFIX SCANS over the lanes: a product scan of (1 - alpha) and a sum scan
//     of  w S  (S = the pixel's combined upstream gradient, one scalar for colour, depth, alpha, normal and distortion
//     weight: they all follow the same linear recurrence).  DPP row shifts + row broadcasts, no LDS;
//   * every gradient of a (block, splat) pair accumulates in the registers of the lane that owns the splat while the 64
//     pixels go by: no cross-lane reduction, no partials, no barrier, and the finished 18-value row is stored once.
// Each of a tile's four blocks writes its own row per splat (slot 4 s + block of the partial-row scratch; only pairs that
// were evaluated are flagged - 1.17 rows per tile instance on the C3 scene), and k_preprocess_bwd sums a Gaussian's rows
// in slot order as before: still no atomics, bit-reproducible.
#include "isr_common.hpp"
#include "isr_fast_pair.hpp"

namespace isr {

constexpr int GEO_SEG = 256;        // tile-list entries culled per round (four per lane)
constexpr int GEO_QCAP = 512;       // queue capacity (entries that touch the block, waiting for their chunk of 64)
constexpr int GEO_ROWS = 8;         // partial rows per (tile, Gaussian) instance: one per 8x4 half of each 8x8 block (row 2 block + half; the
                                    // kernels that keep a block together write the first of its pair)

__device__ __forceinline__ float wave_scan_add(float v) {      // inclusive sum scan over the 64 lanes
    v += dpp_fetch<0x111, 0xF>(v, 0.0f);     // row_shr:1
    v += dpp_fetch<0x112, 0xF>(v, 0.0f);     // row_shr:2
    v += dpp_fetch<0x114, 0xF>(v, 0.0f);     // row_shr:4
    v += dpp_fetch<0x118, 0xF>(v, 0.0f);     // row_shr:8
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 - as adds that write only those rows (the builtin form is a
    // move into a zeroed register plus an add: two instructions more per step, in a loop that is bound by instruction issue);
    // s_nop 1 = the two wait states between a vector write of a register and its read through DPP
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}

// NW = waves per 8x8 block: 1, or 2 on small grids (4 measured: no further gain at 779x519, 0.304 against 0.298 ms) (a 779x519 view is 6 468 blocks for 4 096 wave slots: its time is the
// longest list's) - the waves of a workgroup take 8 / NW pixel rows of the block each, walk the same chunks of 64 splats in
// lockstep, and hand their 21 partial sums per splat to wave 0 through LDS before the row is stored.
// STATS (isr_backward_set_counters; bench.py's lane-utilisation table of this kernel): u64 counters, per wave and summed -
//   [0] chunks of 64 splat slots walked   [1] slots that hold a splat   [2] (chunk, pixel row) pairs   [3] ... of them reached by a splat
//   [4] (chunk, pixel) iterations with a candidate lane   [5] candidate lanes in them   [6] ... with a blending lane   [7] blending lanes
//   [8] chunks whose always-EXACT splat was pre-evaluated   [9] partial rows stored
// HALF (NW = 1): a workgroup = one wave = one 8x4 HALF of a block with its own culled list (k_pack_hits' per-half words): a splat
// reaches ~60 % of the halves of the blocks it reaches, so the chunks of 64 - whose every lane is carried through every pixel
// iteration whether its splat is near the pixel or not - are fewer per pixel.  The two halves of a block write a row each.
template <int NW, bool STATS = false, bool HALF = false>
__global__ __launch_bounds__(64 * NW, 4) void k_render_bwd_geo(
    int W, int H, int gx, const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ point_list,
    const uint32_t* __restrict__ box4, const float* __restrict__ rec, const float* __restrict__ col_pre,
    const float* __restrict__ tm_pre, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dC, const float* __restrict__ dO,
    const uint32_t* __restrict__ point_offsets, const Rect16* __restrict__ rects, float* __restrict__ partial,
    uint8_t* __restrict__ row_flags, int row_stride, int geom_off, int64_t capacity, const uint32_t* __restrict__ tile_order,
    const unsigned long long* __restrict__ hit_mask, unsigned long long* __restrict__ stats = nullptr) {
    static_assert(!HALF || NW == 1, "a half is one wave");
    constexpr int NP = HALF ? 32 : 64 / NW;         // pixels per wave
    unsigned st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) float s_pix_all[64 * 16];
    __shared__ int s_q[GEO_QCAP];
    __shared__ __attribute__((aligned(16))) float s_acc[NW > 1 ? (NW - 1) * 64 * 24 : 4];
    __shared__ unsigned s_last[4];
    // per pixel: (transmittance behind, sum of w S behind) the splats walked so far - read by every lane (one broadcast LDS read),
    // rewritten by the chunk's front-most lane; as registers of "lane p" they cost two v_readlane, two moves and two selects per
    // (chunk, pixel) in a loop that is bound by vector-instruction issue
    __shared__ __attribute__((aligned(8))) float2 s_carry_all[64];
    // A splat whose every pair takes EXACT's instruction sequence (band = +inf: edge-on, horizon inside its footprint; ~0.4 % of
    // the splats, i.e. one in a quarter of all chunks of 64) would make the whole wave run that sequence - for one lane - at every
    // pixel of its box.  It is evaluated here instead, once per chunk, a lane per PIXEL of the block, and handed to its lane
    // through LDS: 12 floats per pixel (dx dy rz sx | sy depth G alpha | pass, use3d).
    __shared__ __attribute__((aligned(16))) float s_ex_all[64 * 12];

    constexpr int SH = HALF ? 3 : 2;      // workgroups per tile: 8 halves or 4 blocks
    const int tile = tile_order != nullptr ? (int)tile_order[blockIdx.x >> SH] : (int)(blockIdx.x >> SH);
    const int blk = HALF ? (int)((blockIdx.x >> 1) & 3) : (int)(blockIdx.x & 3);
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform: scalar row arithmetic)
    const int hf = HALF ? (int)(blockIdx.x & 1) : 0;                               // which half of the block (HALF)
    const int bxo = (blk & 1) * 8, byo = (blk >> 1) * 8 + (HALF ? hf * 4 : wv * (8 / NW));       // origin of this wave's pixels inside the tile
    const int mask_word = HALF ? 4 + 2 * blk + hf : blk;                           // k_pack_hits' word of this wave's pixels
    const int row_id = HALF ? 2 * blk + hf : 2 * blk;                              // partial row of the instance this wave (workgroup) writes
    float* const s_pix = s_pix_all + wv * NP * 16;
    float* const s_ex = s_ex_all + wv * NP * 12;
    float2* const s_carry = s_carry_all + wv * NP;
    auto block_sync = [&]() { if constexpr (NW > 1) __syncthreads(); else wave_lds_sync(); };
    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    if (len <= 0) return;
    const size_t N = (size_t)W * H;
    // ---- the block's pixels: lane p loads pixel p, LDS hands them to everybody (wave-uniform reads in the pixel loop)
    unsigned mylast = 0u;
    bool reg_here = false;
    {
        const unsigned px = tx * TILE + bxo + (lane & 7), py = ty * TILE + byo + (lane >> 3);
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = 0.0f;
        if (lane < NP && px < (unsigned)W && py < (unsigned)H) {
            const size_t pix = (size_t)W * py + px;
            if (dC) { v[0] = dC[pix]; v[1] = dC[N + pix]; v[2] = dC[2 * N + pix]; }
            if (dO) {
#pragma unroll
                for (int k = 0; k < 7; k++) v[3 + k] = dO[(size_t)k * N + pix];
            }
            bool any = false;
#pragma unroll
            for (int k = 0; k < 10; k++) any = any || (v[k] != 0.0f);
            v[10] = final_T[pix]; v[11] = final_T[pix + N]; v[12] = final_T[pix + 2 * N];
            mylast = any ? n_contrib[pix] : 0u;          // a pixel without upstream gradient contributes exact zeros
            reg_here = v[9] != 0.0f;
            v[13] = __uint_as_float(mylast);
            v[14] = __uint_as_float(n_contrib[pix + N]);
            v[15] = (bg[0] * v[0] + bg[1] * v[1]) + bg[2] * v[2];
        }
        if (lane < NP) {
            float4* dst = reinterpret_cast<float4*>(s_pix + lane * 16);
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            dst[2] = make_float4(v[8], v[9], v[10], v[11]);
            dst[3] = make_float4(v[12], v[13], v[14], v[15]);
        }
    }
    unsigned block_last = mylast;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) block_last = max(block_last, (unsigned)__shfl_xor((int)block_last, o));
    if constexpr (NW > 1) {             // the two waves walk the same list prefix: the deeper of their last contributors
        if (lane == 0) s_last[wv] = block_last;
        __syncthreads();
        block_last = s_last[0];
#pragma unroll
        for (int w = 1; w < NW; w++) block_last = max(block_last, s_last[w]);
    }
    if (block_last == 0u) return;
    block_sync();
    const int len_eff = min(len, (int)block_last);
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);
    const bool any_reg = __ballot(reg_here) != 0ull;       // (lambda_dist = 0: no pixel carries a distortion gradient)
    if (lane < NP) s_carry[lane] = make_float2(s_pix[lane * 16 + 10], 0.0f);
    wave_lds_sync();

    int n_q = 0;                    // queued entries (uniform); s_q holds tile-list positions in DESCENDING order
    int seg_hi = len_eff;           // entries [0, seg_hi) are still to be culled
    while (seg_hi > 0 || n_q > 0) {
        while (seg_hi > 0 && n_q <= GEO_QCAP - GEO_SEG) {
            const int seg_lo = max(0, seg_hi - GEO_SEG);
#pragma unroll
            for (int u = 0; u < GEO_SEG / 64; u++) {
                const int i = seg_hi - 1 - (64 * u + lane);
                bool hit = false;
                if (i >= seg_lo) {
                    // k_pack_hits' word for the entry's 64-chunk and this block: the bounding OCTAGON of the alpha >= 1/255 region
                    // against the block (the packed box alone keeps ~1/6 more entries; each costs a lane in a 64-pixel loop)
                    const unsigned long long m = hit_mask[hit_mask_word(r0, tile, i >> 6) + mask_word];
                    hit = (m >> (i & 63)) & 1ull;
                }
                const unsigned long long b = __ballot(hit);
                if (hit) s_q[n_q + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = i;
                n_q += __popcll(b);
            }
            seg_hi = seg_lo;
        }
        block_sync();
        const bool drained = seg_hi == 0;
        int done = 0;
        while (n_q - done >= 64 || (drained && done < n_q)) {
            const int li = done + lane < n_q ? s_q[done + lane] : -1;       // this lane's splat: position in the tile's list
            done += 64;
            if (STATS) { st[0]++; st[1] += (unsigned)__popcll(__ballot(li >= 0)); st[2] += NP / 8; }
            // ---- the splat (one per lane; lane order = back to front)
            F3 Tu = {0, 0, 0}, Tv = {0, 0, 0}, Tw = {0, 0, 1}, nrm = {0, 0, 0}, col = {0, 0, 0};
            float cx = 0, cy = 0, opa = 0, band = 0;
            int bxl = 127, bxh = -128, byl = 127, byh = -128;           // empty box: a lane without a splat touches no pixel
            unsigned slot = 0;
            if (li >= 0) {
                const int id = (int)point_list[r0 + li];
                const unsigned bx = box4[r0 + li];
                bxl = (int)(signed char)(bx & 255u); bxh = (int)(signed char)((bx >> 8) & 255u);
                byl = (int)(signed char)((bx >> 16) & 255u); byh = (int)(signed char)(bx >> 24);
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
                Tu = {a.x, a.y, a.z}; Tv = {a.w, b.x, b.y}; Tw = {b.z, b.w, c.x};
                cx = c.y; cy = c.z; nrm = {c.w, d.x, d.y}; opa = d.z; col = {d.w, e.x, e.y};
                band = e.w;                     // K1's guard band (isr_fast_pair.hpp)
                const Rect16 rc = rects[id];
                slot = point_offsets[id] + (unsigned)(ty - rc.y0) * (unsigned)(rc.x1 - rc.x0) + (unsigned)(tx - rc.x0);
            }
            // The ray-splat intersection p = (px Tw - Tu) x (py Tw - Tv) is affine in the pixel: p = lx A + ly B + C with tile-relative
            // pixel coordinates (A = Tv x Tw, B = Tw x Tu, C = p at the tile origin), and so is its adjoint: dL/dA = sum lx dL/dp, dL/dB = sum ly dL/dp, dL/dC = sum dL/dp.  The pixel
            // loop accumulates those nine sums (instead of two cross products and nine FMAs per pair for dL/dTu, dL/dTv, dL/dTw);
            // they are turned into the gradient of the three rows once per (block, splat), after the loop.
            const float det = fast_det(Tu, Tv, Tw, cx, cy);
            const FastBand fb = fast_band(opa, band);
            const unsigned long long m_forced = __ballot(li >= 0 && !(band < __builtin_inff()));
            const bool pre = __popcll(m_forced) == 1;              // (two or more in one chunk: the per-pair path below, as before)
            const int f_lane = pre ? __builtin_ctzll(m_forced) : 0;
            if (pre) {
                auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), f_lane)); };
                const F3 fTu = {bc(Tu.x), bc(Tu.y), bc(Tu.z)}, fTv = {bc(Tv.x), bc(Tv.y), bc(Tv.z)}, fTw = {bc(Tw.x), bc(Tw.y), bc(Tw.z)};
                FastRay er; FastHit eh;
                const bool ep = exact_pair(tile_x0 + (float)(bxo + (lane & 7)), tile_y0 + (float)(byo + ((lane & (NP - 1)) >> 3)), fTu, fTv, fTw, bc(cx), bc(cy),
                                           bc(opa), er, eh);
                wave_lds_sync();                                   // (the previous chunk's readers - this wave - are done)
                if (lane < NP) {
                    float4* o = reinterpret_cast<float4*>(s_ex + lane * 12);
                    o[0] = make_float4(er.dx, er.dy, er.rz, er.sx);
                    o[1] = make_float4(er.sy, eh.depth, eh.G, eh.alpha);
                    o[2] = make_float4(ep ? 1.0f : 0.0f, eh.use3d ? 1.0f : 0.0f, 0.0f, 0.0f);
                }
                wave_lds_sync();
            }
            const bool pre_mine = pre && lane == f_lane;
            if (STATS && pre) st[8]++;
            // The sums a lane keeps for its splat, as the register PAIRS the packed fp32 instructions take (v_pk_fma_f32 / v_pk_add_f32:
            // one instruction for two sums, the same IEEE operation per component); the pairing follows the operands that are
            // already neighbours - (dx, dy), (sx, sy), dL/dp.xy, the even-aligned halves of the pixel's float4s - so that no
            // register has to be moved to form a pair.
            v2f P01 = {0, 0}, X01 = {0, 0}, Y01 = {0, 0};      // sum dL/dp.xy, sum lx dL/dp.xy, sum ly dL/dp.xy
            v2f XY2 = {0, 0};                                  // (sum lx dL/dp.z, sum ly dL/dp.z)
            v2f PZ2 = {0, 0};                                  // (sum dL/dp.z, sum dL/dz)
            v2f Z01 = {0, 0};                                  // sum dL/dz (sx, sy)
            v2f C01 = {0, 0};                                  // dL/dcentre (low-pass branch)
            v2f N12 = {0, 0}, RG = {0, 0};                     // dL/dnormal.yz, dL/dcolour.rg
            float aN0 = 0, aO = 0, aB = 0;
            bool touched = false;
            for (int prow = 0; prow < NP / 8; prow++) {
              // a pixel row: l = py Tw - Tv, the row's float coordinate and "does my box reach this row" once per row and splat
              const int lyi = byo + prow;
              const bool row_in = li >= 0 && byl <= lyi && byh >= lyi;
              if (__ballot(row_in) == 0ull) continue;             // no splat of the chunk reaches the row: its eight pixels at once
              if (STATS) st[3]++;
              const float ly = (float)lyi;
              const float pyf_row = tile_y0 + ly;
              const FastHalf lrow = fast_l(pyf_row, Tv, Tw);
              for (int pc = 0; pc < 8; pc++) {
                const int p = prow * 8 + pc;
                const float4* pq = reinterpret_cast<const float4*>(s_pix + p * 16);
                const float4 q3 = pq[3];
                const unsigned last_p = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q3.y));
                const int lxi = bxo + pc;
                const bool cand = row_in && (unsigned)li < last_p && bxl <= lxi && bxh >= lxi;
                if (__ballot(cand) != 0ull) {       // (one latch for the loop: `continue`s here made the compiler rotate the accumulators)
                    if (STATS) { st[4]++; st[5] += (unsigned)__popcll(__ballot(cand)); }
                    const float lx = (float)lxi;
                    // the forward's own evaluation of the pair (isr_fast_pair.hpp; EXACT inside the guard bands): same decisions, bit for bit
                    FastRay fr; FastHit fh;
                    bool pass = fast_pair_lane(Tu, Tv, Tw, cx, cy, opa, det, fb, tile_x0 + lx, pyf_row, fr, fh, pre_mine, &lrow);
                    if (pre_mine && cand) {                        // the chunk's always-EXACT splat: evaluated above
                        const float4* e4 = reinterpret_cast<const float4*>(s_ex + p * 12);
                        const float4 e0 = e4[0], e1 = e4[1], e2 = e4[2];
                        fr.dx = e0.x; fr.dy = e0.y; fr.rz = e0.z; fr.sx = e0.w; fr.sy = e1.x;
                        fh.depth = e1.y; fh.G = e1.z; fh.alpha = e1.w;
                        pass = e2.x != 0.0f; fh.use3d = e2.y != 0.0f;
                    }
                    const float dx = fr.dx, dy = fr.dy, rz = fr.rz, sx = fr.sx, sy = fr.sy;
                    const bool use3d = fh.use3d;
                    const float c_d = fh.depth, G = fh.G, alpha = fh.alpha;
                    const bool act = cand && pass;
                    if (__ballot(act) != 0ull) {
                        if (STATS) { st[6]++; st[7] += (unsigned)__popcll(__ballot(act)); }
                        const float4 q0 = pq[0], q1 = pq[1], q2 = pq[2];
                        // q0 = dC.rgb, d_depth   q1 = d_accum, dN.xyz   q2 = d_median, d_reg, T_final, final_D   q3 = final_D2, last, median, bg_dot
                        const float2 cc = s_carry[p];
                        const float om = act ? 1.0f - alpha : 1.0f;
                        // transmittance in front of this lane's splat: carry / prod_{lanes <= l} (1 - alpha), the product
                        // as a SUM scan of logarithms (6 fused DPP adds + v_log + v_exp instead of 18 instructions)
                        const float lg = __builtin_amdgcn_logf(om);                      // log2, <= 0
                        const float Linc = wave_scan_add(lg);
                        const float cT = cc.x, cR = cc.y;
                        const float Tb = cT * __builtin_amdgcn_exp2f(-Linc);
                        const float w = act ? alpha * Tb : 0.0f;
                        const float T_final = q2.z, final_A = 1.0f - q2.z, final_D = q2.w, final_D2 = q3.x;
                        const float dL_dreg = q2.y;
                        const bool has_reg = any_reg && __builtin_amdgcn_readfirstlane((int)__float_as_uint(dL_dreg)) != 0;    // uniform
                        float dL_dweight = 0.0f, dz_reg = 0.0f;
                        if (has_reg) {          // the pixel's distortion gradient (lambda_dist = 0: never)
                            const float inv_cd = __builtin_amdgcn_rcpf(c_d);
                            const float m_d = mscale * (1.0f - NEAR_N * inv_cd);
                            dL_dweight = (__builtin_fmaf(m_d * m_d, final_A, final_D2) - 2.0f * m_d * final_D) * dL_dreg;
                            dz_reg = 2.0f * w * (m_d * final_A - final_D) * dL_dreg * ((FAR_N * NEAR_N / (FAR_N - NEAR_N)) * inv_cd * inv_cd);
                        }
                        float S = col.x * q0.x;
                        S = __builtin_fmaf(col.y, q0.y, S); S = __builtin_fmaf(col.z, q0.z, S);
                        S = __builtin_fmaf(c_d, q0.w, S); S += q1.x;
                        S = __builtin_fmaf(nrm.x, q1.y, S); S = __builtin_fmaf(nrm.y, q1.z, S); S = __builtin_fmaf(nrm.z, q1.w, S);
                        S += dL_dweight;
                        const float wS = w * S;             // (0 for an inactive lane: w is)
                        const float incl = wave_scan_add(wS);
                        const float Rl = cR + (incl - wS);                               // sum of w S over the splats behind this one
                        // (lane 63 is the chunk's front-most splat: its T_before is what the next chunk sees behind it)
                        if (lane == 63) s_carry[p] = make_float2(Tb, cR + incl);
                        touched = touched || act;
                        // every addend below is selected to zero for an inactive lane (its inputs may be inf / NaN)
                        const float inv_om = __builtin_amdgcn_exp2f(-lg);                // 1 / (1 - alpha)
                        float dL_dalpha = __builtin_fmaf(Tb, S, -(Rl * inv_om));
                        dL_dalpha = __builtin_fmaf(-T_final * inv_om, q3.w, dL_dalpha);
                        dL_dalpha = act ? dL_dalpha : 0.0f;
                        float dL_dz = __builtin_fmaf(w, q0.w, dz_reg);
                        const unsigned median_p = __float_as_uint(q3.z);
                        dL_dz += ((unsigned)li == median_p - 1u) ? q2.x : 0.0f;
                        dL_dz = act ? dL_dz : 0.0f;
                        const float gG = (opa * dL_dalpha) * -G;
                        const bool a3 = act && use3d;
                        const float dsx = __builtin_fmaf(gG, sx, dL_dz * Tw.x);
                        const float dsy = __builtin_fmaf(gG, sy, dL_dz * Tw.y);
                        const v2f ss = {a3 ? sx : 0.0f, a3 ? sy : 0.0f};
                        const v2f dp = {a3 ? dsx * rz : 0.0f, a3 ? dsy * rz : 0.0f};
                        const float dpz_ = -__builtin_fmaf(dp.x, ss.x, dp.y * ss.y);
                        const v2f lxy = {lx, ly}, lx2 = {lx, lx}, ly2 = {ly, ly};
                        P01 = P01 + dp;
                        X01 = __builtin_elementwise_fma(lx2, dp, X01);
                        Y01 = __builtin_elementwise_fma(ly2, dp, Y01);
                        XY2 = __builtin_elementwise_fma(lxy, (v2f){dpz_, dpz_}, XY2);
                        PZ2 = PZ2 + (v2f){dpz_, dL_dz};
                        const float dz3 = a3 ? dL_dz : 0.0f;
                        Z01 = __builtin_elementwise_fma((v2f){dz3, dz3}, ss, Z01);
                        const float g2 = (act && !use3d) ? gG * FILTER_INV_SQ : 0.0f;
                        C01 = __builtin_elementwise_fma((v2f){g2, g2}, (v2f){dx, dy}, C01);
                        const v2f w2 = {w, w};
                        aN0 = __builtin_fmaf(w, q1.y, aN0);
                        N12 = __builtin_elementwise_fma(w2, (v2f){q1.z, q1.w}, N12);
                        aO = __builtin_fmaf(G, dL_dalpha, aO);
                        RG = __builtin_elementwise_fma(w2, (v2f){q0.x, q0.y}, RG);
                        aB = __builtin_fmaf(w, q0.z, aB);
                    }
                }
              }
            }
            float aP0 = P01.x, aP1 = P01.y, aP2 = PZ2.x, aX0 = X01.x, aX1 = X01.y, aX2 = XY2.x, aY0 = Y01.x, aY1 = Y01.y, aY2 = XY2.y;
            float aZ0 = Z01.x, aZ1 = Z01.y, aZ2 = PZ2.y, aC0 = C01.x, aC1 = C01.y, aN1 = N12.x, aN2 = N12.y, aR = RG.x, aG = RG.y;
            if constexpr (NW > 1) {             // the other waves' partial sums -> wave 0 (fixed order: pixel rows top to bottom)
                if (wv > 0) {
                    float4* a4 = reinterpret_cast<float4*>(s_acc + ((wv - 1) * 64 + lane) * 24);
                    a4[0] = make_float4(aP0, aP1, aP2, aX0); a4[1] = make_float4(aX1, aX2, aY0, aY1);
                    a4[2] = make_float4(aY2, aZ0, aZ1, aZ2); a4[3] = make_float4(aC0, aC1, aN0, aN1);
                    a4[4] = make_float4(aN2, aO, aR, aG); a4[5] = make_float4(aB, touched ? 1.0f : 0.0f, 0.0f, 0.0f);
                }
                __syncthreads();
                if (wv == 0) {
#pragma unroll
                    for (int w = 1; w < NW; w++) {
                        const float4* a4 = reinterpret_cast<const float4*>(s_acc + ((w - 1) * 64 + lane) * 24);
                        const float4 q0 = a4[0], q1 = a4[1], q2 = a4[2], q3 = a4[3], q4 = a4[4], q5 = a4[5];
                        aP0 += q0.x; aP1 += q0.y; aP2 += q0.z; aX0 += q0.w; aX1 += q1.x; aX2 += q1.y; aY0 += q1.z; aY1 += q1.w;
                        aY2 += q2.x; aZ0 += q2.y; aZ1 += q2.z; aZ2 += q2.w; aC0 += q3.x; aC1 += q3.y; aN0 += q3.z; aN1 += q3.w;
                        aN2 += q4.x; aO += q4.y; aR += q4.z; aG += q4.w; aB += q5.x;
                        touched = touched || q5.y != 0.0f;
                    }
                } else touched = false;
                __syncthreads();
            }
            if (touched) {
                // adjoint of  A = Tv x Tw,  B = Tw x Tu,  C = (X0 Tw - Tu) x (Y0 Tw - Tv),  with the pixel sums taken to absolute
                // coordinates:  SP = sum dL/dp,  SX = sum px dL/dp,  SY = sum py dL/dp:
                //   dL/dTu = Tv x SP - Tw x SY,   dL/dTv = SP x Tu - SX x Tw,   dL/dTw = SX x Tv + Tu x SY + sum dL/dz (sx, sy, 1)
                const F3 SP = {aP0, aP1, aP2};
                const F3 SX = {__builtin_fmaf(tile_x0, aP0, aX0), __builtin_fmaf(tile_x0, aP1, aX1), __builtin_fmaf(tile_x0, aP2, aX2)};
                const F3 SY = {__builtin_fmaf(tile_y0, aP0, aY0), __builtin_fmaf(tile_y0, aP1, aY1), __builtin_fmaf(tile_y0, aP2, aY2)};
                const F3 dTu = cross3(Tv, SP) - cross3(Tw, SY);
                const F3 dTv = cross3(SP, Tu) - cross3(SX, Tw);
                const F3 dTw = cross3(SX, Tv) + cross3(Tu, SY);
                float4* o4 = reinterpret_cast<float4*>(partial + ((size_t)slot * GEO_ROWS + row_id) * row_stride + geom_off);
                o4[0] = make_float4(dTu.x, dTu.y, dTu.z, dTv.x);
                o4[1] = make_float4(dTv.y, dTv.z, dTw.x + aZ0, dTw.y + aZ1);
                o4[2] = make_float4(dTw.z + aZ2, aC0, aC1, aN0);
                o4[3] = make_float4(aN1, aN2, aO, aR);
                o4[4] = make_float4(aG, aB, 0.0f, 0.0f);
                row_flags[(size_t)slot * GEO_ROWS + row_id] = 1;
            }
            if (STATS) st[9] += (unsigned)__popcll(__ballot(touched));
        }
        // what is left (less than a chunk, unless the list is drained) moves to the front of the queue
        const int rem = n_q - done;                    // < 64
        const int keep = lane < rem ? s_q[done + lane] : 0;
        block_sync();
        if (lane < rem) s_q[lane] = keep;
        n_q = rem > 0 ? rem : 0;
        block_sync();
    }
    if (STATS && lane == 0 && stats != nullptr) {
#pragma unroll
        for (int k = 0; k < 10; k++) atomicAdd(stats + k, (unsigned long long)st[k]);
    }
}
template __global__ void k_render_bwd_geo<1, false, false>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);
template __global__ void k_render_bwd_geo<1, true, false>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);
template __global__ void k_render_bwd_geo<2, false, false>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);
template __global__ void k_render_bwd_geo<2, true, false>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);
template __global__ void k_render_bwd_geo<1, false, true>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);
template __global__ void k_render_bwd_geo<1, true, true>(int, int, int, const uint32_t*, const uint32_t*, const uint32_t*, const float*, const float*,
                                              const float*, const float*, const float*, const uint32_t*, const float*, const float*,
                                              const uint32_t*, const Rect16*, float*, uint8_t*, int, int, int64_t, const uint32_t*,
                                              const unsigned long long*, unsigned long long*);

}  // namespace isr

// Per-tile bucket binning of the MI355X surfel rasterizer (replaces the reference's duplicateWithKeys + 64-bit global
// radix sort + identifyTileRanges, rasterizer_impl.cu:70-138,283-323): the u32 scans, the scatter into tile buckets and
// the per-bucket sorts.  Design notes: top of isr_forward.hip.
#include "isr_common.hpp"
#include "isr_fast_pair.hpp"

namespace isr {

// ----------------------------------------------------------------------------
// Scans (exact u32).  Small single-block scan over tiles; three-pass scan over Gaussians.
// dense copy of the padded sub-counters (and reset of the scatter cursors), then a single-workgroup scan
__global__ __launch_bounds__(256) void k_gather_counts(int E, const uint32_t* __restrict__ count, uint32_t* __restrict__ dense,
                                                       uint32_t* __restrict__ cursor, const uint32_t* __restrict__ maxima, int nmax,
                                                       int64_t* __restrict__ header) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < E) {
        dense[i] = count[(size_t)i * CNT_STRIDE];
        cursor[(size_t)i * CNT_STRIDE] = 0u;
    }
    // (late Gaussian scan, isr_api_binning.hip) header[1] = the largest number of tiles any splat of the view touches, from K1's
    // per-workgroup maxima: k_scatter needs it, the scan that used to reduce it now runs behind the chain.  K1 has zeroed it.
    if (maxima != nullptr) {
        __shared__ uint32_t s_m[4];
        uint32_t m = 0;
        for (int b = i; b < nmax; b += (int)gridDim.x * 256) m = max(m, maxima[b]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
            if (m > 0u) atomicMax(reinterpret_cast<unsigned long long*>(header + 1), (unsigned long long)m);
        }
    }
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

// k_tile_scan for views of 4 096 .. 8 192 tiles (1080p: 8 160) with every thread's eight tiles - 32 sub-counters - held in
// registers: ONE round of loads (eight independent 16-byte requests per thread), one workgroup scan, and the launch order
// from the tile sizes the thread already holds.  The general kernel above walks the counters in four dependent rounds and
// reads the offsets it has just written back from memory three times (83 us alone at 1080p, on the binning chain of every
// view).  Same outputs, bit for bit (the class of a tile, the stable order inside a class, the thread that owns a tile).
__global__ __launch_bounds__(1024) void k_tile_scan_regs(int T, uint32_t* __restrict__ sub_offset, uint32_t* __restrict__ offset,
                                                         int64_t* header, uint32_t* __restrict__ tile_order, int order_classes) {
    constexpr int PER = 8;                  // tiles per thread
    static_assert(CNT_SUB == 4, "one uint4 per tile");
    __shared__ uint32_t s_warp[32];
    __shared__ uint32_t s_cls[16 * 1024];   // [class][thread] counts, then offsets
    __shared__ uint32_t s_max[1];
    const int t0 = threadIdx.x * PER;
    uint4 v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++)
        v[j] = t0 + j < T ? *reinterpret_cast<const uint4*>(sub_offset + (size_t)(t0 + j) * 4) : make_uint4(0u, 0u, 0u, 0u);
    const int ncls = order_classes < 1 ? 1 : (order_classes > 16 ? 16 : order_classes);
    if (threadIdx.x == 0) s_max[0] = 1;
    for (int e = threadIdx.x; e < ncls * 1024; e += 1024) s_cls[e] = 0;
    uint32_t n[PER], run = 0, mx = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint4 c = v[j];
        n[j] = c.x + c.y + c.z + c.w;
        mx = max(mx, n[j]);
        v[j].x = run; run += c.x;
        v[j].y = run; run += c.y;
        v[j].z = run; run += c.z;
        v[j].w = run; run += c.w;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_1024(run, s_warp, total);      // (its barriers also publish s_max / s_cls)
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (t0 + j < T) {
            const uint4 o = make_uint4(ex + v[j].x, ex + v[j].y, ex + v[j].z, ex + v[j].w);
            *reinterpret_cast<uint4*>(sub_offset + (size_t)(t0 + j) * 4) = o;
            offset[t0 + j] = o.x;
        }
    if (threadIdx.x == 0) {
        offset[T] = total;
        header[0] = (int64_t)total;
    }
    atomicMax(&s_max[0], mx);
    __syncthreads();
    const uint32_t top = s_max[0];
    auto cls_of = [&](uint32_t m) { return (int)min((uint32_t)(ncls - 1), (uint32_t)(((unsigned long long)(top - m) * ncls) / (top + 1u))); };
    int cls[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        cls[j] = cls_of(n[j]);
        if (t0 + j < T) s_cls[cls[j] * 1024 + threadIdx.x]++;
    }
    __syncthreads();
    uint32_t mine[16], acc = 0;
    for (int k = 0; k < ncls; k++) { const int e = threadIdx.x * ncls + k; mine[k] = s_cls[e]; }
    for (int k = 0; k < ncls; k++) { const uint32_t x = mine[k]; mine[k] = acc; acc += x; }
    uint32_t all;
    __syncthreads();
    const uint32_t exc = block_exclusive_scan_1024(acc, s_warp, all);
    for (int k = 0; k < ncls; k++) s_cls[threadIdx.x * ncls + k] = exc + mine[k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (t0 + j < T) {
            uint32_t* slot = &s_cls[cls[j] * 1024 + threadIdx.x];       // this thread's own cursor of the class
            tile_order[(*slot)++] = (uint32_t)(t0 + j);
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
// k_scan_tops + k_scan_add in one launch: every workgroup sums the totals of the workgroups before it itself (at most a few
// thousand coalesced loads: P / 1024 totals), workgroup 0 also reduces the block maxima into header[1].  One launch less on the
// binning chain; same integers.
__global__ __launch_bounds__(1024) void k_scan_add_tops(int n, int nb, uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sums,
                                                        const uint32_t* __restrict__ block_max, int64_t* __restrict__ header) {
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_prefix;
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 1024) acc += block_sums[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += s_part[w];
        s_prefix = t;
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += s_prefix;
    if (blockIdx.x == 0 && block_max != nullptr && header != nullptr && threadIdx.x < 64) {
        uint32_t m = 0;
        for (int b = threadIdx.x; b < nb; b += 64) m = max(m, block_max[b]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) header[1] = (int64_t)m;
    }
}

// Second (and last) level of the Gaussian offset scan when K1 has done the first per 256 Gaussians (k_preprocess): a workgroup
// covers four of K1's, sums the totals before them itself and adds; workgroup 0 reduces the maxima into header[1].
__global__ __launch_bounds__(1024) void k_scan_add_tops256(int n, int nb, uint32_t* __restrict__ out, const uint32_t* __restrict__ sums,
                                                           const uint32_t* __restrict__ maxima, int64_t* __restrict__ header) {
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_prefix;
    const int b4 = (int)blockIdx.x * 4;
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < b4 && j < nb; j += 1024) acc += sums[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += s_part[w];
        s_prefix = t;
    }
    __syncthreads();
    const int q = threadIdx.x >> 8;
    uint32_t extra = 0;
    for (int k = 0; k < q; k++) extra += (b4 + k < nb) ? sums[b4 + k] : 0u;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += s_prefix + extra;
    if (blockIdx.x == 0 && header != nullptr && threadIdx.x < 64) {
        uint32_t m = 0;
        for (int b = threadIdx.x; b < nb; b += 64) m = max(m, maxima[b]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) header[1] = (int64_t)m;
    }
}

// The late form of the Gaussian offset scan as two launches whose cost does not grow with the square of the view: one workgroup
// scans K1's per-256 totals in place (a contiguous run per thread, one workgroup scan), then every offset adds its block's entry.
// (k_scan_add_tops256 has each of its P / 1024 workgroups sum all totals before it: 190 MB of L2 reads at P = 5 M - 19 us alone,
// 104 us beside the sampled backward.)
__global__ __launch_bounds__(1024) void k_scan_tops_inplace(int nb, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_warp[32];
    const int per = (nb + 1023) / 1024;
    const int i0 = (int)threadIdx.x * per;
    uint32_t run = 0;
    for (int j = 0; j < per; j++) run += (i0 + j < nb) ? sums[i0 + j] : 0u;
    uint32_t total;
    uint32_t acc = block_exclusive_scan_1024(run, s_warp, total);
    for (int j = 0; j < per; j++)
        if (i0 + j < nb) { const uint32_t v = sums[i0 + j]; sums[i0 + j] = acc; acc += v; }
}
__global__ __launch_bounds__(256) void k_scan_add256(int n, uint32_t* __restrict__ out, const uint32_t* __restrict__ tops) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // (K1's workgroup of this element: blockIdx.x)
    if (i < n) out[i] += tops[blockIdx.x];
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
    if (i < P) {
        // rectangle and depth are requested WITH the instance count, not behind it (K1 writes an empty rectangle for a splat
        // that touches nothing; its record's depth is then whatever the buffer held - and not used): one memory trip, not two
        const uint32_t tt = g.tiles_touched[i];
        const Rect16 r_ = g.rect[i];
        const unsigned dz = __float_as_uint(g.rec[(size_t)i * REC + 18]);
        if (tt != 0) { rc = r_; key = ((unsigned long long)dz << 32) | (unsigned)i; }
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
            // (the reserved range's start in the key array, sub-range offset included: the placement loop below then needs no
            // global load per tile - it was one dependent trip per (splat, tile), serial in every lane)
            const size_t ee = (size_t)th_key[e] * CNT_SUB + sub;
            const uint32_t so = sub_offset[ee];
            th_base[e] = so + atomicAdd(tile_cursor + ee * CNT_STRIDE, th_cnt[e]);
            th_cnt[e] = 0u;                 // becomes the fill counter of the reserved range
        }
    __syncthreads();
    for (int y = rc.y0; y < rc.y1; y++)
        for (int x = rc.x0; x < rc.x1; x++) {
            const uint32_t tile = (uint32_t)y * gx + x;
            const size_t e = (size_t)tile * CNT_SUB + sub;
            const int slot = th_find(th_key, tile);
            int64_t at;
            if (slot >= 0) at = (int64_t)th_base[slot] + atomicAdd(&th_cnt[slot], 1u);
            else at = (int64_t)sub_offset[e] + atomicAdd(tile_cursor + e * CNT_STRIDE, 1u);
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

// Buckets of radix_min < n <= SORT_RADIX_KEYS keys: a stable LSD radix sort in LDS instead of the O(n log^2 n) network (dense
// scenes: C5's lists average 3 100 entries, and the side chain that holds this kernel is that step's critical path).
//   * Only the 32 depth bits are sorted by radix passes (8 bits each); a pass whose digit is the same for all keys - the high
//     byte(s) of the depths of one tile - is skipped.  Keys of EQUAL depth (a third of C5's tiles have a pair) are then put into
//     ascending id order by odd-even transposition inside their runs (runs of 2 or 3: as many phases).  Together that is the order
//     of the reference's stable radix sort on (tile, depth) over keys emitted in id order (rasterizer_impl.cu:70-111,309-314).
//   * A pass: every wave owns a contiguous segment of the array (stability = segment order, then position inside it); per wave a
//     256-bin histogram (LDS atomics); the bins are scanned digit-major / wave-minor; then every wave re-walks its segment, 64
//     keys at a time: the lanes holding the same digit find each other with eight ballots, rank themselves by lane order, and the
//     group's first lane advances the wave's running offset of that digit.
// 1024 threads, two key buffers of SORT_RADIX_KEYS u64 + 16 KB of counters = 144 KB of dynamic LDS, one workgroup per CU.
// Measured at C5 (tools/fwd_ab.py, the sort kernels of a view together): 0.381 ms against 0.586 with the network for these
// buckets; taking the (2 048, 4 096] buckets from the one-wave register sort as well: 0.415 - not done.
constexpr int SORT_RADIX_KEYS = 8192;
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(1024) void k_tile_sort_radix(const uint32_t* __restrict__ tile_offset, const unsigned long long* __restrict__ keys,
                                                          uint32_t* __restrict__ point_list, int64_t capacity, int min_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_big[];
    unsigned long long* buf0 = s_big;
    unsigned long long* buf1 = s_big + SORT_RADIX_KEYS;
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_big + 2 * SORT_RADIX_KEYS);       // [16 waves][256 bins]
    __shared__ uint32_t s_total[256];
    __shared__ int s_flag;
    const uint32_t t = blockIdx.x;
    const int64_t r0 = tile_offset[t];
    int64_t r1 = tile_offset[t + 1];
    if (r1 > capacity) r1 = capacity;
    const int n = (int)(r1 - r0);
    if (n <= min_n || n > SORT_RADIX_KEYS) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = (n + 1023) / 1024 * 64;                     // keys per wave segment (a multiple of 64)
    const int seg0 = wv * S, nchunk = S / 64;
    for (int i = threadIdx.x; i < n; i += 1024) buf0[i] = keys[r0 + i];
    unsigned long long* src = buf0;
    unsigned long long* dst = buf1;
    for (int shift = 32; shift < 64; shift += 8) {
        for (int e = threadIdx.x; e < 16 * 256; e += 1024) hist[e] = 0u;
        __syncthreads();                                      // (also: the loads above / the previous pass's scatter)
        for (int c = 0; c < nchunk; c++) {
            const int i = seg0 + c * 64 + lane;
            if (i < n) atomicAdd(&hist[wv * 256 + (int)((src[i] >> shift) & 255ull)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 256) {                              // digit d: exclusive offsets of the 16 waves inside the digit, and its total
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) { const uint32_t v = hist[w * 256 + threadIdx.x]; hist[w * 256 + threadIdx.x] = run; run += v; }
            s_total[threadIdx.x] = run;
        }
        if (threadIdx.x == 0) s_flag = 0;
        __syncthreads();
        if (wv == 0) {                                        // exclusive scan of the 256 totals (four per lane)
            uint32_t c4[4], run = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) { c4[u] = s_total[lane * 4 + u]; if (c4[u] == (uint32_t)n) s_flag = 1; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t x = c4[u]; c4[u] = run; run += x; }
            uint32_t inc = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += y; }
#pragma unroll
            for (int u = 0; u < 4; u++) s_total[lane * 4 + u] = inc - run + c4[u];
        }
        __syncthreads();
        if (s_flag) { __syncthreads(); continue; }            // every key has the same digit: nothing moves
        for (int c = 0; c < nchunk; c++) {
            const int i = seg0 + c * 64 + lane;
            const bool valid = i < n;
            const unsigned long long key = valid ? src[i] : 0ull;
            const int d = (int)((key >> shift) & 255ull);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool bit = (d >> k) & 1;
                const unsigned long long b = __ballot(bit);
                peers &= bit ? b : ~b;
            }
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            if (valid) {
                const uint32_t at = s_total[d] + hist[wv * 256 + d] + (uint32_t)rank;
                dst[at] = key;
            }
            wave_lds_fence();                                 // (every lane has read the running offset)
            if (valid && rank == 0) hist[wv * 256 + d] += (uint32_t)__popcll(peers);
            wave_lds_fence();
        }
        __syncthreads();
        unsigned long long* tmp = src; src = dst; dst = tmp;
    }
    // equal depths -> ascending id: odd-even transposition inside the runs, until an even and an odd phase in a row moved nothing
    int clean = 0;
    for (int phase = 0; phase < n + 2 && clean < 2; phase++) {
        if (threadIdx.x == 0) s_flag = 0;
        __syncthreads();
        for (int j = (phase & 1) + 2 * (int)threadIdx.x; j + 1 < n; j += 2048) {
            const unsigned long long a = src[j], b = src[j + 1];
            if ((a >> 32) == (b >> 32) && (uint32_t)a > (uint32_t)b) { src[j] = b; src[j + 1] = a; s_flag = 1; }
        }
        __syncthreads();
        clean = s_flag ? 0 : clean + 1;                       // (uniform: every thread reads the same flag)
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 1024) point_list[r0 + i] = (uint32_t)src[i];
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


// After the sort: what the blend's four 8x8 blocks of a tile need to know about every list entry BEFORE touching its record.
// Per 64-entry chunk of a tile's list HM_WORDS 64-bit words - bit l of word b (0..3): entry l meets block b; of word 4 + 2 b + h:
// it meets rows 4 h .. 4 h + 3 of block b - so that a block's wave finds its hits with one scalar load per chunk instead of
// gathering 32 bytes per entry (four times per tile); and the tile-relative int8 box the backward kernels test (box4).
// "Meets" (ROW_EXACT, the default): some pixel CENTRE of the block half lies inside the splat's alpha >= 1/255 region - the
// ellipse of the 3-D branch (splat_conic, K1: GeomView::ellipse, the only 32 bytes gathered per entry)
// or the low-pass disc; the minimum of the ellipse's form over the rectangle of a half's pixel centres lies on one of two lines
// (isr_common.hpp).  With ROW_EXACT off (ISR_PACK_EXACT=0), and for the few splats without a certified ellipse, the bounding octagon
// of that region (K1's cull bounds, GeomView::cull) decides as in rounds 1-5: it keeps ~1/3 more (half, splat) pairs -
// pairs the blend kernels evaluate only to find every lane beyond band.hi, and that hold a lane of the splat-major backward for
// every pixel iteration of its chunk.
template <bool ROW_EXACT>
__global__ __launch_bounds__(256) void k_pack_hits(int gx, int64_t capacity, const uint32_t* __restrict__ tile_offset,
                                                   const uint32_t* __restrict__ point_list, const float* __restrict__ cull,
                                                   const float* __restrict__ ellipse, uint32_t* __restrict__ box4,
                                                   unsigned long long* __restrict__ hit_mask) {
    const int tile = blockIdx.x;
    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    if (len <= 0) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float X0 = (float)((tile % gx) * TILE), Y0 = (float)((tile / gx) * TILE);
    // the gathers of a chunk are issued one chunk ahead (ids two ahead): a wave's chunks are a serial chain of two dependent memory
    // trips and ~350 instructions otherwise
    auto load_id = [&](int c) { const int i = c * 64 + lane; return i < len ? (int)point_list[r0 + i] : -1; };
    int id_n = load_id(wv), id_nn = load_id(wv + 4);
    float4 e0_n = make_float4(0.f, 0.f, 0.f, 0.f), e1_n = e0_n;
    if (id_n >= 0) {
        e0_n = reinterpret_cast<const float4*>(ellipse + (size_t)id_n * CULL_STRIDE)[0];
        e1_n = reinterpret_cast<const float4*>(ellipse + (size_t)id_n * CULL_STRIDE)[1];
    }
    for (int c = wv; c * 64 < len; c += 4) {
        const int i = c * 64 + lane;
        unsigned hb = 0u;            // bit 2 b + h: the entry meets half h of block b
        const int id = id_n;
        const float4 c0 = e0_n, c1 = e1_n;
        id_n = id_nn;
        if (id_n >= 0) {
            e0_n = reinterpret_cast<const float4*>(ellipse + (size_t)id_n * CULL_STRIDE)[0];
            e1_n = reinterpret_cast<const float4*>(ellipse + (size_t)id_n * CULL_STRIDE)[1];
        }
        id_nn = load_id(c + 8);
        if (i < len) {
            const float4* cr = reinterpret_cast<const float4*>(cull + (size_t)id * CULL_STRIDE);
            const float Mx = c0.x, My = c0.y, l11 = c0.z, l12 = c0.w, l22 = c1.x, cx = c1.y, cy = c1.z, r2 = c1.w;
            if (!ROW_EXACT || !(l11 > 0.0f)) {
                // the pass-all form (ill-conditioned splat, no finite bound: 0.4 % of a C3 view) or the octagon-only build: K1's box and
                // the same along the diagonals, the blend kernels' own float tests
                const float4 bb = cr[0], dg = cr[1];
                box4[r0 + i] = pack_box4(bb, X0, Y0);
                const bool xa = !(bb.x > X0 + 7.0f) && !(bb.y < X0), xb = !(bb.x > X0 + 15.0f) && !(bb.y < X0 + 8.0f);
                const bool ya = !(bb.z > Y0 + 7.0f) && !(bb.w < Y0), yb = !(bb.z > Y0 + 15.0f) && !(bb.w < Y0 + 8.0f);
                auto diag = [&](float bx0, float by0) {
                    const float bx1 = bx0 + 7.0f, by1 = by0 + 7.0f;
                    return !(dg.x > bx1 + by1) && !(dg.y < bx0 + by0) && !(dg.z > bx1 - by0) && !(dg.w < bx0 - by1);
                };
                const bool o0 = xa && ya && diag(X0, Y0), o1 = xb && ya && diag(X0 + 8.0f, Y0);
                const bool o2 = xa && yb && diag(X0, Y0 + 8.0f), o3 = xb && yb && diag(X0 + 8.0f, Y0 + 8.0f);
                hb = (o0 ? 3u : 0u) | (o1 ? 12u : 0u) | (o2 ? 48u : 0u) | (o3 ? 192u : 0u);
            } else {
                // e(dx, dy) = (l11 dx + l12 dy)^2 + (l22 dy)^2 <= 1 (relative to M).  Along x = const the vertex is dy = -nxy dx / nyy,
                // along y = const dx = -l12 dy / l11.
                const float nyy = __builtin_fmaf(l12, l12, l22 * l22);
                const float i11 = __builtin_amdgcn_rcpf(l11), i22 = __builtin_amdgcn_rcpf(l22);
                const float ryx = -(l11 * l12) * __builtin_amdgcn_rcpf(nyy), rxy = -l12 * i11;
                {   // the tile-relative box the backward kernels test: the ellipse's and the low-pass disc's bounding box (the region
                    // outside of which alpha < 1/255 is certain), + 1 % + 0.25 px for the approximate reciprocals
                    const float hx = __builtin_sqrtf(nyy) * i11 * i22 * 1.01f + 0.25f, hy = i22 * 1.01f + 0.25f;
                    const float rd = __builtin_sqrtf(r2) * 1.01f + 0.25f;
                    box4[r0 + i] = pack_box4(make_float4(fminf(Mx - hx, cx - rd), fmaxf(Mx + hx, cx + rd), fminf(My - hy, cy - rd),
                                                         fmaxf(My + hy, cy + rd)), X0, Y0);
                }
                float xa[2], xb[2], ux[2], vy[2], ddx[2];       // per column range: ends relative to M; l11 dx and the vertex on the line x = clamp(M.x)
#pragma unroll
                for (int sg = 0; sg < 2; sg++) {
                    const float x0 = X0 + 8.0f * (float)sg;
                    xa[sg] = x0 - Mx; xb[sg] = (x0 + 7.0f) - Mx;
                    const float dxe = fminf(fmaxf(0.0f, xa[sg]), xb[sg]);
                    ux[sg] = l11 * dxe;
                    vy[sg] = ryx * dxe;
                    const float dn = fminf(fmaxf(0.0f, x0 - cx), (x0 + 7.0f) - cx);      // the low-pass disc: the range's column nearest to the centre
                    ddx[sg] = dn * dn;
                }
                float ya[4], yb[4], uy[4], wy[4], vx[4], ddy[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float y0 = Y0 + 4.0f * (float)g;
                    ya[g] = y0 - My; yb[g] = (y0 + 3.0f) - My;
                    const float dye = fminf(fmaxf(0.0f, ya[g]), yb[g]);
                    uy[g] = l12 * dye;
                    wy[g] = l22 * dye; wy[g] *= wy[g];
                    vx[g] = rxy * dye;
                    const float dn = fminf(fmaxf(0.0f, y0 - cy), (y0 + 3.0f) - cy);
                    ddy[g] = dn * dn;
                }
                // a lower bound of e: the first square's two addends may cancel, |u| is taken 4 ulps of their magnitudes smaller
                auto e_low = [](float a, float b, float v2) {
                    const float u = fmaxf(fabsf(a + b) - 4.8e-7f * (fabsf(a) + fabsf(b)), 0.0f);
                    return __builtin_fmaf(u, u, v2) * 0.999999f;
                };
#pragma unroll
                for (int g = 0; g < 4; g++) {
#pragma unroll
                    for (int sg = 0; sg < 2; sg++) {
                        const int blk = (g >> 1) * 2 + sg;
                        const float t1 = fminf(fmaxf(vy[sg], ya[g]), yb[g]);                  // on x = clamp(M.x): dy of the minimum
                        const float w1 = l22 * t1;
                        const float e1 = e_low(ux[sg], l12 * t1, w1 * w1);
                        const float t2 = fminf(fmaxf(vx[g], xa[sg]), xb[sg]);                 // on y = clamp(M.y): dx of the minimum
                        const float e2 = e_low(l11 * t2, uy[g], wy[g]);
                        // (negated comparisons: a NaN anywhere keeps the pair)
                        const bool near = !(fminf(e1, e2) > 1.0f) || !(e1 == e1) || !(e2 == e2) || !(ddx[sg] + ddy[g] > r2);
                        if (near) hb |= 1u << (2 * blk + (g & 1));
                    }
                }
            }
        }
        unsigned long long mh[8];
#pragma unroll
        for (int k = 0; k < 8; k++) mh[k] = __ballot((hb >> k) & 1u);
        unsigned long long out = 0ull;
#pragma unroll
        for (int k = 0; k < 4; k++) if (lane == k) out = mh[2 * k] | mh[2 * k + 1];
#pragma unroll
        for (int k = 0; k < 8; k++) if (lane == 4 + k) out = mh[k];
        if (lane < HM_WORDS) hit_mask[hit_mask_word(r0, tile, c) + lane] = out;
    }
}

// Test infrastructure (isr_debug_check_hit_masks): every (tile entry, 8x4 block half) whose bit in k_pack_hits' masks is CLEAR is
// evaluated on all 32 pixels of the half the way the blend kernels do - FAST's rho against band.hi and EXACT's own pair test - and
// pairs that a kernel would have blended are counted: [0] halves checked, [1] pairs FAST's test passes (fast_pair_lane), [2] pairs
// EXACT's test passes; also [3] halves whose bit is set, [4] of them: halves where no pixel is near in either
// arithmetic (what an ideal test would have cleared).  One workgroup per tile, a wave per entry, lane = pixel of a half (two rounds).
__global__ __launch_bounds__(256) void k_check_hit_masks(int W, int H, int gx, int64_t capacity, const uint32_t* __restrict__ tile_offset,
                                                         const uint32_t* __restrict__ point_list, const float* __restrict__ rec,
                                                         const unsigned long long* __restrict__ hit_mask,
                                                         unsigned long long* __restrict__ counters) {
    const int tile = blockIdx.x;
    const int64_t r0 = tile_offset[tile];
    int64_t r1 = tile_offset[tile + 1];
    if (r1 > capacity) r1 = capacity;
    const int len = (int)(r1 - r0);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    unsigned long long n_clear = 0, n_fast = 0, n_exact = 0, n_set = 0, n_idle = 0;
    for (int i = wv; i < len; i += 4) {
        const int id = (int)point_list[r0 + i];
        const float* q = rec + (size_t)id * REC;
        const F3 Tu = {q[0], q[1], q[2]}, Tv = {q[3], q[4], q[5]}, Tw = {q[6], q[7], q[8]};
        const float cx = q[9], cy = q[10], opa = q[14], band = q[19];
        const FastBand fb = fast_band(opa, band);
        const size_t w0 = hit_mask_word(r0, tile, i >> 6);
        for (int blk = 0; blk < 4; blk++) {
            const bool bit_blk = (hit_mask[w0 + blk] >> (i & 63)) & 1ull;
            // lane: half = lane >> 5, pixel of the half = lane & 31
            const int half = lane >> 5, pr = (lane & 31) >> 3, pc = lane & 7;
            const bool bit = (hit_mask[w0 + 4 + 2 * blk + half] >> (i & 63)) & 1ull;
            const float pxf = (float)(tx * TILE + (blk & 1) * 8 + pc), pyf = (float)(ty * TILE + (blk >> 1) * 8 + half * 4 + pr);
            const bool inside = pxf < (float)W && pyf < (float)H;
            FastRay fr, er; FastHit fh, eh;
            const bool fp = fast_pair_lane(Tu, Tv, Tw, cx, cy, opa, fast_det(Tu, Tv, Tw, cx, cy), fb, pxf, pyf, fr, fh);      // FAST's decision (EXACT's inside the bands)
            const bool ep = exact_pair(pxf, pyf, Tu, Tv, Tw, cx, cy, opa, er, eh);
            const bool fnear = inside && fp, enear = inside && ep;
            for (int hf = 0; hf < 2; hf++) {
                const unsigned long long sel = hf == 0 ? 0xffffffffull : 0xffffffff00000000ull;
                const unsigned long long mf = __ballot(fnear) & sel, me = __ballot(enear) & sel;
                const bool b = __builtin_amdgcn_readlane((int)bit, hf * 32) != 0;
                if (!b) {
                    n_clear++; n_fast += __popcll(mf); n_exact += __popcll(me);
                    if ((mf | me) != 0ull && lane == 0) {          // one offender for the report: Gaussian id + 1, its pixel
                        const int l0 = __builtin_ctzll(mf | me);
                        counters[5] = (unsigned long long)id + 1ull;
                        counters[6] = (unsigned long long)(tx * TILE + (blk & 1) * 8 + (l0 & 7)) |
                                      ((unsigned long long)(ty * TILE + (blk >> 1) * 8 + (l0 >> 5) * 4 + ((l0 & 31) >> 3)) << 32);
                    }
                }
                else { n_set++; if (mf == 0ull && me == 0ull) n_idle++; }
                // a block word must be the OR of its halves
                if (b && !bit_blk) n_fast += 1000000ull;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(counters + 0, n_clear); atomicAdd(counters + 1, n_fast); atomicAdd(counters + 2, n_exact);
        atomicAdd(counters + 3, n_set); atomicAdd(counters + 4, n_idle);
    }
}

}  // namespace isr

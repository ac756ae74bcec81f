// Exact mean squared distance to the 3 nearest neighbours — replaces simple_knn.distCUDA2
// (reference submodules/simple-knn/simple_knn.cu:46-222: Morton sort + 1024-point box pruning,
// O(P^2/1024) box tests per point, two blocking D2H copies, thrust allocations).
//
// MI355X design: uniform-grid bucketing (~8 points per cell) built with one count/scan/scatter (the points end up sorted
// by cell, cells in z-y-x order).  Default query (kk_query): a thread per point walks expanding Chebyshev rings of cells
// straight from global memory - the cell-sorted array is L2-resident and neighbouring threads read neighbouring cells:
// 0.58 ms for 10^6 uniform points, 4.3 ms for a clustered (SfM-like) cloud of 10^6 (profiles/r02_knn_timing.json).
// Opt-in (ISO_KNN_LDS=1), the LDS-bucketed form north_star names (kk_query_lds): a workgroup owns a SEGMENT of KNN_SEG consecutive
// cells of one grid row; the points of the segment's 3 x 3 x (KNN_SEG + 2) cell neighbourhood are nine contiguous runs of
// the sorted array, copied coalesced into LDS once, and every query point of the segment scans its own 27 cells there
// (the reference stages its 1024-point boxes in shared memory the same way, simple_knn.cu:148-184, but tests every box
// against every point).  A point whose third neighbour is not provably inside those 27 cells - sparse regions, the
// hull - continues with expanding Chebyshev rings of cells from global memory (knn_rings) and stops as soon as the
// third-best distance is inside the searched box, so the result is exactly the brute-force answer for any distribution.
// Measured: 0.8-1.4x the default's time on uniform clouds, 4-8x on clustered ones (dense cells overflow the LDS budget and
// most segments of the grid are empty) - hence not the default.  No host synchronisation in either.
#include "isr_common.hpp"

namespace iso {

struct KnnGrid {          // device-resident grid description (written by kk_setup)
    float minx, miny, minz, cell, inv_cell;
    int gx, gy, gz, ncell;
};

struct KnnView {
    float* partial;        // [256][6] per-block min/max
    KnnGrid* grid;
    uint32_t* count;       // [ccap]
    uint32_t* offset;      // [ccap]
    uint32_t* cursor;      // [ccap]
    uint32_t* sums;        // scan block sums
    float4* sorted;        // [P] (x,y,z, bitcast index)
};
inline int knn_cell_cap(int P) { return 2 * (P / 8 + 1) + 4096; }
inline KnnView knn_view(void* buf, int P) {
    char* p = (char*)buf;
    const int cc = knn_cell_cap(P);
    KnnView v;
    v.partial = isr::carve<float>(p, 256 * 6);
    v.grid = isr::carve<KnnGrid>(p, 1);
    v.count = isr::carve<uint32_t>(p, cc);
    v.offset = isr::carve<uint32_t>(p, cc);
    v.cursor = isr::carve<uint32_t>(p, cc);
    v.sums = isr::carve<uint32_t>(p, cc / 1024 + 2);
    v.sorted = isr::carve<float4>(p, P > 0 ? P : 1);
    return v;
}
inline size_t knn_bytes(int P) {
    KnnView v = knn_view((void*)0, P);
    return (size_t)(v.sorted + (P > 0 ? P : 1)) + 256;
}

__global__ __launch_bounds__(256) void kk_minmax(int P, const float* __restrict__ pts, float* __restrict__ partial) {
    __shared__ float s[6][4];
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += 256 * gridDim.x)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; a++) { s[a][threadIdx.x >> 6] = mn[a]; s[3 + a][threadIdx.x >> 6] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) partial[blockIdx.x * 6 + threadIdx.x] = fminf(fminf(s[threadIdx.x][0], s[threadIdx.x][1]), fminf(s[threadIdx.x][2], s[threadIdx.x][3]));
    else if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = fmaxf(fmaxf(s[threadIdx.x][0], s[threadIdx.x][1]), fmaxf(s[threadIdx.x][2], s[threadIdx.x][3]));
}

__global__ void kk_setup(int P, int nblk, int ccap, const float* __restrict__ partial, KnnGrid* grid) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int b = 0; b < nblk; b++)
        for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], partial[b * 6 + a]); mx[a] = fmaxf(mx[a], partial[b * 6 + 3 + a]); }
    float ext[3], emax = 0.0f;
    for (int a = 0; a < 3; a++) { ext[a] = mx[a] - mn[a]; emax = fmaxf(emax, ext[a]); }
    if (!(emax > 0.0f)) emax = 1.0f;
    float vol = 1.0f;
    for (int a = 0; a < 3; a++) vol *= fmaxf(ext[a], 1e-3f * emax);
    const float target = fmaxf(1.0f, (float)P / 8.0f);
    float cell = cbrtf(vol / target);
    int g[3];
    for (int it = 0; it < 64; it++) {
        long long prod = 1;
        for (int a = 0; a < 3; a++) {
            g[a] = (int)fminf(1024.0f, floorf(ext[a] / cell) + 1.0f);
            if (g[a] < 1) g[a] = 1;
            prod *= g[a];
        }
        if (prod <= ccap) break;
        cell *= 1.26f;
    }
    grid->minx = mn[0]; grid->miny = mn[1]; grid->minz = mn[2];
    grid->cell = cell; grid->inv_cell = 1.0f / cell;
    grid->gx = g[0]; grid->gy = g[1]; grid->gz = g[2];
    grid->ncell = g[0] * g[1] * g[2];
}

__device__ __forceinline__ void cell_of(const KnnGrid& G, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = min(G.gx - 1, max(0, (int)((x - G.minx) * G.inv_cell)));
    cy = min(G.gy - 1, max(0, (int)((y - G.miny) * G.inv_cell)));
    cz = min(G.gz - 1, max(0, (int)((z - G.minz) * G.inv_cell)));
}

__global__ __launch_bounds__(256) void kk_count(int P, const float* __restrict__ pts, const KnnGrid* __restrict__ grid,
                                                uint32_t* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const KnnGrid G = *grid;
    int cx, cy, cz;
    cell_of(G, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
    atomicAdd(count + ((size_t)cz * G.gy + cy) * G.gx + cx, 1u);
}

__global__ __launch_bounds__(256) void kk_scatter(int P, const float* __restrict__ pts, const KnnGrid* __restrict__ grid,
                                                  const uint32_t* __restrict__ offset, uint32_t* __restrict__ cursor,
                                                  float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const KnnGrid G = *grid;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    int cx, cy, cz;
    cell_of(G, x, y, z, cx, cy, cz);
    const size_t c = ((size_t)cz * G.gy + cy) * G.gx + cx;
    const uint32_t pos = atomicAdd(cursor + c, 1u);
    sorted[offset[c] + pos] = make_float4(x, y, z, __uint_as_float((unsigned)i));
}

// distance formula and top-3 insertion identical to the reference (simple_knn.cu:133-146)
__device__ __forceinline__ void knn_update3(float rx, float ry, float rz, const float4 q, float* best) {
    const float dx = q.x - rx, dy = q.y - ry, dz = q.z - rz;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

// True when everything outside the box of cells [c - r, c + r] is farther than the current third-best distance: it is at
// least `m` away; faces beyond the grid do not bound anything.  The margin covers cell-assignment rounding.
__device__ __forceinline__ bool knn_done(const KnnGrid& G, const float4 me, int cx, int cy, int cz, int r, const float* best) {
    float m = 3.402823466e+38f;
    const float eps = 1e-4f * G.cell;
    if (cx - r > 0) m = fminf(m, me.x - (G.minx + (cx - r) * G.cell));
    if (cx + r < G.gx - 1) m = fminf(m, (G.minx + (cx + r + 1) * G.cell) - me.x);
    if (cy - r > 0) m = fminf(m, me.y - (G.miny + (cy - r) * G.cell));
    if (cy + r < G.gy - 1) m = fminf(m, (G.miny + (cy + r + 1) * G.cell) - me.y);
    if (cz - r > 0) m = fminf(m, me.z - (G.minz + (cz - r) * G.cell));
    if (cz + r < G.gz - 1) m = fminf(m, (G.minz + (cz + r + 1) * G.cell) - me.z);
    if (m == 3.402823466e+38f) return true;            // whole grid searched
    m -= eps;
    return m > 0.0f && best[2] <= m * m;
}

// Rings r_first, r_first + 1, ... around the cell (cx, cy, cz) from global memory, until the third-best distance is provably
// inside the searched box.
__device__ __forceinline__ void knn_rings(const KnnGrid& G, const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                          const float4* __restrict__ sorted, const float4 me, unsigned self, int cx, int cy,
                                          int cz, int r_first, float* best) {
    const int rmax = max(G.gx, max(G.gy, G.gz));
    for (int r = r_first; r <= rmax; r++) {
        // visit the shell of Chebyshev radius r around (cx,cy,cz)
        const int z0 = max(0, cz - r), z1 = min(G.gz - 1, cz + r);
        const int y0 = max(0, cy - r), y1 = min(G.gy - 1, cy + r);
        const int x0 = max(0, cx - r), x1 = min(G.gx - 1, cx + r);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool face = (abs(z - cz) == r) || (abs(y - cy) == r);
                for (int x = x0; x <= x1; x++) {
                    if (!face && abs(x - cx) != r) { if (x < x1 && x < cx + r) x = min(x1, cx + r) - 1; continue; }
                    const size_t c = ((size_t)z * G.gy + y) * G.gx + x;
                    const uint32_t o = offset[c], n = count[c];
                    for (uint32_t k = 0; k < n; k++) {
                        const float4 q = sorted[o + k];
                        if (__float_as_uint(q.w) == self) continue;
                        knn_update3(me.x, me.y, me.z, q, best);
                    }
                }
            }
        if (knn_done(G, me, cx, cy, cz, r, best)) break;
    }
}

__global__ __launch_bounds__(256) void kk_query(int P, const KnnGrid* __restrict__ grid, const uint32_t* __restrict__ offset,
                                                const uint32_t* __restrict__ count, const float4* __restrict__ sorted,
                                                float* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= P) return;
    const KnnGrid G = *grid;
    const float4 me = sorted[s];
    const unsigned self = __float_as_uint(me.w);
    int cx, cy, cz;
    cell_of(G, me.x, me.y, me.z, cx, cy, cz);
    float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    knn_rings(G, offset, count, sorted, me, self, cx, cy, cz, 0, best);
    out[self] = (best[0] + best[1] + best[2]) / 3.0f;
}

// LDS-bucketed query: a fixed-size grid of workgroups strides over the (segment, row) work items - the grid's dimensions
// live on the device (kk_setup), so the host cannot size the launch by them; see the file header.
constexpr int KNN_SEG = 16;                 // cells of a row per workgroup
constexpr int KNN_LDS_PTS = 3072;           // staged points (48 KB); a denser neighbourhood falls back to the global walk
__global__ __launch_bounds__(256) void kk_query_lds(int P, const KnnGrid* __restrict__ grid, const uint32_t* __restrict__ offset,
                                                    const uint32_t* __restrict__ count, const float4* __restrict__ sorted,
                                                    float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float4 s_pts[KNN_LDS_PTS];
    __shared__ uint32_t s_off[9][KNN_SEG + 3];      // per neighbour row: LDS position of the first point of cells x0-1 .. x1+1, + end
    __shared__ uint32_t s_base[10];
    const KnnGrid G = *grid;
    const int segs = (G.gx + KNN_SEG - 1) / KNN_SEG;
    const long long nwork = (long long)segs * G.gy * G.gz;
  for (long long wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    __syncthreads();                                                 // the previous work item is done with the LDS
    const int sg = (int)(wk % segs);
    const int cy = (int)((wk / segs) % G.gy), cz = (int)(wk / ((long long)segs * G.gy));
    const int x0 = sg * KNN_SEG;
    const int x1 = min(G.gx - 1, x0 + KNN_SEG - 1);                 // the segment's cells [x0, x1]
    const int nx0 = max(0, x0 - 1), nx1 = min(G.gx - 1, x1 + 1);    // the neighbourhood's cells along x
    const size_t row_c = ((size_t)cz * G.gy + cy) * G.gx;
    const uint32_t seg_lo = offset[row_c + x0], seg_hi = offset[row_c + x1] + count[row_c + x1];
    if (seg_hi == seg_lo) continue;                                  // no query point here (uniform)
    // nine neighbour rows: global start / length of their runs, and where they go in LDS
    if (threadIdx.x < 9) {
        const int dz = (int)threadIdx.x / 3 - 1, dy = (int)threadIdx.x % 3 - 1;
        const int z = cz + dz, y = cy + dy;
        uint32_t n = 0;
        if (z >= 0 && z < G.gz && y >= 0 && y < G.gy) {
            const size_t rc = ((size_t)z * G.gy + y) * G.gx;
            n = offset[rc + nx1] + count[rc + nx1] - offset[rc + nx0];
        }
        s_base[threadIdx.x] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int r = 0; r < 9; r++) { const uint32_t n = s_base[r]; s_base[r] = acc; acc += n; }
        s_base[9] = acc;
    }
    __syncthreads();
    const uint32_t total = s_base[9];
    const bool staged = total <= (uint32_t)KNN_LDS_PTS;
    if (staged) {
        for (int r = 0; r < 9; r++) {
            const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
            if (z < 0 || z >= G.gz || y < 0 || y >= G.gy) {
                for (int e = threadIdx.x; e < KNN_SEG + 3; e += 256) s_off[r][e] = s_base[r];
                continue;
            }
            const size_t rc = ((size_t)z * G.gy + y) * G.gx;
            const uint32_t g0 = offset[rc + nx0], n = s_base[r + 1] - s_base[r];
            for (uint32_t e = threadIdx.x; e < n; e += 256) s_pts[s_base[r] + e] = sorted[g0 + e];
            // cell starts: entry k <-> cell x0 - 1 + k (cells outside the grid are empty)
            for (int e = threadIdx.x; e < KNN_SEG + 3; e += 256) {
                const int x = x0 - 1 + e;
                uint32_t at;
                if (x < nx0) at = 0; else if (x > nx1) at = n; else at = offset[rc + x] - g0;
                s_off[r][e] = s_base[r] + at;
            }
        }
    }
    __syncthreads();
    for (uint32_t s = seg_lo + threadIdx.x; s < seg_hi; s += 256) {
        const float4 me = sorted[s];
        const unsigned self = __float_as_uint(me.w);
        int cx, cyy, czz;
        cell_of(G, me.x, me.y, me.z, cx, cyy, czz);                 // (cyy, czz) == (cy, cz); cx in [x0, x1]
        float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        int next_ring = 0;
        if (staged) {
            const int k = cx - (x0 - 1);                             // s_off entry of cell cx
#pragma unroll
            for (int r = 0; r < 9; r++) {
                const uint32_t a = s_off[r][k - 1], b = s_off[r][k + 2];     // cells cx-1, cx, cx+1 of this row
                for (uint32_t e = a; e < b; e++) {
                    const float4 q = s_pts[e];
                    if (__float_as_uint(q.w) == self) continue;
                    knn_update3(me.x, me.y, me.z, q, best);
                }
            }
            next_ring = knn_done(G, me, cx, cy, cz, 1, best) ? -1 : 2;
        }
        if (next_ring >= 0) knn_rings(G, offset, count, sorted, me, self, cx, cy, cz, next_ring, best);
        out[self] = (best[0] + best[1] + best[2]) / 3.0f;
    }
  }
}

}  // namespace iso

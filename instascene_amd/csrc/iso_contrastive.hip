// Prototype-contrastive (ProtoNCE) loss — reference utils/contrastive_utils.py:18-73, entirely on the
// device: label filtering (:25-40), prototypes (:43-58), concentration (:60-66), similarity + loss (:68-71)
// and the full backward, with no host synchronisation (the reference calls torch.unique three times per
// loss, each a device->host sync) and no float atomics (bit-reproducible).
//
// Columns: the reference relabels the labels that survive filtering to dense ids.  Here column c is simply
// (label - 1) (or label when consider_negative) in [0, K) for a caller-supplied bound K; columns without a
// surviving sample are masked out of every sum, which is the same arithmetic as dropping them.
//
// Pipeline (all launches on the caller's stream):
//   ck_count       integer histogram of the raw labels                       (int atomics: exact)
//   ck_normalize   f = x/(|x|+1e-9), 1/(|x|+1e-9), column id or -1 per sample
//   ck_gemm_tn     cluster sums  onehot^T.f   on the matrix cores (exact-fp32 MFMA 32x32x2), split over samples
//   ck_finish_u    fixed-order sum of the split partials -> U (mean or predefined), present mask, counts
//   ck_phi         phi_c = clip(10 * sum|f-u_c| / (n_c log(n_c+lambda)), .5, 1)
//   ck_similarity  Z = f.U^T (MFMA), exp, masked row sums, G = softmax - onehot, loss partials
//   ck_loss_reduce fixed-order sum
// backward: ck_gemm_tn (G^T.f) + ck_finish_du, ck_grad_f (G.(U/phi) MFMA + prototype path, scaled by 1/(|x|+1e-9)).
#include "isr_common.hpp"

namespace iso {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CK_NSPLIT = 64;

struct CState {          // carved from the caller's state buffer
    float* f;            // [N,F] normalised features
    float* inv;          // [N]
    int* col;            // [N] column id or -1
    int* hist;           // [K+6] raw-label histogram (K+2 slots) + tickets: phi, loss, batch total (problem 0's)
    float* U;            // [K,F]
    float* phi;          // [K]
    float* cnt;          // [K] surviving samples per column (0 = column absent)
    float* G;            // [N,K] softmax - onehot (0 for dropped samples / absent columns)
    float* part;         // [blocks] loss partials
    float* dU;           // [K,F]
    float* split;        // [CK_NSPLIT, K, F] GEMM partials
    float* phi_part;     // [ceil(N/256), K] per-workgroup partial sums of |f - u|
    float* Us;           // [K,F] U / phi (0 for absent columns): B operand of the backward
};
inline CState cstate(void* buf, int N, int F, int K) {
    char* p = (char*)buf;
    CState s;
    s.f = isr::carve<float>(p, (size_t)N * F);
    s.inv = isr::carve<float>(p, N);
    s.col = isr::carve<int>(p, N);
    s.hist = isr::carve<int>(p, K + 6);
    s.U = isr::carve<float>(p, (size_t)K * F);
    s.phi = isr::carve<float>(p, K);
    s.cnt = isr::carve<float>(p, K);
    s.G = isr::carve<float>(p, (size_t)N * K);
    s.part = isr::carve<float>(p, (size_t)(N + 31) / 32 + 1);
    s.dU = isr::carve<float>(p, (size_t)K * F);
    s.split = isr::carve<float>(p, (size_t)CK_NSPLIT * K * F);
    s.phi_part = isr::carve<float>(p, (size_t)((N + 255) / 256) * K);
    s.Us = isr::carve<float>(p, (size_t)K * F);
    return s;
}
inline size_t cstate_bytes(int N, int F, int K) {
    CState s = cstate((void*)0, N, F, K);
    return (size_t)(s.Us + (size_t)K * F) + 256;
}

// Several losses of identical shape (N, F, K and flags) in ONE sequence of launches: every kernel takes the problem
// index from a spare grid dimension; the problems' state blocks are `stride` bytes apart, their inputs / outputs are
// separate tensors.  A single loss is a batch of one.
constexpr int CK_MAXB = 4;
struct CKBatch {
    long long stride;                   // bytes between consecutive problems' state blocks
    const float* x[CK_MAXB];            // features [N,F]
    const void* labels[CK_MAXB];        // [N] int64 / int32
    const float* predef[CK_MAXB];       // [K,F] predefined prototypes, or NULL (cluster means)
    float* dX[CK_MAXB];                 // backward: dL/dfeatures [N,F]
    float w[CK_MAXB];                   // loss weights
};
#define CK_AT(ptr, pb) ptr = (decltype(ptr))((const char*)(ptr) + (size_t)(pb) * (size_t)bt.stride)

__global__ __launch_bounds__(256) void ck_zero(int n, int* __restrict__ hist, CKBatch bt) {
    CK_AT(hist, blockIdx.y);
    for (int e = threadIdx.x; e < n; e += 256) hist[e] = 0;
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// raw labels are int64 (torch.long) or int32; shift = 1 unless consider_negative
__device__ __forceinline__ long long load_label(const void* labels, int is64, int i) {
    return is64 ? ((const long long*)labels)[i] : (long long)((const int*)labels)[i];
}

constexpr int CK_LDS_HIST = 2048;
__global__ __launch_bounds__(256) void ck_count(int N, int K, int shift, int is64, int* __restrict__ hist, CKBatch bt) {
    const void* labels = bt.labels[blockIdx.y];
    CK_AT(hist, blockIdx.y);
    // per-workgroup histogram in LDS, one global atomic per non-empty bin (a few dozen labels: 8192 global atomics
    // on ~65 addresses serialise in L2)
    __shared__ int s_h[CK_LDS_HIST];
    const bool lds = K + 2 <= CK_LDS_HIST;
    if (lds) {
        for (int b = threadIdx.x; b < K + 2; b += 256) s_h[b] = 0;
        __syncthreads();
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        const long long c = load_label(labels, is64, i) - shift;      // histogram slot c+1 (slot 0: label == shift-1, e.g. 0)
        if (c >= -1 && c < K) {
            if (lds) atomicAdd(&s_h[(int)(c + 1)], 1);
            else atomicAdd(hist + (int)(c + 1), 1);
        }
    }
    if (lds) {
        __syncthreads();
        for (int b = threadIdx.x; b < K + 2; b += 256)
            if (s_h[b] != 0) atomicAdd(hist + b, s_h[b]);
    }
}

// LPR lanes per sample (float4 each per step) when F % 4 == 0, one lane per sample otherwise
__global__ __launch_bounds__(256) void ck_normalize(int N, int F, int K, int shift, int consider_negative, int min_pixnum,
                                                    int is64, const int* __restrict__ hist, float* __restrict__ f,
                                                    float* __restrict__ inv, int* __restrict__ col, int lpr, CKBatch bt) {
    const float* x = bt.x[blockIdx.y];
    const void* labels = bt.labels[blockIdx.y];
    CK_AT(hist, blockIdx.y); CK_AT(f, blockIdx.y); CK_AT(inv, blockIdx.y); CK_AT(col, blockIdx.y);
    const int i = (int)(((long long)blockIdx.x * 256 + threadIdx.x) / lpr);
    const int sub = threadIdx.x & (lpr - 1);
    const bool ok_row = i < N;
    const float* xi = x + (size_t)(ok_row ? i : 0) * F;
    float s = 0.0f;
    if (lpr > 1) {
        const int q = F >> 2;
        for (int c = sub; c < q; c += lpr) {
            const float4 v = ok_row ? reinterpret_cast<const float4*>(xi)[c] : make_float4(0, 0, 0, 0);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int o = lpr >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    } else if (ok_row) {
        for (int ch = 0; ch < F; ch++) s += xi[ch] * xi[ch];
    }
    if (!ok_row) return;
    const float r = 1.0f / (__builtin_sqrtf(s) + 1e-9f);
    if (sub == 0) {
        const long long lab = load_label(labels, is64, i);
        const long long c = lab - shift;
        bool ok = (consider_negative || lab > 0) && c >= 0 && c < K;
        if (ok) ok = hist[(int)c + 1] > min_pixnum;
        col[i] = ok ? (int)c : -1;
        inv[i] = r;
    }
    if (lpr > 1) {
        const int q = F >> 2;
        float4* fo = reinterpret_cast<float4*>(f + (size_t)i * F);
        for (int c = sub; c < q; c += lpr) {
            const float4 v = reinterpret_cast<const float4*>(xi)[c];
            fo[c] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
        }
    } else {
        for (int ch = 0; ch < F; ch++) f[(size_t)i * F + ch] = xi[ch] * r;
    }
}

// C[k][c] (partial over a slice of the samples) = sum_i A[i][k] * f[i][c]
//   onehot != 0 : A[i][k] = (col[i] == k)      (cluster sums)
//   onehot == 0 : A[i][k] = G[i][k]            (prototype gradient)
// grid (ceil(K/32), ceil(F/32), CK_NSPLIT); the block's 4 waves stripe its slice; fixed-order combine.
__global__ __launch_bounds__(256) void ck_gemm_tn(int N, int F, int K, int onehot, const int* __restrict__ col,
                                                  const float* __restrict__ G, const float* __restrict__ f,
                                                  float* __restrict__ split, CKBatch bt) {
    const int pb = blockIdx.z / CK_NSPLIT, zs = blockIdx.z - pb * CK_NSPLIT;
    if (bt.predef[pb] != nullptr) return;           // predefined prototypes: no cluster sums, no prototype gradient
    CK_AT(col, pb); CK_AT(G, pb); CK_AT(f, pb); CK_AT(split, pb);
    __shared__ float s_acc[4][32][33];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int kidx = k0 + (lane & 31), cidx = c0 + (lane & 31), kk = lane >> 5;
    const int chunk = (N + CK_NSPLIT - 1) / CK_NSPLIT;
    const int i_lo = zs * chunk, i_hi = min(N, i_lo + chunk);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = i_lo + 2 * wv; i < i_hi; i += 8) {       // A[m = cluster][k = sample], B[k = sample][n = channel]
        const int smp = i + kk;
        float a = 0.0f, b = 0.0f;
        if (smp < i_hi) {
            if (kidx < K) a = onehot ? (col[smp] == kidx ? 1.0f : 0.0f) : G[(size_t)smp * K + kidx];
            if (cidx < F) b = f[(size_t)smp * F + cidx];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) s_acc[wv][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
        const int m = e >> 5, c = e & 31;
        if (k0 + m < K && c0 + c < F)
            split[((size_t)zs * K + k0 + m) * F + c0 + c] =
                (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]);
    }
}

// U = cluster mean (or predefined prototype), cnt = surviving samples per column
__global__ __launch_bounds__(256) void ck_finish_u(int F, int K, int min_pixnum, const int* __restrict__ hist,
                                                   const float* __restrict__ split, float* __restrict__ U,
                                                   float* __restrict__ cnt, CKBatch bt) {
    const float* predef = bt.predef[blockIdx.y];
    CK_AT(hist, blockIdx.y); CK_AT(split, blockIdx.y); CK_AT(U, blockIdx.y); CK_AT(cnt, blockIdx.y);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K * F) return;
    const int k = e / F;
    const int h = hist[k + 1];
    const float n = h > min_pixnum ? (float)h : 0.0f;     // every sample of a surviving label survives
    if (e == k * F) cnt[k] = n;
    float v = 0.0f;
    if (n > 0.0f) {
        if (predef != nullptr) v = predef[e];
        else {
            float s = 0.0f;
            for (int z = 0; z < CK_NSPLIT; z++) s += split[(size_t)z * K * F + e];
            v = s / n;
        }
    }
    U[e] = v;
}

__global__ __launch_bounds__(256) void ck_finish_du(int F, int K, const float* __restrict__ split,
                                                    const float* __restrict__ phi, const float* __restrict__ cnt,
                                                    float* __restrict__ dU, CKBatch bt) {
    if (bt.predef[blockIdx.y] != nullptr) return;
    CK_AT(split, blockIdx.y); CK_AT(phi, blockIdx.y); CK_AT(cnt, blockIdx.y); CK_AT(dU, blockIdx.y);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K * F) return;
    const int k = e / F;
    float s = 0.0f;
    for (int z = 0; z < CK_NSPLIT; z++) s += split[(size_t)z * K * F + e];
    dU[e] = cnt[k] > 0.0f ? s / phi[k] : 0.0f;
}

__device__ __forceinline__ float ld_agent(const float* p) {      // bypass the non-coherent per-CU cache
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// phi_c = clip(10 * sum_{i in c} |f_i - u_c| / (n_c log(n_c + lambda)), .5, 1)      (:60-66)
// One thread per sample computes its distance; the workgroup sums them per column in sample order (fixed order,
// no float atomics), and the LAST workgroup to finish adds the per-workgroup partials in workgroup order and writes
// phi and Us = U / phi.  (The previous layout — one workgroup per column scanning all N samples — was a 35 us
// latency chain for 8192 samples.)
__global__ __launch_bounds__(256) void ck_phi(int N, int F, int K, const float* __restrict__ f, const int* __restrict__ col,
                                              const float* __restrict__ U, const float* __restrict__ cnt,
                                              float temp_lambda, float* __restrict__ part, int* __restrict__ ticket,
                                              float* __restrict__ phi, float* __restrict__ Us, CKBatch bt) {
    CK_AT(f, blockIdx.y); CK_AT(col, blockIdx.y); CK_AT(U, blockIdx.y); CK_AT(cnt, blockIdx.y); CK_AT(part, blockIdx.y);
    CK_AT(ticket, blockIdx.y); CK_AT(phi, blockIdx.y); CK_AT(Us, blockIdx.y);
    __shared__ __attribute__((aligned(16))) float s_d[256];
    __shared__ __attribute__((aligned(16))) int s_c[256];
    __shared__ int s_last;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = i < N ? col[i] : -1;
    float d = 0.0f;
    if (c >= 0) {
        const float* fi = f + (size_t)i * F;
        const float* uc = U + (size_t)c * F;
        float s = 0.0f;
        if ((F & 3) == 0) {
            for (int q = 0; q < (F >> 2); q++) {
                const float4 a = reinterpret_cast<const float4*>(fi)[q], b = reinterpret_cast<const float4*>(uc)[q];
                const float t0 = a.x - b.x, t1 = a.y - b.y, t2 = a.z - b.z, t3 = a.w - b.w;
                s += t0 * t0; s += t1 * t1; s += t2 * t2; s += t3 * t3;
            }
        } else {
            for (int ch = 0; ch < F; ch++) { const float t = fi[ch] - uc[ch]; s += t * t; }
        }
        d = __builtin_sqrtf(s);
    }
    s_d[threadIdx.x] = d;
    s_c[threadIdx.x] = c;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        float acc = 0.0f;
#pragma unroll 4
        for (int j = 0; j < 256; j += 4) {      // 128-bit LDS broadcast reads; sample order
            const int4 cc = *reinterpret_cast<const int4*>(s_c + j);
            const float4 dd = *reinterpret_cast<const float4*>(s_d + j);
            if (cc.x == k) acc += dd.x;
            if (cc.y == k) acc += dd.y;
            if (cc.z == k) acc += dd.z;
            if (cc.w == k) acc += dd.w;
        }
        part[(size_t)blockIdx.x * K + k] = acc;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int k = threadIdx.x; k < K; k += 256) {
        float dsum = 0.0f;
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 8) {     // eight loads in flight, added in workgroup order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (b0 + u < gridDim.x) ? ld_agent(part + (size_t)(b0 + u) * K + k) : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++) dsum += v[u];
        }
        const float n = cnt[k];
        float p = 1.0f;
        if (n > 0.0f) {
            p = dsum / (n * __logf(n + temp_lambda)) * 10.0f;
            p = p < 0.5f ? 0.5f : (p > 1.0f ? 1.0f : p);
        }
        phi[k] = p;
        s_d[k & 255] = p;        // only used when K <= 256 (below)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * F; e += 256) {
        const int k = e / F;
        const float p = K <= 256 ? s_d[k] : ld_agent(phi + k);
        Us[e] = cnt[k] > 0.0f ? U[e] / p : 0.0f;
    }
}

// One wave (32 samples) per workgroup — N/32 small workgroups spread over all CUs instead of N/128 on a quarter of
// them.  A[i][k] = f[row i][chan k]; B[k][j] = U[proto j][chan k].
__global__ __launch_bounds__(64) void ck_similarity(int N, int F, int K, const float* __restrict__ f,
                                                     const float* __restrict__ U, const float* __restrict__ phi,
                                                     const float* __restrict__ cnt, const int* __restrict__ colid,
                                                     float* __restrict__ G, float* __restrict__ part, CKBatch bt) {
    CK_AT(f, blockIdx.y); CK_AT(U, blockIdx.y); CK_AT(phi, blockIdx.y); CK_AT(cnt, blockIdx.y); CK_AT(colid, blockIdx.y);
    CK_AT(G, blockIdx.y); CK_AT(part, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 32;
    const int arow = i0 + (lane & 31), kk = lane >> 5;
    const int ksteps = (F + 1) / 2, ntile = (K + 31) / 32;
    float rsum[16];
    int lab[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        rsum[r] = 0.0f;
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        lab[r] = row < N ? colid[row] : -1;
    }
    float lpart = 0.0f;
    for (int sweep = 0; sweep < 2; sweep++) {
        for (int jt = 0; jt < ntile; jt++) {
            const int col = jt * 32 + (lane & 31);
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 0; s < ksteps; s++) {
                const int ch = 2 * s + kk;
                const float a = (arow < N && ch < F) ? f[(size_t)arow * F + ch] : 0.0f;
                const float b = (col < K && ch < F) ? U[(size_t)col * F + ch] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            const bool present = col < K && cnt[col] > 0.0f;
            const float ph = present ? phi[col] : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float z = acc[r] / ph;
                const float e = (present && lab[r] >= 0) ? __expf(z) : 0.0f;
                if (sweep == 0) {
                    float t = e;   // sum over the 32 columns held by the 32 lanes of this half-wave
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) t += __shfl_xor(t, o);
                    rsum[r] += t;
                } else if (row < N && col < K) {
                    float g = 0.0f;
                    if (lab[r] >= 0 && present) {
                        const float S = rsum[r] + 1e-9f;
                        const bool pos = (col == lab[r]);
                        g = e / S - (pos ? 1.0f : 0.0f);
                        if (pos) lpart += __logf(S) - z;     // -log(e_pos / S)
                    }
                    G[(size_t)row * K + col] = g;
                }
            }
        }
    }
    float tot = lpart;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// Fast path of ck_similarity for F <= 32 and K <= 32*NT (NT <= 3): the sample fragment (A operand) is loaded once,
// all NT column tiles are kept in registers (one sweep instead of two), row sums use DPP adds inside a 16-lane row plus
// one cross-row exchange, and the LAST workgroup reduces the loss partials in workgroup order (no extra launch).
template <int CTRL>
__device__ __forceinline__ float ck_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {     // sum over the 32 lanes of this lane's half-wave
    v = ck_dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = ck_dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = ck_dpp_add<0x141>(v);      // row_half_mirror
    v = ck_dpp_add<0x140>(v);      // row_mirror
    return v + __shfl_xor(v, 16);
}

// A problem's loss is final: weight it, and let the LAST problem of the batch add the weighted losses in problem order.
__device__ __forceinline__ void ck_publish_loss(float t, int pb, int nb, float w, float* __restrict__ loss,
                                                float* __restrict__ loss_total, int* __restrict__ ticket_total) {
    loss[pb] = t * w;
    if (loss_total == nullptr) return;
    __threadfence();
    if (atomicAdd(ticket_total, 1) != nb - 1) return;
    __threadfence();
    float tot = ld_agent(loss);
    for (int p = 1; p < nb; p++) tot += ld_agent(loss + p);
    loss_total[0] = tot;
}

// LAST (rounds 3-4, now opt-in ISR_CK_LAST_MAX): the last workgroup to finish adds the loss partials and saves a launch - but every
// workgroup then pays an agent-scope release fence (on this part: a write-back of its XCD's L2) and an atomic on one ticket:
// 13 us of the C3 step at the 256 x 3 workgroups of 8 192 samples, 100 us at the 1 024 x 3 of the reference's default 32 768.
// Default: the partials are added by ck_loss_reduce in a launch of its own.
// FS = MFMA steps over the channels (two channels per step): 16 for F <= 32, 32 for F <= 64 (C5's feature width; the generic
// kernel reloads its operands from memory inside the channel loop and sweeps the columns twice: 58 us alone against 25).
template <int NT, bool LAST, int FS = 16>
__global__ __launch_bounds__(64) void ck_similarity_small(int N, int F, int K, const float* __restrict__ f,
                                                           const float* __restrict__ U, const float* __restrict__ phi,
                                                           const float* __restrict__ cnt, const int* __restrict__ colid,
                                                           float* __restrict__ G, float* __restrict__ part,
                                                           int* __restrict__ ticket, float* __restrict__ loss,
                                                           float* __restrict__ loss_total, int* __restrict__ ticket_total,
                                                           CKBatch bt) {
    CK_AT(f, blockIdx.y); CK_AT(U, blockIdx.y); CK_AT(phi, blockIdx.y); CK_AT(cnt, blockIdx.y); CK_AT(colid, blockIdx.y);
    CK_AT(G, blockIdx.y); CK_AT(part, blockIdx.y); CK_AT(ticket, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 32;
    const int arow = i0 + (lane & 31), kk = lane >> 5;
    float a[FS];
#pragma unroll
    for (int s = 0; s < FS; s++) {
        const int ch = 2 * s + kk;
        a[s] = (arow < N && ch < F) ? f[(size_t)arow * F + ch] : 0.0f;
    }
    int lab[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        lab[r] = row < N ? colid[row] : -1;
    }
    f32x16 acc[NT];
    bool present[NT];
    float ph[NT];
#pragma unroll
    for (int jt = 0; jt < NT; jt++) {
        const int col = jt * 32 + (lane & 31);
        float b[FS];
#pragma unroll
        for (int s = 0; s < FS; s++) {
            const int ch = 2 * s + kk;
            b[s] = (col < K && ch < F) ? U[(size_t)col * F + ch] : 0.0f;
        }
        present[jt] = col < K && cnt[col < K ? col : 0] > 0.0f;
        ph[jt] = present[jt] ? phi[col] : 1.0f;
        f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < FS; s++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
        acc[jt] = c;
    }
    float lpart = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float z[NT], e[NT], esum = 0.0f;
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            z[jt] = acc[jt][r] / ph[jt];
            e[jt] = (present[jt] && lab[r] >= 0) ? __expf(z[jt]) : 0.0f;
            esum += e[jt];
        }
        const float S = half_wave_sum(esum) + 1e-9f;
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            const int col = jt * 32 + (lane & 31);
            if (row < N && col < K) {
                float g = 0.0f;
                if (lab[r] >= 0 && present[jt]) {
                    const bool pos = (col == lab[r]);
                    g = e[jt] / S - (pos ? 1.0f : 0.0f);
                    if (pos) lpart += __logf(S) - z[jt];     // -log(e_pos / S)
                }
                G[(size_t)row * K + col] = g;
            }
        }
    }
    float tot = lpart;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
    if constexpr (!LAST) {
        if (lane == 0) part[blockIdx.x] = tot;
        return;
    }
    int last = 0;
    if (lane == 0) {
        part[blockIdx.x] = tot;
        __threadfence();
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    last = __shfl(last, 0);
    if (!last) return;
    __threadfence();
    float t = 0.0f;                      // fixed order: lane-strided, then the same butterfly
    for (unsigned b = lane; b < gridDim.x; b += 64) t += ld_agent(part + b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) ck_publish_loss(t, blockIdx.y, gridDim.y, bt.w[blockIdx.y], loss, loss_total, ticket_total);
}

__global__ __launch_bounds__(256) void ck_loss_reduce(int nblk, const float* __restrict__ part, float* __restrict__ loss,
                                                      float* __restrict__ loss_total, int* __restrict__ ticket_total,
                                                      CKBatch bt) {
    CK_AT(part, blockIdx.y);
    __shared__ float s_red[4];
    float a = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += part[i];
    const float t = block_sum_256(a, s_red);
    if (threadIdx.x == 0) ck_publish_loss(t, blockIdx.y, gridDim.y, bt.w[blockIdx.y], loss, loss_total, ticket_total);
}

// dF = G.(U/phi) [+ dU[y]/n_y];  dX = g * dF * inv.   one wave x 32 samples per workgroup, 32-channel tiles.
__global__ __launch_bounds__(64) void ck_grad_f(int N, int F, int K, const float* __restrict__ G,
                                                 const float* __restrict__ Us,
                                                 const float* __restrict__ cnt, const float* __restrict__ dU,
                                                 const int* __restrict__ colid, const float* __restrict__ inv,
                                                 const float* __restrict__ gloss, CKBatch bt) {
    CK_AT(G, blockIdx.y); CK_AT(Us, blockIdx.y); CK_AT(cnt, blockIdx.y); CK_AT(dU, blockIdx.y); CK_AT(colid, blockIdx.y);
    CK_AT(inv, blockIdx.y);
    float* dX = bt.dX[blockIdx.y];
    const bool use_mean = bt.predef[blockIdx.y] == nullptr;
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 32;
    const int arow = i0 + (lane & 31), kk = lane >> 5;
    const float g = gloss[0] * bt.w[blockIdx.y];
    for (int c0 = 0; c0 < F; c0 += 32) {
        const int ch = c0 + (lane & 31);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k0 = 0; k0 < K; k0 += 32) {       // A[m = sample][k = cluster], B[k = cluster][n = channel]; 32 clusters
            float a[16], b[16];                       // per round: all 32 loads are issued before the 16 MFMAs
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int kc = k0 + 2 * s + kk;
                a[s] = (arow < N && kc < K) ? G[(size_t)arow * K + kc] : 0.0f;
                b[s] = (kc < K && ch < F) ? Us[(size_t)kc * F + ch] : 0.0f;      // U / phi, 0 for absent columns
            }
#pragma unroll
            for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < N && ch < F) {
                float v = acc[r];
                const int y = colid[row];
                if (use_mean && y >= 0) v += dU[(size_t)y * F + ch] / cnt[y];
                dX[(size_t)row * F + ch] = y >= 0 ? g * v * inv[row] : 0.0f;
            }
        }
    }
}

}  // namespace iso

// ---------------------------------------------------------------------------
// Row normalisation y = x / (|x| + eps) and its exact backward — the reference applies it twice to the
// [P,F] feature every step in torch (scene/gaussian_model.py:122-125 eps 1e-6, gaussian_renderer/__init__.py:61-62
// eps 1e-9): ~10 elementwise/reduction launches forward+backward; here one streaming kernel each way.
namespace iso {

template <bool BWD>
__global__ __launch_bounds__(256) void rn_kernel(long long N, int F, float eps, const float* __restrict__ x,
                                                 const float* __restrict__ dy, float* __restrict__ out) {
    // LPR lanes cooperate on one row (float4 per lane per step); LPR = power of two >= F/4, <= 64
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q && lpr < 64) lpr <<= 1;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1);
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool ok = row < N;
    const float4* xr = reinterpret_cast<const float4*>(x + (ok ? row : 0) * F);
    const float4* dr = BWD ? reinterpret_cast<const float4*>(dy + (ok ? row : 0) * F) : nullptr;
    float ss = 0.0f, sd = 0.0f;
    for (int c = sub; c < q; c += lpr) {
        const float4 v = ok ? xr[c] : make_float4(0, 0, 0, 0);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (BWD) {
            const float4 g = ok ? dr[c] : make_float4(0, 0, 0, 0);
            sd += v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
        }
    }
    for (int o = lpr >> 1; o >= 1; o >>= 1) {
        ss += __shfl_xor(ss, o);
        if (BWD) sd += __shfl_xor(sd, o);
    }
    if (!ok) return;
    const float n = __builtin_sqrtf(ss);
    const float r = 1.0f / (n + eps);
    float4* o4 = reinterpret_cast<float4*>(out + row * F);
    if (!BWD) {
        for (int c = sub; c < q; c += lpr) {
            const float4 v = xr[c];
            o4[c] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
        }
    } else {
        const float k = n > 0.0f ? r * r * sd / n : 0.0f;     // d(1/(n+eps))/dx = -r^2 x/n ; subgradient 0 at x = 0
        for (int c = sub; c < q; c += lpr) {
            const float4 v = xr[c], g = dr[c];
            o4[c] = make_float4(r * g.x - k * v.x, r * g.y - k * v.y, r * g.z - k * v.z, r * g.w - k * v.w);
        }
    }
}

// Two chained row normalisations in one pass (F % 4 == 0, F <= 256: one float4 per lane, LPR lanes per row):
//   y = x / (|x| + eps1),  z = y / (|y| + eps2)
// forward : out1 = y, out2 = z (bit-identical to two rn_kernel<false> passes)
// backward: out1 = dL/dx given gy = dL/dy (may be null) and gz = dL/dz (may be null); x is read once, y and both
//           norms are recomputed in registers: 3 streams in, 1 out instead of 6 in, 3 out for the unfused chain.
template <bool BWD>
__global__ __launch_bounds__(256) void rn2_kernel(long long N, int F, float eps1, float eps2, const float* __restrict__ x,
                                                  const float* __restrict__ gy, const float* __restrict__ gz,
                                                  float* __restrict__ out1, float* __restrict__ out2) {
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q) lpr <<= 1;
    const int sub = (threadIdx.x & 63) & (lpr - 1);
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool ok = row < N && sub < q;
    const float4 z4 = make_float4(0, 0, 0, 0);
    const size_t off = (size_t)(row < N ? row : 0) * F + 4 * sub;
    float4 v = z4;                       // (`if`, not `ok ? *p : z4`: see k_feature_rows_step)
    if (ok) v = *reinterpret_cast<const float4*>(x + off);
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    if (!BWD) {
        for (int o = lpr >> 1; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        const float r1 = 1.0f / (__builtin_sqrtf(ss) + eps1);
        const float4 y = make_float4(v.x * r1, v.y * r1, v.z * r1, v.w * r1);
        float s2 = y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        for (int o = lpr >> 1; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
        const float r2 = 1.0f / (__builtin_sqrtf(s2) + eps2);
        if (ok) {
            if (out1 != nullptr) *reinterpret_cast<float4*>(out1 + off) = y;        // (y is optional: a sharded tail wants z only)
            *reinterpret_cast<float4*>(out2 + off) = make_float4(y.x * r2, y.y * r2, y.z * r2, y.w * r2);
        }
    } else {
        float4 a = z4, b = z4;
        if (ok && gy != nullptr) a = *reinterpret_cast<const float4*>(gy + off);
        if (ok && gz != nullptr) b = *reinterpret_cast<const float4*>(gz + off);
        float sa = v.x * a.x + v.y * a.y + v.z * a.z + v.w * a.w;      // <x, gy>
        float sb = v.x * b.x + v.y * b.y + v.z * b.z + v.w * b.w;      // <x, gz>
        for (int o = lpr >> 1; o >= 1; o >>= 1) {
            ss += __shfl_xor(ss, o); sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o);
        }
        const float nx = __builtin_sqrtf(ss), r1 = 1.0f / (nx + eps1);
        const float ny = nx * r1, r2 = 1.0f / (ny + eps2);
        // t = dL/dy through z:  r2*gz - k2*y,  k2 = r2^2 <y,gz>/|y|;   u = gy + t;   dx = r1*u - k1*x,  k1 = r1^2 <x,u>/|x|
        const float k2 = ny > 0.0f ? r2 * r2 * (r1 * sb) / ny : 0.0f;
        const float sxu = sa + r2 * sb - k2 * r1 * ss;
        const float k1 = nx > 0.0f ? r1 * r1 * sxu / nx : 0.0f;
        const float cy = k2 * r1;                                       // t = r2*gz - cy*x
        if (ok) {
            float4 o;
            o.x = r1 * (a.x + r2 * b.x - cy * v.x) - k1 * v.x;
            o.y = r1 * (a.y + r2 * b.y - cy * v.y) - k1 * v.y;
            o.z = r1 * (a.z + r2 * b.z - cy * v.z) - k1 * v.z;
            o.w = r1 * (a.w + r2 * b.w - cy * v.w) - k1 * v.w;
            *reinterpret_cast<float4*>(out1 + off) = o;
        }
    }
}

// Adam step on a [N,F] parameter fused with the two chained row normalisations of the updated rows (F % 4 == 0,
// F <= 256; one float4 per lane): the optimiser pass (4 streams in, 3 out) already has every new row in registers, so the
// next forward's y = p/(|p|+eps1), z = y/(|y|+eps2) cost two more streams out instead of a separate 1-in 2-out pass.
// Arithmetic of torch.optim.Adam (no weight decay, no amsgrad):
//   m = m + (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// (streamed once per step - as in k_feature_rows_step, the moments are loaded and x / m / v / z stored non-temporally)
typedef float iso_nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 iso_nt_load4(const float* p) {
    const iso_nt_f4 q = __builtin_nontemporal_load(reinterpret_cast<const iso_nt_f4*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ void iso_nt_store4(float* p, const float4& v) {
    const iso_nt_f4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<iso_nt_f4*>(p));
}
__global__ __launch_bounds__(256) void adam_rn2_kernel(long long N, int F, float lr_over_bc1, float om1, float beta2, float om2,
                                                       float inv_sqrt_bc2, float eps, float eps1, float eps2,
                                                       float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ y, float* __restrict__ z) {
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q) lpr <<= 1;
    const int sub = (threadIdx.x & 63) & (lpr - 1);
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool ok = row < N && sub < q;
    const size_t off = (size_t)(row < N ? row : 0) * F + 4 * sub;
    float4 np4 = make_float4(0, 0, 0, 0);
    if (ok) {
        const float4 p4 = *reinterpret_cast<const float4*>(p + off), g4 = *reinterpret_cast<const float4*>(g + off);
        float4 m4 = iso_nt_load4(m + off), v4 = iso_nt_load4(v + off);
#define ISO_ADAM1(c)                                                                   \
        m4.c = m4.c + om1 * (g4.c - m4.c);                                             \
        v4.c = beta2 * v4.c + om2 * (g4.c * g4.c);                                     \
        np4.c = p4.c - lr_over_bc1 * (m4.c / (__builtin_sqrtf(v4.c) * inv_sqrt_bc2 + eps));
        ISO_ADAM1(x) ISO_ADAM1(y) ISO_ADAM1(z) ISO_ADAM1(w)
#undef ISO_ADAM1
        iso_nt_store4(m + off, m4);
        iso_nt_store4(v + off, v4);
        iso_nt_store4(p + off, np4);
    }
    float ss = np4.x * np4.x + np4.y * np4.y + np4.z * np4.z + np4.w * np4.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float r1 = 1.0f / (__builtin_sqrtf(ss) + eps1);
    const float4 y4 = make_float4(np4.x * r1, np4.y * r1, np4.z * r1, np4.w * r1);
    float s2 = y4.x * y4.x + y4.y * y4.y + y4.z * y4.z + y4.w * y4.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
    const float r2 = 1.0f / (__builtin_sqrtf(s2) + eps2);
    if (ok) {
        if (y != nullptr) *reinterpret_cast<float4*>(y + off) = y4;
        iso_nt_store4(z + off, make_float4(y4.x * r2, y4.y * r2, y4.z * r2, y4.w * r2));
    }
}

// Sparse row gradient -> one entry per distinct row.  `vals[i, :]` is the gradient of row `idx[i]` of a [P, F] tensor
// (the backward of a row gather; indices repeat when rows were drawn with replacement).  The first occurrence of a row
// ("head") receives the sum of all its occurrences, in ascending i — the order of index_put_(accumulate=True) — and
// slot[row] = i points at merged[i, :].  n is a few thousand, so instead of sorting the index list every workgroup keeps
// all of it in LDS and its 64 samples are compared against everything: the sixteen waves scan a sixteenth each (one
// broadcast LDS read + one compare per candidate, a scalar branch skips the rest when no lane matches) and combine
// "smallest matching position" / "number of matches" with integer LDS atomics (order-independent).
constexpr int ROWS_COMPACT_MAX = 16384;
constexpr int ROWS_COMPACT_DUPS = 8;        // repeats of one row remembered per sample before falling back to a rescan
__global__ __launch_bounds__(1024) void rows_compact_kernel(int n, int F, long long P, const long long* __restrict__ idx,
                                                           const float* __restrict__ vals, int* __restrict__ slot,
                                                           float* __restrict__ merged) {
    __shared__ __attribute__((aligned(16))) int s_idx[ROWS_COMPACT_MAX];
    __shared__ int s_first[64], s_count[64], s_dup[64 * ROWS_COMPACT_DUPS];
    const int n4 = (n + 3) & ~3;
    for (int e0 = threadIdx.x; e0 < n4; e0 += 8 * 1024) {      // eight independent loads in flight per thread
        long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = e0 + u * 1024; v[u] = e < n ? idx[e] : -1; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * 1024;
            if (e < n4) s_idx[e] = (v[u] >= 0 && v[u] < P) ? (int)v[u] : -1 - e;   // invalid entries match nothing
        }
    }
    if (threadIdx.x < 64) { s_first[threadIdx.x] = 0x7fffffff; s_count[threadIdx.x] = 0; }
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const int mine = i < n ? s_idx[i] : -0x7fffffff;
    // each of the 16 waves scans every 16th block of four 16-byte groups: four independent LDS reads in flight per step
    const int4* s4 = reinterpret_cast<const int4*>(s_idx);
    for (int g = wv * 4; g < n4 / 4; g += 64) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (g + u < n4 / 4) ? s4[g + u] : make_int4(-2, -2, -2, -2);
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; u++) any = any | (v[u].x == mine) | (v[u].y == mine) | (v[u].z == mine) | (v[u].w == mine);
        if (__ballot(any) == 0ull) continue;
        if (any) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (e[t] == mine) {
                        const int j = 4 * (g + u) + t;
                        atomicMin(&s_first[lane], j);
                        const int k = atomicAdd(&s_count[lane], 1);
                        if (k < ROWS_COMPACT_DUPS) s_dup[lane * ROWS_COMPACT_DUPS + k] = j;
                    }
            }
        }
    }
    // the block's 64 rows are copied as they are (coalesced); heads with repeats are finished below
    const int i0 = blockIdx.x * 64, nrow = min(64, n - i0);
    for (int e = threadIdx.x; e < nrow * F; e += 1024) merged[(size_t)i0 * F + e] = vals[(size_t)i0 * F + e];
    __syncthreads();
    if (wv != 0) return;
    const bool head = i < n && mine >= 0 && s_first[lane] == i;
    if (head) slot[mine] = i;
    // heads whose row was drawn again (rare): the whole wave finishes them one at a time, lanes = channels
    unsigned long long todo = __ballot(head && s_count[lane] > 1);
    while (todo != 0ull) {
        const int L = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int hi = blockIdx.x * 64 + L, hrow = s_idx[hi], cnt = s_count[L];
        float* dst = merged + (size_t)hi * F;
        if (cnt <= ROWS_COMPACT_DUPS) {
            int prev = hi;                              // ascending positions: repeated selection of the next larger one
            for (int k = 1; k < cnt; k++) {
                int nxt = 0x7fffffff;
                for (int t = 0; t < cnt; t++) {
                    const int j = s_dup[L * ROWS_COMPACT_DUPS + t];
                    if (j > prev && j < nxt) nxt = j;
                }
                for (int c = lane; c < F; c += 64) dst[c] += vals[(size_t)nxt * F + c];
                prev = nxt;
            }
        } else {
            for (int j = hi + 1; j < n; j++)
                if (s_idx[j] == hrow)
                    for (int c = lane; c < F; c += 64) dst[c] += vals[(size_t)j * F + c];
        }
    }
}

// The same result in three small launches without the index list in LDS (the default: rows_compact_kernel's 1 024-thread /
// 64 KB workgroups wait for a compute unit with that much room whenever the next view's binning chain is running beside
// the step - 59 us on average in the C3 step against 16 us alone, profiles/r03c_kernel_stats_C3_seg.csv):
//   rows_first_kernel   slot[row] = min i over the occurrences of the row (unsigned atomicMin on the 0xFFFFFFFF-filled table:
//                       -1 stays "no entry");
//   rows_copy_kernel    merged = vals, and every later occurrence e of a row hangs itself into the row's chain: the slot's
//                       upper field becomes e, chain[e] = what it held before (0 = end of the chain: position 0 is always a
//                       first occurrence);
//   rows_merge_kernel   a wave per sample; only heads with a chain work: they walk it (<= ROWS_CHAIN_MAX hops), put the
//                       positions in ascending order - the order of index_put_(accumulate=True) and of rows_compact_kernel,
//                       hence the same bits - and add those rows; a longer chain falls back to a scan of the index list.
// Round 4 counted the repeats in the upper field and let the head of a row drawn three times or more scan the whole list for
// them: one such row (expected: four per launch at the reference's 32 768 samples) held the launch for 110-190 us.
// slot while the three kernels run: last chained position << pos_bits | first position.  pos_bits = 14 up to 16 384 samples
// and ceil(log2 n) beyond: both fields fit 32 bits up to ROWS_SPLIT_MAX samples (the reference's default sample_batchsize is
// 32 768, arguments/__init__.py:103), and 0xFFFFFFFF stays "no entry" (last > first).
constexpr int ROWS_SPLIT_MAX = 65536;
constexpr int ROWS_CHAIN_MAX = 48;          // occurrences of one row merged from its chain (one lane each); more: the scan
inline int rows_pos_bits(int n) { int b = 14; while ((1 << b) < n) b++; return b; }
__global__ __launch_bounds__(256) void rows_first_kernel(int n, long long P, const long long* __restrict__ idx, unsigned* __restrict__ slot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long v = idx[i];
    if (v >= 0 && v < P) atomicMin(slot + v, (unsigned)i);
}
__global__ __launch_bounds__(256) void rows_copy_kernel(int n, int F, long long P, const long long* __restrict__ idx,
                                                        const float* __restrict__ vals, unsigned* __restrict__ slot,
                                                        float* __restrict__ merged, int pos_bits, unsigned* __restrict__ chain) {
    const unsigned POS_MASK = (1u << pos_bits) - 1u;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)n * F;
    if ((F & 3) == 0) {
        if (4 * e < total) reinterpret_cast<float4*>(merged)[e] = reinterpret_cast<const float4*>(vals)[e];
    } else {
        for (long long k = 4 * e; k < total && k < 4 * e + 4; k++) merged[k] = vals[k];
    }
    if (e < n) {
        const long long v = idx[e];
        unsigned before = 0u;
        if (v >= 0 && v < P) {
            unsigned old = slot[v];
            if ((old & POS_MASK) != (unsigned)e) {           // a later occurrence: the new end of its row's chain
                while (true) {
                    const unsigned seen = atomicCAS(slot + v, old, ((unsigned)e << pos_bits) | (old & POS_MASK));
                    if (seen == old) break;
                    old = seen;
                }
                before = old >> pos_bits;
            }
        }
        chain[e] = before;
    }
}
__global__ __launch_bounds__(256) void rows_merge_kernel(int n, int F, long long P, const long long* __restrict__ idx,
                                                         const float* __restrict__ vals, unsigned* __restrict__ slot,
                                                         float* __restrict__ merged, int pos_bits, const unsigned* __restrict__ chain) {
    const unsigned POS_MASK = (1u << pos_bits) - 1u;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const long long v = idx[i];
    if (v < 0 || v >= P) return;
    const unsigned sv = slot[v];
    const unsigned first = sv & POS_MASK;
    unsigned e = sv >> pos_bits;
    if (e == 0u || (unsigned)i != first) return;                  // drawn once, or not the row's head
    float* dst = merged + (size_t)i * F;
    // walk the chain (arrival order); lane k keeps the k-th position
    unsigned mine = 0xffffffffu;
    int cnt = 0;
    while (e != 0u && cnt < ROWS_CHAIN_MAX) {
        if (lane == cnt) mine = e;
        cnt++;
        e = chain[e];
    }
    if (e == 0u) {
        // ascending position: rank of every kept position among the kept ones (positions are distinct)
        int rank = 0;
        for (int k = 0; k < cnt; k++) rank += (unsigned)__shfl((int)mine, k) < mine ? 1 : 0;
        for (int r = 0; r < cnt; r++) {
            const unsigned long long who = __ballot(lane < cnt && rank == r);
            const unsigned jj = (unsigned)__shfl((int)mine, __builtin_ctzll(who));
            for (int c = lane; c < F; c += 64) dst[c] += vals[(size_t)jj * F + c];
        }
    } else {
        // a row drawn more than ROWS_CHAIN_MAX times: scan the rest of the index list, U loads in flight per lane
        // (the row's sum lives in registers while the list goes by - F <= 256: four channels per lane - so that the adds of its
        // thousands of occurrences are not a chain of dependent read-modify-writes of global memory; same order, same bits)
        constexpr int U = 32;
        float acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = lane + 64 * k < F ? dst[lane + 64 * k] : 0.0f;
        for (int j0 = i + 1; j0 < n; j0 += 64 * U) {
            long long w[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + 64 * u + lane;
                w[u] = j < n ? idx[j] : -1;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                unsigned long long m = __ballot(w[u] == v);
                while (m != 0ull) {                 // ascending position
                    const int jj = j0 + 64 * u + __builtin_ctzll(m);
                    m &= m - 1ull;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (lane + 64 * k < F) acc[k] += vals[(size_t)jj * F + lane + 64 * k];
                    for (int c = lane + 256; c < F; c += 64) dst[c] += vals[(size_t)jj * F + c];      // (rows wider than 256: in memory)
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (lane + 64 * k < F) dst[lane + 64 * k] = acc[k];
    }
    if (lane == 0) slot[v] = (unsigned)i;
}

// out[i, :] = x[idx[i], :] / (|x[idx[i], :]| + eps): rows of the normalised table without the table (the same lanes-per-row
// layout and summation order as rn2_kernel / adam_rn2_kernel, hence the same bits as their `y` rows).  Rows with an index
// outside [0, P) are zero.
__global__ __launch_bounds__(256) void gather_rownorm_kernel(int n, int F, long long P, float eps, const float* __restrict__ x,
                                                             const long long* __restrict__ idx, float* __restrict__ out) {
    const int q = F >> 2;
    int lpr = 1;
    while (lpr < q) lpr <<= 1;
    const int sub = (threadIdx.x & 63) & (lpr - 1);
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) / lpr;
    const long long row = i < n ? idx[i] : -1;
    const bool ok = i < n && sub < q && row >= 0 && row < P;
    float4 v = make_float4(0, 0, 0, 0);
    if (ok) v = *reinterpret_cast<const float4*>(x + (size_t)row * F + 4 * sub);
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    for (int o = lpr >> 1; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float r1 = 1.0f / (__builtin_sqrtf(ss) + eps);
    if (i < n && sub < q)
        *reinterpret_cast<float4*>(out + (size_t)i * F + 4 * sub) = make_float4(v.x * r1, v.y * r1, v.z * r1, v.w * r1);
}

// The sampling of one train_semantic iteration (train_semantic.py:118-129,163-168,183-190) in ONE launch: 2*B labelled
// pixels of the view (B for each of the two single-view losses) with their labels from the two label maps, and B visible
// labelled Gaussians with their labels - eight torch kernels (randint x2, five gathers, an index_select) otherwise.
// Uniform draws with replacement from a counter-based generator: splitmix64 of (seed, step, draw), mapped to [0, n) by
// the high half of a 64 x 64-bit product.
__device__ __forceinline__ unsigned long long sm64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void sample_step_kernel(unsigned long long seed, unsigned long long step, int B,
                                                          long long n_pool2d, const long long* __restrict__ pool2d,
                                                          const long long* __restrict__ segmap_a,
                                                          const long long* __restrict__ segmap_b, long long n_pool3d,
                                                          const long long* __restrict__ pool3d,
                                                          const long long* __restrict__ labels3d,
                                                          long long* __restrict__ pix, long long* __restrict__ lab_a,
                                                          long long* __restrict__ lab_b, long long* __restrict__ pick3d,
                                                          long long* __restrict__ lab3d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * B) return;
    const unsigned long long r = sm64(sm64(seed ^ (step * 0xd1342543de82ef95ull)) + (unsigned long long)i);
    if (i < 2 * B) {
        if (n_pool2d <= 0) return;
        const long long q = pool2d[(long long)__umul64hi(r, (unsigned long long)n_pool2d)];
        pix[i] = q;
        if (i < B) lab_a[i] = segmap_a[q];
        else lab_b[i - B] = segmap_b[q];
    } else {
        if (n_pool3d <= 0) return;
        const int k = i - 2 * B;
        const long long g = pool3d[(long long)__umul64hi(r, (unsigned long long)n_pool3d)];
        pick3d[k] = g;
        lab3d[k] = labels3d[g];
    }
}

__global__ __launch_bounds__(256) void rn_scalar(long long N, int F, float eps, int bwd, const float* __restrict__ x,
                                                 const float* __restrict__ dy, float* __restrict__ out) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const float* xr = x + row * F;
    float ss = 0.0f, sd = 0.0f;
    for (int c = 0; c < F; c++) { ss += xr[c] * xr[c]; if (bwd) sd += xr[c] * dy[row * F + c]; }
    const float n = __builtin_sqrtf(ss), r = 1.0f / (n + eps);
    const float k = n > 0.0f ? r * r * sd / n : 0.0f;
    for (int c = 0; c < F; c++) out[row * F + c] = bwd ? r * dy[row * F + c] - k * xr[c] : xr[c] * r;
}

}  // namespace iso

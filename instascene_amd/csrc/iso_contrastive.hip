// Prototype-contrastive (ProtoNCE) loss core — reference utils/contrastive_utils.py:41-71.
//
// Deterministic, atomic-free pipeline:
//   ck_normalize   : f = x / (|x| + 1e-9), 1/(|x|+1e-9)            (thread per sample)
//   ck_clusters    : n_k, u_k (mean or predefined), phi_k           (workgroup per cluster, fixed-order sums)
//   ck_similarity  : Z = F.U^T on the matrix cores (exact-fp32 MFMA 32x32x2), exp, row sums,
//                    G = softmax - onehot, per-workgroup loss partials
//   ck_loss_reduce : fixed-order sum of the partials
// backward:
//   ck_grad_u      : dU = G^T.F / phi   (MFMA over the sample dimension, fixed-order combine)
//   ck_grad_f      : dF = G.(U/phi) (+ dU[y]/n_y when prototypes are computed), dX = g * dF / (|x|+1e-9)
#include "isr_common.hpp"

namespace iso {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CState {          // carved from the caller's state buffer
    float* f;            // [N,F] normalised features
    float* inv;          // [N]
    float* U;            // [K,F]
    float* phi;          // [K]
    float* cnt;          // [K] n_k as float
    float* G;            // [N,K] softmax - onehot
    float* part;         // [blocks] loss partials
    float* dU;           // [K,F]
};
inline CState cstate(void* buf, int N, int F, int K) {
    char* p = (char*)buf;
    CState s;
    s.f = isr::carve<float>(p, (size_t)N * F);
    s.inv = isr::carve<float>(p, N);
    s.U = isr::carve<float>(p, (size_t)K * F);
    s.phi = isr::carve<float>(p, K);
    s.cnt = isr::carve<float>(p, K);
    s.G = isr::carve<float>(p, (size_t)N * K);
    s.part = isr::carve<float>(p, (size_t)(N + 127) / 128 + 1);
    s.dU = isr::carve<float>(p, (size_t)K * F);
    return s;
}
inline size_t cstate_bytes(int N, int F, int K) {
    CState s = cstate((void*)0, N, F, K);
    return (size_t)(s.dU + (size_t)K * F) + 256;
}

__global__ __launch_bounds__(256) void ck_normalize(int N, int F, const float* __restrict__ x, float* __restrict__ f,
                                                    float* __restrict__ inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* xi = x + (size_t)i * F;
    float s = 0.0f;
    for (int c = 0; c < F; c++) s += xi[c] * xi[c];
    const float r = 1.0f / (__builtin_sqrtf(s) + 1e-9f);
    inv[i] = r;
    for (int c = 0; c < F; c++) f[(size_t)i * F + c] = xi[c] * r;
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// one workgroup per cluster; channels handled in chunks of 256 threads / members striped over threads
__global__ __launch_bounds__(256) void ck_clusters(int N, int F, const float* __restrict__ f,
                                                   const int32_t* __restrict__ labels, const float* __restrict__ predef,
                                                   float temp_lambda, float* __restrict__ U, float* __restrict__ phi,
                                                   float* __restrict__ cnt) {
    __shared__ float s_red[4];
    __shared__ float s_u[1024];
    const int k = blockIdx.x;
    float c = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) c += (labels[i] == k) ? 1.0f : 0.0f;
    const float n = block_sum_256(c, s_red);
    if (predef != nullptr) {
        for (int ch = threadIdx.x; ch < F; ch += 256) s_u[ch] = predef[(size_t)k * F + ch];
    } else {
        // mean: thread t owns channels t, t+256, ...; members visited in index order (deterministic)
        for (int ch = threadIdx.x; ch < F; ch += 256) {
            float a = 0.0f;
            for (int i = 0; i < N; i++)
                if (labels[i] == k) a += f[(size_t)i * F + ch];
            s_u[ch] = a / n;
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < F; ch += 256) U[(size_t)k * F + ch] = s_u[ch];
    float d = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        if (labels[i] != k) continue;
        float s = 0.0f;
        for (int ch = 0; ch < F; ch++) {
            const float t = f[(size_t)i * F + ch] - s_u[ch];
            s += t * t;
        }
        d += __builtin_sqrtf(s);
    }
    const float dsum = block_sum_256(d, s_red);
    if (threadIdx.x == 0) {
        float p = dsum / (n * __logf(n + temp_lambda)) * 10.0f;
        p = p < 0.5f ? 0.5f : (p > 1.0f ? 1.0f : p);
        phi[k] = p;
        cnt[k] = n;
    }
}

// 4 waves x 32 samples per workgroup.  A[i][k] = f[row i][chan k]; B[k][j] = U[proto j][chan k].
__global__ __launch_bounds__(256) void ck_similarity(int N, int F, int K, const float* __restrict__ f,
                                                     const float* __restrict__ U, const float* __restrict__ phi,
                                                     const int32_t* __restrict__ labels, float* __restrict__ G,
                                                     float* __restrict__ part) {
    __shared__ float s_red[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wv) * 32;
    const int arow = i0 + (lane & 31), kk = lane >> 5;
    const int ksteps = (F + 1) / 2, ntile = (K + 31) / 32;
    float rsum[16];
    int lab[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        rsum[r] = 0.0f;
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        lab[r] = row < N ? labels[row] : -1;
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
            const float ph = col < K ? phi[col] : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float z = acc[r] / ph;
                const float e = (col < K && row < N) ? __expf(z) : 0.0f;
                if (sweep == 0) {
                    float t = e;   // sum over the 32 columns held by the 32 lanes of this half-wave
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) t += __shfl_xor(t, o);
                    rsum[r] += t;
                } else if (row < N && col < K) {
                    const float S = rsum[r] + 1e-9f;
                    const bool pos = (col == lab[r]);
                    G[(size_t)row * K + col] = e / S - (pos ? 1.0f : 0.0f);
                    if (pos) lpart += __logf(S) - z;     // -log(e_pos / S)
                }
            }
        }
    }
    const float tot = block_sum_256(lpart, s_red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void ck_loss_reduce(int nb, const float* __restrict__ part, float* __restrict__ loss) {
    __shared__ float s_red[4];
    float a = 0.0f;
    for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
    const float t = block_sum_256(a, s_red);
    if (threadIdx.x == 0) loss[0] = t;
}

// dU[k][c] = sum_i G[i][k] f[i][c] / phi_k.  Workgroup per (32-cluster tile, 32-channel tile); its 4 waves stripe the samples.
__global__ __launch_bounds__(256) void ck_grad_u(int N, int F, int K, const float* __restrict__ f,
                                                 const float* __restrict__ G, const float* __restrict__ phi,
                                                 float* __restrict__ dU) {
    __shared__ float s_acc[4][32][33];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int kidx = k0 + (lane & 31), cidx = c0 + (lane & 31), kk = lane >> 5;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 2 * wv; i < N; i += 8) {       // A[m = cluster][k = sample], B[k = sample][n = channel]
        const int smp = i + kk;
        const float a = (smp < N && kidx < K) ? G[(size_t)smp * K + kidx] : 0.0f;
        const float b = (smp < N && cidx < F) ? f[(size_t)smp * F + cidx] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) s_acc[wv][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
        const int m = e >> 5, c = e & 31;
        if (k0 + m < K && c0 + c < F) {
            const float v = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]);
            dU[(size_t)(k0 + m) * F + c0 + c] = v / phi[k0 + m];
        }
    }
}

// dF = G.(U/phi) [+ dU[y]/n_y];  dX = g * dF * inv.   4 waves x 32 samples, 32-channel tiles.
__global__ __launch_bounds__(256) void ck_grad_f(int N, int F, int K, const float* __restrict__ G,
                                                 const float* __restrict__ U, const float* __restrict__ phi,
                                                 const float* __restrict__ cnt, const float* __restrict__ dU,
                                                 const int32_t* __restrict__ labels, const float* __restrict__ inv,
                                                 const float* __restrict__ gloss, int use_mean, float* __restrict__ dX) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wv) * 32;
    const int arow = i0 + (lane & 31), kk = lane >> 5;
    const float g = gloss[0];
    const int ksteps = (K + 1) / 2;
    for (int c0 = 0; c0 < F; c0 += 32) {
        const int ch = c0 + (lane & 31);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < ksteps; s++) {       // A[m = sample][k = cluster], B[k = cluster][n = channel]
            const int kc = 2 * s + kk;
            const float a = (arow < N && kc < K) ? G[(size_t)arow * K + kc] : 0.0f;
            const float b = (kc < K && ch < F) ? U[(size_t)kc * F + ch] / phi[kc] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < N && ch < F) {
                float v = acc[r];
                if (use_mean) {
                    const int y = labels[row];
                    v += dU[(size_t)y * F + ch] / cnt[y];
                }
                dX[(size_t)row * F + ch] = g * v * inv[row];
            }
        }
    }
}

}  // namespace iso

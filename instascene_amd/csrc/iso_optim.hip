// The optimiser step of the train.py loop (reference scene/gaussian_model.py:206-253: torch.optim.Adam(lr=0, eps=1e-15) over
// six parameter groups with their own learning rates; train.py:153-156) as ONE pass over the Gaussians, together with what
// surrounds it in every iteration:
//   * in front of it, the chain rule from the gradients of the ACTIVATED tensors the rasterizer consumed - exp(scaling),
//     sigmoid(opacity), normalize(rotation), cat(f_dc, f_rest) (scene/gaussian_model.py:109-138) - to the raw parameters
//     (autograd runs ~10 elementwise kernels for it);
//   * behind it, those activations of the UPDATED parameters for the next forward (7 more kernels and a concatenation).
// torch's fused Adam alone is 12 launches for the six groups (0.32 ms at 300 k Gaussians: 1.5 TB/s); everything here is one
// launch at stream speed.  Dense, like torch's: a Gaussian that was not visible (zero gradient) still decays its moments and
// moves by its momentum - skipping such rows would not be the reference's optimiser.
// Arithmetic of torch.optim.Adam (no weight decay, no amsgrad), as in adam_rn2_kernel:
//   m = m + (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#pragma once
#include <hip/hip_runtime.h>

namespace iso {

struct GaussAdamArgs {
    // groups: 0 xyz [P,3], 1 f_dc [P,3], 2 f_rest [P,R] (R = 3 * (M - 1)), 3 opacity [P,1], 4 scaling [P,2], 5 rotation [P,4]
    float* p[6];
    float* m[6];
    float* v[6];
    float lr_over_bc1[6];
    // gradients of the activated tensors: xyz [P,3], shs [P,M,3], sigmoid(opacity) [P,1], exp(scaling) [P,2],
    // normalize(rotation) [P,4]; a NULL entry leaves its group(s) untouched (torch skips parameters without .grad)
    const float* g_xyz;
    const float* g_shs;
    const float* g_opa;
    const float* g_scale;
    const float* g_rot;
    // activations of the updated parameters (any may be NULL): shs [P,M,3], opacity, scale, rotation
    float* a_shs;
    float* a_opa;
    float* a_scale;
    float* a_rot;
    float om1, beta2, om2, inv_sqrt_bc2, eps;
    int P, M;      // M = SH coefficients per colour (16 for degree 3); R = 3 (M - 1)
};

__device__ __forceinline__ float adam1(float& p, float g, float& m, float& v, float lr_over_bc1, const GaussAdamArgs& a) {
    m = m + a.om1 * (g - m);
    v = a.beta2 * v + a.om2 * (g * g);
    p = p - lr_over_bc1 * (m / (__builtin_sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
    return p;
}

// thread space: [0, 3P) xyz | [3P, 3P + 3MP) sh coefficients | P opacity | 2P scaling | P rotation rows
__global__ __launch_bounds__(256) void gaussian_adam_kernel(GaussAdamArgs a) {
    const long long P = a.P, S = 3LL * a.M;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < 3 * P) {                                        // xyz: identity activation
        if (a.g_xyz == nullptr) return;
        float p = a.p[0][e], m = a.m[0][e], v = a.v[0][e];
        adam1(p, a.g_xyz[e], m, v, a.lr_over_bc1[0], a);
        a.p[0][e] = p; a.m[0][e] = m; a.v[0][e] = v;
        return;
    }
    e -= 3 * P;
    if (e < S * P) {                                        // SH: shs = cat(f_dc, f_rest) along the coefficient axis
        const long long row = e / S;
        const int k = (int)(e - row * S);
        const int grp = k < 3 ? 1 : 2;
        const size_t off = k < 3 ? (size_t)row * 3 + k : (size_t)row * (S - 3) + (k - 3);
        float p = a.p[grp][off];
        if (a.g_shs != nullptr) {
            float m = a.m[grp][off], v = a.v[grp][off];
            adam1(p, a.g_shs[e], m, v, a.lr_over_bc1[grp], a);
            a.p[grp][off] = p; a.m[grp][off] = m; a.v[grp][off] = v;
        }
        if (a.a_shs != nullptr) a.a_shs[e] = p;
        return;
    }
    e -= S * P;
    if (e < P) {                                            // opacity: sigmoid
        float p = a.p[3][e];
        if (a.g_opa != nullptr) {
            const float s = 1.0f / (1.0f + expf(-p));
            float m = a.m[3][e], v = a.v[3][e];
            adam1(p, a.g_opa[e] * (s * (1.0f - s)), m, v, a.lr_over_bc1[3], a);
            a.p[3][e] = p; a.m[3][e] = m; a.v[3][e] = v;
        }
        if (a.a_opa != nullptr) a.a_opa[e] = 1.0f / (1.0f + expf(-p));
        return;
    }
    e -= P;
    if (e < 2 * P) {                                        // scaling: exp
        float p = a.p[4][e];
        if (a.g_scale != nullptr) {
            float m = a.m[4][e], v = a.v[4][e];
            adam1(p, a.g_scale[e] * expf(p), m, v, a.lr_over_bc1[4], a);
            a.p[4][e] = p; a.m[4][e] = m; a.v[4][e] = v;
        }
        if (a.a_scale != nullptr) a.a_scale[e] = expf(p);
        return;
    }
    e -= 2 * P;
    if (e < P) {                                            // rotation: y = q / max(|q|, 1e-12)  (torch.nn.functional.normalize)
        float4 q = reinterpret_cast<const float4*>(a.p[5])[e];
        if (a.g_rot != nullptr) {
            const float4 g = reinterpret_cast<const float4*>(a.g_rot)[e];
            const float n = fmaxf(__builtin_sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f), r = 1.0f / n;
            const float4 y = make_float4(q.x * r, q.y * r, q.z * r, q.w * r);
            const float d = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
            const float4 gq = make_float4((g.x - y.x * d) * r, (g.y - y.y * d) * r, (g.z - y.z * d) * r, (g.w - y.w * d) * r);
            float4 m = reinterpret_cast<const float4*>(a.m[5])[e], v = reinterpret_cast<const float4*>(a.v[5])[e];
            adam1(q.x, gq.x, m.x, v.x, a.lr_over_bc1[5], a);
            adam1(q.y, gq.y, m.y, v.y, a.lr_over_bc1[5], a);
            adam1(q.z, gq.z, m.z, v.z, a.lr_over_bc1[5], a);
            adam1(q.w, gq.w, m.w, v.w, a.lr_over_bc1[5], a);
            reinterpret_cast<float4*>(a.p[5])[e] = q;
            reinterpret_cast<float4*>(a.m[5])[e] = m;
            reinterpret_cast<float4*>(a.v[5])[e] = v;
        }
        if (a.a_rot != nullptr) {
            const float r = 1.0f / fmaxf(__builtin_sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            reinterpret_cast<float4*>(a.a_rot)[e] = make_float4(q.x * r, q.y * r, q.z * r, q.w * r);
        }
    }
}


// Direct exchange over mapped peer buffers (xGMI): dst[i] = sum over the W sources of src_w[begin + i], added in rank order
// - the reduce half of a reduce-scatter in which every rank PULLS its own shard from all peers at once (all seven links of
// an MI355X busy, where a ring keeps one busy per step).  The W pointers are device addresses of the peers' gradient
// buffers (hipIpcOpenMemHandle); a float4 per thread and source, fully coalesced.
struct PeerPtrs { const float* src[16]; };
__global__ __launch_bounds__(256) void peer_sum_kernel(int W, PeerPtrs p, long long begin, long long count, float* __restrict__ dst) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= count) return;
    if (i4 + 4 <= count && ((begin + i4) & 3) == 0) {
        float4 acc = *reinterpret_cast<const float4*>(p.src[0] + begin + i4);
        for (int w = 1; w < W; w++) {
            const float4 v = *reinterpret_cast<const float4*>(p.src[w] + begin + i4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + i4) = acc;
    } else {
        for (long long i = i4; i < count && i < i4 + 4; i++) {
            float acc = p.src[0][begin + i];
            for (int w = 1; w < W; w++) acc += p.src[w][begin + i];
            dst[i] = acc;
        }
    }
}

// ---- device-side phase flags of the direct exchange -------------------------------------------------------------------------------
// Every rank owns a small array of 32-bit generation counters in fine-grained device memory that its peers have mapped
// (iso_ipc_alloc / iso_ipc_open).  A rank publishes "phase ph of call g is complete on my stream" by a one-thread kernel on that
// stream (release at system scope, then the store), and consumes its peers' progress by a one-wave kernel on its own stream that
// polls their counters (acquire at system scope): the kernels enqueued behind it read what the peers wrote before their publish.
// No host round trip; the host only enqueues.  A poll gives up after `timeout_ms` (a peer died): it sets *status and returns, so a
// broken run ends with an error on the next host check instead of a wedged GPU.
__global__ void flag_set_kernel(unsigned* flag, unsigned value) {
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct FlagPtrs { const unsigned* f[16]; };
__global__ __launch_bounds__(64) void flag_wait_kernel(int W, FlagPtrs p, int skip, unsigned value, unsigned* status, long long timeout_ticks) {
    const int w = threadIdx.x;
    if (w < W && w != skip) {
        const long long t0 = wall_clock64();
        // generations only grow; compared as a signed distance so that the counter may wrap
        while ((int)(__hip_atomic_load(p.f[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > timeout_ticks) { atomicOr(status, 1u << w); break; }
        }
    }
    __threadfence_system();
}

// ---- compacted exchange: only the rows a rank touched -------------------------------------------------------------------------
// pack: the rows r with touched[r] != 0 of grad[P,F] -> idx[n], rows[n,F], *count = n (order = the order of the position atomics:
// irrelevant, every row appears once).  One wave per 64 rows; a lane copies a float4 column of a row.
__global__ __launch_bounds__(256) void rows_pack_kernel(int P, int F, const unsigned char* __restrict__ touched, const float* __restrict__ grad,
                                                        int* __restrict__ idx, float* __restrict__ rows, int* __restrict__ count) {
    __shared__ int s_base;
    __shared__ int s_n;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool on = r < P && touched[r] != 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int at = 0;
    if (on) at = atomicAdd(&s_n, 1);
    __syncthreads();
    if (threadIdx.x == 0) s_base = atomicAdd(count, s_n);
    __syncthreads();
    if (on) {
        const int pos = s_base + at;
        idx[pos] = r;
        const float* src = grad + (size_t)r * F;
        float* dst = rows + (size_t)pos * F;
        if ((F & 3) == 0)
            for (int c = 0; c < F; c += 4) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
        else
            for (int c = 0; c < F; c++) dst[c] = src[c];
    }
}
// dst[idx[e]][:] += rows[e][:] for e < *count (count, idx, rows may live in a peer's memory): rows of one list are distinct, so no
// atomics; lists are applied one launch after the other in rank order, i.e. every row is summed in rank order on every rank.
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(int F, int P, const int* __restrict__ count, const int* __restrict__ idx,
                                                               const float* __restrict__ rows, float* __restrict__ dst, int assign) {
    const int n = *count;
    const int q4 = (F + 3) / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * q4; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / q4), c = (int)(e - (long long)i * q4) * 4;
        const int r = idx[i];
        if (r < 0 || r >= P) continue;
        if ((F & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(rows + (size_t)i * F + c);
            float4* d = reinterpret_cast<float4*>(dst + (size_t)r * F + c);
            if (assign) *d = v;
            else { float4 o = *d; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; *d = o; }
        } else {
            for (int k = c; k < F && k < c + 4; k++) {
                const float v = rows[(size_t)i * F + k];
                if (assign) dst[(size_t)r * F + k] = v; else dst[(size_t)r * F + k] += v;
            }
        }
    }
}
}  // namespace iso

// Microbenchmark: rate of global atomic increments on MI355X as a function of memory scope, return value and address
// pattern (the binning kernels issue ~4.75 M of them per view at C3).  hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE, bool RET>
__global__ void k_atomics(int n, const unsigned* __restrict__ addr, unsigned* __restrict__ ctr, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned a = addr[(size_t)k * n + i];
        if (RET) acc += __hip_atomic_fetch_add(ctr + a, 1u, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(ctr + a, 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET) out[i] = acc;
}

int main(int argc, char** argv) {
    const int n = 1200000;
    const int slots = argc > 1 ? atoi(argv[1]) : 32640, stride = argc > 2 ? atoi(argv[2]) : 16;
    printf("slots %d stride %d\n", slots, stride);
    std::vector<unsigned> h((size_t)4 * n);
    unsigned *addr, *ctr, *out;
    hipMalloc(&addr, sizeof(unsigned) * 4 * n);
    hipMalloc(&ctr, sizeof(unsigned) * slots * stride);
    hipMalloc(&out, sizeof(unsigned) * n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pattern = 0; pattern < 2; pattern++) {
        srand(1);
        for (size_t i = 0; i < h.size(); i++) {
            const int gaussian = (int)(i % n);
            unsigned s = pattern == 0 ? (unsigned)(rand() % slots)                                     // random tiles
                                      : (unsigned)(((gaussian / 256) * 7 + rand() % 12) % slots);      // neighbours share tiles
            h[i] = s * stride;
        }
        hipMemcpy(addr, h.data(), sizeof(unsigned) * h.size(), hipMemcpyHostToDevice);
#define RUN(SC, RT, name)                                                                                   \
        for (int rep = 0; rep < 3; rep++) {                                                                  \
            hipMemset(ctr, 0, sizeof(unsigned) * slots * stride);                                            \
            hipEventRecord(e0);                                                                              \
            hipLaunchKernelGGL((k_atomics<SC, RT>), dim3((n + 255) / 256), dim3(256), 0, 0, n, addr, ctr, out); \
            hipEventRecord(e1); hipEventSynchronize(e1);                                                     \
            float ms; hipEventElapsedTime(&ms, e0, e1);                                                      \
            if (rep == 2) {                                                                                  \
                std::vector<unsigned> c(slots * stride);                                                     \
                hipMemcpy(c.data(), ctr, sizeof(unsigned) * c.size(), hipMemcpyDeviceToHost);                \
                unsigned long long tot = 0; for (unsigned v : c) tot += v;                                   \
                printf("pattern %d %-28s %.3f ms  %.1f G atomics/s  sum %llu (expect %d)\n", pattern, name, ms, \
                       4.0 * n / ms * 1e-6, tot, 4 * n);                                                     \
            }                                                                                                \
        }
        RUN(__HIP_MEMORY_SCOPE_AGENT, false, "agent, no return")
        RUN(__HIP_MEMORY_SCOPE_AGENT, true, "agent, returning")
    }
    return 0;
}

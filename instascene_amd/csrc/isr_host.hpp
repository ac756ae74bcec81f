// Host-side glue shared by the translation units of libinstascene_hip.so.
//
// The library is built from five translation units (Makefile, `make -j5`), each = the kernels of one stage plus the C-ABI
// entry points that launch them:
//   isr_api_forward.hip       K1 preprocess, the EXACT forward blend (K8), state read-back, profiling, error text
//   isr_api_binning.hip       K2-K7: scans, scatter into tile buckets, per-bucket sorts
//   isr_api_forward_fast.hip  the FAST forward blend (K8, k_render_fwd_fast)
//   isr_api_backward.hip      K9-K11: the blend backward kernels, row reductions, fused feature tail, markVisible
//   isr_api_ops.hip           include/instascene_ops.h: k-NN, contrastive loss, render() post-processing, SSIM, optimisers
// No device code crosses a unit (no -fgpu-rdc); what the units share on the host is declared here.
#pragma once
#include <atomic>

#include "isr_common.hpp"
#include "../../include/instascene_rasterizer.h"
#include "../../include/instascene_ops.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace isr {

extern thread_local char g_err[512];          // isr_api_forward.hip

static inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define ISR_HIP(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(ISR_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

// Debug mode (the reference's `debug=True`, DSR/cuda_rasterizer/auxiliary.h:297-304 CHECK_CUDA: synchronise after every
// launch and throw with the name of the kernel that faulted): on for the calling thread while isr_set_debug(1) holds,
// or for the process with ISR_DEBUG_SYNC=1.
extern thread_local int g_debug;              // isr_api_forward.hip
extern thread_local int g_fault_after;        // isr_set_debug's injected fault (tests)
static inline bool debug_sync() {
    static const bool env = [] { const char* e = getenv("ISR_DEBUG_SYNC"); return e && e[0] == '1'; }();
    return env || g_debug != 0;
}
// debug mode: the kernel just launched must have RUN without a fault before the next one is enqueued
static inline const char* debug_check(hipStream_t stream) {
    if (!debug_sync()) return nullptr;
    if (g_fault_after > 0 && --g_fault_after == 0) return "injected fault (isr_set_debug)";
    const hipError_t e = hipStreamSynchronize(stream);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#define ISR_STAGE(name, stream)                                                                             \
    do {                                                                                                    \
        const char* m_ = debug_check(stream);                                                               \
        if (m_ != nullptr) return fail(ISR_EHIP, "[debug] kernel %s failed: %s", name, m_);                 \
    } while (0)
// after a launch: the launch error, and in debug mode the execution error of the kernel itself
#define ISR_LAUNCH_CHECK_S(name, stream)                                                                    \
    do {                                                                                                    \
        hipError_t e_ = hipGetLastError();                                                                  \
        if (e_ != hipSuccess) return fail(ISR_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
        ISR_STAGE(name, stream);                                                                            \
    } while (0)
#define ISR_LAUNCH_CHECK(name) ISR_LAUNCH_CHECK_S(name, s)

// ---- isr_api_binning.hip
// exclusive u32 scan of in[0, n) into out, sums = scratch of ceil(n / 1024) + 1 words
int launch_scan_u32(int n, const uint32_t* in, uint32_t* out, uint32_t* sums, hipStream_t s);
int launch_prepare_scans(int P, int T, const GeomView& g, const ImageView& iv, hipStream_t s);

// ---- isr_api_forward_fast.hip
extern thread_local unsigned long long* g_fwd_counters;     // isr_forward_set_counters: consumed by the next FAST forward
extern std::atomic<unsigned long long*> g_bwd_counters;     // isr_backward_set_counters: consumed by the next k_render_bwd_geo launch (any
                                                            // host thread: torch runs backward passes on its own device threads)
int launch_render_fwd_fast(int P, int tiles, hipStream_t s, int W, int H, int ED, int gx, const ImageView& iv, const BinView& bv,
                           const float* rec, const float* cull, const float* col_pre, const float* tm_pre, const float* extras,
                           const float* bg, float* out_color, float* out_others, float* out_extra, int32_t* tracer,
                           long long tcap, int32_t* tcount, int64_t capacity, bool aux, const float* xscale = nullptr);

// ---- isr_api_backward.hip
size_t backward_scratch_bytes(int64_t R, int ED, unsigned mask);
size_t backward_sampled_scratch_bytes(int64_t R, int ED, int n, int W, int H);

}  // namespace isr

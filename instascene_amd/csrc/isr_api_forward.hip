// C-ABI entry points of libinstascene_hip.so, forward half: K1-K8 (declared in include/instascene_rasterizer.h).
// Host-side only: argument checking, workspace carving and kernel launches on the
// caller's stream.  No torch types, no allocation, no hidden synchronisation except
// where the header says so.
#include <type_traits>
#include "isr_host.hpp"
#include "isr_forward.hip"    // the unit's kernels are defined before the entry points that launch them

namespace isr {

thread_local char g_err[512] = "";
thread_local int g_debug = 0;
thread_local int g_fault_after = 0;

template <class Math>
static int launch_render_fwd(int tiles, hipStream_t s, int W, int H, int ED, int gx, const ImageView& iv,
                             const BinView& bv, const float* rec, const float* cull, const float* col_pre, const float* tm_pre,
                             const float* extras, const float* bg, float* out_color, float* out_others, float* out_extra,
                             int32_t* tracer, long long tcap, int32_t* tcount, int64_t capacity) {
    // first pass: geometry/colour/aux + the first feature chunk; further passes add 32 channels each
    int ch = 0, first = 1;
    static const int per_block = [] { const char* e = getenv("ISR_FWD_WAVE"); return e ? atoi(e) : 1; }();
    if (per_block && std::is_same<Math, ExactMath>::value) {
        // k_render_fwd_w: one wave per 8x8 block (isr_forward_fast.hip's decomposition), longest lists first
        const int grid = (tiles + 7) / 8 * 32;
        do {
            ProfScope ps_("k_render_fwd", s);
            const int rem = ED - ch;
#define ISR_GOW(F)                                                                                                    \
    hipLaunchKernelGGL((k_render_fwd_w<F>), dim3(grid), dim3(64), 0, s, W, H, ED, ch, first, gx, tiles, iv.tile_offset,  \
                       bv.point_list, rec, col_pre, tm_pre, extras, bg, iv.final_T, iv.n_contrib, out_color, out_others, \
                       out_extra, tracer, tcap, tcount, bv.hit_mask, capacity, iv.tile_order)
            if (rem <= 0) ISR_GOW(0);
            else if (rem <= 8) ISR_GOW(8);
            else if (rem <= 16) ISR_GOW(16);
            else ISR_GOW(32);
#undef ISR_GOW
            ISR_LAUNCH_CHECK("k_render_fwd_w");
            ch += MAX_FCHUNK;
            first = 0;
        } while (ch < ED);
        return ISR_OK;
    }
    do {
        ProfScope ps_("k_render_fwd", s);
        const int rem = ED - ch;
#define ISR_GO(F, B)                                                                                                 \
    hipLaunchKernelGGL((k_render_fwd<Math, F, B>), dim3(tiles), dim3(256), 0, s, W, H, ED, ch, first, gx,             \
                       iv.tile_offset, bv.point_list, rec, cull, col_pre, tm_pre, extras, bg, iv.final_T, iv.n_contrib,    \
                       out_color, out_others, out_extra, tracer, tcap, tcount, bv.box4, capacity)
        if (rem <= 0) ISR_GO(0, 256);
        else if (rem <= 8) ISR_GO(8, 256);
        else if (rem <= 16) ISR_GO(16, 256);
        else ISR_GO(32, 128);
#undef ISR_GO
        ISR_LAUNCH_CHECK("k_render_fwd");
        ch += MAX_FCHUNK;
        first = 0;
    } while (ch < ED);
    return ISR_OK;
}

}  // namespace isr

using namespace isr;

extern "C" {

const char* isr_last_error(void) { return g_err; }
int isr_version(void) { return 1; }

int isr_set_debug(int on, int fault_after) {
    const int was = g_debug;
    g_debug = on != 0;
    g_fault_after = (on != 0 && fault_after > 0) ? fault_after : 0;
    return was;
}

void isr_profile_enable(int on) {
    Prof& p = prof();
    for (auto& r : p.recs) { p.pool.push_back(r.a); p.pool.push_back(r.b); }
    p.recs.clear();
    p.on = on != 0;
    p.dominant_only = on == 2;
}

/* Writes "name count total_ms" lines for everything recorded since isr_profile_enable(1); synchronises
 * on the recorded events.  Returns the number of bytes written (0 if nothing / buffer too small). */
size_t isr_profile_summary(char* buf, size_t len) {
    Prof& p = prof();
    std::vector<std::string> names;
    std::vector<double> tot;
    std::vector<int> cnt;
    for (auto& r : p.recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        size_t k = 0;
        for (; k < names.size(); k++) if (names[k] == r.name) break;
        if (k == names.size()) { names.push_back(r.name); tot.push_back(0); cnt.push_back(0); }
        tot[k] += ms; cnt[k] += 1;
    }
    std::string out;
    for (size_t k = 0; k < names.size(); k++) {
        char line[256];
        snprintf(line, sizeof(line), "%s %d %.6f\n", names[k].c_str(), cnt[k], tot[k]);
        out += line;
    }
    if (out.size() + 1 > len) return 0;
    memcpy(buf, out.c_str(), out.size() + 1);
    return out.size();
}

size_t isr_geom_bytes(int P) { return geom_bytes(P < 1 ? 1 : P); }
size_t isr_image_bytes(int width, int height) { return image_bytes(width, height); }
size_t isr_binning_bytes(int64_t num_rendered, int width, int height) { return bin_bytes(num_rendered, tiles_x(width) * tiles_y(height)); }

int isr_forward_prepare(int P, int D, int M, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                        const float* projmatrix, const float* cam_pos, float, float, int prefiltered, int* radii,
                        void* geom_buffer, void* image_buffer, int64_t* num_rendered_host, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || width <= 0 || height <= 0) return fail(ISR_EINVAL, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (!geom_buffer || !image_buffer || (P > 0 && !radii)) return fail(ISR_EINVAL, "null workspace/radii");
    if (P > 0 && (!means3D || !opacities || !viewmatrix || !projmatrix))
        return fail(ISR_EINVAL, "means3D/opacities/viewmatrix/projmatrix must be given");
    if ((shs == nullptr) == (colors_precomp == nullptr) && P > 0)
        return fail(ISR_EINVAL, "provide exactly one of shs / colors_precomp");
    if (P > 0 && (((scales == nullptr) || (rotations == nullptr)) == (transMat_precomp == nullptr)))
        return fail(ISR_EINVAL, "provide exactly one of (scales, rotations) / transMat_precomp");
    if (shs && !cam_pos) return fail(ISR_EINVAL, "cam_pos required with shs");
    if (shs && (M < 1 || (D > 0 && M < 4) || (D > 1 && M < 9) || (D > 2 && M < 16) || D > 3))
        return fail(ISR_EINVAL, "sh degree %d needs more coefficients than M=%d (max degree 3)", D, M);
    const int gx = tiles_x(width), gy = tiles_y(height), T = gx * gy;
    if (gx > 65535 || gy > 65535) return fail(ISR_EINVAL, "image too large for 16-bit tile coordinates");
    GeomView g = geom_view(geom_buffer, P < 1 ? 1 : P);
    ImageView iv = image_view(image_buffer, width, height);
    ISR_HIP(hipMemsetAsync(iv.tile_count, 0, sizeof(uint32_t) * (size_t)T * CNT_SUB * CNT_STRIDE, s));
    if (P > 0) {
        { ProfScope ps_("k_preprocess", s);
        const int tight = (prefiltered & ISR_PREPARE_TIGHT_RECTS) ? 1 : 0;
        if (M == 16 && colors_precomp == nullptr && shs != nullptr)      // SH rows staged through LDS (coalesced reads)
            hipLaunchKernelGGL(k_preprocess<true>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, scales,
                               scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, viewmatrix,
                               projmatrix, cam_pos, width, height, gx, gy, radii, g, iv.tile_count, tight);
        else
            hipLaunchKernelGGL(k_preprocess<false>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, scales,
                               scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, viewmatrix,
                               projmatrix, cam_pos, width, height, gx, gy, radii, g, iv.tile_count, tight); }
        ISR_LAUNCH_CHECK("k_preprocess");
    }
    if (launch_prepare_scans(P, T, g, iv, s) != 0) return fail(ISR_EHIP, "launch of the prepare scans failed");
    ISR_STAGE("k_scan / k_tile_scan", s);
    if (num_rendered_host) return isr_read_num_rendered(geom_buffer, num_rendered_host, stream);
    return ISR_OK;
}

int isr_read_num_rendered(const void* geom_buffer, int64_t* num_rendered_host, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!geom_buffer || !num_rendered_host) return fail(ISR_EINVAL, "null argument");
    ISR_HIP(hipMemcpyAsync(num_rendered_host, geom_buffer, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    ISR_HIP(hipStreamSynchronize(s));
    return ISR_OK;
}

int isr_forward_render(int P, int ED, int width, int height, int mode, const float* background,
                       const float* colors_precomp, const float* transMat_precomp, const float* extra_attrs,
                       void* geom_buffer, void* binning_buffer, int64_t binning_capacity, void* image_buffer,
                       float* out_color, float* out_others, float* out_extra, int32_t* tracer_pairs,
                       int64_t tracer_capacity, int32_t* tracer_count, void* stream) {
    return isr_forward_render_scaled(P, ED, width, height, mode, background, colors_precomp, transMat_precomp, extra_attrs, nullptr,
                                     geom_buffer, binning_buffer, binning_capacity, image_buffer, out_color, out_others, out_extra,
                                     tracer_pairs, tracer_capacity, tracer_count, stream);
}

int isr_forward_render_scaled(int P, int ED, int width, int height, int mode, const float* background,
                              const float* colors_precomp, const float* transMat_precomp, const float* extra_attrs,
                              const float* extra_row_scale, void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                              void* image_buffer, float* out_color, float* out_others, float* out_extra, int32_t* tracer_pairs,
                              int64_t tracer_capacity, int32_t* tracer_count, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool prebinned = (mode & ISR_MODE_PREBINNED) != 0;
    const bool feature_only = (mode & ISR_MODE_FEATURE_ONLY) != 0;
    mode &= ~(ISR_MODE_PREBINNED | ISR_MODE_FEATURE_ONLY);
    if (feature_only && (mode != ISR_MODE_FAST || ED <= 0)) return fail(ISR_EINVAL, "ISR_MODE_FEATURE_ONLY needs ISR_MODE_FAST and ED > 0");
    if (!geom_buffer || !binning_buffer || !image_buffer || !background || (!feature_only && (!out_color || !out_others)))
        return fail(ISR_EINVAL, "null buffer");
    if (ED < 0 || (ED > 0 && (!extra_attrs || !out_extra))) return fail(ISR_EINVAL, "extra_attrs/out_extra required when ED>0");
    if (mode != ISR_MODE_EXACT && mode != ISR_MODE_FAST) return fail(ISR_EINVAL, "unknown mode %d", mode);
    if (tracer_pairs && !tracer_count) return fail(ISR_EINVAL, "tracer_count required with tracer_pairs");
    const int gx = tiles_x(width), gy = tiles_y(height), T = gx * gy;
    GeomView g = geom_view(geom_buffer, P < 1 ? 1 : P);
    ImageView iv = image_view(image_buffer, width, height);
    BinView bv = bin_view(binning_buffer, binning_capacity);
    if (tracer_pairs) ISR_HIP(hipMemsetAsync(tracer_count, 0xFF, sizeof(int32_t), s));     // -1: the counter ends at (pairs - 1)
    if (!prebinned) {
        const int rc = isr_forward_bin(P, width, height, geom_buffer, binning_buffer, binning_capacity, image_buffer, stream);
        if (rc != ISR_OK) return rc;
    }
    if (extra_row_scale != nullptr && (mode != ISR_MODE_FAST || ED <= 0))
        return fail(ISR_EINVAL, "extra_row_scale needs ISR_MODE_FAST and ED > 0");
    if (mode == ISR_MODE_EXACT)
        return launch_render_fwd<ExactMath>(T, s, width, height, ED, gx, iv, bv, g.rec, g.cull, colors_precomp, transMat_precomp,
                                            extra_attrs, background, out_color, out_others, out_extra, tracer_pairs,
                                            (long long)tracer_capacity, tracer_count, binning_capacity);
    return launch_render_fwd_fast(P, T, s, width, height, ED, gx, iv, bv, g.rec, g.cull, colors_precomp, transMat_precomp,
                                  extra_attrs, background, out_color, out_others, out_extra, feature_only ? nullptr : tracer_pairs,
                                  (long long)tracer_capacity, tracer_count, binning_capacity, !feature_only, extra_row_scale);
}

int isr_debug_state(int P, int width, int height, int64_t num_rendered, const void* geom_buffer,
                    const void* binning_buffer, const void* image_buffer, uint32_t* tiles_touched, uint32_t* point_list,
                    uint32_t* ranges, uint32_t* n_contrib, float* final_T, float* splat_records, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    GeomView g = geom_view(const_cast<void*>(geom_buffer), P < 1 ? 1 : P);
    ImageView iv = image_view(const_cast<void*>(image_buffer), width, height);
    const size_t N = (size_t)width * height, T = (size_t)tiles_x(width) * tiles_y(height);
    if (tiles_touched && P) ISR_HIP(hipMemcpyAsync(tiles_touched, g.tiles_touched, 4 * (size_t)P, hipMemcpyDeviceToHost, s));
    if (splat_records && P) ISR_HIP(hipMemcpyAsync(splat_records, g.rec, 4 * (size_t)P * REC, hipMemcpyDeviceToHost, s));
    if (point_list && num_rendered > 0 && binning_buffer) {
        BinView bv = bin_view(const_cast<void*>(binning_buffer), num_rendered);
        ISR_HIP(hipMemcpyAsync(point_list, bv.point_list, 4 * (size_t)num_rendered, hipMemcpyDeviceToHost, s));
    }
    if (n_contrib) ISR_HIP(hipMemcpyAsync(n_contrib, iv.n_contrib, 8 * N, hipMemcpyDeviceToHost, s));
    if (final_T) ISR_HIP(hipMemcpyAsync(final_T, iv.final_T, 12 * N, hipMemcpyDeviceToHost, s));
    ISR_HIP(hipStreamSynchronize(s));
    if (ranges) {
        uint32_t* off = new uint32_t[T + 1];
        hipError_t e = hipMemcpy(off, iv.tile_offset, 4 * (T + 1), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete[] off; return fail(ISR_EHIP, "copy of tile offsets failed"); }
        for (size_t t = 0; t < T; t++) {
            const bool empty = off[t] == off[t + 1];
            ranges[2 * t] = empty ? 0u : off[t];        // the reference leaves empty tiles at (0,0)
            ranges[2 * t + 1] = empty ? 0u : off[t + 1];
        }
        delete[] off;
    }
    return ISR_OK;
}

}  // extern "C"

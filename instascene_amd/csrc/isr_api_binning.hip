// libinstascene_hip.so, binning (K2-K7): scans, scatter, per-tile sort.  Host side: isr_host.hpp.
#include "isr_host.hpp"
#include "isr_binning.hip"

namespace isr {

int launch_scan_u32(int n, const uint32_t* in, uint32_t* out, uint32_t* sums, hipStream_t s) {
    const int nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, s, n, in, out, sums, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_scan_add_tops, dim3(nb), dim3(1024), 0, s, n, nb, out, sums, (const uint32_t*)nullptr, (int64_t*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}


// The Gaussian offset scan (tiles touched -> row offsets of the backward's partial rows) is not an input of the binning chain:
// only its by-product header[1] is (k_scatter).  By default it is therefore launched BEHIND the chain (isr_forward_bin_event's
// last launch: a trainer's side stream reaches the hit masks 19 us (C3) / 68 us (C5) sooner) and k_gather_counts reduces
// header[1].  ISR_SCAN_LATE=0 or ISR_SCAN_LAUNCHES > 1: inside isr_forward_prepare as in rounds 1-5.  Either way ONE
// isr_forward_bin per isr_forward_prepare (the scatter cursors are consumed by it, and the late scan adds in place).
static int scan_launches() {
    static const int launches = [] { const char* e = getenv("ISR_SCAN_LAUNCHES"); return e ? atoi(e) : 1; }();
    return launches;
}
bool scan_late() {
    static const bool on = [] { const char* e = getenv("ISR_SCAN_LATE"); return !(e && e[0] == '0'); }();
    return on && scan_launches() <= 1;
}
int launch_late_scan(int P, const GeomView& g, hipStream_t s) {
    if (P <= 0 || !scan_late()) return 0;
    const int nb = (P + 1023) / 1024, nb256 = (P + 255) / 256;
    ProfScope ps2_("k_scan_gaussians", s);
    static const bool two = [] { const char* e = getenv("ISR_SCAN_LATE_TWO"); return !(e && e[0] == '0'); }();
    if (two) {
        hipLaunchKernelGGL(k_scan_tops_inplace, dim3(1), dim3(1024), 0, s, nb256, g.scan_tmp);
        hipLaunchKernelGGL(k_scan_add256, dim3(nb256), dim3(256), 0, s, P, g.point_offsets, g.scan_tmp);
    } else {
        hipLaunchKernelGGL(k_scan_add_tops256, dim3(nb), dim3(1024), 0, s, P, nb256, g.point_offsets, g.scan_tmp, g.scan_tmp + nb256 + 1, (int64_t*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the two scans of isr_forward_prepare: Gaussians (tiles touched -> row offsets, and the largest rectangle into header[1])
// and tiles (sub-counter totals -> bucket offsets, launch order)
int launch_prepare_scans(int P, int T, const GeomView& g, const ImageView& iv, hipStream_t s) {
    if (P > 0) {
        const int nb = (P + 1023) / 1024;
        ProfScope ps2_("k_scan_gaussians", s);
        // ISR_SCAN_LAUNCHES: 1 (default) K1 has scanned inside its workgroups, one launch adds the totals; 2 / 3: the scan of
        // rounds 1-3 in two / three launches of its own (they overwrite what K1 wrote)
        const int launches = scan_launches();
        if (launches <= 1) {
            const int nb256 = (P + 255) / 256;
            if (!scan_late())
                hipLaunchKernelGGL(k_scan_add_tops256, dim3(nb), dim3(1024), 0, s, P, nb256, g.point_offsets, g.scan_tmp, g.scan_tmp + nb256 + 1, g.header);
        } else {
            hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, s, P, g.tiles_touched, g.point_offsets, g.scan_tmp, g.scan_tmp + nb + 1);
            if (launches >= 3) {
                hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, nb, g.scan_tmp, g.scan_tmp + nb + 1, g.header);
                hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, s, P, g.point_offsets, g.scan_tmp);
            } else {
                hipLaunchKernelGGL(k_scan_add_tops, dim3(nb), dim3(1024), 0, s, P, nb, g.point_offsets, g.scan_tmp, g.scan_tmp + nb + 1, g.header);
            }
        }
    }
    static const int order_classes = [] { const char* e = getenv("ISR_ORDER_CLASSES"); return e ? atoi(e) : 16; }();
    { ProfScope ps3_("k_tile_scan", s);
    const bool late = P > 0 && scan_late();
    const int nb256_ = (P + 255) / 256;
    hipLaunchKernelGGL(k_gather_counts, dim3((T * CNT_SUB + 255) / 256), dim3(256), 0, s, T * CNT_SUB, iv.tile_count, iv.sub_offset, iv.tile_cursor,
                       late ? (const uint32_t*)(g.scan_tmp + nb256_ + 1) : (const uint32_t*)nullptr, nb256_, g.header);
    static const bool regs_scan = [] { const char* e = getenv("ISR_TILE_SCAN_REGS"); return !(e && e[0] == '0'); }();
    if (regs_scan && T >= 4096 && T <= 8192)
        hipLaunchKernelGGL(k_tile_scan_regs, dim3(1), dim3(1024), 0, s, T, iv.sub_offset, iv.tile_offset, g.header, iv.tile_order, order_classes);
    else
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, T, iv.sub_offset, iv.tile_offset, g.header, iv.tile_order, order_classes); }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace isr

using namespace isr;

extern "C" {

int isr_forward_bin(int P, int width, int height, void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                    void* image_buffer, void* stream) {
    return isr_forward_bin_event(P, width, height, geom_buffer, binning_buffer, binning_capacity, image_buffer, nullptr, stream);
}

int isr_forward_bin_event(int P, int width, int height, void* geom_buffer, void* binning_buffer, int64_t binning_capacity,
                          void* image_buffer, void* scatter_done_event, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(ISR_EINVAL, "null buffer");
    const int gx = tiles_x(width), gy = tiles_y(height), T = gx * gy;
    GeomView g = geom_view(geom_buffer, P < 1 ? 1 : P);
    ImageView iv = image_view(image_buffer, width, height);
    BinView bv = bin_view(binning_buffer, binning_capacity);
    if (P > 0 && binning_capacity > 0) {
        { ProfScope ps_("k_scatter", s);
        hipLaunchKernelGGL(k_scatter, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, g, iv.sub_offset, iv.tile_cursor,
                           bv.keys, binning_capacity); }
        ISR_LAUNCH_CHECK("k_scatter");
        // (a trainer that issues this chain on a side stream makes its HBM-saturating per-Gaussian tail wait for THIS point: beside
        // that tail the scatter's 8-byte key writes to half-evicted lines take 7x as long - 1.7 ms instead of 0.23 at C5)
        if (scatter_done_event != nullptr) ISR_HIP(hipEventRecord((hipEvent_t)scatter_done_event, s));
        { ProfScope ps_("k_tile_sort", s);
        // dense scenes (more than ~1 500 instances per tile on average): buckets beyond the 4 096-key LDS budget get their
        // own launch with 128 KB of LDS instead of the global-memory network
        const int big = (binning_capacity / (T > 0 ? T : 1)) > 1500 ? 1 : 0;
        static const bool wave_sort = [] { const char* e = getenv("ISR_WAVE_SORT"); return !(e && e[0] == '0'); }();
        // buckets of up to 2 048 keys: one wave each, in registers; the LDS network takes the rest
        // (only 32, 64 and 128 keys per lane exist as kernels: anything else falls back to 64 - a value without a kernel would
        // leave the buckets between the LDS network's range and the big kernel's unsorted)
        static const int wave_max = [] { const char* e = getenv("ISR_WAVE_SORT_MAX"); const int v = e ? atoi(e) : 64;
                                         return (v == 32 || v == 64 || v == 128) ? v : 64; }();
        static const bool radix = [] { const char* e = getenv("ISR_SORT_RADIX"); return !(e && e[0] == '0'); }();
        const int wk = !wave_sort ? 0 : (!big ? 32 : wave_max);           // keys per lane of the widest variant launched
        const int wflags = wk == 0 ? 0 : wk == 32 ? 2 : wk == 64 ? 6 : 14;
        if (wk == 128)
            hipLaunchKernelGGL(k_tile_sort_wave<128>, dim3(T), dim3(64), 0, s, iv.tile_offset, bv.keys, bv.point_list, binning_capacity);
        else if (wk == 64)
            hipLaunchKernelGGL(k_tile_sort_wave<64>, dim3(T), dim3(64), 0, s, iv.tile_offset, bv.keys, bv.point_list, binning_capacity);
        else if (wk == 32)
            hipLaunchKernelGGL(k_tile_sort_wave<32>, dim3(T), dim3(64), 0, s, iv.tile_offset, bv.keys, bv.point_list, binning_capacity);
        hipLaunchKernelGGL(k_tile_sort, dim3(T), dim3(256), 0, s, iv.tile_offset, bv.keys, bv.point_list, binning_capacity,
                           big | wflags);
        if (big) {
            // buckets beyond the wave kernels' range: LDS radix sort up to 8 192 keys, the bitonic network in 128 KB of LDS beyond
            constexpr int RADIX_LDS = 2 * SORT_RADIX_KEYS * (int)sizeof(unsigned long long) + 16 * 256 * (int)sizeof(uint32_t);
            static const bool attr_ok = [] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_sort_big), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SORT_BIG_KEYS * (int)sizeof(unsigned long long)) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_sort_radix), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           RADIX_LDS) == hipSuccess;
            }();
            if (!attr_ok) return fail(ISR_EHIP, "k_tile_sort_big / _radix: cannot reserve %d bytes of LDS", RADIX_LDS);
            const int beyond_wave = wk > 64 ? wk * 64 : SORT_LDS_KEYS;
            if (radix)
                hipLaunchKernelGGL(k_tile_sort_radix, dim3(T), dim3(1024), RADIX_LDS, s, iv.tile_offset, bv.keys, bv.point_list,
                                   binning_capacity, beyond_wave);
            hipLaunchKernelGGL(k_tile_sort_big, dim3(T), dim3(1024), SORT_BIG_KEYS * sizeof(unsigned long long), s, iv.tile_offset,
                               bv.keys, bv.point_list, binning_capacity, radix && SORT_RADIX_KEYS > beyond_wave ? SORT_RADIX_KEYS : beyond_wave);
        } }
        ISR_LAUNCH_CHECK("k_tile_sort");
        { ProfScope ps_("k_pack_hits", s);
        // ISR_PACK_EXACT=0: the bounding-octagon test alone (rounds 1-5); default: the row-exact conic test behind it
        static const bool row_exact = [] { const char* e = getenv("ISR_PACK_EXACT"); return !(e && e[0] == '0'); }();
        if (row_exact)
            hipLaunchKernelGGL(k_pack_hits<true>, dim3(T), dim3(256), 0, s, gx, binning_capacity, iv.tile_offset, bv.point_list, g.cull, g.ellipse, bv.box4, bv.hit_mask);
        else
            hipLaunchKernelGGL(k_pack_hits<false>, dim3(T), dim3(256), 0, s, gx, binning_capacity, iv.tile_offset, bv.point_list, g.cull, g.ellipse, bv.box4, bv.hit_mask); }
        ISR_LAUNCH_CHECK("k_pack_hits");
    }
    if (launch_late_scan(P, g, s) != 0) return fail(ISR_EHIP, "launch of the Gaussian offset scan failed");
    return ISR_OK;
}

int isr_debug_check_hit_masks(int P, int width, int height, int64_t num_rendered, const void* geom_buffer, const void* binning_buffer,
                              const void* image_buffer, unsigned long long* device_counters, void* stream) {
    if (!geom_buffer || !binning_buffer || !image_buffer || !device_counters) return fail(ISR_EINVAL, "check_hit_masks: null buffer");
    if (P <= 0 || num_rendered <= 0) return ISR_OK;
    const GeomView g = geom_view(const_cast<void*>(geom_buffer), P);
    const ImageView iv = image_view(const_cast<void*>(image_buffer), width, height);
    const BinView bv = bin_view(const_cast<void*>(binning_buffer), num_rendered);
    const int gx = tiles_x(width), T = gx * tiles_y(height);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_check_hit_masks, dim3(T), dim3(256), 0, s, width, height, gx, num_rendered, iv.tile_offset, bv.point_list, g.rec,
                       bv.hit_mask, device_counters);
    ISR_LAUNCH_CHECK("k_check_hit_masks");
    return ISR_OK;
}

}  // extern "C"

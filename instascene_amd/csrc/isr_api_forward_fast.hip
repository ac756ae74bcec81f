// libinstascene_hip.so, the FAST forward blend (K8, k_render_fwd_fast).  Host side: isr_host.hpp.
#include "isr_host.hpp"
#include "isr_forward_fast.hip"

namespace isr {

thread_local unsigned long long* g_fwd_counters = nullptr;    // isr_forward_set_counters: consumed by the next FAST forward

// FAST arithmetic: k_render_fwd_fast (isr_forward_fast.hip), 32 feature channels per pass
int launch_render_fwd_fast(int P, int tiles, hipStream_t s, int W, int H, int ED, int gx, const ImageView& iv,
                                  const BinView& bv, const float* rec, const float* cull, const float* col_pre, const float* tm_pre,
                                  const float* extras, const float* bg, float* out_color, float* out_others, float* out_extra,
                                  int32_t* tracer, long long tcap, int32_t* tcount, int64_t capacity, bool aux, const float* xscale) {
    unsigned long long* counters = g_fwd_counters;
    g_fwd_counters = nullptr;
    static const int per_block = [] { const char* e = getenv("ISR_FWD_WAVE"); return e ? atoi(e) : 1; }();
    // launch order: longest lists first.  The per-block kernel always (-2 % at C3, -4 % at C5: its tail is one wave deep);
    // the tile-wide kernel only on small grids (heaviest-first costs it 4 % at 1080p)
    static const int order_below = [] { const char* e = getenv("ISR_FWD_ORDER_BELOW"); return e ? atoi(e) : 0; }();
    const uint32_t* order = tiles < (order_below > 0 ? order_below : per_block ? (1 << 30) : 4096) ? iv.tile_order : nullptr;
    int ch = 0, first = 1;
    if (per_block && P <= (1 << 26)) {           // (the per-block kernel packs a tracer pair into 32 bits: 26 for the Gaussian)
        // k_render_fwd_fast_w: one wave per 8x8 block; the four blocks of a tile on one XCD (workgroup v -> XCD v % 8)
        const int grid = (tiles + 7) / 8 * 32;
        static const bool wide_ok = [] { const char* e = getenv("ISR_FWD_WIDE"); return !(e && e[0] == '0'); }();
        // ISR_FWD_CN=0: colour and normal by the vector FMAs also for narrow feature chunks (k_render_fwd_fast_w<.., CN = false>)
        static const bool cn_ok = [] { const char* e = getenv("ISR_FWD_CN"); return !(e && e[0] == '0'); }();
        do {
            ProfScope ps_("k_render_fwd", s);
#define ISR_GW2(FEAT, STATS, AUX_, ORD, NC_)                                                                                       \
    hipLaunchKernelGGL((k_render_fwd_fast_w<FEAT, STATS, AUX_, ORD, NC_>), dim3(grid), dim3(64), 0, s, W, H, ED, ch, first, gx, tiles, \
                       iv.tile_offset, bv.point_list, rec, cull, col_pre, tm_pre, extras, xscale, bg, iv.final_T, iv.n_contrib,        \
                       out_color, out_others, out_extra, tracer, tcap, tcount, bv.hit_mask, capacity, counters, order)
#define ISR_GW(FEAT, STATS, NC_)                                                                                           \
    do { if (aux) { if (order) ISR_GW2(FEAT, STATS, true, true, NC_); else ISR_GW2(FEAT, STATS, true, false, NC_); }            \
         else { if (order) ISR_GW2(FEAT, STATS, false, true, NC_); else ISR_GW2(FEAT, STATS, false, false, NC_); } } while (0)
            // 64 channels per pass while at least 64 remain (and the rows are float4-aligned): section 9.11
            const bool wide = wide_ok && ED - ch >= 64 && (ED & 3) == 0;
            if (ED - ch <= 0) { if (counters) ISR_GW(false, true, 1); else ISR_GW(false, false, 1); }
            else if (wide) { if (counters) ISR_GW(true, true, 2); else ISR_GW(true, false, 2); }
            else if (cn_ok && aux && first && ED <= 24 && (ED & 3) == 0) {
                // a single narrow pass with the aux outputs (the reference's default seg_feat_dim = 16): rgb and normal ride on the
                // feature MFMAs' spare rows (k_render_fwd_fast_w<.., CN = true>; the STATS build likewise: the same bits)
#define ISR_GCN(STATS, ORD)                                                                                                                 \
    hipLaunchKernelGGL((k_render_fwd_fast_w<true, STATS, true, ORD, 1, true>), dim3(grid), dim3(64), 0, s, W, H, ED, ch, first, gx, tiles,     \
                       iv.tile_offset, bv.point_list, rec, cull, col_pre, tm_pre, extras, xscale, bg, iv.final_T, iv.n_contrib,                     \
                       out_color, out_others, out_extra, tracer, tcap, tcount, bv.hit_mask, capacity, counters, order)
                if (counters) { if (order) ISR_GCN(true, true); else ISR_GCN(true, false); }
                else { if (order) ISR_GCN(false, true); else ISR_GCN(false, false); }
#undef ISR_GCN
            }
            else { if (counters) ISR_GW(true, true, 1); else ISR_GW(true, false, 1); }
#undef ISR_GW2
#undef ISR_GW
            ISR_LAUNCH_CHECK("k_render_fwd_fast_w");
            ch += wide ? 2 * MAX_FCHUNK : MAX_FCHUNK;
            first = 0;
        } while (ch < ED);
        return ISR_OK;
    }
    if (xscale != nullptr) return fail(ISR_EINVAL, "extra_row_scale is taken by the per-block FAST blend only (ISR_FWD_WAVE=1, P <= 2^26)");
    do {
        ProfScope ps_("k_render_fwd", s);
#define ISR_GO2(FEAT, STATS, AUX_, ORD)                                                                                   \
    hipLaunchKernelGGL((k_render_fwd_fast<FEAT, STATS, AUX_, ORD>), dim3(tiles), dim3(256), 0, s, W, H, ED, ch, first, gx, \
                       iv.tile_offset, bv.point_list, rec, cull, col_pre, tm_pre, extras, bg, iv.final_T, iv.n_contrib,    \
                       out_color, out_others, out_extra, tracer, tcap, tcount, bv.box4, capacity, counters, order)
#define ISR_GO(FEAT, STATS)                                                                                           \
    do { if (aux) { if (order) ISR_GO2(FEAT, STATS, true, true); else ISR_GO2(FEAT, STATS, true, false); }            \
         else { if (order) ISR_GO2(FEAT, STATS, false, true); else ISR_GO2(FEAT, STATS, false, false); } } while (0)
        if (ED - ch <= 0) { if (counters) ISR_GO(false, true); else ISR_GO(false, false); }
        else { if (counters) ISR_GO(true, true); else ISR_GO(true, false); }
#undef ISR_GO2
#undef ISR_GO
        ISR_LAUNCH_CHECK("k_render_fwd_fast");
        ch += MAX_FCHUNK;
        first = 0;
    } while (ch < ED);
    return ISR_OK;
}


}  // namespace isr

extern "C" void isr_forward_set_counters(unsigned long long* device_counters) { isr::g_fwd_counters = device_counters; }

// diff_surfel_rasterization._C as a COMPILED torch extension for ROCm: the reference's pybind boundary (ext.cpp:15-18;
// rasterize_points.cu:39-151 RasterizeGaussiansCUDA, :153-270 RasterizeGaussiansBackwardCUDA, :272-295 markVisible), the same
// three functions with the same argument order and return tuples, on top of the C ABI of include/instascene_rasterizer.h
// (libinstascene_hip.so: hand-written gfx950 kernels, no torch types).  No device code here: this file is host glue built by
// g++ against torch's headers - torch tensors in, device pointers and the current HIP stream out.
//
// What it is for: the drop-in module `diff_surfel_rasterization._C` (dropin/diff_surfel_rasterization/_C.py) binds these when
// the extension is built; `instascene_amd.rasterizer` (ctypes over the same C ABI) stays the feature-complete host layer
// (arena, asynchronous binning, prefetched geometry passes, sampled backward) that the trainers of harness.py use.
//
// Semantics = the reference's: a blocking read of the instance count in the middle of the forward (rasterizer_impl.cu:287),
// fresh tensors from torch's allocator, state handed to the backward as three byte tensors.  Deviations (the same as the Python
// mirror's): the tracer list holds H*W*10 rows, not pre-filled (a pixel has at most 9 entries with
// w > 0.1; the reference allocates and -1-fills H*W*100); `gau_pixel_indices` is the index of the last valid pair.
// ISR_MODE = fast_reflists (default) | fast | exact selects the arithmetic / tile lists like everywhere else in the library.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <string>
#include <tuple>

#include "instascene_rasterizer.h"

namespace {

struct Mode { int mode; int tight; };

Mode current_mode() {
    static const Mode m = [] {
        const char* e = std::getenv("ISR_MODE");
        const std::string s = e ? e : "fast_reflists";
        if (s == "exact") return Mode{ISR_MODE_EXACT, 0};
        if (s == "fast" || s == "fast_tight") return Mode{ISR_MODE_FAST, ISR_PREPARE_TIGHT_RECTS};
        return Mode{ISR_MODE_FAST, 0};
    }();
    return m;
}
// debug=true: per-launch synchronisation for THIS call only - restored on every way out, also when check() throws (advisor, round 5:
// a failing call used to leave the thread in the slow mode for every later call)
struct DebugScope {
    const bool on; int prev = 0;
    explicit DebugScope(bool debug) : on(debug) { if (on) prev = isr_set_debug(1, 0); }
    ~DebugScope() { if (on) isr_set_debug(prev, 0); }
    DebugScope(const DebugScope&) = delete;
    DebugScope& operator=(const DebugScope&) = delete;
};

int g_mode_override = -1, g_tight_override = -1;      // set_mode(): the Python layer's rasterizer.set_mode reaches here too

void check(int rc, const char* what) {
    if (rc != ISR_OK) throw std::runtime_error(std::string(what) + ": " + isr_last_error());
}

torch::Tensor f32c(const torch::Tensor& t, const char* name) {
    if (!t.defined() || t.numel() == 0) return t;
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA (HIP) tensor");
    return t.scalar_type() == torch::kFloat32 && t.is_contiguous() ? t : t.to(torch::kFloat32).contiguous();
}
const float* fptr(const torch::Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }
float* fptr_w(torch::Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }

void* stream_of(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansHIP(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& colors_,
                      const torch::Tensor& opacity_, const torch::Tensor& scales_, const torch::Tensor& rotations_,
                      const float scale_modifier, const torch::Tensor& transMat_precomp_, const torch::Tensor& extra_attrs_,
                      const int attr_degree, const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, const float tan_fovx,
                      const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh_, const int degree,
                      const torch::Tensor& campos_, const bool prefiltered, const bool debug) {
    if (means3D_.ndimension() != 2 || means3D_.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    c10::DeviceGuard guard(means3D_.device());
    const int P = (int)means3D_.size(0), H = image_height, W = image_width, F = attr_degree;
    const Mode env = current_mode();
    const int mode = g_mode_override >= 0 ? g_mode_override : env.mode;
    const int tight = mode == ISR_MODE_EXACT ? 0 : (g_tight_override >= 0 ? g_tight_override : env.tight);
    const auto bg = f32c(background, "background"), means3D = f32c(means3D_, "means3D"), colors = f32c(colors_, "colors");
    const auto opacity = f32c(opacity_, "opacity"), scales = f32c(scales_, "scales"), rotations = f32c(rotations_, "rotations");
    const auto transMat = f32c(transMat_precomp_, "transMat_precomp"), view = f32c(viewmatrix_, "viewmatrix");
    const auto proj = f32c(projmatrix_, "projmatrix"), sh = f32c(sh_, "sh"), campos = f32c(campos_, "campos");
    const auto extra = F > 0 ? f32c(extra_attrs_, "extra_attrs") : torch::Tensor();
    const auto fopt = means3D.options().dtype(torch::kFloat32), iopt = means3D.options().dtype(torch::kInt32);
    const auto bopt = means3D.options().dtype(torch::kUInt8);
    void* st = stream_of(means3D);
    const DebugScope debug_scope(debug);

    torch::Tensor out_color = torch::empty({3, H, W}, fopt), out_others = torch::empty({3 + 3 + 1, H, W}, fopt);
    torch::Tensor out_extra = F > 0 ? torch::empty({F, H, W}, fopt) : torch::empty({0}, fopt);
    torch::Tensor radii = torch::empty({P}, iopt);
    torch::Tensor geom = torch::empty({(int64_t)isr_geom_bytes(P)}, bopt), img = torch::empty({(int64_t)isr_image_bytes(W, H)}, bopt);
    torch::Tensor binning = torch::empty({0}, bopt);
    torch::Tensor pairs = torch::empty({(int64_t)H * W * 10, 2}, iopt), last = torch::full({1}, -1, iopt);
    int64_t R = 0;
    if (P != 0) {
        const int M = sh.defined() && sh.dim() == 3 ? (int)sh.size(1) : 0;
        check(isr_forward_prepare(P, degree, M, W, H, fptr(means3D), fptr(sh), fptr(colors), fptr(opacity), fptr(scales), scale_modifier,
                                  fptr(rotations), fptr(transMat), fptr(view), fptr(proj), fptr(campos), tan_fovx, tan_fovy,
                                  (prefiltered ? 1 : 0) | tight, radii.data_ptr<int>(), geom.data_ptr(), img.data_ptr(), &R, st),
              "isr_forward_prepare");
        binning = torch::empty({(int64_t)isr_binning_bytes(R, W, H)}, bopt);
        check(isr_forward_render(P, F, W, H, mode, fptr(bg), fptr(colors), fptr(transMat), fptr(extra), geom.data_ptr(), binning.data_ptr(), R,
                                 img.data_ptr(), fptr_w(out_color), fptr_w(out_others), fptr_w(out_extra), pairs.data_ptr<int>(),
                                 (int64_t)H * W * 10, last.data_ptr<int>(), st),
              "isr_forward_render");
    } else {
        out_color.copy_(bg.view({3, 1, 1}).expand({3, H, W}));
        out_others.zero_();
        out_extra.zero_();
        radii.zero_();
    }
    return std::make_tuple((int)R, out_color, out_others, radii, out_extra, geom, binning, img, pairs, last);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardHIP(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& radii,
                              const torch::Tensor& colors_, const torch::Tensor& scales_, const torch::Tensor& rotations_,
                              const torch::Tensor& extra_attrs_, const float scale_modifier, const torch::Tensor& transMat_precomp_,
                              const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, const float tan_fovx, const float tan_fovy,
                              const torch::Tensor& dL_dout_color_, const torch::Tensor& dL_dout_others_, const torch::Tensor& dL_dout_extra_,
                              const torch::Tensor& sh_, const int degree, const torch::Tensor& campos_, const torch::Tensor& geomBuffer,
                              const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug) {
    c10::DeviceGuard guard(means3D_.device());
    const auto means3D = f32c(means3D_, "means3D");
    const int P = (int)means3D.size(0);
    const auto bg = f32c(background, "background"), colors = f32c(colors_, "colors"), scales = f32c(scales_, "scales");
    const auto rotations = f32c(rotations_, "rotations"), transMat = f32c(transMat_precomp_, "transMat_precomp");
    const auto view = f32c(viewmatrix_, "viewmatrix"), proj = f32c(projmatrix_, "projmatrix"), sh = f32c(sh_, "sh");
    const auto campos = f32c(campos_, "campos"), extra = f32c(extra_attrs_, "extra_attrs");
    const auto dC = f32c(dL_dout_color_, "dL_dout_color"), dO = f32c(dL_dout_others_, "dL_dout_others"), dE = f32c(dL_dout_extra_, "dL_dout_extra");
    const int H = dC.defined() && dC.numel() ? (int)dC.size(1) : (int)dE.size(1);
    const int W = dC.defined() && dC.numel() ? (int)dC.size(2) : (int)dE.size(2);
    const int M = sh.defined() && sh.dim() == 3 ? (int)sh.size(1) : 0;
    const int F = extra.defined() && extra.dim() == 2 ? (int)extra.size(1) : 0;
    const auto fopt = means3D.options().dtype(torch::kFloat32);
    // the reference zero-initialises its gradient tensors (rasterize_points.cu:196-205); isr_backward writes every entry
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, fopt), dL_dmeans2D = torch::empty({P, 3}, fopt), dL_dcolors = torch::empty({P, 3}, fopt);
    torch::Tensor dL_dnormal = torch::empty({P, 3}, fopt), dL_dopacity = torch::empty({P, 1}, fopt), dL_dtransMat = torch::empty({P, 9}, fopt);
    torch::Tensor dL_dsh = torch::empty({P, M, 3}, fopt), dL_dscales = torch::empty({P, 2}, fopt), dL_drotations = torch::empty({P, 4}, fopt);
    torch::Tensor dL_dextra = torch::empty({P, F}, fopt);
    if (P != 0) {
        const int mode = g_mode_override >= 0 ? g_mode_override : current_mode().mode;
        const unsigned mask = ISR_GRAD_GEOMETRY | (F > 0 ? ISR_GRAD_EXTRA : 0u);
        const size_t sb = isr_backward_scratch_bytes(R, F, mask);
        torch::Tensor scratch = torch::empty({(int64_t)sb}, means3D.options().dtype(torch::kUInt8));
        const DebugScope debug_scope(debug);
        check(isr_backward(P, degree, M, R, F, W, H, mode, mask, fptr(bg), fptr(means3D), fptr(sh), fptr(colors), fptr(scales), scale_modifier,
                           fptr(rotations), fptr(transMat), fptr(extra), fptr(view), fptr(proj), fptr(campos), tan_fovx, tan_fovy,
                           radii.data_ptr<int>(), geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(), fptr(dC), fptr(dO),
                           fptr(dE), fptr_w(dL_dmeans2D), fptr_w(dL_dnormal), fptr_w(dL_dopacity), fptr_w(dL_dcolors), fptr_w(dL_dmeans3D),
                           fptr_w(dL_dtransMat), fptr_w(dL_dsh), fptr_w(dL_dscales), fptr_w(dL_drotations), fptr_w(dL_dextra),
                           scratch.data_ptr(), sb, stream_of(means3D)),
              "isr_backward");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations, dL_dextra);
}

torch::Tensor markVisible(torch::Tensor& means3D_, torch::Tensor& viewmatrix_, torch::Tensor& projmatrix_) {
    c10::DeviceGuard guard(means3D_.device());
    const auto means3D = f32c(means3D_, "means3D"), view = f32c(viewmatrix_, "viewmatrix"), proj = f32c(projmatrix_, "projmatrix");
    const int P = (int)means3D.size(0);
    torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
    if (P != 0)
        check(isr_mark_visible(P, fptr(means3D), fptr(view), fptr(proj), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), stream_of(means3D)),
              "isr_mark_visible");
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "diff_surfel_rasterization._C for MI355X: the reference's three entry points on libinstascene_hip.so";
    m.def("rasterize_gaussians", &RasterizeGaussiansHIP);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardHIP);
    m.def("mark_visible", &markVisible);
    m.def("set_mode", [](const std::string& s) {
        if (s == "exact") { g_mode_override = ISR_MODE_EXACT; g_tight_override = 0; }
        else if (s == "fast" || s == "fast_tight") { g_mode_override = ISR_MODE_FAST; g_tight_override = ISR_PREPARE_TIGHT_RECTS; }
        else if (s == "fast_reflists") { g_mode_override = ISR_MODE_FAST; g_tight_override = 0; }
        else throw std::runtime_error("mode must be exact | fast | fast_reflists");
    }, "arithmetic mode / tile lists of the calls that follow (default: ISR_MODE, else fast_reflists)");
    m.def("library_version", [] { return isr_version(); });
}

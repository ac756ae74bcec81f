// The FAST arithmetic's evaluation of one (pixel, splat) pair - ONE definition, used by the forward blend
// (isr_forward_fast.hip) and by every FAST backward kernel (isr_backward.hip: k_render_bwd<FastMath>,
// k_render_bwd_sparse<FastMath>; isr_backward_geo.hip).
//
// The reference decides per pair whether the splat is blended at all (forward.cu:356-393: p.z == 0, depth < near_n,
// alpha < 1/255) and its backward re-derives the same decisions from the same expressions (backward.cu:284-345).  A
// backward that evaluated the pair in a different - even if mathematically equal - form would flip some of those
// decisions against its own forward: a splat near the 1/255 or near-plane threshold is then blended by one pass and
// skipped by the other, and the transmittance the backward reconstructs is off by a factor (1 - alpha).  Hence: the same
// instruction sequence on the same inputs in both passes (the library is built with -ffp-contract=off, every fused
// operation below is explicit, v_rcp_f32 / v_exp_f32 are deterministic), so the decisions replay bit for bit.
//
//   p(px, py) = (px Tw - Tu) x (py Tw - Tv) is affine in the pixel:  p = lx A + ly B + C  with tile-relative lx, ly,
//   A = Tv x Tw,  B = Tw x Tu,  C = p at the tile origin;  depth = <p, Tw> / p.z = det / p.z  (A, B are orthogonal to Tw).
#pragma once

#include "isr_common.hpp"

namespace isr {

// alpha = opa exp(-rho / 2) < 1/255 for every rho > skip (1 % + 0.05 margin); never skips when opa > 1
__device__ __forceinline__ float fast_skip(float opa) {
    float skip = __builtin_inff();
    if (opa <= 1.0f) {
        const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
        skip = 2.0f * l * 1.01f + 0.05f;
    }
    return skip;
}

struct FastSplat { F3 A, B, C; float det; };

// per (tile, splat) instance; X0, Y0 = pixel coordinates of the tile's origin
__device__ __forceinline__ FastSplat fast_splat(const F3 Tu, const F3 Tv, const F3 Tw, float X0, float Y0) {
    FastSplat f;
    f.A = {__builtin_fmaf(Tv.y, Tw.z, -(Tv.z * Tw.y)), __builtin_fmaf(Tv.z, Tw.x, -(Tv.x * Tw.z)),
           __builtin_fmaf(Tv.x, Tw.y, -(Tv.y * Tw.x))};
    f.B = {__builtin_fmaf(Tw.y, Tu.z, -(Tw.z * Tu.y)), __builtin_fmaf(Tw.z, Tu.x, -(Tw.x * Tu.z)),
           __builtin_fmaf(Tw.x, Tu.y, -(Tw.y * Tu.x))};
    const F3 k0 = {__builtin_fmaf(X0, Tw.x, -Tu.x), __builtin_fmaf(X0, Tw.y, -Tu.y), __builtin_fmaf(X0, Tw.z, -Tu.z)};
    const F3 l0 = {__builtin_fmaf(Y0, Tw.x, -Tv.x), __builtin_fmaf(Y0, Tw.y, -Tv.y), __builtin_fmaf(Y0, Tw.z, -Tv.z)};
    f.C = {__builtin_fmaf(k0.y, l0.z, -(k0.z * l0.y)), __builtin_fmaf(k0.z, l0.x, -(k0.x * l0.z)),
           __builtin_fmaf(k0.x, l0.y, -(k0.y * l0.x))};
    f.det = __builtin_fmaf(f.C.x, Tw.x, __builtin_fmaf(f.C.y, Tw.y, f.C.z * Tw.z));
    return f;
}

struct FastRay { float p_x, p_y, p_z, dx, dy, rho2d, rz, sx, sy, rho3d, rho; };

// first half: the intersection and the two squared distances.  lx, ly: tile-relative pixel (0..15 as float);
// cxr, cyr: the splat's low-pass centre relative to the tile origin
__device__ __forceinline__ FastRay fast_ray(float lx, float ly, float Ax, float Ay, float Az, float Bx, float By, float Bz,
                                            float Cx, float Cy, float Cz, float cxr, float cyr) {
    FastRay r;
    r.p_x = __builtin_fmaf(lx, Ax, __builtin_fmaf(ly, Bx, Cx));
    r.p_y = __builtin_fmaf(lx, Ay, __builtin_fmaf(ly, By, Cy));
    r.p_z = __builtin_fmaf(lx, Az, __builtin_fmaf(ly, Bz, Cz));
    r.dx = cxr - lx;
    r.dy = cyr - ly;
    const float hh = __builtin_fmaf(r.dy, r.dy, r.dx * r.dx);
    r.rho2d = hh + hh;                                  // FilterInvSquare = 2
    r.rz = __builtin_amdgcn_rcpf(r.p_z);
    r.sx = r.p_x * r.rz;
    r.sy = r.p_y * r.rz;
    r.rho3d = __builtin_fmaf(r.sy, r.sy, r.sx * r.sx);
    r.rho = fminf(r.rho3d, r.rho2d);
    return r;
}
// decision 1 (forward.cu:358 and the certain alpha < 1/255): the pair can contribute at all
__device__ __forceinline__ bool fast_near(const FastRay& r, float skip) { return r.rho <= skip && r.p_z != 0.0f; }

struct FastHit { bool use3d; float depth, G, alpha; };

// second half: depth along the ray, Gaussian weight, alpha.  det = FastSplat::det, Twz = Tw.z, opa = opacity
__device__ __forceinline__ FastHit fast_hit(const FastRay& r, float det, float Twz, float opa) {
    FastHit h;
    h.use3d = r.rho3d <= r.rho2d;
    h.depth = h.use3d ? det * r.rz : Twz;
    h.G = __builtin_amdgcn_exp2f(r.rho * -0.72134752f);                 // exp(-rho / 2)
    h.alpha = fminf(0.99f, opa * h.G);
    return h;
}
// decision 2 (forward.cu:372, :386)
__device__ __forceinline__ bool fast_pass(const FastHit& h) { return !(h.depth < NEAR_N) && !(h.alpha < 1.0f / 255.0f); }

}  // namespace isr

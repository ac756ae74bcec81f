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

// ---- guard bands ----------------------------------------------------------------------------------------------------
// FAST's rho differs from the two-rounding EXACT / oracle evaluation by rounding noise (dominated by EXACT's own
// fl(px Tw.z) - Tu.z, an absolute error of ulp(px Tw.z) on a difference of a few pixels).  A pair whose rho lies within that
// noise of a DECISION threshold - alpha = 1/255 (forward.cu:386), rho3d = rho2d (forward.cu:365-372), depth = near_n
// (forward.cu:372) - is re-evaluated with EXACT's instruction sequence (exact_pair below), in the forward and in every backward
// kernel alike, so the decision taken is the oracle's.  The noise bound `band` is per Gaussian and view (splat_band, K1,
// stored in rec[19]); band = +inf forces the EXACT sequence for every pair of the splat (ill-conditioned, near the near plane,
// horizon inside the footprint).  Outside the band FAST's decision equals EXACT's by the bound (checked on the device by the
// STATS build of the forward kernel: counter 7 counts pairs outside the band whose decision differs - it must stay 0).
struct FastBand { float hi, lo, bw; };      // alpha < 1/255 is certain for rho > hi, alpha >= 1/255 for rho <= lo
__device__ __forceinline__ FastBand fast_band(float opa, float band) {
    const float l = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
    const float thr = l + l;                            // opa exp(-rho / 2) = 1/255
    const float hi = thr + band, lo = thr - band;
    return {hi, lo, hi - lo};                           // (bw from hi and lo: a kernel that stages only those two recomputes the same bits)
}
__device__ __forceinline__ FastBand fast_band_of(float hi, float lo) { return {hi, lo, hi - lo}; }

// Bound on |rho_FAST - rho_EXACT| (both branches) over the splat's footprint - its alpha >= 1/255 box `cb` (splat_cull_box)
// clipped to the image - plus the slack of exp / log / the threshold itself.  First-order worst-case rounding analysis:
//   EXACT  k = fl(fl(px Tw) - Tu), l likewise, p = fl(fl(k.y l.z) - fl(k.z l.y)) ..., s = p.xy / p.z
//   FAST   A, B, C (fast_splat) by fma, p = fma(lx, A, fma(ly, B, C)), s = p.xy * rcp(p.z)
// u = 2^-24.  Returns +inf (always EXACT) when the bound is not small or a depth of the footprint may lie within it of near_n.
__device__ __forceinline__ float splat_band(F3 Tu, F3 Tv, F3 Tw, float cx, float cy, float opa, float4 cb, int W, int H) {
    const float inf = __builtin_inff();
    const float u = 5.9604645e-8f;
    const float xa = fmaxf(cb.x, 0.0f), xb = fminf(cb.y, (float)(W - 1)), ya = fmaxf(cb.z, 0.0f), yb = fminf(cb.w, (float)(H - 1));
    if (!(xa <= xb && ya <= yb)) return inf;            // no pixel (or NaN): never evaluated
    const float l0 = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
    const float S2 = l0 + l0 + 0.1f, S = __builtin_sqrtf(S2);
    const F3 aw = {fabsf(Tw.x), fabsf(Tw.y), fabsf(Tw.z)};
    // pixels of the footprint and the origins of their tiles: [xa - 16, xb + 16]
    const float PX = xb + 16.0f, PY = yb + 16.0f;
    const F3 ek = {u * (2.0f * PX * aw.x + fabsf(Tu.x)), u * (2.0f * PX * aw.y + fabsf(Tu.y)), u * (2.0f * PX * aw.z + fabsf(Tu.z))};
    const F3 el = {u * (2.0f * PY * aw.x + fabsf(Tv.x)), u * (2.0f * PY * aw.y + fabsf(Tv.y)), u * (2.0f * PY * aw.z + fabsf(Tv.z))};
    const F3 ka = {__builtin_fmaf(xa, Tw.x, -Tu.x), __builtin_fmaf(xa, Tw.y, -Tu.y), __builtin_fmaf(xa, Tw.z, -Tu.z)};
    const F3 kb = {__builtin_fmaf(xb, Tw.x, -Tu.x), __builtin_fmaf(xb, Tw.y, -Tu.y), __builtin_fmaf(xb, Tw.z, -Tu.z)};
    const F3 la = {__builtin_fmaf(ya, Tw.x, -Tv.x), __builtin_fmaf(ya, Tw.y, -Tv.y), __builtin_fmaf(ya, Tw.z, -Tv.z)};
    const F3 lb = {__builtin_fmaf(yb, Tw.x, -Tv.x), __builtin_fmaf(yb, Tw.y, -Tv.y), __builtin_fmaf(yb, Tw.z, -Tv.z)};
    const F3 K = {fmaxf(fabsf(ka.x), fabsf(kb.x)) + 16.0f * aw.x, fmaxf(fabsf(ka.y), fabsf(kb.y)) + 16.0f * aw.y,
                  fmaxf(fabsf(ka.z), fabsf(kb.z)) + 16.0f * aw.z};
    const F3 L = {fmaxf(fabsf(la.x), fabsf(lb.x)) + 16.0f * aw.x, fmaxf(fabsf(la.y), fabsf(lb.y)) + 16.0f * aw.y,
                  fmaxf(fabsf(la.z), fabsf(lb.z)) + 16.0f * aw.z};
    const F3 aC = {K.y * L.z + K.z * L.y, K.z * L.x + K.x * L.z, K.x * L.y + K.y * L.x};
    const F3 aA = {fabsf(Tv.y * Tw.z) + fabsf(Tv.z * Tw.y), fabsf(Tv.z * Tw.x) + fabsf(Tv.x * Tw.z), fabsf(Tv.x * Tw.y) + fabsf(Tv.y * Tw.x)};
    const F3 aB = {fabsf(Tw.y * Tu.z) + fabsf(Tw.z * Tu.y), fabsf(Tw.z * Tu.x) + fabsf(Tw.x * Tu.z), fabsf(Tw.x * Tu.y) + fabsf(Tw.y * Tu.x)};
    const float Px = (L.z * ek.y + K.y * el.z + L.y * ek.z + K.z * el.y) + u * (10.0f * aC.x + 64.0f * (aA.x + aB.x));
    const float Py = (L.x * ek.z + K.z * el.x + L.z * ek.x + K.x * el.z) + u * (10.0f * aC.y + 64.0f * (aA.y + aB.y));
    const float Pz = (L.y * ek.x + K.x * el.y + L.x * ek.y + K.y * el.x) + u * (10.0f * aC.z + 64.0f * (aA.z + aB.z));
    // p.z is affine in the pixel: its extremes over the footprint are at the corners
    const float z00 = __builtin_fmaf(ka.x, la.y, -(ka.y * la.x)), z10 = __builtin_fmaf(kb.x, la.y, -(kb.y * la.x));
    const float z01 = __builtin_fmaf(ka.x, lb.y, -(ka.y * lb.x)), z11 = __builtin_fmaf(kb.x, lb.y, -(kb.y * lb.x));
    const float zlo = fminf(fminf(z00, z10), fminf(z01, z11)), zhi = fmaxf(fmaxf(z00, z10), fmaxf(z01, z11));
    if (!(zlo > 0.0f || zhi < 0.0f)) return inf;        // the splat's horizon crosses the footprint
    const float zmin = fminf(fabsf(zlo), fabsf(zhi)) - (Pz + Pz);
    if (!(zmin > 0.0f)) return inf;
    const float r = 1.0f / zmin;
    const float es = ((Px + Py) + 2.0f * S * Pz) * r;                   // |delta sx| + |delta sy|
    const float e3 = 2.0f * S * es + 16.0f * u * S2;
    const float Dm = __builtin_sqrtf(0.5f * S2);
    const float ox = fmaxf(fabsf(cx - xa), fabsf(cx - xb)) + 16.0f, oy = fmaxf(fabsf(cy - ya), fabsf(cy - yb)) + 16.0f;
    const float e2 = 4.0f * Dm * u * ((ox + oy) + 4.0f * Dm) + 4.0f * u * S2;
    const float band = 1.25f * (e3 + e2) + 1e-5f;
    // depth = <p, Tw> / p.z = det / p.z (3-D branch) or Tw.z: may any depth of the footprint lie within its error of near_n?
    const F3 c0 = {__builtin_fmaf(ka.y, la.z, -(ka.z * la.y)), __builtin_fmaf(ka.z, la.x, -(ka.x * la.z)), z00};
    const float det = __builtin_fmaf(c0.x, Tw.x, __builtin_fmaf(c0.y, Tw.y, c0.z * Tw.z));
    const float d0 = det / z00, d1 = det / z10, d2 = det / z01, d3 = det / z11;
    const float dlo = fminf(fminf(fminf(d0, d1), fminf(d2, d3)), Tw.z), dhi = fmaxf(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)), Tw.z);
    const float ddet = 16.0f * u * (aC.x * aw.x + aC.y * aw.y + aC.z * aw.z);
    const float derr = (ddet + fmaxf(fabsf(dlo), fabsf(dhi)) * 2.0f * Pz) * r + (aw.x + aw.y) * es +
                       8.0f * u * (S * (aw.x + aw.y) + aw.z) + 1e-4f;
    if (!(dlo - derr > NEAR_N || dhi + derr < NEAR_N)) return inf;
    if (!(band < 0.04f)) return inf;                    // (the hit masks' own margin is 0.05 in rho; also catches NaN)
    return band;
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
__device__ __forceinline__ bool fast_near(const FastRay& r, float hi) { return r.rho <= hi && r.p_z != 0.0f; }
// ... and lies within the rounding noise of the alpha threshold or of the branch rho3d = rho2d (only asked of `near` pairs)
__device__ __forceinline__ bool fast_in_band(const FastRay& r, const FastBand& b) {
    return r.rho > b.lo || fabsf(r.rho3d - r.rho2d) <= b.bw;
}

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
// decision 2 (forward.cu:372) of a near pair outside the band (there alpha >= 1/255 is certain)
__device__ __forceinline__ bool fast_pass(const FastHit& h) { return !(h.depth < NEAR_N); }

// The pair in EXACT arithmetic - k_render_fwd_w<ExactMath>'s instruction sequence (the reference's operation order,
// forward.cu:340-393; bit-identical to the oracle) - written into the FAST structures.  pxf, pyf: absolute pixel.
__device__ __forceinline__ bool exact_pair(float pxf, float pyf, const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy, float opa,
                                           FastRay& r, FastHit& h) {
    typedef ExactMath M;
    const F3 kq = {M::msub(pxf, Tw.x, Tu.x), M::msub(pxf, Tw.y, Tu.y), M::msub(pxf, Tw.z, Tu.z)};
    const F3 lq = {M::msub(pyf, Tw.x, Tv.x), M::msub(pyf, Tw.y, Tv.y), M::msub(pyf, Tw.z, Tv.z)};
    r.p_x = M::msub(kq.y, lq.z, kq.z * lq.y);
    r.p_y = M::msub(kq.z, lq.x, kq.x * lq.z);
    r.p_z = M::msub(kq.x, lq.y, kq.y * lq.x);
    r.dx = cx - pxf;
    r.dy = cy - pyf;
    r.rho2d = FILTER_INV_SQ * M::mad(r.dy, r.dy, r.dx * r.dx);
    r.rz = M::div(1.0f, r.p_z);
    r.sx = M::div(r.p_x, r.p_z);
    r.sy = M::div(r.p_y, r.p_z);
    r.rho3d = M::mad(r.sy, r.sy, r.sx * r.sx);
    r.rho = fminf(r.rho3d, r.rho2d);
    h.use3d = r.rho3d <= r.rho2d;
    h.depth = h.use3d ? M::mad(r.sy, Tw.y, r.sx * Tw.x) + Tw.z : Tw.z;
    const float power = -0.5f * r.rho;
    h.G = M::ex(power);
    h.alpha = fminf(0.99f, opa * h.G);
    return r.p_z != 0.0f && !(h.depth < NEAR_N) && !(power > 0.0f) && !(h.alpha < 1.0f / 255.0f);
}
// the splat's record by (wave-uniform) id: scalar loads
__device__ __forceinline__ bool exact_pair_rec(float pxf, float pyf, const float* __restrict__ rec, int id, FastRay& r, FastHit& h) {
    const float* q = rec + (size_t)id * REC;
    return exact_pair(pxf, pyf, {q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}, q[9], q[10], q[14], r, h);
}
// select the EXACT evaluation for the lanes of the band
__device__ __forceinline__ void fast_take(bool inb, const FastRay& er, const FastHit& eh, FastRay& r, FastHit& h) {
    r.p_x = inb ? er.p_x : r.p_x; r.p_y = inb ? er.p_y : r.p_y; r.p_z = inb ? er.p_z : r.p_z;
    r.dx = inb ? er.dx : r.dx; r.dy = inb ? er.dy : r.dy; r.rho2d = inb ? er.rho2d : r.rho2d;
    r.rz = inb ? er.rz : r.rz; r.sx = inb ? er.sx : r.sx; r.sy = inb ? er.sy : r.sy;
    r.rho3d = inb ? er.rho3d : r.rho3d; r.rho = inb ? er.rho : r.rho;
    h.use3d = inb ? eh.use3d : h.use3d; h.depth = inb ? eh.depth : h.depth; h.G = inb ? eh.G : h.G; h.alpha = inb ? eh.alpha : h.alpha;
}
// One pair for a lane that owns its splat's record (the splat-major backward kernels): FAST, EXACT inside the band.
// Returns whether the pair blends (before the T < 1e-4 stop).
__device__ __forceinline__ bool fast_pair_lane(const FastSplat& fs, const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy, float opa,
                                               const FastBand& b, float cxr, float cyr, float lx, float ly, float pxf, float pyf,
                                               FastRay& fr, FastHit& fh) {
    fr = fast_ray(lx, ly, fs.A.x, fs.A.y, fs.A.z, fs.B.x, fs.B.y, fs.B.z, fs.C.x, fs.C.y, fs.C.z, cxr, cyr);
    fh = fast_hit(fr, fs.det, Tw.z, opa);
    const bool near = fast_near(fr, b.hi);
    bool pass = near && fast_pass(fh);
    if (near && fast_in_band(fr, b)) pass = exact_pair(pxf, pyf, Tu, Tv, Tw, cx, cy, opa, fr, fh);
    return pass;
}

}  // namespace isr

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
//   p(px, py) = k x l,  k = px Tw - Tu,  l = py Tw - Tv  (forward.cu:340-352).  The two-rounding EXACT / oracle evaluation
//   of k.z = fl(fl(px Tw.z) - Tu.z) cancels ~3 decimal digits (px Tw.z ~ Tu.z ~ 1e4, their difference a few pixels times the
//   depth) and that rounding dominates everything else in rho by two orders of magnitude.  FAST therefore computes k.z and
//   l.z with EXACTLY those two roundings (bit-identical to EXACT's), the well-conditioned k.xy, l.xy and the cross product
//   with fused multiply-adds, two components per v_pk_* instruction (8 vector instructions for p), s = p.xy * rcp(p.z), and
//   depth = <p, Tw> / p.z = det / p.z with the per-splat det = det[Tu, Tv, Tw]  (<k x l, Tw> = det[k, l, Tw], and adding
//   multiples of Tw to the other two rows does not change it).  Its rho then follows EXACT's to ~1e-6 relative, so the guard
//   bands below are narrow.
#pragma once

#include "isr_common.hpp"

namespace isr {

// ---- guard bands ----------------------------------------------------------------------------------------------------
// FAST's rho still differs from the two-rounding EXACT / oracle evaluation by rounding noise (fma against mul + add in the
// well-conditioned terms, rcp against IEEE division).  A pair whose rho lies within that noise of a DECISION threshold - alpha = 1/255 (forward.cu:386), rho3d = rho2d (forward.cu:365-372), depth = near_n
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

// det = <Tu, Tv x Tw> = <k x l, Tw> for every pixel; evaluated at the splat's centre (cx, cy), where k and l are small: no
// cancellation of the large Tu.z ~ cx Tw.z, Tv.z ~ cy Tw.z
__device__ __forceinline__ float fast_det(const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy) {
    const F3 k = {__builtin_fmaf(cx, Tw.x, -Tu.x), __builtin_fmaf(cx, Tw.y, -Tu.y), __builtin_fmaf(cx, Tw.z, -Tu.z)};
    const F3 l = {__builtin_fmaf(cy, Tw.x, -Tv.x), __builtin_fmaf(cy, Tw.y, -Tv.y), __builtin_fmaf(cy, Tw.z, -Tv.z)};
    const F3 c = {__builtin_fmaf(k.y, l.z, -(k.z * l.y)), __builtin_fmaf(k.z, l.x, -(k.x * l.z)), __builtin_fmaf(k.x, l.y, -(k.y * l.x))};
    return __builtin_fmaf(c.x, Tw.x, __builtin_fmaf(c.y, Tw.y, c.z * Tw.z));
}

// Bound on |rho_FAST - rho_EXACT| (both branches) over the splat's footprint - its alpha >= 1/255 box `cb` (splat_cull_box)
// clipped to the image - plus the slack of exp / log / the threshold itself.  First-order worst-case rounding analysis, u = 2^-24:
//   k.z, l.z                identical in both;
//   k.x  EXACT fl(fl(px Tw.x) - Tu.x): u (|px Tw.x| + |k.x|);  FAST fma: u |k.x|   (k.y, l.x, l.y alike)
//   p.x  EXACT fl(fl(k.y l.z) - fl(k.z l.y)), FAST fma(k.y, l.z, -fl(k.z l.y)): u (|k.y l.z| + 2 |k.z l.y| + 2 |p.x|) + inputs
//   s    EXACT p.xy / p.z, FAST p.xy * rcp(p.z): 3 u |s|;   rho3d, rho2d: fma against mul + add, 3 u rho
// Returns +inf (always EXACT) when the bound is not small, the horizon crosses the footprint, or a depth of the footprint may
// lie below near_n + its error - so that a finite band also certifies p.z != 0 and depth >= near_n for every pair (fast_near, fast_pass).
// `exact_noise` (optional): a bound on |rho3d_EXACT - rho3d| over the same footprint, i.e. on what EXACT's two-rounding k.z, l.z
// (shared by FAST, hence not part of the band) cost against the real-number value - what a test that evaluates the splat's conic
// in centre-relative coordinates (splat_conic, k_pack_hits) has to allow for; +inf whenever the band is.
__device__ __forceinline__ float splat_band(F3 Tu, F3 Tv, F3 Tw, float cx, float cy, float opa, float4 cb, int W, int H,
                                            float* exact_noise = nullptr) {
    const float inf = __builtin_inff();
    if (exact_noise) *exact_noise = inf;
    const float u = 5.9604645e-8f;
    const float xa = fmaxf(cb.x, 0.0f), xb = fminf(cb.y, (float)(W - 1)), ya = fmaxf(cb.z, 0.0f), yb = fminf(cb.w, (float)(H - 1));
    if (!(xa <= xb && ya <= yb)) return inf;            // no pixel (or NaN): never evaluated
    const float l0 = opa * 255.0f > 1.0f ? __logf(opa * 255.0f) : 0.0f;
    const float S2 = l0 + l0 + 0.1f, S = __builtin_sqrtf(S2);
    const F3 aw = {fabsf(Tw.x), fabsf(Tw.y), fabsf(Tw.z)};
    const F3 ka = {__builtin_fmaf(xa, Tw.x, -Tu.x), __builtin_fmaf(xa, Tw.y, -Tu.y), __builtin_fmaf(xa, Tw.z, -Tu.z)};
    const F3 kb = {__builtin_fmaf(xb, Tw.x, -Tu.x), __builtin_fmaf(xb, Tw.y, -Tu.y), __builtin_fmaf(xb, Tw.z, -Tu.z)};
    const F3 la = {__builtin_fmaf(ya, Tw.x, -Tv.x), __builtin_fmaf(ya, Tw.y, -Tv.y), __builtin_fmaf(ya, Tw.z, -Tv.z)};
    const F3 lb = {__builtin_fmaf(yb, Tw.x, -Tv.x), __builtin_fmaf(yb, Tw.y, -Tv.y), __builtin_fmaf(yb, Tw.z, -Tv.z)};
    // |k|, |l| over the footprint (affine: extremes at its ends; + a pixel for the clipping of the box to whole pixels)
    const F3 K = {fmaxf(fabsf(ka.x), fabsf(kb.x)) + aw.x, fmaxf(fabsf(ka.y), fabsf(kb.y)) + aw.y, fmaxf(fabsf(ka.z), fabsf(kb.z)) + aw.z};
    const F3 L = {fmaxf(fabsf(la.x), fabsf(lb.x)) + aw.x, fmaxf(fabsf(la.y), fabsf(lb.y)) + aw.y, fmaxf(fabsf(la.z), fabsf(lb.z)) + aw.z};
    const float dkx = u * (xb * aw.x + 2.0f * K.x), dky = u * (xb * aw.y + 2.0f * K.y);
    const float dlx = u * (yb * aw.x + 2.0f * L.x), dly = u * (yb * aw.y + 2.0f * L.y);
    const F3 aC = {K.y * L.z + K.z * L.y, K.z * L.x + K.x * L.z, K.x * L.y + K.y * L.x};
    const float Px = (L.z * dky + K.z * dly) + 4.0f * u * aC.x;
    const float Py = (K.z * dlx + L.z * dkx) + 4.0f * u * aC.y;
    const float Pz = (L.y * dkx + K.x * dly + L.x * dky + K.y * dlx) + 4.0f * u * aC.z;
    // p.z is affine in the pixel: its extremes over the footprint are at the corners
    const float z00 = __builtin_fmaf(ka.x, la.y, -(ka.y * la.x)), z10 = __builtin_fmaf(kb.x, la.y, -(kb.y * la.x));
    const float z01 = __builtin_fmaf(ka.x, lb.y, -(ka.y * lb.x)), z11 = __builtin_fmaf(kb.x, lb.y, -(kb.y * lb.x));
    const float zlo = fminf(fminf(z00, z10), fminf(z01, z11)), zhi = fmaxf(fmaxf(z00, z10), fmaxf(z01, z11));
    if (!(zlo > 0.0f || zhi < 0.0f)) return inf;        // the splat's horizon crosses the footprint
    const float zmin = fminf(fabsf(zlo), fabsf(zhi)) - (Pz + Pz);
    if (!(zmin > 0.0f)) return inf;
    const float r = 1.0f / zmin;
    const float es = ((Px + Py) + 2.0f * S * Pz) * r + 8.0f * u * S;      // |delta sx| + |delta sy|
    const float e3 = 2.0f * S * es + 8.0f * u * S2;
    const float e2 = 8.0f * u * S2;                     // dx, dy are EXACT's own; fma against mul + add
    const float band = 1.5f * (e3 + e2) + 4e-6f;        // (+ exp2 / __logf / the product opa * G: < 2e-6 in rho)
    // depth = <p, Tw> / p.z = det / p.z (3-D branch) or Tw.z: may any depth of the footprint lie within its error of near_n?
    const float det = fast_det(Tu, Tv, Tw, cx, cy);
    const float d0 = det / z00, d1 = det / z10, d2 = det / z01, d3 = det / z11;
    const float dlo = fminf(fminf(fminf(d0, d1), fminf(d2, d3)), Tw.z), dhi = fmaxf(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)), Tw.z);
    const float kc = fabsf(__builtin_fmaf(cx, Tw.z, -Tu.z)) + fabsf(__builtin_fmaf(cy, Tw.z, -Tv.z));      // conditioning of det
    const float ddet = 16.0f * u * ((K.y + K.x + kc) * (L.x + L.y + kc) * (aw.x + aw.y + aw.z));
    const float derr = (ddet + fmaxf(fabsf(dlo), fabsf(dhi)) * 2.0f * Pz) * r + (aw.x + aw.y) * es +
                       8.0f * u * (S * (aw.x + aw.y) + aw.z) + 1e-5f;
    if (!(dlo - derr > NEAR_N)) return inf;             // (with a finite band every depth of the footprint passes the near-plane test)
    if (!(band < 0.04f)) return inf;                    // (the hit masks' own margin is 0.05 in rho; also catches NaN)
    if (exact_noise) {
        const float dkz = u * (xb * aw.z + fabsf(Tu.z) + K.z), dlz = u * (yb * aw.z + fabsf(Tv.z) + L.z);
        *exact_noise = 2.0f * S * ((K.y + K.x) * dlz + (L.y + L.x) * dkz) * r;
    }
    return band;
}

typedef float v2f __attribute__((ext_vector_type(2)));

struct FastRay { float p_x, p_y, p_z, dx, dy, rho2d, rz, sx, sy, rho3d, rho; };

// first half: the intersection and the two squared distances.  pq = the pixel (absolute x, y); operands as the register
// pairs the packed instructions take them in (a staged record is laid out so that ds_read_b128 delivers exactly these):
//   Tuxy = (Tu.x, Tu.y)  Tvxy = (Tv.x, Tv.y)  Twxy = (Tw.x, Tw.y)  Tuvz = (Tu.z, Tv.z)  cxy = the low-pass centre
__device__ __forceinline__ FastRay fast_ray(v2f pq, v2f Tuxy, v2f Tvxy, v2f Twxy, v2f Tuvz, float Twz, v2f cxy) {
    FastRay r;
    const v2f klz = pq * (v2f){Twz, Twz} - Tuvz;                         // (k.z, l.z): EXACT's two roundings (-ffp-contract=off)
    const v2f kxy = __builtin_elementwise_fma((v2f){pq.x, pq.x}, Twxy, -Tuxy);
    const v2f lxy = __builtin_elementwise_fma((v2f){pq.y, pq.y}, Twxy, -Tvxy);
    // (p.x, p.y) = l.z (k.y, -k.x) - k.z (l.y, -l.x): two packed instructions whose operand swizzles and signs are instruction
    // modifiers (the compiler materialises the swapped, half-negated pairs with four extra moves / xors)
    //   t   = (k.z l.y, -k.z l.x)            src0 = klz.lo twice, src1 = (lxy.hi, -lxy.lo)
    //   pxy = (l.z k.y - t.lo, -l.z k.x - t.hi)   src0 = klz.hi twice, src1 = (kxy.hi, -kxy.lo), src2 = -t
    v2f t, pxy;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(klz), "v"(lxy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,1,1]" : "=v"(pxy) : "v"(klz), "v"(kxy), "v"(t));
    r.p_x = pxy.x;
    r.p_y = pxy.y;
    r.p_z = __builtin_fmaf(kxy.x, lxy.y, -(kxy.y * lxy.x));
    const v2f d = cxy - pq;
    r.dx = d.x;
    r.dy = d.y;
    const float hh = __builtin_fmaf(r.dy, r.dy, r.dx * r.dx);
    r.rho2d = hh + hh;                                  // FilterInvSquare = 2
    r.rz = __builtin_amdgcn_rcpf(r.p_z);
    const v2f sxy = pxy * (v2f){r.rz, r.rz};
    r.sx = sxy.x;
    r.sy = sxy.y;
    r.rho3d = __builtin_fmaf(r.sy, r.sy, r.sx * r.sx);
    r.rho = fminf(r.rho3d, r.rho2d);
    return r;
}
// The same split by pixel column and row, for a kernel that walks the pixels of a block for one splat per lane (k depends on
// the column only, l on the row only): bit-identical to fast_ray (the packed instructions round each half like the scalar ones).
struct FastHalf { v2f xy; float z; };
__device__ __forceinline__ FastHalf fast_k(float pxf, const F3 Tu, const F3 Tw) {
    return {__builtin_elementwise_fma((v2f){pxf, pxf}, (v2f){Tw.x, Tw.y}, -(v2f){Tu.x, Tu.y}), pxf * Tw.z - Tu.z};
}
__device__ __forceinline__ FastHalf fast_l(float pyf, const F3 Tv, const F3 Tw) {
    return {__builtin_elementwise_fma((v2f){pyf, pyf}, (v2f){Tw.x, Tw.y}, -(v2f){Tv.x, Tv.y}), pyf * Tw.z - Tv.z};
}
__device__ __forceinline__ FastRay fast_ray_kl(const FastHalf& k, const FastHalf& l, float dx, float dy) {
    FastRay r;
    const v2f klz = {k.z, l.z};
    v2f t, pxy;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(klz), "v"(l.xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,1,1]" : "=v"(pxy) : "v"(klz), "v"(k.xy), "v"(t));
    r.p_x = pxy.x;
    r.p_y = pxy.y;
    r.p_z = __builtin_fmaf(k.xy.x, l.xy.y, -(k.xy.y * l.xy.x));
    r.dx = dx;
    r.dy = dy;
    const float hh = __builtin_fmaf(r.dy, r.dy, r.dx * r.dx);
    r.rho2d = hh + hh;
    r.rz = __builtin_amdgcn_rcpf(r.p_z);
    const v2f sxy = pxy * (v2f){r.rz, r.rz};
    r.sx = sxy.x;
    r.sy = sxy.y;
    r.rho3d = __builtin_fmaf(r.sy, r.sy, r.sx * r.sx);
    r.rho = fminf(r.rho3d, r.rho2d);
    return r;
}
__device__ __forceinline__ FastRay fast_ray(float pxf, float pyf, const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy) {
    return fast_ray((v2f){pxf, pyf}, (v2f){Tu.x, Tu.y}, (v2f){Tv.x, Tv.y}, (v2f){Tw.x, Tw.y}, (v2f){Tu.z, Tv.z}, Tw.z, (v2f){cx, cy});
}
// decision 1 (forward.cu:358 and the certain alpha < 1/255): the pair can contribute at all.  The reference's p.z == 0 test needs
// no instruction here: a splat with a finite band has |p.z| >= zmin > 0 on its whole footprint (splat_band), outside of it
// rho2d > hi, and p.z = 0 makes rho3d inf or NaN, never the smaller of the two; a splat whose horizon crosses its footprint has
// band = +inf and every pair of it takes exact_pair, which tests p.z itself.
__device__ __forceinline__ bool fast_near(const FastRay& r, float hi) { return r.rho <= hi; }
// ... and lies within the rounding noise of the alpha threshold or of the branch rho3d = rho2d (only asked of `near` pairs)
__device__ __forceinline__ bool fast_in_band(const FastRay& r, const FastBand& b) {
    return r.rho > b.lo || fabsf(r.rho3d - r.rho2d) <= b.bw;
}

struct FastHit { bool use3d; float depth, G, alpha; };

// second half: depth along the ray, Gaussian weight, alpha.  det = fast_det, Twz = Tw.z, opa = opacity
__device__ __forceinline__ FastHit fast_hit(const FastRay& r, float det, float Twz, float opa) {
    FastHit h;
    h.use3d = r.rho3d <= r.rho2d;
    h.depth = h.use3d ? det * r.rz : Twz;
    h.G = __builtin_amdgcn_exp2f(r.rho * -0.72134752f);                 // exp(-rho / 2)
    h.alpha = fminf(0.99f, opa * h.G);
    return h;
}
// decision 2 (forward.cu:372, depth < near_n) of a near pair outside the band needs no instruction either: a splat with a finite
// band has every depth of its footprint above near_n by more than its error (splat_band returns +inf otherwise), and alpha >= 1/255
// is certain there.  (Kept as a function so that the call sites read like the reference's sequence of tests.)
__device__ __forceinline__ bool fast_pass(const FastHit&) { return true; }

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
// `elsewhere`: this lane's EXACT evaluation is supplied by the caller (k_render_bwd_geo's always-EXACT splat of a chunk).
__device__ __forceinline__ bool fast_pair_lane(const F3 Tu, const F3 Tv, const F3 Tw, float cx, float cy, float opa, float det,
                                               const FastBand& b, float pxf, float pyf, FastRay& fr, FastHit& fh, bool elsewhere = false,
                                               const FastHalf* row = nullptr) {
    if (row != nullptr) fr = fast_ray_kl(fast_k(pxf, Tu, Tw), *row, cx - pxf, cy - pyf);       // (row = fast_l(pyf, Tv, Tw), hoisted by the caller)
    else fr = fast_ray(pxf, pyf, Tu, Tv, Tw, cx, cy);
    fh = fast_hit(fr, det, Tw.z, opa);
    const bool near = fast_near(fr, b.hi);
    bool pass = near && fast_pass(fh);
#ifndef ISR_AB_NO_LANE_EXACT        // (A/B builds only, tools/build_variant.sh: what the EXACT path costs the splat-major kernels)
    if (!elsewhere && near && fast_in_band(fr, b)) pass = exact_pair(pxf, pyf, Tu, Tv, Tw, cx, cy, opa, fr, fh);
#endif
    return pass;
}

}  // namespace isr

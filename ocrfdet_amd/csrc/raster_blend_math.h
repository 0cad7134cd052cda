// Per-record arithmetic of the alpha blend (forward.cu:320-352 + the w-depth fork's depth channel), written ONCE for the
// two forward kernels (raster_blend_kernel of rasterize.hip: per call; raster_blend_sorted_kernel of raster_plan.hip:
// planned), so that both produce the same bits for the same record sequence.  A lane carries TWO pixels with the same
// x as the halves of packed fp32 operations.
//
// Every decision of the reference's loop is kept, record for record; what differs from the expression order of
// forward.cu is rounding only (north_star's tolerance is 1e-4; the tests bound the pixels whose DECISIONS flip against
// the oracle's threshold-ambiguity map):
//   * a SIMPLE record (opacity in (0, 0.99], conic with B^2 <= (1 - 2^-10) A C — so that the computed exponent is <= 0 at
//     every pixel and min(0.99, alpha) is the identity) is staged with log2(e) folded into the conic and log2(opacity)
//     as the constant term: alpha = exp2(L + A2 dx^2 + C2 dy^2 + B2 dx dy) as two packed FMAs on top of four scalar
//     operations — no multiply by log2(e), no multiply by the opacity, no clamp, no `power > 0` test;
//     any other record (GENERIC) is evaluated in the reference's order with both tests;
//   * "alpha < 1/255 -> skip" (forward.cu:331) is ONE packed FMA with the clamp modifier: s = clamp((alpha - k) 2^40)
//     with k the float below 1/255f is exactly 0 for alpha < 1/255f and exactly 1 otherwise (alpha - k is a multiple of
//     2^-31 there); the skipped record's weight is alpha T s = 0 and T - 0 = T: no compare, no select;
//   * T' = T - alpha T (the reference: T (1 - alpha));
//   * median depth (T > 0.5 and T' < 0.5 -> this record's depth): the records whose INCOMING T is above 0.5 form a
//     prefix of a pixel's sequence, and the last of them is the crossing record iff the pixel ends below 0.5.  The loop
//     only COUNTS them (one packed clamp-FMA + one packed add); the depth is looked up once per batch (median_index).
//     [A T' of exactly 0.5 is assigned the depth of the record that produced it — wherever the batch boundaries of a
//     kernel fall: median_index tests T <= 0.5 (round 6: with T < 0.5 a batch that ENDED on the exact tie lost the
//     crossing, a batch that went on kept it — one pixel of cfg2's 1.08 M on the object-centric set differed between the
//     planned and the per-call kernel).  The reference's strict compares leave such a pixel at the default: a tie of its
//     OWN arithmetic, T (1 - alpha), which rounds differently anyway; the oracle's ambiguity map covers both.]
//   * the stop test (T' < 1e-4 -> done, this record not blended; forward.cu:340-345) and the "stopped" state (sign of
//     T) are compare + select, in the trips that can stop only: see no_stop_need().
#pragma once
#include <hip/hip_runtime.h>

#include "raster_common.h"

namespace rb {

using rc::f2;
using rc::fma2;
using rc::splat;

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kHuge = 1099511627776.0f;                     // 2^40
constexpr unsigned kBelow255Bits = 0x3B808080u;               // the float below 1.0f / 255.0f (= 0x3B808081)

// d = clamp(a * b + c) on both halves; b wave-uniform (an SGPR pair: one constant-bus read).  hipcc does not fold a
// clamp into a packed fp32 op (it emits v_pk_fma + two v_max ... clamp), hence the inline instruction.
__device__ __forceinline__ f2 pk_fma_clamp(f2 a, f2 b, f2 c) {
  f2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
// the same where `a` may be the result of the transcendental unit (v_exp_f32) of the PREVIOUS instruction: gfx940+ needs
// one wait state between a trans op and a VALU op that reads its result; hipcc's hazard recogniser inserts it for its
// own instructions but does not look into inline assembly (without it the high half was read stale on part of the
// lanes: found as wrong pixels (x & 4) == 0 of every wave's second pixel rows)
__device__ __forceinline__ f2 pk_fma_clamp_after_trans(f2 a, f2 b, f2 c) {
  f2 d;
  asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}

// loop constants (hoisted by the callers: two VGPR pairs)
struct Consts {
  f2 neg_k255;      // -k 2^40, k = the float below 1/255f
  f2 neg_half;      // -0.5 2^40
};
__device__ __forceinline__ Consts consts() {
  Consts c;
  c.neg_k255 = splat(-(__uint_as_float(kBelow255Bits) * kHuge));
  c.neg_half = splat(-0.5f * kHuge);
  return c;
}

// is (opacity, conic) a SIMPLE record?  con = (-0.5 conic.x, -0.5 conic.z, conic.y, opacity)
__device__ __forceinline__ bool is_simple(const float4 con) {
  const float qa = -2.f * con.x, qc = -2.f * con.y, qb = con.z, o = con.w;
  return (o > 0.f) && (o <= 0.99f) && (qa > 0.f) && (qc > 0.f) && (qb * qb <= 0.9990234375f * (qa * qc));
}

// The staged form of a record, 12 floats (three 16-byte LDS words).  Operands that are broadcast into packed
// operations sit in slots 0-2 of a word (hipcc copies a broadcast operand out of slot 3), scalars in slot 3:
//   a = (y, C, red, x)   b = (green, blue, A, B)   c = (L | opacity, need factor, depth, 0)
// SIMPLE:  A = -0.5 conic.x log2e, C = -0.5 conic.z log2e, B = -conic.y log2e, L = log2(opacity)
// GENERIC: A = -0.5 conic.x,       C = -0.5 conic.z,       B = conic.y,        opacity
struct Staged { float4 a, b, c; };

// `need factor`: sqrt(1.25e-4) / (1 - amax), amax >= every alpha this record can produce (see no_stop_need)
__device__ __forceinline__ float need_factor(float o, bool simple) {
  const float amax = simple ? o : fminf(0.99f, o);               // NaN opacity -> NaN factor -> "may stop"
  return 0.011180340f * __frcp_rn(1.0f - amax);
}

__device__ __forceinline__ Staged stage(const float4 con, float px, float py, float r, float g, float b, float depth,
                                        bool simple) {
  Staged s;
  const float o = con.w;
  const float A = simple ? con.x * kLog2e : con.x;
  const float C = simple ? con.y * kLog2e : con.y;
  const float B = simple ? -(con.z * kLog2e) : con.z;
  const float L = simple ? __log2f(o) : o;                       // v_log_f32: 1 ulp, one instruction (libm's log2f: ~20)
  s.a = make_float4(py, C, r, px);
  s.b = make_float4(g, b, A, B);
  s.c = make_float4(L, need_factor(o, simple), depth, 0.f);
  return s;
}
// a record that changes nothing (pads a list): SIMPLE with opacity 0 -> L = -inf -> alpha = 0
__device__ __forceinline__ Staged stage_noop() {
  Staged s;
  s.a = make_float4(0.f, 0.f, 0.f, 0.f);
  s.b = make_float4(0.f, 0.f, 0.f, 0.f);
  s.c = make_float4(-INFINITY, 0.011180340f, 0.f, 0.f);
  return s;
}

// While every pixel of a wave (inside the image) has T > need0 need1 = 1.25e-4 / ((1 - amax0)(1 - amax1)), neither of
// a trip's two records can trip the stop test: after the first T >= T (1 - amax0) (1 - 2^-22), after the second
// T' >= T (1 - amax0)(1 - amax1)(1 - 2^-21) > 1.2e-4 > 1e-4.  (a per-trip test against the constant
// 1/80, which covers ONE record of alpha <= 0.99, is wrong for two.)  The kernels use the largest factor of a wave's
// list (of a staged batch) for both records: one number per batch, no per-trip operand.
__device__ __forceinline__ float no_stop_need(float f0, float f1) { return f0 * f1; }

// exponent (base 2, log2(opacity) included) of a SIMPLE record at the lane's two pixels.
//   dx = x - px;  t = A dx;  nb = B dx;  qxl = fma(t, dx, L) per pixel (callers: L may differ per pixel)
__device__ __forceinline__ f2 p2_simple(float nb, float C, f2 qxl, f2 dy) {
  const f2 u = fma2(splat(C), dy, splat(nb));
  return fma2(u, dy, qxl);
}
// ... its alpha and the 0/1 "not skipped" factor
__device__ __forceinline__ void alpha_of_p2(f2 p2, const Consts& k, f2* alpha, f2* s) {
  f2 G;
  G.x = __builtin_amdgcn_exp2f(p2.x);
  G.y = __builtin_amdgcn_exp2f(p2.y);
  *alpha = G;
  *s = pk_fma_clamp_after_trans(G, splat(kHuge), k.neg_k255);
}
__device__ __forceinline__ void alpha_simple(float nb, float C, f2 qxl, f2 dy, const Consts& k, f2* alpha, f2* s) {
  alpha_of_p2(p2_simple(nb, C, qxl, dy), k, alpha, s);
}
// exponent below which alpha = exp2(p2) is under 1/255 with a margin far beyond v_exp_f32's error: log2(1/255) - 0.02
constexpr float kSkipP2 = -8.0144f;

// alpha of a GENERIC record, in the reference's order (forward.cu:320-333); o per pixel
__device__ __forceinline__ void alpha_generic(float dx, float A, float B, float C, f2 o, f2 dy, f2* alpha, f2* s) {
  const float qx = (A * dx) * dx;
  const float bx = B * dx;
  const f2 qy = (splat(C) * dy) * dy;
  const f2 power = (splat(qx) + qy) - splat(bx) * dy;          // A, C carry the -0.5
  const f2 p2 = power * splat(kLog2e);                         // __expf(x) = v_exp_f32(log2(e) x)
  f2 G;
  G.x = __builtin_amdgcn_exp2f(p2.x);
  G.y = __builtin_amdgcn_exp2f(p2.y);
  f2 a = o * G;
  a.x = fminf(0.99f, a.x);
  a.y = fminf(0.99f, a.y);
  *alpha = a;
  s->x = ((power.x > 0.0f) | (a.x < 1.0f / 255.0f)) ? 0.f : 1.f;
  s->y = ((power.y > 0.0f) | (a.y < 1.0f / 255.0f)) ? 0.f : 1.f;
}

// [T > 0.5] of the two pixels, counted (what chain<MEDCNT> does first; for records a wave skips as a whole)
struct Px;
__device__ __forceinline__ void count_above_half(Px& p, const Consts& k);

// running state of the lane's two pixels.  T < 0 (sign bit) <=> the pixel has stopped or lies outside the image; |T| is
// its transmittance.
struct Px {
  f2 T, C0, C1, C2, D, cnt;
};

__device__ __forceinline__ void count_above_half(Px& p, const Consts& k) {
  p.cnt += pk_fma_clamp(p.T, splat(kHuge), k.neg_half);
}

__device__ __forceinline__ bool dead(float T) { return __float_as_int(T) < 0; }

// the T / colour / depth chain of one record.  MEDCNT: count this record if the incoming T is above 0.5;
// MEAN: depth channel = sum of depth * weight; NOSTOP: this record cannot trip the stop test for any pixel of the wave.
// -> the weights alpha T of the two pixels (0 where the record was skipped or not blended)
template <bool MEDCNT, bool MEAN, bool NOSTOP>
__device__ __forceinline__ f2 chain(Px& p, f2 alpha, f2 s, float cr, float cg, float cb, float dep, const Consts& k) {
  if constexpr (MEDCNT) count_above_half(p, k);
  const f2 aT = alpha * p.T;
  const f2 cand = aT * s;
  const f2 Tn = p.T - cand;
  f2 wgt = cand;
  if constexpr (NOSTOP) {
    p.T = Tn;
  } else {
    // a live pixel has T >= 1e-4, so Tn < 1e-4 alone means "stop" (this record is not blended); a stopped pixel
    // (T < 0) has Tn <= 0: it "stops" again, which changes nothing
    const bool stop0 = Tn.x < 0.0001f, stop1 = Tn.y < 0.0001f;
    wgt.x = stop0 ? 0.f : cand.x;
    wgt.y = stop1 ? 0.f : cand.y;
    p.T.x = stop0 ? -fabsf(p.T.x) : Tn.x;
    p.T.y = stop1 ? -fabsf(p.T.y) : Tn.y;
  }
  p.C0 = fma2(splat(cr), wgt, p.C0);
  p.C1 = fma2(splat(cg), wgt, p.C1);
  p.C2 = fma2(splat(cb), wgt, p.C2);
  if constexpr (MEAN) p.D = fma2(splat(dep), wgt, p.D);
  return wgt;
}

// After a batch: index (within the batch's record sequence of this pixel) of the record at which the pixel crossed 0.5,
// or -1.  cnt = records of the batch whose incoming T was above 0.5 (float, exact); T = the pixel's state now.
__device__ __forceinline__ int median_index(float cnt, float T) {
  const int c = (int)cnt;
  return (c > 0 && T <= 0.5f) ? c - 1 : -1;
}

}  // namespace rb

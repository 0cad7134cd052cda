// One BATCH of the forward alpha blend, shared by the two forward kernels (raster_blend_sorted_kernel of raster_plan.hip:
// planned; raster_blend_kernel<..., LISTS> of rasterize.hip: per call), so that both spend their time the same way and
// produce the same bits for the same record sequence (forward.cu:261-374 + the w-depth fork's depth channel).
//
// A workgroup of 256 threads = one vertical pair of 16 x 16 tiles; wave w owns the 16 x 8 pixel block of rows
// [8 w, 8 w + 8) of the pair (waves 0-1: upper tile, 2-3: lower tile), thread (lx, r) the pixels (lx, 8 w + r) and
// (lx, 8 w + r + 4): same x, so the dx-only part of the exponent is shared and the two pixels run as the halves of packed
// fp32 operations (raster_blend_math.h).  blend_batch():
//   1. stages ns <= kStageB records (the caller's `fetch(ri)` hands over conic | opacity, pixel centre, colour, depth and
//      which of the pair's tiles the record's rect covers) in their loop form (three 16-byte LDS words);
//   2. decides per (record, wave) whether the record's alpha >= 1/255 ellipse can reach the wave's pixel block — an
//      exact conservative test, so a wave skips records by construction — and builds the four waves' ordered lists;
//   3. blends each wave's list front to back, two records per trip, in loops without wave-uniform decisions;
//   4. -> true when every pixel of the tile pair has stopped (forward.cu:304-307).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "raster_blend_math.h"
#include "raster_common.h"

namespace rbody {

using rc::f2;
using rc::kBlock;
using rc::kTileY;
using rc::splat;

#ifndef OCRF_PLAN_STAGE
#define OCRF_PLAN_STAGE 128
#endif
constexpr int kStageB = OCRF_PLAN_STAGE;          // records staged per batch (a tile pair saturates after ~110 at cfg2)
constexpr int kTripB = 2;                         // records per trip of the blend loop (the no-stop bound is per pair)
constexpr int kStagePartsB = kBlock / kStageB;    // threads per staged record: each tests 4 / kStageParts waves
constexpr int kReachPerThreadB = 4 / kStagePartsB;
constexpr int kSrcWavesB = kStageB / 64;          // waves that hold one copy of the staged batch
static_assert(kStageB == 128 || kStageB == 256, "one or two threads per staged record");
constexpr int kListLen = kStageB + 2 * kTripB + 2;      // (aligned: a trip reads an entry PAIR as one 32-bit word)

// the LDS a batch lives in (the caller declares it: static arrays in both kernels)
struct Lds {
  float4 *l_a, *l_b, *l_c;                        // [kStageB + 1]; slot kStageB: rb::stage_noop() (pads odd lists)
  unsigned short (*l_list)[kListLen];             // [4]
  int (*l_lcnt)[4];                               // [kSrcWavesB]: [source wave of the batch copy][destination wave]
  int* l_generic;                                 // [4]: the wave's list of this batch holds a GENERIC record
  int* l_fmax;                                    // [4]: largest need factor (float bits) of its list of this batch
};

// the lane's place in the tile pair
struct Tile {
  int tyA, wave, lane, tid;
  float pixf_x;
  f2 pixf_y;
  bool inside0, inside1;
  float bx0, bx1;                                 // the pair's pixel columns (pixel centres are integer coordinates)
};

struct Fetched {
  float4 con;                                     // (-0.5 conic.x, -0.5 conic.z, conic.y, opacity)
  float2 pix;                                     // projected centre, pixels
  const float* col;                               // the Gaussian's RGB
  float depth;                                    // view-space depth (median / mean depth channel)
  bool covA, covB;                                // the record's tile rect covers the pair's upper / lower tile
};

// diagnostic builds only (STATS): phase stamps and record counts of the caller
struct NoStats {
  unsigned n_staged = 0, n_listed = 0, n_eval = 0;
  __device__ __forceinline__ void stamp(int) {}
};

template <bool MEDIAN, bool STATS, class Fetch, class Stats>
__device__ __forceinline__ bool blend_batch(rb::Px& px, const rb::Consts& kc, const Tile& t, const Lds& L, int ns,
                                            Fetch&& fetch, int variant, Stats& diag) {
  const int tid = t.tid, wave = t.wave, lane = t.lane;
  if constexpr (STATS) diag.n_staged += ns;
  // ---- stage ns records; which waves can each one reach?  kStageBarts threads per record, each testing
  // kReachPerThreadB of the four waves (the copies of the batch live in waves [part * kSrcWavesB, ...)) ----
  const int part = tid / kStageB, ri = tid % kStageB;
  bool reach[kReachPerThreadB];
  bool simple = true;
  float nfac = 0.f;
#pragma unroll
  for (int j = 0; j < kReachPerThreadB; ++j) reach[j] = false;
  if (tid < 4) {
    L.l_generic[tid] = 0;
    L.l_fmax[tid] = 0;
  }
  if (ri < ns) {
    const Fetched fr = fetch(ri);              // (every thread of a record: the part-0 thread also stages it)
    const float4 con = fr.con;
    const float2 pix = fr.pix;
    const float o = con.w;
    // the power below which alpha = o exp(power) is under 1/255 whatever the pixel (1 % margin for v_exp_f32
    // and the log2(e) multiply); o <= 0: +inf (never rendered); NaN opacity: NaN (evaluated in full)
    const float thr = (o > 0.f) ? (__logf(1.0f / (255.0f * o)) - 0.01f) : ((o <= 0.f) ? INFINITY : o);
    simple = rb::is_simple(con);
    nfac = rb::need_factor(o, simple);
    if (part == 0) {
      const rb::Staged st = rb::stage(con, pix.x, pix.y, fr.col[0], fr.col[1], fr.col[2], fr.depth, simple);
      L.l_a[ri] = st.a;
      L.l_b[ri] = st.b;
      L.l_c[ri] = st.c;
    }
    // alpha >= 1/255 needs power >= thr, i.e. Q(dx, dy) = 0.5 (A dx^2 + C dy^2) + B dx dy <= -thr.  The minimum of
    // the convex Q over a wave's pixel block (a box in (dx, dy)) is 0 if the centre lies inside, else it is on
    // the box's boundary: per edge a clamped 1-D minimiser.  The block is skipped only if that minimum exceeds
    // -thr by more than the rounding of both evaluations (<= 1e-6 of the sum of the terms' magnitudes; 4e-6
    // is allowed for, and 1e-3 absolute: more than the difference between the two evaluation orders of
    // raster_blend_math.h) — so a skipped record has alpha < 1/255 at every pixel of the block, where the
    // reference skips it too (forward.cu:331-333).  Anything unusual (NaN, non-convex conic) is evaluated in full.
    const float qa = -2.f * con.x, qc = -2.f * con.y, qb = con.z;
    const bool convex = (qa > 0.f) && (qc > 0.f) && (qa * qc - qb * qb > 0.f);
    const bool never = thr >= 0.f;                         // opacity < 1/255: no pixel ever blends it
    const float lim = -thr;
    const float inv_a = 1.f / qa, inv_c = 1.f / qc;
    const float dxlo = pix.x - t.bx1, dxhi = pix.x - t.bx0;
    const float Dx = fmaxf(fabsf(dxlo), fabsf(dxhi));
#pragma unroll
    for (int j = 0; j < kReachPerThreadB; ++j) {
      const int w = part * kReachPerThreadB + j;
      const bool cov = (w < 2) ? fr.covA : fr.covB;
      const float by0 = (float)(t.tyA * kTileY + 8 * w), by1 = by0 + 7.f;
      const float dylo = pix.y - by1, dyhi = pix.y - by0;
      const float Dy = fmaxf(fabsf(dylo), fabsf(dyhi));
      bool skip = false;
      if (convex && !(thr != thr)) {
        const bool in_x = (dxlo <= 0.f) && (dxhi >= 0.f), in_y = (dylo <= 0.f) && (dyhi >= 0.f);
        if (!(in_x && in_y)) {
          auto Q = [&](float dx, float dy) { return 0.5f * (qa * dx * dx + qc * dy * dy) + qb * dx * dy; };
          auto clampf = [](float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); };
          const float q0 = Q(dxlo, clampf(-qb * dxlo * inv_c, dylo, dyhi));
          const float q1 = Q(dxhi, clampf(-qb * dxhi * inv_c, dylo, dyhi));
          const float q2 = Q(clampf(-qb * dylo * inv_a, dxlo, dxhi), dylo);
          const float q3 = Q(clampf(-qb * dyhi * inv_a, dxlo, dxhi), dyhi);
          const float qmin = fminf(fminf(q0, q1), fminf(q2, q3));
          const float M = 0.5f * (qa * Dx * Dx + qc * Dy * Dy) + fabsf(qb) * Dx * Dy;
          skip = (qmin - 4e-6f * M - 1e-3f) > lim;
        }
      }
      reach[j] = cov && !never && !skip;
    }
  }
  // ordered per-wave lists of staged indices: the kSrcWavesB waves of a part hold the batch in order
  const int src = wave % kSrcWavesB;
  int lrank[kReachPerThreadB];
#pragma unroll
  for (int j = 0; j < kReachPerThreadB; ++j) {
    const unsigned long long m = __ballot(reach[j]);
    lrank[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) L.l_lcnt[src][part * kReachPerThreadB + j] = __popcll(m);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kReachPerThreadB; ++j) {
    const int w = part * kReachPerThreadB + j;
    int base = 0, tot = 0;
#pragma unroll
    for (int sw = 0; sw < kSrcWavesB; ++sw) {
      const int c = L.l_lcnt[sw][w];
      if (sw < src) base += c;
      tot += c;
    }
    if (reach[j]) {
      // entry = the record's byte offset in the staged arrays | SIMPLE flag
      L.l_list[w][base + lrank[j]] = (unsigned short)((ri << 4) | (simple ? 0x8000 : 0));
      if (!simple) L.l_generic[w] = 1;
      // positive floats order like their bits; a NaN factor (NaN opacity) wins: "may stop" throughout
      atomicMax(&L.l_fmax[w], __float_as_int(nfac));
    }
    if (ri == 0) {                         // pad: a trip reads kTripB entries whatever the list's length
#pragma unroll
      for (int q = 0; q < 2 * kTripB; ++q) L.l_list[w][tot + q] = (unsigned short)((kStageB << 4) | 0x8000);
    }
  }
  int n_mine = 0;                          // length of THIS wave's list
#pragma unroll
  for (int sw = 0; sw < kSrcWavesB; ++sw) n_mine += L.l_lcnt[sw][wave];
  __syncthreads();
  n_mine = __builtin_amdgcn_readfirstlane(n_mine);
  const bool generic_batch = __builtin_amdgcn_readfirstlane(L.l_generic[wave]) != 0 || (variant & 2);
  // no record of this wave's list can trip the stop test while every pixel inside the image has T > need
  // (raster_blend_math.h: no_stop_need, with the list's largest factor for both records of a trip)
  const float fmax_w = __int_as_float(__builtin_amdgcn_readfirstlane(L.l_fmax[wave]));
  const float need = rb::no_stop_need(fmax_w, fmax_w);
  if constexpr (STATS) diag.n_listed += n_mine;
  diag.stamp(1);

  // ---- blend this wave's records front to back (raster_blend_math.h), two records per trip: two independent
  // exponent / alpha chains in flight.  Measured issue costs on gfx950 (tools/ubench/valu_cost.hip; plain VALU = 1):
  // packed f32 1.4 (for two pixels), v_exp_f32 2.4, a compare + select pair 2.5 — so the loop carries no decision
  // whose outcome is known for the whole wave: the loops run in the order (median count, no stop) -> (median
  // count, stop) -> (no count, no stop) -> (no count, stop), each leaving when its own condition ends.  A SIMD
  // holds five waves of this kernel; they hide the two dependent LDS reads of a trip (list entry -> record).
  {
    const unsigned short* mylist = L.l_list[wave];
    const char* la = reinterpret_cast<const char*>(L.l_a);
    const char* lb = reinterpret_cast<const char*>(L.l_b);
    const char* lc = reinterpret_cast<const char*>(L.l_c);
    struct Rec { float4 a, b; float L, dep; };
    auto load1 = [&](unsigned byte_off, Rec* q) {
      q->a = *reinterpret_cast<const float4*>(la + byte_off);
      q->b = *reinterpret_cast<const float4*>(lb + byte_off);
      if constexpr (MEDIAN) {
        q->L = *reinterpret_cast<const float*>(lc + byte_off);
        q->dep = 0.f;
      } else {
        const float4 c = *reinterpret_cast<const float4*>(lc + byte_off);
        q->L = c.x;
        q->dep = c.z;
      }
    };
    auto load = [&](int k, Rec* rec) {
      const unsigned pair = *reinterpret_cast<const unsigned*>(mylist + k);
      load1(pair & 0x7FF0u, &rec[0]);
      load1((pair >> 16) & 0x7FF0u, &rec[1]);
    };
    auto one = [&](auto med_tag, auto nostop_tag, auto generic_tag, const Rec& q, bool simple_rec) {
      constexpr bool MED = decltype(med_tag)::value, NOSTOP = decltype(nostop_tag)::value;
      constexpr bool GEN = decltype(generic_tag)::value;
      const float dx = q.a.w - t.pixf_x;
      const f2 dy = splat(q.a.x) - t.pixf_y;
      f2 alpha, s;
      auto fast = [&]() {
        const float t = q.b.z * dx;
        const float nb = q.b.w * dx;
        const float qxl = __builtin_fmaf(t, dx, q.L);
        rb::alpha_simple(nb, q.a.y, splat(qxl), dy, kc, &alpha, &s);
      };
      if constexpr (GEN) {
        if (simple_rec) fast();
        else rb::alpha_generic(dx, q.b.z, q.b.w, q.a.y, splat(q.L), dy, &alpha, &s);
      } else {
        fast();
      }
      rb::chain<MEDIAN && MED, !MEDIAN, NOSTOP>(px, alpha, s, q.a.z, q.b.x, q.b.y, q.dep, kc);
    };
    // wave-level tests, each two compares on the VALU and scalar logic (a ballot of a combined predicate costs a
    // select + a compare more).  Stopped and outside pixels carry T < 0: as integers their bits are negative.
    const unsigned long long in0 = __ballot(t.inside0), in1 = __ballot(t.inside1);
    auto may_stop = [&]() {
      return ((__ballot(!(px.T.x > need)) & in0) | (__ballot(!(px.T.y > need)) & in1)) != 0ull || (variant & 1);
    };
    auto any_above_half = [&]() {
      return __ballot(max(__float_as_int(px.T.x), __float_as_int(px.T.y)) > 0x3F000000) != 0ull;
    };
    auto any_alive = [&]() { return __ballot((__float_as_int(px.T.x) & __float_as_int(px.T.y)) >= 0) != 0ull; };
    int k = 0;
    Rec rec[kTripB];
    if (generic_batch) {
      // rare (an opacity above 0.99, a nearly singular conic, NaNs): one loop, one record per trip, every decision
      // per record (a second record in flight here costs the whole kernel a wave per SIMD in registers)
      for (; k < n_mine; ++k) {
        if (!any_alive()) break;
        if constexpr (STATS) diag.n_eval += 1;
        const unsigned ent = __builtin_amdgcn_readfirstlane((unsigned)mylist[k]);
        Rec q;
        load1(ent & 0x7FF0u, &q);
        if constexpr (MEDIAN) q.dep = 0.f;
        one(std::true_type{}, std::false_type{}, std::true_type{}, q, (ent & 0x8000u) != 0u);
      }
    } else {
      if constexpr (MEDIAN) {
        for (; k < n_mine; k += kTripB) {
          if (!any_above_half() || may_stop()) break;
          if constexpr (STATS) diag.n_eval += kTripB;
          load(k, rec);
          one(std::true_type{}, std::true_type{}, std::false_type{}, rec[0], true);
          one(std::true_type{}, std::true_type{}, std::false_type{}, rec[1], true);
        }
        for (; k < n_mine; k += kTripB) {
          if (!any_alive()) { k = n_mine; break; }
          if (!any_above_half()) break;
          if constexpr (STATS) diag.n_eval += kTripB;
          load(k, rec);
          one(std::true_type{}, std::false_type{}, std::false_type{}, rec[0], true);
          one(std::true_type{}, std::false_type{}, std::false_type{}, rec[1], true);
        }
      }
      for (; k < n_mine; k += kTripB) {
        if (may_stop()) break;
        if constexpr (STATS) diag.n_eval += kTripB;
        load(k, rec);
        one(std::false_type{}, std::true_type{}, std::false_type{}, rec[0], true);
        one(std::false_type{}, std::true_type{}, std::false_type{}, rec[1], true);
      }
      for (; k < n_mine; k += kTripB) {
        if (!any_alive()) break;                                        // every pixel stopped
        if constexpr (STATS) diag.n_eval += kTripB;
        load(k, rec);
        one(std::false_type{}, std::false_type{}, std::false_type{}, rec[0], true);
        one(std::false_type{}, std::false_type{}, std::false_type{}, rec[1], true);
      }
    }
    if constexpr (MEDIAN) {
      // the record at which a pixel crossed 0.5 in this batch, if it did: looked up once (raster_blend_math.h)
      const int m0 = rb::median_index(px.cnt.x, px.T.x), m1 = rb::median_index(px.cnt.y, px.T.y);
      if (m0 >= 0) px.D.x = L.l_c[(mylist[m0] & 0x7FF0u) >> 4].z;
      if (m1 >= 0) px.D.y = L.l_c[(mylist[m1] & 0x7FF0u) >> 4].z;
      px.cnt = splat(0.f);
    }
  }
  // every pixel saturated -> stop (forward.cu:304-307)
  return __syncthreads_count(rb::dead(px.T.x) && rb::dead(px.T.y)) == kBlock;
}

}  // namespace rbody

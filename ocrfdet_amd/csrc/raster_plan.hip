// Static render plans for the Gaussian rasteriser (MI355X, gfx950).
//
// In OcRFDet the Gaussian MEANS are the fixed voxel grid (view_transformer_ocrf.py:651-673,690-692) and the
// cameras are fixed per calibration; only scales / rotations / opacities / colours change from step to step
// (the S/R/A/C heads, :1130-1133).  Everything of the rasteriser's front end that depends on (mean, camera)
// alone is therefore a constant of the calibration: the near-plane cull and view-space depth
// (forward.cu:166-171 in_frustum, auxiliary.h:139-164), the projected centre (:196-199), the Jacobian of
// computeCov2D (:83-98), and with them the whole front-to-back ORDER of a view's Gaussians — the reference's
// sort key is (tile | depth bits), ties by Gaussian id (rasterizer_impl.cu:226-267, stable radix sort of
// duplicateWithKeys' id-ordered output).
//
// A plan (built once per (means, cameras)) keeps, per view, the Gaussians that can ever be visible — in front
// of the near plane and inside the frame for any world-space extent up to a stated bound — ALREADY SORTED by
// (depth bits, id), with their static per-record data, and per bin of tiles the list positions of the records that
// can ever reach it (candidate lists).  A render then is
//   raster_plan_head_kernel     conic / tile rect (tightened by the opacity) of the HEAD of every rendered view's list, in
//                               list order — as many entries as the last render of the view needed, up to all of them;
//   raster_blend_sorted_kernel  (first pass) a tile pair filters the head of the list, then its bin's candidates, by tile
//                               rect — the survivors arrive in the reference's per-tile order, so there is no sort, no
//                               depth bucket, no carry — and each of its four waves blends only the records whose
//                               alpha >= 1/255 ellipse reaches the wave's own 16x8 pixel block (exact conservative test
//                               at staging, so a wave skips records by construction); a tile pair that needs records
//                               behind the prepared head is handed to the second pass;
//   raster_plan_check_kernel    the extent check of all Gaussians (status bit 4) and, only if a tile pair asked, the rest
//                               of the lists — prepared ONCE, never per tile pair;
//   raster_blend_sorted_kernel  (second pass) the tile pairs handed over (none in the steady state: it retires at once)
// against zero-fill -> preprocess (all P x V pairs) -> bucket scan -> scatter -> blend (with an in-LDS bitonic
// sort) of rasterize.hip.  Per-pixel arithmetic and its order are those of raster_blend_kernel: colour, depth
// and final_T are bit-identical to the per-call pipeline (tests/test_raster_plan_gpu.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "launch.h"
#include "ocrf_hip.h"
#include "raster_blend_body.h"
#include "raster_blend_math.h"
#include "radix_emit.h"
#include "raster_common.h"

namespace {

using namespace rc;

constexpr unsigned kPlanMagic = 0x4F435250u;      // "OCRP"
constexpr int kHeaderInts = 16;                   // magic, P, V, H, W, gx, gy, bound bits, total lo, total hi, ...
#ifndef OCRF_PLAN_SCAN
#define OCRF_PLAN_SCAN 1      // 256 rects per scan round.  Measured on one box (tools/ab_gauss.sh), cfg2 step / blend alone: init
                              // 0.215 ms / 137 us with 1, 0.219 / 141 with 2, 0.225 / 144 with 4 (the unrolled round's registers
                              // and instructions are paid in the direct region too); objects 2.07 / 1.53 ms, 2.04 / 1.51, 2.00 / 1.49:
                              // a candidate costs ~ 4 ns per workgroup whatever the round's width — issue-bound, not latency-bound
#endif
constexpr int kStageP = rbody::kStageB;           // records staged per batch (raster_blend_body.h)
constexpr int kScanUnrollP = OCRF_PLAN_SCAN;      // rect batches in flight in the scan
constexpr int kSrcWaves = rbody::kSrcWavesB;
constexpr int kCapPos = kStageP + kScanUnrollP * kBlock;
constexpr unsigned kPosMask = 0x3FFFFFFFu;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PlanLayout {
  size_t header, cams, g_mask, g_off, e_q0, e_q1, s_id, s_key, s_pix, s_e, bytes;
};

inline void plan_layout(int P, int V, long T, PlanLayout* L) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  L->header = take((size_t)(kHeaderInts + V + 1) * 4);
  L->cams = take((size_t)V * sizeof(Camera));
  L->g_mask = take((size_t)P * 4);          // per Gaussian: bit v = kept in view v
  L->g_off = take((size_t)P * 4);           //               first record in Gaussian-major order
  L->e_q0 = take((size_t)T * 16);           // per record, Gaussian-major: (A00, A01, A02, A10) of A = J W,
  L->e_q1 = take((size_t)T * 16);           //   (A11, A12, pixel x, pixel y)
  L->s_id = take((size_t)T * 4);            // per record, sorted (view-major, depth bits then id): Gaussian id,
  L->s_key = take((size_t)T * 4);           //   depth bits,
  L->s_pix = take((size_t)T * 8);           //   pixel centre,
  L->s_e = take((size_t)T * 4);             //   the record's index in Gaussian-major order
  L->bytes = off;
}

struct BuildLayout {      // workspace of the build
  size_t keys, rec_id, counts, sort, scan, bytes;
};

constexpr int kCountInts = 32 + 4;               // per-view kept counts | total | "a depth outside the key range" | -, -

inline void build_layout(int P, long T, BuildLayout* L) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  L->keys = take((size_t)T * 4);                 // sort key of every record (Gaussian-major): view | depth bits
  L->rec_id = take((size_t)T * 4);               // Gaussian id of every record
  L->counts = take((size_t)kCountInts * 4);
  L->sort = take(ocrf::radix_sort_ids_bytes((int)T));
  L->scan = take(ocrf::exclusive_scan_bytes(P));
  L->bytes = off;
}

struct DynLayout {        // per-call scratch of the planned render
  size_t rect, con, deferred, ccnt, ccand, flag, bytes;
};

constexpr int kSegC = 1024;                       // candidates per segment of the per-call compaction (deep views)
inline long max_segments(long cand_cap, long n_bins_total) { return cand_cap / kSegC + n_bins_total + 1; }

constexpr int kMaxDeferred = 1 << 20;             // tile pairs (of all items of a call) a second pass can take over

inline void dyn_layout(long T, int n_sets, DynLayout* L, long cand_cap = 0, long n_bins_total = 0) {
  size_t off = 0;
  auto take = [&](size_t bts) { size_t o = off; off += align_up(bts, 256); return o; };
  const size_t n = (size_t)T * n_sets;
  L->rect = take(n * sizeof(Rect));
  L->con = take(n * 16);
  L->deferred = take((size_t)kMaxDeferred * 4);   // tile pairs that ran out of prepared records (second pass)
  // deep views (raster_plan_compact_kernel): per segment of a bin's candidates how many reach the bin THIS call, and
  // those — (list position, mask of the bin's tiles the record's rect covers) — at the front of the segment's range
  L->ccnt = take(cand_cap > 0 ? (size_t)max_segments(cand_cap, n_bins_total) * 4 : 0);
  L->ccand = take(cand_cap > 0 ? (size_t)cand_cap * 8 : 0);
  L->flag = take(1024);                           // control words (kCtl*): guard flag, ticket queue, heads, reach — LAST
  L->bytes = off;
}

// ---------------------------------------------------------------------------------------------
// The build, without a host read anywhere (a plan per SAMPLE — the reference recomputes the render cameras from the
// dataloader's c2w for every sample, view_transformer_ocrf.py:1140-1152 — is five kernels + one radix sort into
// buffers of a fixed record CAPACITY; hipGraph-capturable):
//   1. classify       one thread per Gaussian, all views: the set of views that can ever see it (bit mask) and how
//                     many — in front of the near plane and inside the frame for every world-space extent <= bound
//   2. exclusive scan of the per-Gaussian counts -> the Gaussian's first record, in GAUSSIAN-MAJOR record order e
//   3. fill records   per record: the rows of J W and the pixel centre (static per (mean, camera)), its Gaussian id and
//                     its sort key (view << 27 | depth bits - 0x3E000000): depths in [0.125, 8191) m order by their
//                     27 low-order-relevant bits, so ONE 32-bit key sorts every view's list at once
//   4. stable radix sort of the keys (index_prep.hip), 4 passes.  Equal (view, depth) keep their Gaussian-major order
//                     = ascending id: the reference's order (stable sort of duplicateWithKeys' id-ordered output,
//                     rasterizer_impl.cu:226-267).  Records beyond the total carry the key ~0 and stay at the end.
//   5. gather         sorted position -> (record, id, depth bits, pixel centre); header: view offsets, magic.
// The magic word is written LAST and only if everything held: total <= capacity, every depth inside the key range,
// no look-back scan gave up.  A plan without it is refused by every render (status bit 8).
// ---------------------------------------------------------------------------------------------
constexpr unsigned kKeyBase = 0x3E000000u, kKeyDepthBits = 27, kKeyDepthMask = (1u << kKeyDepthBits) - 1u;

__global__ __launch_bounds__(kBlock) void plan_classify_kernel(int P, int V, int gx, int gy, int W, int H,
                                                               const float* __restrict__ means3D,
                                                               const Camera* __restrict__ cams, float bound,
                                                               unsigned* __restrict__ g_mask, int* __restrict__ g_cnt) {
  const int id = blockIdx.x * kBlock + threadIdx.x;
  const bool live = id < P;
  const int idc = live ? id : P - 1;
  const float px = means3D[3 * idc], py = means3D[3 * idc + 1], pz = means3D[3 * idc + 2];
  unsigned m = 0;
  for (int v = 0; v < V; ++v) {
    const Camera& cam = cams[v];
    StaticPoint sp;
    bool keep = false;
    if (static_point(cam, px, py, pz, &sp)) {
      float A[2][3];
      jacobian_rows(cam, sp.j00, sp.j02, sp.j11, sp.j12, A);
      const float rb = radius_bound(A, bound);
      keep = live && !surely_outside(ndc2pix(sp.projx, W), ndc2pix(sp.projy, H), rb, gx, gy);
    }
    m |= keep ? (1u << v) : 0u;      // (per-view counts: from the sorted keys, plan_gather_kernel — 8 000 waves adding
  }                                  // to one word per view cost 450 us in atomics)
  if (live) {
    g_mask[id] = m;
    g_cnt[id] = __popc(m);
  }
}

__global__ __launch_bounds__(kBlock) void plan_fill_records_kernel(int P, int W, int H, long cap,
                                                                   const float* __restrict__ means3D,
                                                                   const Camera* __restrict__ cams,
                                                                   const unsigned* __restrict__ g_mask,
                                                                   const int* __restrict__ g_off,
                                                                   const int* __restrict__ counts,
                                                                   float4* __restrict__ e_q0, float4* __restrict__ e_q1,
                                                                   unsigned* __restrict__ keys, int* __restrict__ rec_id,
                                                                   int* __restrict__ bad) {
  const int id = blockIdx.x * kBlock + threadIdx.x;
  const long total = counts[32];
  // keys of the unused tail of the capacity: sorted behind every record
  for (long e = total + (long)blockIdx.x * kBlock + threadIdx.x; e < cap; e += (long)gridDim.x * kBlock) keys[e] = 0xFFFFFFFFu;
  if (id >= P) return;
  unsigned m = g_mask[id];
  long e = g_off[id];
  const float px = means3D[3 * id], py = means3D[3 * id + 1], pz = means3D[3 * id + 2];
  bool out_of_range = false;
  while (m) {
    const int v = __ffs(m) - 1;
    m &= m - 1;
    const Camera& cam = cams[v];
    StaticPoint sp;
    static_point(cam, px, py, pz, &sp);
    float A[2][3];
    jacobian_rows(cam, sp.j00, sp.j02, sp.j11, sp.j12, A);
    const unsigned bits = __float_as_uint(sp.vz);                  // vz > 0.2: bits order like the value
    out_of_range |= !(bits >= kKeyBase && bits - kKeyBase < kKeyDepthMask);
    if (e < cap) {
      e_q0[e] = make_float4(A[0][0], A[0][1], A[0][2], A[1][0]);
      e_q1[e] = make_float4(A[1][1], A[1][2], ndc2pix(sp.projx, W), ndc2pix(sp.projy, H));
      keys[e] = ((unsigned)v << kKeyDepthBits) | ((bits - kKeyBase) & kKeyDepthMask);
      rec_id[e] = id;
    }
    ++e;
  }
  if (out_of_range) atomicOr(bad, 1);
}

// `states`: the look-back state blocks of the build's scans (n_states blocks, stride_words apart; bit 63 of a block's
// first word = that scan gave up)
__global__ __launch_bounds__(kBlock) void plan_gather_kernel(int P, int V, int H, int W, int gx, int gy, float bound,
                                                             long cap, const int* __restrict__ counts,
                                                             const float* __restrict__ cameras,
                                                             const unsigned* __restrict__ sorted_keys,
                                                             const int* __restrict__ sorted_e,
                                                             const int* __restrict__ rec_id,
                                                             const float4* __restrict__ e_q1,
                                                             const unsigned long long* __restrict__ sort_states,
                                                             int n_sort_states, long sort_stride,
                                                             const unsigned long long* __restrict__ scan_state,
                                                             int* __restrict__ header, float* __restrict__ cams_out,
                                                             unsigned* __restrict__ s_id, unsigned* __restrict__ s_key,
                                                             float2* __restrict__ s_pix, unsigned* __restrict__ s_e) {
  const long total = counts[32];
  if (blockIdx.x == 0) {
    __shared__ int s_off[33];
    const long nn = total < cap ? total : cap;
    if (threadIdx.x <= (unsigned)V) {
      // first sorted position whose key belongs to view >= v (the keys are view-major): V + 1 binary searches
      const unsigned long long want = (unsigned long long)threadIdx.x << kKeyDepthBits;
      long lo = 0, hi = nn;
      while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((unsigned long long)sorted_keys[mid] < want) lo = mid + 1; else hi = mid;
      }
      s_off[threadIdx.x] = (int)(threadIdx.x == (unsigned)V ? nn : lo);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int* view_off = header + kHeaderInts;
      for (int v = 0; v <= V; ++v) view_off[v] = s_off[v];
      bool ok = total <= cap && counts[33] == 0 && (scan_state[0] >> 63) == 0;
      for (int i = 0; i < n_sort_states; ++i) ok = ok && (sort_states[i * sort_stride] >> 63) == 0;
      header[1] = P; header[2] = V; header[3] = H; header[4] = W; header[5] = gx; header[6] = gy;
      header[7] = __float_as_int(bound);
      header[8] = (int)(total & 0xFFFFFFFFl); header[9] = (int)(total >> 32);
      header[0] = ok ? (int)kPlanMagic : 0;                       // a plan that did not fit (or failed) is unusable
    }
    for (int i = threadIdx.x; i < V * 36; i += kBlock) cams_out[i] = cameras[i];
  }
  // (the per-view list arrays — s_e, s_id, s_key, s_pix — are written by the sort's last pass: ocrf::RadixPlanEmit)
  if (sorted_e == nullptr) return;
  const long n = total < cap ? total : cap;
  for (long pos = (long)blockIdx.x * kBlock + threadIdx.x; pos < n; pos += (long)gridDim.x * kBlock) {
    const int e = sorted_e[pos];
    const float4 q1 = e_q1[e];
    s_e[pos] = (unsigned)e;
    s_id[pos] = (unsigned)rec_id[e];
    s_key[pos] = (sorted_keys[pos] & kKeyDepthMask) + kKeyBase;
    s_pix[pos] = make_float2(q1.z, q1.w);
  }
}


// ---------------------------------------------------------------------------------------------
// Candidate lists (what a tile may see: rasterizer_impl.cu:70-138 duplicateWithKeys + identifyTileRanges build, per CALL,
// the list of every tile; consumed forward.cu:261-374).  Here the part of that structure that depends on (means, cameras,
// extent bound) alone is a constant of the plan: per (view, bin of bw x bh tile PAIRS) the positions — ascending, i.e. in
// the view's blend order — of the records whose BOUND-inflated tile rect reaches the bin.  The rect a record gets in a call
// (conic_radius_rect with the call's parameters, tightened by its opacity) lies inside the inflated one, so a tile pair
// that walks its bin's candidates sees every record the reference's tile list holds, in the same order, and tests O(own
// records) rects instead of the view's whole list (cfg2: a 4 x 4-tile bin lists ~ 24 000 of a view's ~ 115 000 records).
//   bins buffer: header (16 ints: magic, V, gx, gy, bw, bh, nbx, nby, total, capacity) | b_tab (V * nbx * nby + 1 int
//   pairs: the bin's first candidate, its candidates inside the direct region) | cand (capacity x u32).  The magic word is written last and only if the lists fit: a bins buffer without it is ignored
//   (the blend then walks whole lists: slower, same image).
// Build (no host read, kernels only): inflated bin range per list entry -> per (view, bin, segment) counts -> exclusive
// scan -> ordered compaction of every segment into its bin's list.
// ---------------------------------------------------------------------------------------------
constexpr unsigned kBinsMagic = 0x4F435242u;      // "OCRB"
constexpr int kBinsHeaderInts = 16;
constexpr int kBinSeg = 8192;                     // list entries per (view, bin) workgroup of the count / fill kernels
// The first kBinDirect entries of a view's list — its nearest records, each of which covers a good part of the image — are
// tested by every tile pair directly, in list order (one memory round trip per 256 rects, no candidate indirection: a
// scene whose pixels saturate early never gets past them); a bin's candidates are followed from the first one behind them
// (b_tab[bin].y = how many of its candidates lie inside the direct region).
constexpr int kBinDirect = 512;
static_assert(kBinDirect % kBlock == 0 && kBinDirect <= kBinSeg, "the direct region is whole slices of the first segment");

struct BinsLayout {
  size_t header, b_off, b_seg, seg_bin, cand, bytes;
};

inline void bins_layout(int V, int nbx, int nby, long cand_cap, BinsLayout* L) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  const long nb = (long)V * nbx * nby;
  L->header = take((size_t)kBinsHeaderInts * 4);
  L->b_off = take(((size_t)nb + 1) * 8);
  // segments of kSegC candidates (behind the direct region) for the per-call compaction of deep views: a bin's first
  // segment (b_seg, nb + 1 ints) and the bin of every segment (seg_bin)
  L->b_seg = take(((size_t)nb + 1) * 4);
  L->seg_bin = take((size_t)max_segments(std::max<long>(cand_cap, 1), nb) * 4);
  L->cand = take((size_t)std::max<long>(cand_cap, 1) * 4);
  L->bytes = off;
}

struct BinsBuildLayout {
  size_t range, counts, scan, total, bytes;
};

inline int bins_segments(int P, long T) { return (int)((std::min<long>(P, T) + kBinSeg - 1) / kBinSeg); }

inline void bins_build_layout(int P, int V, long T, int nbins, BinsBuildLayout* L) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  const long n_counts = (long)V * nbins * bins_segments(P, T);
  L->range = take((size_t)T * 4);                 // per list entry: its inflated bin range (bx0, by0, bx1, by1 inclusive; u8 each)
  L->counts = take((size_t)n_counts * 4);
  L->scan = take(ocrf::exclusive_scan_bytes(n_counts));
  L->total = take(256);
  L->bytes = off;
}

// sigma_max(A)^2 of the 2x3 matrix A = J W: the larger eigenvalue of A A^T.  lambda_max(A Sigma A^T + 0.3 I) <= 0.3 +
// sigma_max(A)^2 |Sigma|_2 — the bound radius_bound() takes with the Frobenius norm (kept for the static cull, whose kept
// sets the committed figures were made with), here with the spectral norm: ~ 1.4 x tighter, so shorter candidate lists.
// >= the reference's integer radius (forward.cu:219-232: ceil(3 sqrt(lambda_max))) + 1 px.
__device__ __forceinline__ float radius_bound_spectral(const float A[2][3], float rn) {
  const float a = A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2];
  const float c = A[1][0] * A[1][0] + A[1][1] * A[1][1] + A[1][2] * A[1][2];
  const float b = A[0][0] * A[1][0] + A[0][1] * A[1][1] + A[0][2] * A[1][2];
  const float mid = 0.5f * (a + c);
  const float smax2 = (mid + sqrtf(fmaxf(0.f, mid * mid - (a * c - b * b)))) * 1.0001f;
  return ceilf(3.f * sqrtf(0.3f + smax2 * rn * rn) * 1.001f + 2.f);
}

__global__ __launch_bounds__(kBlock) void plan_bin_range_kernel(const int* __restrict__ header, float bound, int bw,
                                                                int bh, const unsigned* __restrict__ s_e,
                                                                const float4* __restrict__ e_q0,
                                                                const float4* __restrict__ e_q1,
                                                                unsigned* __restrict__ range) {
  if (header[0] != (int)kPlanMagic) return;
  const int v = blockIdx.y;
  const int gx = header[5], gy = header[6];
  const int off = header[kHeaderInts + v], nv = header[kHeaderInts + v + 1] - off;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nv) return;
  const unsigned e = s_e[off + i];
  const float4 q0 = e_q0[e], q1 = e_q1[e];
  const float A[2][3] = {{q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}};
  const float rb = radius_bound_spectral(A, bound);
  unsigned r = 0x000000FFu;                       // bx0 = 255 > bx1 = 0: reaches no bin
  if (rb == rb && rb < 1.0e9f) {
    // the reference's rect formula (auxiliary.h:46-56) with the bound radius: monotone in the radius, so it contains the
    // rect of every call
    const int x0 = min(gx, max(0, (int)((q1.z - rb) / (float)kTileX)));
    const int y0 = min(gy, max(0, (int)((q1.w - rb) / (float)kTileY)));
    const int x1 = min(gx, max(0, (int)((q1.z + rb + (float)(kTileX - 1)) / (float)kTileX)));
    const int y1 = min(gy, max(0, (int)((q1.w + rb + (float)(kTileY - 1)) / (float)kTileY)));
    if (x1 > x0 && y1 > y0)
      r = (unsigned)(x0 / bw) | ((unsigned)(y0 / (2 * bh)) << 8) | ((unsigned)((x1 - 1) / bw) << 16) |
          ((unsigned)((y1 - 1) / (2 * bh)) << 24);
  } else {
    r = 0xFFFF0000u;                              // NaN / Inf: a candidate of every bin
  }
  range[off + i] = r;
}

__device__ __forceinline__ bool bin_in_range(unsigned r, int bx, int by) {
  return bx >= (int)(r & 255u) && by >= (int)((r >> 8) & 255u) && bx <= (int)((r >> 16) & 255u) && by <= (int)(r >> 24);
}

// grid (segments, bins of a view, views): how many entries of the segment reach the bin
__global__ __launch_bounds__(kBlock) void plan_bin_count_kernel(const int* __restrict__ header, int nbx, int n_seg,
                                                                const unsigned* __restrict__ range,
                                                                int* __restrict__ counts) {
  const int seg = blockIdx.x, bin = blockIdx.y, v = blockIdx.z;
  const int nbins = gridDim.y;
  int n = 0;
  if (header[0] == (int)kPlanMagic) {
    const int off = header[kHeaderInts + v], nv = header[kHeaderInts + v + 1] - off;
    const int bx = bin % nbx, by = bin / nbx;
    const int lo = seg * kBinSeg, hi = min(nv, lo + kBinSeg);
    for (int i = lo + (int)threadIdx.x; i < hi; i += kBlock) n += bin_in_range(range[off + i], bx, by) ? 1 : 0;
  }
  __shared__ int l_n[4];
  for (int d = 32; d > 0; d >>= 1) n += __shfl_down(n, d);
  if ((threadIdx.x & 63) == 0) l_n[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[((long)v * nbins + bin) * n_seg + seg] = l_n[0] + l_n[1] + l_n[2] + l_n[3];
}

// the same grid: the segment's entries that reach the bin, in order, behind those of the bin's earlier segments
__global__ __launch_bounds__(kBlock) void plan_bin_fill_kernel(const int* __restrict__ header, int nbx, int n_seg,
                                                               long cand_cap, const unsigned* __restrict__ range,
                                                               const int* __restrict__ offsets,
                                                               const int* __restrict__ total, int2* __restrict__ b_tab,
                                                               unsigned* __restrict__ cand) {
  const int seg = blockIdx.x, bin = blockIdx.y, v = blockIdx.z;
  const int nbins = gridDim.y;
  if (header[0] != (int)kPlanMagic) return;
  const long slot = ((long)v * nbins + bin) * n_seg + seg;
  if (seg == 0 && threadIdx.x == 0 && bin == nbins - 1 && v == (int)gridDim.z - 1)
    b_tab[v * nbins + bin + 1] = make_int2(*total, 0);
  if ((long)*total > cand_cap) return;            // the lists do not fit: nothing is written, the buffer stays without magic
  const int off = header[kHeaderInts + v], nv = header[kHeaderInts + v + 1] - off;
  const int bx = bin % nbx, by = bin / nbx;
  const int lo = seg * kBinSeg, hi = min(nv, lo + kBinSeg);
  __shared__ int l_w[4];
  int base = offsets[slot];
  const int first = base;
  int direct = 0;                                 // candidates of the bin inside the direct region (segment 0 only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = lo; i0 < hi; i0 += kBlock) {
    const int i = i0 + (int)threadIdx.x;
    const bool hit = i < hi && bin_in_range(range[off + i], bx, by);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) l_w[wave] = __popcll(m);
    __syncthreads();
    int mine = base, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) mine += l_w[w];
      tot += l_w[w];
    }
    if (hit) cand[mine + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
    base += tot;
    if (i0 < kBinDirect) direct += tot;
    __syncthreads();
  }
  if (seg == 0 && threadIdx.x == 0) b_tab[v * nbins + bin] = make_int2(first, direct);
}

// the segment tables (one workgroup: plan time).  A bin's candidates behind the direct region, in segments of kSegC.
__global__ __launch_bounds__(kBlock) void plan_bins_segments_kernel(int nb, long max_segs, const int2* __restrict__ b_tab,
                                                                    const int* __restrict__ total, long cand_cap,
                                                                    int* __restrict__ b_seg, int* __restrict__ seg_bin) {
  if ((long)*total > cand_cap) return;
  __shared__ int l_carry;
  if (threadIdx.x == 0) l_carry = 0;
  __syncthreads();
  // chunks of 256 bins: inclusive scan of their segment counts inside the workgroup (wave scans + carries)
  __shared__ int l_w[4];
  for (int b0 = 0; b0 < nb; b0 += kBlock) {
    const int b = b0 + (int)threadIdx.x;
    int n = 0;
    if (b < nb) {
      const int2 t0 = b_tab[b], t1 = b_tab[b + 1];
      n = (max(0, t1.x - (t0.x + t0.y)) + kSegC - 1) / kSegC;
    }
    int inc = n;
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(inc, d);
      if ((int)(threadIdx.x & 63) >= d) inc += y;
    }
    if ((threadIdx.x & 63) == 63) l_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = l_carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += l_w[w];
    const int first = base + inc - n;
    if (b < nb) {
      b_seg[b] = first;
      for (int k = 0; k < n; ++k)
        if ((long)first + k < max_segs) seg_bin[first + k] = b;
    }
    __syncthreads();
    if (threadIdx.x == kBlock - 1) l_carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) b_seg[nb] = l_carry;
}

__global__ void plan_bins_header_kernel(const int* __restrict__ plan_header, const int* __restrict__ total, long cand_cap,
                                        int V, int gx, int gy, int bw, int bh, int nbx, int nby,
                                        const unsigned long long* __restrict__ scan_state, int* __restrict__ header) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  header[1] = V; header[2] = gx; header[3] = gy; header[4] = bw; header[5] = bh; header[6] = nbx; header[7] = nby;
  header[8] = *total;
  header[9] = (int)std::min<long>(cand_cap, 0x7FFFFFFF);
  const bool ok = plan_header[0] == (int)kPlanMagic && (long)*total <= cand_cap && (scan_state[0] >> 63) == 0;
  header[0] = ok ? (int)kBinsMagic : 0;
}

constexpr int kMaxSets = 32;

// ---------------------------------------------------------------------------------------------
// The parameter-dependent half of ONE record (forward.cu:201-256): conic / tile rect from the Gaussian's parameters and
// the plan's rows of J W (raster_common.h: the per-call preprocess's arithmetic, same bits).
//   con = (-0.5 conic.x, -0.5 conic.z, conic.y, opacity); rect (0,0,0,0): not rendered this step; *rad: the reference's
//   integer radius (0: none) — computed for an invisible Gaussian (opacity < 1/255) only when the caller asks for it.
// ---------------------------------------------------------------------------------------------
struct SetParams {
  const float* opacities;      // (n_sets, P)
  const float* scales;         // (n_sets, P, 3)
  const float* rotations;      // (n_sets, P, 4)
  float scale_modifier;
};

__device__ __forceinline__ void dyn_record(const SetParams& sp, long gi, const float4 q0, const float4 q1, int gx, int gy,
                                           Rect* rect, float4* con, int* rad_out = nullptr) {
  *rect = Rect{0, 0, 0, 0};
  *con = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rad_out) *rad_out = 0;
  const float o = sp.opacities[gi];
  // alpha = min(0.99, o exp(power)) with power <= 0 (forward.cu:327-333): under 1/255 at EVERY pixel when o is (NaN
  // compares false: evaluated in full).  Such a Gaussian gets an empty rect — no tile pair ever scans into it — and, unless
  // its radius is asked for, no covariance work either.  Free space in a trained OcRF is mostly this.
  const bool unseen = o < 1.0f / 255.0f;
  if (unseen && !rad_out) return;
  const float sx = sp.scale_modifier * sp.scales[3 * gi], sy = sp.scale_modifier * sp.scales[3 * gi + 1],
              sz = sp.scale_modifier * sp.scales[3 * gi + 2];
  float c3[6];
  cov3d_from_scale_rot(sx, sy, sz, sp.rotations[4 * gi], sp.rotations[4 * gi + 1], sp.rotations[4 * gi + 2],
                       sp.rotations[4 * gi + 3], c3);
  const float A[2][3] = {{q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}};
  float cov_x, cov_y, cov_z, con_x, con_y, con_z;
  cov2d(A, c3, &cov_x, &cov_y, &cov_z);
  int rad = 0;
  Rect r;
  if (conic_radius_rect(cov_x, cov_y, cov_z, q1.z, q1.w, gx, gy, &con_x, &con_y, &con_z, &rad, &r)) {
    if (rad_out) *rad_out = rad;                     // (the reference's radius: not tightened, not opacity-dependent)
    if (unseen) return;
    tighten_rect(o, cov_x, cov_z, q1.z, q1.w, gx, gy, &r);
    *rect = r;
    *con = make_float4(-0.5f * con_x, -0.5f * con_z, con_y, o);
  }
}

// Control words of a plan's per-call scratch (ints, zero when the scratch is allocated):
//   [0] guard flag  [1] arrival counter of the armed per-call blend  [16] ticket queue of the blend  [17] its arrival
//   counter  [18] tile pairs the first pass handed to the second  [32, 64) head[v]: list entries of plan view v the head
//   kernel prepares (0: never rendered — the default)  [64, 96) reach[v]: how far into view v's list the tile pairs of the
//   running blend scanned
//   [19] some view is deep  [96, 128) deep[v]: the last render of view v walked far into its list (>= kDeepReach entries:
//   its pixels do not saturate early) — the head kernel then prepares the view's WHOLE list and the candidates of its bins
//   are compacted per call (raster_plan_compact_kernel)
//   [20] this call's compaction ran (written by raster_plan_compact_kernel)  [128, 134) its three tables' addresses (b_seg,
//   ccnt, ccand as 64-bit words): the blend reads them from here only when a tile pair is deep — as kernel arguments they
//   cost six more scalar registers everywhere, and the kernel already spills scalars (measured: +25 us on the init set)
constexpr int kCtlQueue = 16, kCtlArrive = 17, kCtlDeferred = 18, kCtlAnyDeep = 19, kCtlCompacted = 20, kCtlHead = 32,
              kCtlReach = 64, kCtlDeep = 96, kCtlTables = 128;      // (1024 bytes: dyn_layout)
constexpr int kDeepReach = 8192;
constexpr int kHeadDefault = 4096;               // list entries per view prepared before anything is known
constexpr int kHeadBlocks = 64;                  // workgroups per item of the head kernel (each strides over the head)

// what the last workgroup of a call leaves for the next one, per plan view (thread `v` of a wave): the head follows the
// reach; a view whose tile pairs walked kDeepReach entries or more is marked deep
__device__ __forceinline__ void close_view(int* ctl, int v) {
  const int reached = atomicExch(ctl + kCtlReach + v, 0);
  int deep = ctl[kCtlDeep + v];
  if (reached > 0) {
    ctl[kCtlHead + v] = (int)min(1l << 30, ((long)reached + reached / 4 + 2 * kBlock - 1) / kBlock * kBlock);
    deep = reached >= kDeepReach ? 1 : 0;
    ctl[kCtlDeep + v] = deep;
  }
  const unsigned long long any = __ballot(deep != 0);
  if (v == 0) {
    ctl[kCtlAnyDeep] = any != 0ull ? 1 : 0;
    // ... and where the host can see it without a copy or a wait (pinned host memory, address left by the head kernel)
    int* hint = reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(ctl + kCtlTables)[3]);
    if (hint) *hint = any != 0ull ? 1 : 0;
  }
}

// head of view v's list this step: `force` > 0: that many, < 0: none (diagnostic / tests), 0: what the last blend wrote.
// No upper limit but the list's length: a scene whose pixels do not saturate (an object-centric opacity field) needs
// every record of a view, and then the head kernel prepares every record ONCE.
__device__ __forceinline__ int head_of(const int* ctl, int v, int nv, int force) {
  int k = force > 0 ? force : (force < 0 ? 0 : ctl[kCtlHead + v]);
  if (force == 0 && k == 0) k = kHeadDefault;
  if (force == 0 && ctl[kCtlDeep + v] != 0) k = nv;
  return min(k, nv);
}

// ---------------------------------------------------------------------------------------------
// step 1 (no radii asked): the HEAD of every rendered view's list.
// A tile pair stops scanning its view's depth-ordered list as soon as all its pixels are saturated — at cfg2 after
// ~500 of 120 000 entries — so computing conic / rect of EVERY record in front of the blend (rounds 3-4's update kernel:
// 120 MB, 32 us, and the blend cannot start before it) prepares a hundred times what is read.  Here only the first
// head[v] entries of each rendered view are prepared, in LIST order (the blend's scan reads them coalesced, no list ->
// record indirection); beyond them the blend computes a record itself when it gets there (same inline arithmetic, same
// bits).  head[v] follows what the previous blend of the view needed (reach[v] + 25 %): no host read, self-adjusting,
// and never a correctness matter.
//   per (item z, 256 list entries): the dynamic arrays d_rect / d_con at [dyn + off + i]; block 0 also does the call's
//   bookkeeping (ticket queues, plan usable, cameras = the plan's, items name valid and distinct views).
// The extent check of ALL Gaussians (status bit 4) is not in front of the blend either: raster_plan_check_kernel runs
// behind it.
// ---------------------------------------------------------------------------------------------
struct HeadArgs {
  int P, vps, n_sets, n_items, blocks_per_item, force_head;
  long set_stride;
  const int* header;
  const int* view_sel;
  const unsigned* s_id;
  const unsigned* s_e;
  const float4* e_q0;
  const float4* e_q1;
  SetParams sp;
  Rect* d_rect;
  float4* d_con;
  int* status;
  int* ctl;
  int guard;
  const unsigned* call_cams;
  const unsigned* plan_cams;
  int* radii;                    // (n_items, P) or null: the radius of every listed (item, Gaussian) pair (zero-filled before)
  int all;                       // 1: every entry of every rendered view's list (radii asked for as an output)
  int* hint;                     // host-visible word the blend's close-out writes "some view is deep" into (or null)
};

// entries [lo, hi) of item z's list (plan view v): conic / tile rect into the dynamic arrays, in list order; this
// workgroup takes the chunks of 256 entries `first`, `first + stride`, ... of that range
__device__ __forceinline__ void prepare_entries(const HeadArgs& a, int z, int v, int lo, int hi, int first, int stride) {
  const int* header = a.header;
  const int gx = header[5], gy = header[6];
  const int off = header[kHeaderInts + v];
  const int set = z / a.vps;
  for (int i = lo + first * kBlock + (int)threadIdx.x; i < hi; i += stride * kBlock) {
    const unsigned id = a.s_id[off + i], e = a.s_e[off + i];
    Rect rect;
    float4 con;
    int rad;
    dyn_record(a.sp, (long)set * a.P + id, a.e_q0[e], a.e_q1[e], gx, gy, &rect, &con, a.radii ? &rad : nullptr);
    const long d = (long)set * a.set_stride + off + i;
    a.d_rect[d] = rect;
    a.d_con[d] = con;
    if (a.radii) a.radii[(long)z * a.P + id] = rad;
  }
}

__global__ __launch_bounds__(kBlock) void raster_plan_head_kernel(HeadArgs a) {
  const int* header = a.header;
  int* flag = a.guard ? a.ctl : nullptr;
  if (blockIdx.x == 0) {
    __shared__ int l_owner[32];
    __shared__ int l_seen[kMaxSets * 32];
    if (threadIdx.x == 0) {
      reinterpret_cast<unsigned long long*>(a.ctl + kCtlTables)[3] = reinterpret_cast<unsigned long long>(a.hint);
      a.ctl[kCtlQueue] = 0;                        // ticket counter of the blend that follows
      if (flag) flag[1] = 0;                       // arrival counter of the armed per-call blend (rasterize.hip)
    }
    if (a.call_cams) {
      bool differ = false;
      for (int i = threadIdx.x; i < header[2] * 36; i += kBlock) differ |= a.call_cams[i] != a.plan_cams[i];
      if (__ballot(differ) != 0ull && (threadIdx.x & 63) == 0) {
        atomicOr(a.status, 16);
        if (flag) atomicOr(flag, 1);
      }
    }
    if (header[0] != (int)kPlanMagic && threadIdx.x == 0) {
      atomicOr(a.status, 8);
      if (flag) atomicOr(flag, 1);
    }
    // a view named twice in one set, or by two sets that share the dynamic arrays: refused (status bit 8)
    for (int i = threadIdx.x; i < a.n_sets * 32; i += kBlock) l_seen[i] = -1;
    if (threadIdx.x < 32) l_owner[threadIdx.x] = -1;
    __syncthreads();
    const int V = header[2];
    for (int it = threadIdx.x; it < a.n_items; it += kBlock) {
      const int s = it / a.vps;
      const int v = a.view_sel ? a.view_sel[it] : it % a.vps;
      bool ok = v >= 0 && v < V;
      if (ok) ok = atomicExch(&l_seen[s * 32 + v], it) == -1;
      if (ok && a.set_stride == 0) ok = atomicExch(&l_owner[v], s) == -1;
      if (!ok) atomicOr(a.status, 8);
    }
  }
  if (header[0] != (int)kPlanMagic || a.blocks_per_item == 0) return;
  const int b = blockIdx.x;
  const int z = b / a.blocks_per_item;
  const int V = header[2];
  const int v = a.view_sel ? a.view_sel[z] : z % a.vps;
  if (v < 0 || v >= V) return;
  const int* view_off = header + kHeaderInts;
  const int nv = view_off[v + 1] - view_off[v];
  // entries [0, head) of the item's list, this workgroup every blocks_per_item-th chunk of 256
  prepare_entries(a, z, v, 0, a.all ? nv : head_of(a.ctl, v, nv, a.force_head), b % a.blocks_per_item, a.blocks_per_item);
}

// ---------------------------------------------------------------------------------------------
// Deep views — the last render walked far into the view's list: its pixels do not saturate early (an object-centric
// opacity field: most Gaussians faint or invisible), so every tile pair tests most of its bin's candidates, and most of
// those turn out not to reach it THIS call (cfg2, objects set: 21 000 candidates per tile pair, 1 200 hits).  For such a
// view the head kernel has prepared the whole list; here, per segment of kSegC candidates of a bin, the candidates whose
// opacity-tightened rect of this call reaches the bin at all are moved to the front of the segment's range — in order,
// each with the mask of the bin's tiles its rect covers — so that the blend reads (position, mask) pairs coalesced: no
// rect gather, no misses.  This is the per-call half of the reference's duplicateWithKeys (rasterizer_impl.cu:70-109),
// paid only by views that need it.  One workgroup strides over the segments; all leave at once when no view is deep.
// ---------------------------------------------------------------------------------------------
struct CompactArgs {
  const int* header;             // the plan's
  const int* bins_header;
  const int2* b_tab;
  const int* b_seg;
  const int* seg_bin;
  const unsigned* cand;
  const Rect* d_rect;            // list order (set_stride == 0: one copy)
  int* ctl;
  int* ccnt;
  uint2* ccand;
  int bw, bh, nbx, nbins;
  const int* skip_if;
};

__global__ __launch_bounds__(kBlock) void raster_plan_compact_kernel(CompactArgs a) {
  const bool on = a.ctl[kCtlAnyDeep] != 0 && !(a.skip_if && *a.skip_if != 0) && a.header[0] == (int)kPlanMagic &&
                  a.bins_header[0] == (int)kBinsMagic;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.ctl[kCtlCompacted] = on ? 1 : 0;
    auto* tab = reinterpret_cast<unsigned long long*>(a.ctl + kCtlTables);
    tab[0] = reinterpret_cast<unsigned long long>(a.b_seg);
    tab[1] = reinterpret_cast<unsigned long long>(a.ccnt);
    tab[2] = reinterpret_cast<unsigned long long>(a.ccand);
  }
  if (!on) return;
  const int V = a.header[2];
  const int n_seg_total = a.b_seg[V * a.nbins];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int l_w[4];
  for (int g = blockIdx.x; g < n_seg_total; g += gridDim.x) {
    const int bin = a.seg_bin[g];
    const int v = bin / a.nbins;
    if (a.ctl[kCtlDeep + v] == 0) continue;
    const int lb = bin - v * a.nbins;
    const int tx0 = (lb % a.nbx) * a.bw, ty0 = (lb / a.nbx) * 2 * a.bh;      // the bin's first tile column / row
    const int2 t0 = a.b_tab[bin], t1 = a.b_tab[bin + 1];
    const int first = t0.x + t0.y + (g - a.b_seg[bin]) * kSegC;
    const int end = min(first + kSegC, t1.x);
    const int off = a.header[kHeaderInts + v];
    int base = first;
    for (int c0 = first; c0 < end; c0 += kBlock) {
      const int c = c0 + (int)threadIdx.x;
      unsigned pos = 0, mask = 0;
      if (c < end) {
        pos = a.cand[c];
        const Rect rc = a.d_rect[off + (int)pos];
        const int lx0 = max((int)rc.x0 - tx0, 0), lx1 = min((int)rc.x1 - tx0, a.bw);
        const int ly0 = max((int)rc.y0 - ty0, 0), ly1 = min((int)rc.y1 - ty0, 2 * a.bh);
        if (lx1 > lx0 && ly1 > ly0) {
          const unsigned row = ((1u << (lx1 - lx0)) - 1u) << lx0;
          for (int y = ly0; y < ly1; ++y) mask |= row << (y * a.bw);
        }
      }
      const unsigned long long m = __ballot(mask != 0u);
      if (lane == 0) l_w[wave] = __popcll(m);
      __syncthreads();
      int mine = base, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < wave) mine += l_w[w];
        tot += l_w[w];
      }
      if (mask != 0u) a.ccand[mine + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(pos, mask);
      base += tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) a.ccnt[g] = base - first;
  }
}

// ---------------------------------------------------------------------------------------------
// The extent check of ALL Gaussians (status bit 4: the plan's static cull holds for extents <= its bound; 29 MB of
// parameters at cfg2), one thread per (set, Gaussian) pair.  Its result is a status bit, so it runs BEHIND the blend on
// the render's stream, not in front of it.  NaN / Inf anywhere counts as a violation (fmaxf drops NaNs).
// ---------------------------------------------------------------------------------------------
// `rest` (between the two passes of the blend): when tile pairs of the first pass ran out of prepared records
// (ctl[kCtlDeferred] > 0), the records BEHIND every rendered view's head are prepared here — once, by this launch, never by
// a tile pair — and the second pass renders those tile pairs with the whole list at hand.
__global__ __launch_bounds__(kBlock) void raster_plan_check_kernel(long n_pairs, const int* __restrict__ header,
                                                                   const float* __restrict__ scales, float scale_modifier,
                                                                   const float* __restrict__ rotations,
                                                                   int* __restrict__ status, int* __restrict__ flag,
                                                                   HeadArgs rest, int do_rest) {
  if (header[0] != (int)kPlanMagic) return;
  if (do_rest && rest.ctl[kCtlDeferred] > 0 && !(rest.guard && rest.ctl[0] != 0)) {
    const int V = header[2];
    const int* view_off = header + kHeaderInts;
    for (int z = 0; z < rest.n_items; ++z) {
      const int v = rest.view_sel ? rest.view_sel[z] : z % rest.vps;
      if (v < 0 || v >= V) continue;
      const int nv = view_off[v + 1] - view_off[v];
      const int head = head_of(rest.ctl, v, nv, rest.force_head);
      if (head < nv) prepare_entries(rest, z, v, head, nv, blockIdx.x, gridDim.x);
    }
  }
  const long gi = (long)blockIdx.x * kBlock + threadIdx.x;
  const float bound = __int_as_float(header[7]);
  bool bad = false;
  if (gi < n_pairs) {
    const float sx = scale_modifier * scales[3 * gi], sy = scale_modifier * scales[3 * gi + 1],
                sz = scale_modifier * scales[3 * gi + 2];
    const float rn = extent_bound(sx, sy, sz, rotations[4 * gi], rotations[4 * gi + 1], rotations[4 * gi + 2],
                                  rotations[4 * gi + 3]);
    bad = !(rn <= bound) || !(((sx + sy) + sz) * 0.f == 0.f);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) {
    atomicOr(status, 4);
    if (flag) atomicOr(flag, 1);                 // device guard: the armed per-call chain renders this call
  }
}

// ---------------------------------------------------------------------------------------------
// step 2: blend of a sorted list (forward.cu:261-374 + w-depth README:5-11).
// One workgroup = one vertical pair of 16x16 tiles of one rendered item; wave w owns the 16x8 pixel block of rows
// [8w, 8w+8) of the pair (waves 0-1: upper tile, 2-3: lower tile), thread (lx, r) the pixels (lx, 8w + r) and
// (lx, 8w + r + 4): same x, so the dx-only part of the exponent is shared and the two pixels run as the halves of
// packed fp32 ops.  Per-record arithmetic: raster_blend_math.h (shared with the per-call kernel).
// ---------------------------------------------------------------------------------------------
// five waves per SIMD (<= 96 VGPRs): left alone hipcc takes 121 for the per-tile-pair prologue (staging, list extension);
// with the bound the record loops still hold everything in registers (no scratch)
#ifndef OCRF_PLAN_WAVES
#define OCRF_PLAN_WAVES 5
#endif
#define OCRF_BLEND_BOUNDS __launch_bounds__(kBlock, OCRF_PLAN_WAVES)

constexpr int kItemTable = 64;                   // items whose view the blend keeps in LDS (more: read per tile pair)
constexpr int kMaxSegTab = 128;                  // segments of a bin whose compacted counts a tile pair keeps in LDS
// workgroup barrier that orders LDS traffic only (outstanding global stores are not waited for)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BlendArgs {
  unsigned long long* stats;     // STATS build only
  int P, W, H, gx, gy, n_items, vps;
  long set_stride;
  const int* header;
  const int* view_sel;
  const unsigned* s_id;
  const unsigned* s_key;
  const float2* s_pix;
  const unsigned* s_e;
  const float4* e_q0;
  const float4* e_q1;
  Rect* d_rect;                  // dynamic arrays: read; extended beyond the head by the tile pairs that get there
  float4* d_con;
  const float* colors;
  SetParams sp;
  const float* bg;
  float* out_color;
  float* out_depth;
  float* out_final_T;
  const int* skip_if;
  int* ctl;
  int* chain_hist;
  int chain_hist_words;
  const int* yield_if;
  int base_grid;
  int full;                      // 1: the head kernel prepared EVERY record of every rendered view (radii as an output)
  int force_head;
  int variant;                   // diagnostic (ocrf_tune_set 14): bit 0 = no no-stop loops, bit 1 = the GENERIC loop only
  int n_sets;
  int* status;
  // candidate lists (plan-time, ocrf_raster_plan_bins_build): per (plan view, bin of bw x bh tile PAIRS) the positions, in
  // list (= blend) order, of the view's records whose BOUND-inflated rect reaches the bin; null: a tile pair walks the
  // view's whole list
  const int* bins_header;
  const int2* b_tab;
  const unsigned* cand;
  int bw, bh, nbx, nbins;
  // two passes: the first renders a tile pair as far as the prepared head of its view's list reaches and hands it to the
  // second (deferred[], ctl[kCtlDeferred]) if it needs more; the second finds every record prepared (the launch in between)
  int* deferred;
  int pass;                      // 1, 2
  int last_pass;                 // this launch closes the call (heads from reach, counters back to zero)
};

// DEEP: the instantiation that carries the code for deep views (compacted candidates).  In the plain instantiation that
// code — never executed on a scene that saturates early — cost every tile pair of every scene, wherever in the kernel it
// stood and however little of it the plain path touched (cfg2, init set: step 0.208 -> 0.222-0.230 ms; a `continue` for
// deep views, a start value of `blocked`, a zero head: all the same, tools/ab_libs.sh).  So WHICH instantiation a call
// launches is decided on the host, from a word the last call's close-out wrote into host-visible memory (`hint`: stale by a
// call or two at worst — either instantiation renders any view correctly, only faster or slower).
// (DEEP at four waves per SIMD — what its grid asks for anyway: at five the extra state spills 12 VGPRs to scratch.)
// SECOND: nothing but the kernel's NAME — the second pass (usually an empty launch) as a symbol of its own, so that a
// profile's per-kernel means (rocprofv3 --stats, --pmc) are those of the first pass and not halved by the empty launches.
template <bool MEDIAN, bool STATS = false, bool DEEP = false, bool SECOND = false>
__global__ __launch_bounds__(kBlock, DEEP ? 4 : OCRF_PLAN_WAVES) void raster_blend_sorted_kernel(BlendArgs g) {
  __shared__ unsigned l_pos[kCapPos];
  __shared__ float4 l_a[kStageP + 1], l_b[kStageP + 1], l_c[kStageP + 1];
  // (aligned: a trip reads an entry PAIR as one 32-bit word; rows are 268 bytes)
  __shared__ __attribute__((aligned(16))) unsigned short l_list[4][rbody::kListLen];
  __shared__ int l_wtot[kScanUnrollP * 4];
  __shared__ int l_lcnt[kSrcWaves][4];        // [source wave of the batch copy][destination wave]
  __shared__ int l_generic[4];                // [wave]: its list of this batch holds a GENERIC record
  __shared__ int l_fmax[4];                   // [wave]: largest need factor (float bits) of its list of this batch
  __shared__ int l_work;
  __shared__ int l_reach;                     // list entries this tile pair has looked at (position of its last candidate + 1)
  // what a tile pair needs to know of its item, read from global memory ONCE per workgroup (a persistent workgroup
  // renders ~ 6 tile pairs; item -> view -> list offsets -> head was three dependent round trips at the start of each)
  __shared__ int l_voff[33], l_head[32], l_vsel[kItemTable];
  __shared__ int l_deep[32];                  // [view]: its bins' candidates were compacted for this call
  // deep tile pair, per WAVE (each wave builds its own copy: no workgroup barrier in the set-up): compacted candidates in
  // front of each segment of its bin | its segments, its bin's first candidate behind the direct region, bit of tile A
  __shared__ int l_segpre[DEEP ? 4 : 1][DEEP ? kMaxSegTab + 2 : 2];
  __shared__ int l_tp[4][4];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int* const ctl = g.ctl;
  if (g.pass == 2 && ctl[kCtlDeferred] == 0) {
    // the usual case: no tile pair of the first pass ran out of prepared records — workgroup 0 closes the call (next
    // call's heads from this call's reach; the first pass left the ticket queue and the arrival counter at zero)
    if (blockIdx.x == 0 && tid < 32) close_view(ctl, tid);
    return;
  }
  bool run = true;
  if (g.skip_if && *g.skip_if != 0) {
    // the plan's bound does not hold this step: the armed per-call chain renders.  Its bucket histograms are cleared
    // here — by the kernel that sits in front of it anyway — instead of by a launch of their own in every call.
    for (int i = blockIdx.x * kBlock + tid; i < g.chain_hist_words; i += gridDim.x * kBlock) g.chain_hist[i] = 0;
    run = false;
  }
  // A scheduling hint, not a dependency: launched on every slot the device has, the workgroups beyond `base_grid`
  // leave at once while *yield_if says another stream's chain is still running (it wants those wave slots); the
  // ticket queue makes any number of participants render the same image.
  if (g.yield_if && (int)blockIdx.x >= g.base_grid && *g.yield_if != 0) run = false;
  const int* header = g.header;
  if (header[0] != (int)kPlanMagic) {                                 // reported by step 1 (status bit 8)
    // an unusable plan (a rebuild for a pose that keeps more records than the capacity): nothing is rendered — the
    // images are zero-filled rather than left as they were allocated (a host-guarded caller may
    // consume them before it reads the status word)
    if (run && !(g.skip_if && *g.skip_if != 0)) {
      const long npix = (long)g.W * g.H * g.n_items;
      for (long i = (long)blockIdx.x * kBlock + tid; i < npix; i += (long)gridDim.x * kBlock) {
        g.out_final_T[i] = 0.f;
        g.out_depth[i] = 0.f;
        g.out_color[3 * i] = 0.f; g.out_color[3 * i + 1] = 0.f; g.out_color[3 * i + 2] = 0.f;
      }
    }
    run = false;
  }
  const int V = header[2];
  const bool use_bins = g.cand != nullptr && g.bins_header[0] == (int)kBinsMagic;      // (built for this plan, and it fitted)
  const int* view_off = header + kHeaderInts;
  const int lx = lane & 15, r = lane >> 4;
  const int gx = g.gx, gy = g.gy, W = g.W, H = g.H;
  const int gyp = (gy + 1) / 2;
  // second pass: the tile pairs the first pass could not finish (their number is final: that launch is over)
  const int n_work = g.pass == 2 ? min(ctl[kCtlDeferred], kMaxDeferred) : gx * gyp * g.n_items;
  if (tid == 0) {      // slot kStageP: a record that changes nothing, pads odd list lengths
    const rb::Staged st = rb::stage_noop();
    l_a[kStageP] = st.a;
    l_b[kStageP] = st.b;
    l_c[kStageP] = st.c;
  }
  if (run) {
    // (the heads are what the last blend's close-out left: nothing writes them while tile pairs are being rendered)
    if (tid <= min(V, 32)) l_voff[tid] = view_off[tid];
    if (tid >= 64 && tid < 96) {
      const int v = tid - 64;
      l_head[v] = (v < V && !g.full) ? head_of(ctl, v, view_off[v + 1] - view_off[v], g.force_head) : 0;
      if constexpr (DEEP)
        l_deep[v] = (v < V && use_bins && g.force_head == 0 && ctl[kCtlCompacted] != 0) ? ctl[kCtlDeep + v] : 0;
    }
    if (tid >= 128 && tid - 128 < min(g.n_items, kItemTable)) {
      const int z = tid - 128;
      l_vsel[z] = g.view_sel ? g.view_sel[z] : z % g.vps;
    }
  }
  rb::Consts kc = rb::consts();
  asm volatile("" : "+v"(kc.neg_k255), "+v"(kc.neg_half));      // kept in VGPR pairs (else re-materialised per trip)
  // Persistent workgroups over a ticket queue: the grid is what the chip holds at once, every workgroup takes the next
  // tile pair until none is left.  Every wave reaches the exit: the ticket counter only grows.
  // The NEXT ticket is drawn when a tile pair starts and read when it ends: the atomic's round trip (several
  // microseconds on a word that 900 workgroups share) used to sit between two tile pairs of a slot, 16 % of its time.
  int next_ticket = 0;
  if (run && tid == 0) next_ticket = atomicAdd(ctl + kCtlQueue, 1);
  while (run) {
  // (barriers that order LDS only: a __syncthreads() here would also wait for the previous pair's image stores)
  lds_barrier();                                                      // the previous pair's LDS traffic is over
  if (tid == 0) l_work = next_ticket;
  lds_barrier();
  const int ticket = __builtin_amdgcn_readfirstlane(l_work);      // (wave-uniform values in scalar registers from here on)
  if (ticket >= n_work) break;
  if (tid == 0) next_ticket = atomicAdd(ctl + kCtlQueue, 1);
  const int work = __builtin_amdgcn_readfirstlane(g.pass == 2 ? g.deferred[ticket] : ticket);
  if (tid == 0) l_reach = 0;                   // (ordered before its first use by the scan's barriers)
  const int z = work / (gx * gyp);
  const int tx = (work - z * gx * gyp) % gx, ty2 = (work - z * gx * gyp) / gx;
  const int v = __builtin_amdgcn_readfirstlane(z < kItemTable ? l_vsel[z] : (g.view_sel ? g.view_sel[z] : z % g.vps));
  if (v < 0 || v >= V) continue;                                      // reported by step 1 (status bit 8)
  const int set = z / g.vps;
  const int tyA = 2 * ty2, tyB = tyA + 1;
  const int off = __builtin_amdgcn_readfirstlane(l_voff[v]), nv = __builtin_amdgcn_readfirstlane(l_voff[v + 1]) - off;
  // list entries [0, head) have their conic / rect in the dynamic arrays, in list order.  First pass: a tile pair never
  // prepares records itself — it is handed to the second pass when it needs one behind the head
  const int head = (g.full || g.pass == 2) ? nv : __builtin_amdgcn_readfirstlane(l_head[v]);
  const long dyn = (long)set * g.set_stride;
  const float* set_colors = g.colors + 3 * (long)set * g.P;
  auto dyn_index = [&](int i) { return dyn + (long)(off + i); };
  // what this tile pair scans: the first n_direct entries of the view's list as they come, then its bin's candidates
  // (positions into the view's list, ascending) from the first one behind them; without lists the whole list
  // (read here, used behind the direct region.  Read only by the tile pairs that get there — most of a saturating scene's
  // do not — the scan loop came out slower: 15.5 vs 14.5 us per tile pair at cfg2, tools/ab_gauss.sh)
  int n_direct = nv, nc = nv;
  bool deep_view = false, deep_pair = false;      // compacted candidates behind the direct region (a deep view)
  const unsigned* cand = nullptr;
  if (use_bins) {
    const int bin = v * g.nbins + (ty2 / g.bh) * g.nbx + tx / g.bw;
    const int2 t0 = g.b_tab[bin], t1 = g.b_tab[bin + 1];
    n_direct = min(nv, kBinDirect);
    const int first = __builtin_amdgcn_readfirstlane(t0.x) + __builtin_amdgcn_readfirstlane(t0.y);
    nc = n_direct + (__builtin_amdgcn_readfirstlane(t1.x) - first);
    cand = g.cand + first;
    if constexpr (DEEP) deep_view = __builtin_amdgcn_readfirstlane(l_deep[v]) != 0;
  }
  const int pxi = tx * kTileX + lx;
  const int py0 = tyA * kTileY + 8 * wave + r, py1 = py0 + 4;
  const bool tile_ok = (tyA + (wave >> 1)) < gy;
  const bool inside0 = tile_ok && pxi < W && py0 < H, inside1 = tile_ok && pxi < W && py1 < H;
  const float pixf_x = (float)pxi;
  const f2 pixf_y = f2{(float)py0, (float)py1};
  // the wave's pixel block as float bounds (pixel centres are the integer coordinates, forward.cu:283)
  const float bx0 = (float)(tx * kTileX), bx1 = bx0 + 15.f;

  rb::Px px;
  px.T = f2{inside0 ? 1.0f : -1.0f, inside1 ? 1.0f : -1.0f};
  px.C0 = splat(0.f); px.C1 = splat(0.f); px.C2 = splat(0.f);
  px.D = splat(MEDIAN ? 15.0f : 0.0f);
  px.cnt = splat(0.f);

  int scan = 0, npos = 0;
  bool all_done = false;
  bool blocked = false;                        // the next candidate is not prepared (first pass only)
  // diagnostic build only: phase cycles and record counts of this workgroup
  struct Diag {
    unsigned long long t_prev = 0, t_acc[3] = {0, 0, 0};
    unsigned n_staged = 0, n_listed = 0, n_eval = 0;
    __device__ __forceinline__ void stamp(int slot) {
      if constexpr (STATS) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime()      /* 100 MHz */;
        if (slot >= 0) t_acc[slot] += t - t_prev;
        t_prev = t;
      }
    }
  } diag;
  auto stamp = [&](int slot) { diag.stamp(slot); };
  stamp(-1);
  rbody::Tile tile;
  tile.tyA = tyA; tile.wave = wave; tile.lane = lane; tile.tid = tid;
  tile.pixf_x = pixf_x; tile.pixf_y = pixf_y; tile.inside0 = inside0; tile.inside1 = inside1;
  tile.bx0 = bx0; tile.bx1 = bx1;
  const rbody::Lds lds{l_a, l_b, l_c, l_list, l_lcnt, l_generic, l_fmax};
  while (!all_done) {
    if constexpr (DEEP) {
    if (deep_view && !deep_pair && scan >= n_direct) {
      // a deep view, and this tile pair has got past the direct region: this call's compacted candidates of its bin
      // (position, tile mask), segment by segment.  (Set up HERE — between two scan phases, the first of which stops at
      // the end of the direct region — and not inside the scan loop or in front of it: there the mere presence of this
      // block cost every tile pair of every scene, 132 -> 160 us for cfg2's twelve views on the init set, tools/ab_libs.sh.)
      deep_view = false;
      const int bin = v * g.nbins + (ty2 / g.bh) * g.nbx + tx / g.bw;
      const auto* tab = reinterpret_cast<const unsigned long long*>(ctl + kCtlTables);
      const int* b_seg = reinterpret_cast<const int*>(tab[0]);
      const int* ccnt = reinterpret_cast<const int*>(tab[1]);
      const int s0 = __builtin_amdgcn_readfirstlane(b_seg[bin]);
      const int ns = __builtin_amdgcn_readfirstlane(b_seg[bin + 1]) - s0;
      if (ns <= kMaxSegTab) {
        static_assert(kMaxSegTab <= 128, "two segments per lane");
        const int l2 = 2 * lane;
        const int ca = l2 < ns ? ccnt[s0 + l2] : 0, cb = l2 + 1 < ns ? ccnt[s0 + l2 + 1] : 0;
        int inc = ca + cb;
        for (int d = 1; d < 64; d <<= 1) {
          const int y = __shfl_up(inc, d);
          if (lane >= d) inc += y;
        }
        const int before = inc - (ca + cb);
        l_segpre[wave][l2 + 1] = before + ca;            // (entries beyond ns: never read)
        l_segpre[wave][l2 + 2] = before + ca + cb;
        if (lane == 0) {
          l_segpre[wave][0] = 0;
          // (read back in the scan rounds: kept out of the scalar registers the record loops are short of)
          l_tp[wave][0] = ns;
          l_tp[wave][1] = (int)(cand - g.cand);
          l_tp[wave][2] = ((tyA - (ty2 / g.bh) * 2 * g.bh) * g.bw) + (tx - (tx / g.bw) * g.bw);
        }
        deep_pair = true;
        nc = n_direct + __builtin_amdgcn_readlane(inc, 63);
      }
    }
    }
    // (a deep view's first scan phase ends with the direct region: the compacted lists are set up above, next time round)
    const int lim = deep_view ? n_direct : nc;
    // ---- scan: positions of the candidates whose rect covers this tile pair, in list (= blend) order ----
    while (scan < lim && npos < kStageP && !blocked) {
      // the direct region 256 rects at a time (a dense tile pair fills its first batch from them); behind it the
      // candidates kScanUnrollP x 256 at a time: each round is two dependent memory round trips (candidate -> rect)
      const int n_u = (scan < n_direct) ? 1 : kScanUnrollP;
      unsigned code[kScanUnrollP];
      bool hit[kScanUnrollP], valid[kScanUnrollP];
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        const int ic = scan + u * kBlock + tid;
        hit[u] = false;
        valid[u] = false;
        code[u] = 0u;
        if (DEEP && u < n_u && ic < nc && deep_pair && ic >= n_direct) {
          // deep: the j-th compacted candidate of the bin = entry j - pre[s] of segment s, pre[s] <= j < pre[s + 1]
          const int j = ic - n_direct;
          const int nseg = l_tp[wave][0], tail_first = l_tp[wave][1], bit_a = l_tp[wave][2];
          int lo = 0, hi = nseg;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (l_segpre[wave][mid] <= j) lo = mid; else hi = mid;
          }
          const uint2* ccand = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned long long*>(ctl + kCtlTables)[2]);
          const uint2 e = ccand[tail_first + lo * kSegC + (j - l_segpre[wave][lo])];
          const bool cA = ((e.y >> bit_a) & 1u) != 0u, cB = ((e.y >> (bit_a + g.bw)) & 1u) != 0u;
          valid[u] = true;                      // (a deep view's whole list is prepared)
          hit[u] = cA || cB;
          code[u] = e.x | (cA ? 0x40000000u : 0u) | (cB ? 0x80000000u : 0u);
        } else if (u < n_u && ic < nc) {
          const int i = ic < n_direct ? ic : (int)cand[ic - n_direct];
          // candidates are ascending: the unprepared ones (i >= head) are a suffix of the round
          valid[u] = i < head;
          if (valid[u]) {
            const Rect rc = g.d_rect[dyn_index(i)];
            const bool cA = (tyA >= rc.y0) && (tyA < rc.y1), cB = (tyB >= rc.y0) && (tyB < rc.y1);
            hit[u] = (tx >= rc.x0) && (tx < rc.x1) && (cA || cB);
            code[u] = (unsigned)i | (cA ? 0x40000000u : 0u) | (cB ? 0x80000000u : 0u);
          }
        }
      }
      int rank[kScanUnrollP];
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        const unsigned long long m = __ballot(hit[u]), mv = __ballot(valid[u]);
        rank[u] = __popcll(m & ((1ull << lane) - 1ull));
        // (low half: hits of the wave; high half: its prepared candidates)
        if (lane == 0) l_wtot[u * 4 + wave] = __popcll(m) | (__popcll(mv) << 16);
      }
      __syncthreads();
      int o = npos, n_valid = 0;
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        int mine = o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int c = l_wtot[u * 4 + w] & 0xFFFF;
          n_valid += l_wtot[u * 4 + w] >> 16;
          if (w < wave) mine += c;
          o += c;
        }
        if (hit[u]) l_pos[mine + rank[u]] = code[u];
      }
      npos = o;
      // the round ends at its first unprepared candidate: what lies behind it waits for the second pass
      const int n_round = min(nc - scan, n_u * kBlock);
      blocked = n_valid < n_round;
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u)
        if (valid[u] && u * kBlock + tid == n_valid - 1) l_reach = (int)(code[u] & kPosMask) + 1;
      scan += n_valid;
      __syncthreads();
    }
    stamp(0);
    if (npos == 0) {
      if (deep_view && !blocked && scan < nc) continue;      // nothing in the direct region: on to the compacted lists
      break;                                  // list exhausted
    }
    for (int s0 = 0; s0 < npos && !all_done; s0 += kStageP) {
      const int ns = min(kStageP, npos - s0);
      // ---- stage ns records, build the four waves' lists, blend them (raster_blend_body.h) ----
      auto fetch = [&](int ri) {
        const unsigned code = l_pos[s0 + ri];
        const int li = (int)(code & kPosMask);
        rbody::Fetched fr;
        fr.con = g.d_con[dyn_index(li)];
        fr.pix = g.s_pix[off + li];
        fr.col = set_colors + 3 * (long)g.s_id[off + li];
        fr.depth = __uint_as_float(g.s_key[off + li]);
        fr.covA = (code & 0x40000000u) != 0u;
        fr.covB = (code & 0x80000000u) != 0u;
        return fr;
      };
      all_done = rbody::blend_batch<MEDIAN, STATS>(px, kc, tile, lds, ns, fetch, g.variant, diag);
      stamp(2);
    }
    npos = 0;
    if (scan >= nc || blocked) break;
  }
  // pixels still open and records still to come that nobody has prepared: the tile pair is handed to the second pass —
  // which renders it from its first record, with the view's whole list prepared ONCE by the launch in between — and the
  // view's head covers the whole list from the next call on
  const bool defer = blocked && !all_done;
  if (defer) {
    if (tid == 0) {
      const int k = atomicAdd(ctl + kCtlDeferred, 1);
      if (k < kMaxDeferred) g.deferred[k] = work;
      else atomicOr(g.status, 8);              // (more tile pairs than the list holds: 2^20 — reported, not rendered)
      atomicMax(ctl + kCtlReach + v, nv);
    }
    continue;
  }
  if constexpr (STATS) {
    // per wave: slot = workgroup * 4 + wave: scan, stage, blend cycles | scanned, staged, listed, evaluated records
    if (lane == 0) {
      const long w = (long)work * 4 + wave;
      g.stats[w * 8 + 0] = diag.t_acc[0]; g.stats[w * 8 + 1] = diag.t_acc[1]; g.stats[w * 8 + 2] = diag.t_acc[2];
      g.stats[w * 8 + 3] = (unsigned long long)scan; g.stats[w * 8 + 4] = diag.n_staged;
      g.stats[w * 8 + 5] = diag.n_listed; g.stats[w * 8 + 6] = diag.n_eval;
      g.stats[w * 8 + 7] = diag.t_prev;                  // when this tile pair's last batch ended (100 MHz wall clock)
    }
  }
  // how far this tile pair read into the view's list: the next call's head (the read is a hint: a stale value only
  // costs an atomic)
  // Reported by the tile pairs that came within a chunk of the end of the prepared head (or beyond it), and by every
  // 16th tile pair whatever it read — the sample lets the head shrink again.  (Every tile pair reporting is 4 224 atomics
  // on 12 words per launch: they serialise at the memory side and cost the launch 60 us.)
  // (no look at the word first: the read's round trip would sit in front of the next tile pair's first barrier)
  if (tid == 0 && !g.full) {
    // (in list entries: the position of the last candidate this tile pair looked at)
    const int reached = l_reach;
    if (reached > head - kBlock || (work & 15) == 0) atomicMax(ctl + kCtlReach + v, reached);
  }

  const long npix = (long)W * H;
  auto store = [&](bool inside, int py, float t, float c0, float c1, float c2, float d) {
    if (!inside) return;
    const long pix = (long)py * W + pxi;
    g.out_final_T[z * npix + pix] = t;
    g.out_color[((long)z * 3 + 0) * npix + pix] = c0 + t * g.bg[0];
    g.out_color[((long)z * 3 + 1) * npix + pix] = c1 + t * g.bg[1];
    g.out_color[((long)z * 3 + 2) * npix + pix] = c2 + t * g.bg[2];
    g.out_depth[z * npix + pix] = d;
  };
  store(inside0, py0, fabsf(px.T.x), px.C0.x, px.C1.x, px.C2.x, px.D.x);
  store(inside1, py1, fabsf(px.T.y), px.C0.y, px.C1.y, px.C2.y, px.D.y);
  }   // ticket loop
  // The last workgroup to leave closes the call: next call's heads from this call's reach, counters back to zero (the
  // ticket queues too: a call needs no reset in front of it).  Every workgroup arrives, whichever way it left the loop.
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    l_work = atomicAdd(ctl + kCtlArrive, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (l_work) {
    __threadfence();
    if (tid < 32 && g.last_pass) close_view(ctl, tid);      // one memory round trip for all views, not 32 in a row
    if (tid == 32) atomicExch(ctl + kCtlArrive, 0);
    if (tid == 33) atomicExch(ctl + kCtlQueue, 0);
    if (tid == 34 && g.last_pass) atomicExch(ctl + kCtlDeferred, 0);
  }
}

// workgroups of `kernel` (256 threads, static LDS only) the device holds at once: the size of a persistent grid
template <typename K>
int resident_blocks(K kernel) {
  static int cached[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 8) dev = 0;
  if (cached[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // four per CU even where five fit (96 VGPRs): measured — cfg2's twelve views alone 129 us on 1 024 workgroups, 137 on
    // 1 152, 142-149 on 1 280 (tools/time_blend_r5.py): a fifth wave per SIMD adds more LDS / issue contention to the
    // record loops than latency hiding
    cached[dev] = std::min(per_cu, 4) * cus;
  }
  return cached[dev];
}

int g_plan_grid = 0;        // ocrf_tune_set(11, n): workgroups of the persistent blend (0 = what the device holds at once)
int g_blend_variant = 0;    // ocrf_tune_set(14, bits): diagnostic loop selection of the planned blend
int g_single_pass = 0;      // ocrf_tune_set(15, 1): no second pass (diagnostic: what the two launches behind the first pass cost)
int g_head_force = 0;       // ocrf_tune_set(13, n): list entries per view the head kernel prepares (n > 0), none (n < 0), adaptive (0)
unsigned long long* g_plan_stats = nullptr;      // ocrf_diag_plan_stats: the next planned blends run the STATS build

}  // namespace

namespace ocrf {
void raster_plan_tune(int key, int value) {
  if (key == 11) g_plan_grid = value > 0 ? value : 0;
  if (key == 13) g_head_force = value;
  if (key == 14) g_blend_variant = value;
  if (key == 15) g_single_pass = value;
}
}

extern "C" {

// Diagnostic: the size of the persistent blend's grid (median depth, no wave skip) as the occupancy API reports it
int ocrf_diag_plan_resident(void) { return resident_blocks(raster_blend_sorted_kernel<true>); }

// Diagnostic: when set (device buffer of tile pairs * items * 4 waves * 8 u64), the next planned renders run the
// instrumented build of the sorted blend.  Never used by the product path.
int ocrf_diag_plan_stats(unsigned long long* buf) { g_plan_stats = buf; return 0; }

size_t ocrf_raster_plan_build_workspace_bytes(int P, int n_views, long capacity) {
  if (P <= 0 || n_views <= 0 || n_views > 32 || capacity <= 0 || capacity >= (1l << 30)) return 0;
  BuildLayout L;
  build_layout(P, capacity, &L);
  return L.bytes;
}

size_t ocrf_raster_plan_bytes(int P, int n_views, long capacity) {
  if (P <= 0 || n_views <= 0 || n_views > 32 || capacity < 0) return 0;
  PlanLayout L;
  plan_layout(P, n_views, capacity, &L);
  return L.bytes;
}

namespace {
// steps 1 + 2 of the build; counts (device): [0, 32) per view, [32] total, [33] depth-range flag
hipError_t plan_count(int P, int V, int H, int W, const float* means3D, const float* cameras, float bound,
                      unsigned* g_mask, int* g_off, int* counts, void* scan_ws, size_t scan_bytes, hipStream_t stream) {
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  hipError_t e = ocrf::zero_async(counts, (size_t)kCountInts * 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(plan_classify_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, P, V, gx, gy, W, H,
                     means3D, reinterpret_cast<const Camera*>(cameras), bound, g_mask, g_off);
  return ocrf::exclusive_scan_ints(g_off, P, counts + 32, scan_ws, scan_bytes, stream);
}
}  // namespace

// Sizing pass: the records a plan for (means3D, cameras, extent_bound) holds.  g_mask (device, P words): bit v = view v
// keeps the Gaussian; total (device int): their number.  workspace >= ocrf_raster_plan_count_workspace_bytes(P).
int ocrf_raster_plan_count(int P, int n_views, int H, int W, const float* means3D, const float* cameras,
                           float extent_bound_, unsigned* g_mask, int* total, void* workspace, size_t workspace_bytes,
                           ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (P <= 0 || n_views <= 0 || n_views > 32 || H <= 0 || W <= 0 || !means3D || !cameras || !g_mask || !total ||
      !workspace || !(extent_bound_ >= 0.f) || (long)P * n_views >= (1l << 30) ||
      workspace_bytes < ocrf_raster_plan_count_workspace_bytes(P))
    return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(workspace);
  int* g_off = reinterpret_cast<int*>(base);
  char* scan_ws = base + align_up((size_t)P * 4, 256);
  int* cnt = reinterpret_cast<int*>(scan_ws + ocrf::exclusive_scan_bytes(P));
  hipError_t e = plan_count(P, n_views, H, W, means3D, cameras, extent_bound_, g_mask, g_off, cnt, scan_ws,
                            ocrf::exclusive_scan_bytes(P), stream);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyAsync(total, cnt + 32, sizeof(int), hipMemcpyDeviceToDevice, stream);
}

size_t ocrf_raster_plan_count_workspace_bytes(int P) {
  if (P <= 0) return 0;
  return align_up((size_t)P * 4, 256) + ocrf::exclusive_scan_bytes(P) + align_up(kCountInts * 4, 256);
}

// The build proper: no host read, kernels only (hipGraph-capturable), into a plan of `capacity` records
// (>= ocrf_raster_plan_bytes(P, n_views, capacity) bytes).  A plan whose records do not fit is marked unusable on the
// device (status bit 8 at render time).  `cameras` may be a different calibration on every call: a rebuild per sample.
int ocrf_raster_plan_build(int P, int n_views, int H, int W, const float* means3D, const float* cameras,
                           float extent_bound_, long capacity, void* workspace, size_t workspace_bytes, void* plan,
                           size_t plan_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (P <= 0 || n_views <= 0 || n_views > 32 || H <= 0 || W <= 0 || !means3D || !cameras || !workspace || !plan ||
      capacity <= 0 || capacity >= (1l << 30) || !(extent_bound_ >= 0.f) || (long)P * n_views >= (1l << 30))
    return (int)hipErrorInvalidValue;
  PlanLayout L;
  plan_layout(P, n_views, capacity, &L);
  BuildLayout B;
  build_layout(P, capacity, &B);
  if (plan_bytes < L.bytes || workspace_bytes < B.bytes) return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  if (gx > 65535 || gy > 65535) return (int)hipErrorInvalidValue;
  char* pb = static_cast<char*>(plan);
  char* wb = static_cast<char*>(workspace);
  int* header = reinterpret_cast<int*>(pb + L.header);
  auto* g_mask = reinterpret_cast<unsigned*>(pb + L.g_mask);
  int* g_off = reinterpret_cast<int*>(pb + L.g_off);
  auto* e_q0 = reinterpret_cast<float4*>(pb + L.e_q0);
  auto* e_q1 = reinterpret_cast<float4*>(pb + L.e_q1);
  auto* keys = reinterpret_cast<unsigned*>(wb + B.keys);
  int* rec_id = reinterpret_cast<int*>(wb + B.rec_id);
  int* counts = reinterpret_cast<int*>(wb + B.counts);
  const Camera* cams = reinterpret_cast<const Camera*>(cameras);
  // the magic word falls first: a render that races a failed or interrupted rebuild sees an unusable plan
  hipError_t e = ocrf::zero_async(header, 4, stream);
  if (e != hipSuccess) return (int)e;
  e = plan_count(P, n_views, H, W, means3D, cameras, extent_bound_, g_mask, g_off, counts, wb + B.scan, B.bytes - B.scan,
                 stream);
  if (e != hipSuccess) return (int)e;
  const dim3 pgrid((P + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(plan_fill_records_kernel, pgrid, dim3(kBlock), 0, stream, P, W, H, capacity, means3D, cams,
                     static_cast<const unsigned*>(g_mask), static_cast<const int*>(g_off),
                     static_cast<const int*>(counts), e_q0, e_q1, keys, rec_id, counts + 33);
  const unsigned* sk = nullptr;
  const int* se = nullptr;
  // the sort's last pass writes the per-view lists itself (one launch over the capacity and 46 us less than a gather
  // that read the sorted pairs back)
  ocrf::RadixPlanEmit emit;
  emit.rec_id = rec_id; emit.e_q1 = e_q1;
  emit.s_e = reinterpret_cast<unsigned*>(pb + L.s_e); emit.s_id = reinterpret_cast<unsigned*>(pb + L.s_id);
  emit.s_key = reinterpret_cast<unsigned*>(pb + L.s_key); emit.s_pix = reinterpret_cast<float2*>(pb + L.s_pix);
  emit.depth_mask = kKeyDepthMask; emit.key_base = kKeyBase;
  e = ocrf::radix_sort_ids(keys, (int)capacity, 32, wb + B.sort, B.scan - B.sort, &sk, &se, stream, &emit);
  if (e != hipSuccess) return (int)e;
  const unsigned long long* sort_states = nullptr;
  int n_states = 0;
  long stride = 0;
  ocrf::radix_sort_states(wb + B.sort, (int)capacity, 32, &sort_states, &n_states, &stride);
  // header, view offsets, cameras (one workgroup); with se == nullptr the lists are already in place
  const unsigned ggrid = se ? (unsigned)std::min<long>((capacity + kBlock - 1) / kBlock, 4096) : 1u;
  hipLaunchKernelGGL(plan_gather_kernel, dim3(ggrid), dim3(kBlock), 0, stream, P, n_views, H, W, gx, gy, extent_bound_,
                     capacity, static_cast<const int*>(counts), cameras, sk, se, static_cast<const int*>(rec_id),
                     static_cast<const float4*>(e_q1), sort_states, n_states, stride,
                     reinterpret_cast<const unsigned long long*>(wb + B.scan), header,
                     reinterpret_cast<float*>(pb + L.cams), reinterpret_cast<unsigned*>(pb + L.s_id),
                     reinterpret_cast<unsigned*>(pb + L.s_key), reinterpret_cast<float2*>(pb + L.s_pix),
                     reinterpret_cast<unsigned*>(pb + L.s_e));
  return (int)hipGetLastError();
}

// ---- candidate lists of a built plan (see plan_bin_range_kernel) ---------------------------------------------------
namespace {
bool bins_shape(int H, int W, int bw, int bh, int* gx, int* gy, int* nbx, int* nby) {
  if (H <= 0 || W <= 0 || bw <= 0 || bh <= 0) return false;
  *gx = (W + kTileX - 1) / kTileX;
  *gy = (H + kTileY - 1) / kTileY;
  *nbx = (*gx + bw - 1) / bw;
  *nby = ((*gy + 1) / 2 + bh - 1) / bh;
  return *nbx <= 255 && *nby <= 255 && (long)*nbx * *nby <= 65535;
}
}  // namespace

size_t ocrf_raster_plan_bins_bytes(int n_views, int H, int W, int bin_w, int bin_h, long cand_capacity) {
  int gx, gy, nbx, nby;
  if (n_views <= 0 || n_views > 32 || cand_capacity < 0 || cand_capacity >= (1l << 31) ||
      !bins_shape(H, W, bin_w, bin_h, &gx, &gy, &nbx, &nby))
    return 0;
  BinsLayout L;
  bins_layout(n_views, nbx, nby, cand_capacity, &L);
  return L.bytes;
}

size_t ocrf_raster_plan_bins_workspace_bytes(int P, int n_views, int H, int W, int bin_w, int bin_h, long capacity) {
  int gx, gy, nbx, nby;
  if (P <= 0 || n_views <= 0 || n_views > 32 || capacity <= 0 || !bins_shape(H, W, bin_w, bin_h, &gx, &gy, &nbx, &nby))
    return 0;
  BinsBuildLayout B;
  bins_build_layout(P, n_views, capacity, nbx * nby, &B);
  return B.bytes;
}

// Builds the candidate lists of `plan` (already built on this stream: ocrf_raster_plan_build) into `bins`.  No host read;
// hipGraph-capturable.  cand_capacity = 0: a sizing pass — only *total_out (device int, may be null otherwise) is written:
// the candidates these cameras give.  Lists that do not fit leave the buffer without its magic word (ignored by renders).
int ocrf_raster_plan_bins_build(const void* plan, size_t plan_bytes, int P, int n_views, long capacity, int H, int W,
                                float extent_bound_, int bin_w, int bin_h, long cand_capacity, void* bins,
                                size_t bins_bytes, int* total_out, void* workspace, size_t workspace_bytes,
                                ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int gx, gy, nbx, nby;
  if (!plan || P <= 0 || n_views <= 0 || n_views > 32 || capacity <= 0 || capacity >= (1l << 30) || !workspace ||
      !(extent_bound_ >= 0.f) || cand_capacity < 0 || cand_capacity >= (1l << 31) ||
      !bins_shape(H, W, bin_w, bin_h, &gx, &gy, &nbx, &nby) || (cand_capacity > 0 && !bins) ||
      (cand_capacity == 0 && !total_out))
    return (int)hipErrorInvalidValue;
  PlanLayout L;
  plan_layout(P, n_views, capacity, &L);
  const int nbins = nbx * nby;
  BinsBuildLayout B;
  bins_build_layout(P, n_views, capacity, nbins, &B);
  BinsLayout Q;
  bins_layout(n_views, nbx, nby, cand_capacity, &Q);
  if (plan_bytes < L.bytes || workspace_bytes < B.bytes || (cand_capacity > 0 && bins_bytes < Q.bytes))
    return (int)hipErrorInvalidValue;
  const char* pb = static_cast<const char*>(plan);
  char* wb = static_cast<char*>(workspace);
  const int* header = reinterpret_cast<const int*>(pb + L.header);
  auto* range = reinterpret_cast<unsigned*>(wb + B.range);
  int* counts = reinterpret_cast<int*>(wb + B.counts);
  int* total = reinterpret_cast<int*>(wb + B.total);
  const int n_seg = bins_segments(P, capacity);
  const long n_counts = (long)n_views * nbins * n_seg;
  if (n_counts >= (1l << 31)) return (int)hipErrorInvalidValue;
  hipError_t e = hipSuccess;
  if (cand_capacity > 0) {
    e = ocrf::zero_async(static_cast<char*>(bins) + Q.header, 4, stream);      // the magic word falls first
    if (e != hipSuccess) return (int)e;
  }
  const long per_view = std::min<long>(P, capacity);
  hipLaunchKernelGGL(plan_bin_range_kernel, dim3((unsigned)((per_view + kBlock - 1) / kBlock), n_views), dim3(kBlock), 0,
                     stream, header, extent_bound_, bin_w, bin_h, reinterpret_cast<const unsigned*>(pb + L.s_e),
                     reinterpret_cast<const float4*>(pb + L.e_q0), reinterpret_cast<const float4*>(pb + L.e_q1), range);
  hipLaunchKernelGGL(plan_bin_count_kernel, dim3(n_seg, nbins, n_views), dim3(kBlock), 0, stream, header, nbx, n_seg,
                     static_cast<const unsigned*>(range), counts);
  e = ocrf::exclusive_scan_ints(counts, n_counts, total, wb + B.scan, B.total - B.scan, stream);
  if (e != hipSuccess) return (int)e;
  if (total_out) {
    e = hipMemcpyAsync(total_out, total, sizeof(int), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return (int)e;
  }
  if (cand_capacity == 0) return (int)hipGetLastError();
  char* qb = static_cast<char*>(bins);
  hipLaunchKernelGGL(plan_bin_fill_kernel, dim3(n_seg, nbins, n_views), dim3(kBlock), 0, stream, header, nbx, n_seg,
                     cand_capacity, static_cast<const unsigned*>(range), static_cast<const int*>(counts),
                     static_cast<const int*>(total), reinterpret_cast<int2*>(qb + Q.b_off),
                     reinterpret_cast<unsigned*>(qb + Q.cand));
  hipLaunchKernelGGL(plan_bins_segments_kernel, dim3(1), dim3(kBlock), 0, stream, n_views * nbins,
                     max_segments(cand_capacity, (long)n_views * nbins), reinterpret_cast<const int2*>(qb + Q.b_off),
                     static_cast<const int*>(total), cand_capacity, reinterpret_cast<int*>(qb + Q.b_seg),
                     reinterpret_cast<int*>(qb + Q.seg_bin));
  hipLaunchKernelGGL(plan_bins_header_kernel, dim3(1), dim3(64), 0, stream, header, static_cast<const int*>(total),
                     cand_capacity, n_views, gx, gy, bin_w, bin_h, nbx, nby,
                     reinterpret_cast<const unsigned long long*>(wb + B.scan), reinterpret_cast<int*>(qb + Q.header));
  return (int)hipGetLastError();
}

size_t ocrf_rasterize_planned_workspace_bytes(long total_kept, int n_sets) {
  if (total_kept < 0 || n_sets <= 0) return 0;
  DynLayout L;
  dyn_layout(total_kept, n_sets, &L);
  return L.bytes;
}

// ... with candidate lists (ocrf_raster_plan_bins_build): room for the per-call compaction of deep views too
size_t ocrf_rasterize_planned_bins_workspace_bytes(long total_kept, int n_sets, int n_views, int H, int W, int bin_w, int bin_h,
                                                   long cand_capacity) {
  int gx, gy, nbx, nby;
  if (total_kept < 0 || n_sets <= 0 || n_views <= 0 || cand_capacity <= 0 || !bins_shape(H, W, bin_w, bin_h, &gx, &gy, &nbx, &nby))
    return 0;
  DynLayout L;
  dyn_layout(total_kept, n_sets, &L, cand_capacity, (long)n_views * nbx * nby);
  return L.bytes;
}

int ocrf_rasterize_planned(const void* plan, size_t plan_bytes, int P, int n_plan_views, long total_kept, int H, int W,
                           int n_sets, int n_items, const int* item_view, const float* colors,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* bg, int depth_mode, float* out_color, float* out_depth, float* out_final_T,
                           int* radii, int* status, void* workspace, size_t workspace_bytes, int guard,
                           const float* means3D, void* chain_workspace, size_t chain_workspace_bytes,
                           int blend_workgroups, const int* yield_if, int phase, const float* call_cameras,
                           int views_disjoint, const void* bins, size_t bins_bytes, int bin_w, int bin_h,
                           long cand_capacity, int* hint, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!plan || P <= 0 || n_plan_views <= 0 || n_plan_views > 32 || total_kept < 0 || total_kept >= (1l << 30) ||
      H <= 0 || W <= 0 || n_sets <= 0 || n_sets > kMaxSets || n_items <= 0 || n_items % n_sets || blend_workgroups < 0 ||
      phase < 0 ||
      phase > 2 || (phase != 0 && guard) || !colors ||
      !opacities || !scales ||
      !rotations || !bg || (depth_mode != 0 && depth_mode != 1) || !out_color || !out_depth || !out_final_T ||
      !status || !workspace)
    return (int)hipErrorInvalidValue;
  const int vps = n_items / n_sets;
  if (vps > n_plan_views || (!item_view && vps != n_plan_views)) return (int)hipErrorInvalidValue;
  PlanLayout L;
  plan_layout(P, n_plan_views, total_kept, &L);
  // views_disjoint: every plan view is rendered by at most one set of this call (checked on the device: status bit 8) —
  // all sets share ONE copy of the dynamic arrays
  const long set_stride = views_disjoint ? 0 : total_kept;
  int q_gx, q_gy, q_nbx = 1, q_nby = 1;
  const bool have_bins = bins != nullptr && cand_capacity > 0 && bins_shape(H, W, bin_w, bin_h, &q_gx, &q_gy, &q_nbx, &q_nby);
  DynLayout D;
  dyn_layout(total_kept, views_disjoint ? 1 : n_sets, &D, have_bins ? cand_capacity : 0,
             have_bins ? (long)n_plan_views * q_nbx * q_nby : 0);
  if (plan_bytes < L.bytes || workspace_bytes < D.bytes) return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  const char* pb = static_cast<const char*>(plan);
  char* wb = static_cast<char*>(workspace);
  const int* header = reinterpret_cast<const int*>(pb + L.header);
  const Camera* cams = reinterpret_cast<const Camera*>(pb + L.cams);
  auto* d_rect = reinterpret_cast<Rect*>(wb + D.rect);
  auto* d_con = reinterpret_cast<float4*>(wb + D.con);
  int* ctl = reinterpret_cast<int*>(wb + D.flag);
  int* deferred = reinterpret_cast<int*>(wb + D.deferred);
  if ((long)gx * ((gy + 1) / 2) * n_items > kMaxDeferred) return (int)hipErrorInvalidValue;
  // candidate lists (optional; bin_w / bin_h / cand_capacity as given to ocrf_raster_plan_bins_build).  Whether the
  // buffer holds usable lists is read from ITS header on the device (the blend walks whole lists otherwise)
  const int* bins_header = nullptr;
  const int2* b_tab = nullptr;
  const unsigned* cand = nullptr;
  int nbx = 1, nbins = 1;
  CompactArgs comp{};
  bool compact = false;
  if (bins) {
    int bgx, bgy, nby;
    if (!bins_shape(H, W, bin_w, bin_h, &bgx, &bgy, &nbx, &nby) || cand_capacity <= 0) return (int)hipErrorInvalidValue;
    BinsLayout Q;
    bins_layout(n_plan_views, nbx, nby, cand_capacity, &Q);
    if (bins_bytes < Q.bytes) return (int)hipErrorInvalidValue;
    const char* qb = static_cast<const char*>(bins);
    bins_header = reinterpret_cast<const int*>(qb + Q.header);
    b_tab = reinterpret_cast<const int2*>(qb + Q.b_off);
    cand = reinterpret_cast<const unsigned*>(qb + Q.cand);
    nbins = nbx * nby;
    // per-call compaction of deep views: one copy of the dynamic arrays (set_stride == 0) and a bin of <= 32 tiles
    if (set_stride == 0 && bin_w * 2 * bin_h <= 32) {
      comp.header = header; comp.bins_header = bins_header; comp.b_tab = b_tab;
      comp.b_seg = reinterpret_cast<const int*>(qb + Q.b_seg);
      comp.seg_bin = reinterpret_cast<const int*>(qb + Q.seg_bin);
      comp.cand = cand; comp.d_rect = d_rect; comp.ctl = ctl;
      comp.ccnt = reinterpret_cast<int*>(wb + D.ccnt);
      comp.ccand = reinterpret_cast<uint2*>(wb + D.ccand);
      comp.bw = bin_w; comp.bh = bin_h; comp.nbx = nbx; comp.nbins = nbins;
      comp.skip_if = nullptr;
      compact = true;
    }
  }
  int* flag = nullptr;
  int* chain_hist = nullptr;
  size_t chain_hist_words = 0;
  if (guard) {
    if (!means3D || !radii || !chain_workspace ||
        chain_workspace_bytes < ocrf_rasterize_workspace_bytes(P, n_items))
      return (int)hipErrorInvalidValue;
    // flag[0]: "the extent check fired".  No memset per call: step 1 only RAISES it, the armed blend lowers
    // it after a call that fired, so it is zero on entry unless a fired call was cut short — then this call takes the
    // exact per-call path once more and lowers it.  (The scratch is zero-filled when it is allocated; any other first
    // value only costs one slow call.)  flag[1]: arrival counter of that blend, zeroed by step 1.
    flag = ctl;
    chain_hist = ocrf::raster_chain_hist(chain_workspace, P, n_items, &chain_hist_words);
  }
  // radii asked for as an OUTPUT (guard bit 1, or radii without a guard): EVERY record of every rendered view is prepared
  // by the head kernel, which also writes the radius of every listed (item, Gaussian) pair into the zero-filled output
  // (round 6: the one-thread-per-Gaussian update kernel of rounds 3-5 — 163 VGPRs, Gaussian-major arrays the blend read
  // through an indirection — is gone); else only the head of each view's list.  With the device guard (its radii buffer
  // is the armed chain's own) the extent check runs FIRST: the blend has to know whether to leave the call to the chain
  const bool full = radii != nullptr && (guard == 0 || (guard & 2) != 0);
  const SetParams sp{opacities, scales, rotations, scale_modifier};
  HeadArgs h;
  h.P = P; h.vps = vps; h.n_sets = n_sets; h.n_items = n_items;
  h.blocks_per_item = 0;
  h.force_head = g_head_force;
  h.set_stride = set_stride;
  h.header = header; h.view_sel = item_view;
  h.s_id = reinterpret_cast<const unsigned*>(pb + L.s_id);
  h.s_e = reinterpret_cast<const unsigned*>(pb + L.s_e);
  h.e_q0 = reinterpret_cast<const float4*>(pb + L.e_q0);
  h.e_q1 = reinterpret_cast<const float4*>(pb + L.e_q1);
  h.sp = sp; h.d_rect = d_rect; h.d_con = d_con; h.status = status; h.ctl = ctl; h.guard = guard ? 1 : 0;
  h.call_cams = reinterpret_cast<const unsigned*>(call_cameras);
  h.plan_cams = reinterpret_cast<const unsigned*>(cams);
  h.radii = full ? radii : nullptr;
  h.all = full ? 1 : 0;
  // deep mode: the last call (or the one before) left "some view is deep" in the caller's host-visible word
  compact = compact && !full && g_head_force == 0 && hint != nullptr && *reinterpret_cast<volatile int*>(hint) != 0;
  h.hint = hint;
  if (phase != 2) {
    if (guard) {      // (behind the blend otherwise: there it is a status bit only)
      const long n_pairs = (long)n_sets * P;
      hipLaunchKernelGGL(raster_plan_check_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                         stream, n_pairs, header, scales, scale_modifier, rotations, status, flag, h, 0);
    }
    if (full) {       // (item, Gaussian) pairs outside the lists: radius 0
      const hipError_t ez = ocrf::zero_async(radii, (size_t)n_items * P * 4, stream);
      if (ez != hipSuccess) return (int)ez;
    }
    h.blocks_per_item = (g_head_force < 0 && !full) ? 0 : kHeadBlocks;
    ocrf::launch(OCRF_K_RASTER_PLAN_UPDATE, raster_plan_head_kernel,
                 dim3((unsigned)std::max(1, n_items * h.blocks_per_item)), dim3(kBlock), 0, stream, h);
    if (compact) {      // (behind the head kernel: it reads this call's rects; leaves at once unless a view is deep)
      comp.skip_if = flag;
      hipLaunchKernelGGL(raster_plan_compact_kernel, dim3(1024), dim3(kBlock), 0, stream, comp);
    }
  }   // phase != 2
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (phase == 1) return 0;
  const int n_work = gx * ((gy + 1) / 2) * n_items;
  // yield_if: the grid is everything the device holds, `blend_workgroups` of it stay whatever the hint says
  const int base_grid = (yield_if && blend_workgroups > 0) ? blend_workgroups : (1 << 30);
  const int want_grid = g_plan_grid ? g_plan_grid : ((yield_if && blend_workgroups > 0) ? 0 : blend_workgroups);      // the diagnostic knob wins
  BlendArgs g;
  g.stats = g_plan_stats;
  g.P = P; g.W = W; g.H = H; g.gx = gx; g.gy = gy; g.n_items = n_items; g.vps = vps;
  g.set_stride = set_stride; g.header = header; g.view_sel = item_view;
  g.s_id = reinterpret_cast<const unsigned*>(pb + L.s_id);
  g.s_key = reinterpret_cast<const unsigned*>(pb + L.s_key);
  g.s_pix = reinterpret_cast<const float2*>(pb + L.s_pix);
  g.s_e = reinterpret_cast<const unsigned*>(pb + L.s_e);
  g.e_q0 = reinterpret_cast<const float4*>(pb + L.e_q0);
  g.e_q1 = reinterpret_cast<const float4*>(pb + L.e_q1);
  g.d_rect = d_rect; g.d_con = d_con; g.colors = colors; g.sp = sp; g.bg = bg;
  g.out_color = out_color; g.out_depth = out_depth; g.out_final_T = out_final_T;
  g.skip_if = flag; g.ctl = ctl; g.chain_hist = chain_hist; g.chain_hist_words = (int)chain_hist_words;
  g.yield_if = yield_if; g.base_grid = base_grid; g.full = full ? 1 : 0; g.force_head = g_head_force; g.variant = g_blend_variant;
  g.n_sets = n_sets; g.status = status;
  g.bins_header = bins_header; g.b_tab = b_tab; g.cand = cand; g.bw = bin_w > 0 ? bin_w : 1; g.bh = bin_h > 0 ? bin_h : 1;
  g.nbx = nbx; g.nbins = nbins;
  g.deferred = deferred;
  // `full`: every record was prepared in front of the blend — one pass.  Else two: the first as far as the prepared heads
  // reach; then the extent check, which also prepares the REST of the lists if a tile pair asked for it; then the second
  // pass over those tile pairs (it retires at once when there are none)
  const bool two_pass = !full && !g_single_pass;
  g.pass = 1;
  g.last_pass = two_pass ? 0 : 1;
  auto blend = [&](const BlendArgs& ga, bool first) -> hipError_t {
    if (g_plan_stats) {      // diagnostic build (median depth), never used by the product path
      const dim3 sgrid((unsigned)std::min(n_work, want_grid ? want_grid : resident_blocks(raster_blend_sorted_kernel<true, true>)));
      if (!compact && first) hipLaunchKernelGGL((raster_blend_sorted_kernel<true, true, false, false>), sgrid, dim3(kBlock), 0, stream, ga);
      else if (!compact) hipLaunchKernelGGL((raster_blend_sorted_kernel<true, true, false, true>), sgrid, dim3(kBlock), 0, stream, ga);
      else if (first) hipLaunchKernelGGL((raster_blend_sorted_kernel<true, true, true, false>), sgrid, dim3(kBlock), 0, stream, ga);
      else hipLaunchKernelGGL((raster_blend_sorted_kernel<true, true, true, true>), sgrid, dim3(kBlock), 0, stream, ga);
      return hipGetLastError();
    }
    // (the kernel timer of bench.py's roofline leg sees the FIRST pass; the second has an id of its own)
    const int kid = first ? OCRF_K_RASTER_BLEND_SORTED : OCRF_K_RASTER_BLEND_SECOND;
#define OCRF_BLEND_SORTED(MED, DEEPK, SEC)                                                                             \
  ocrf::launch(kid, raster_blend_sorted_kernel<MED, false, DEEPK, SEC>,                                                \
               dim3((unsigned)std::min(n_work, want_grid ? want_grid : resident_blocks(raster_blend_sorted_kernel<MED, false, DEEPK, SEC>))), \
               dim3(kBlock), 0, stream, ga)
#define OCRF_BLEND_PASS(MED, DEEPK) do { if (first) OCRF_BLEND_SORTED(MED, DEEPK, false); else OCRF_BLEND_SORTED(MED, DEEPK, true); } while (0)
    if (depth_mode == 0 && !compact) OCRF_BLEND_PASS(true, false);
    else if (depth_mode == 0) OCRF_BLEND_PASS(true, true);
    else if (!compact) OCRF_BLEND_PASS(false, false);
    else OCRF_BLEND_PASS(false, true);
#undef OCRF_BLEND_PASS
#undef OCRF_BLEND_SORTED
    return hipGetLastError();
  };
  e = blend(g, true);
  if (e != hipSuccess) return (int)e;
  if (!two_pass && !guard) {
    const long n_pairs = (long)n_sets * P;
    hipLaunchKernelGGL(raster_plan_check_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       n_pairs, header, scales, scale_modifier, rotations, status, static_cast<int*>(nullptr), h, 0);
  }
  if (two_pass) {
    const long n_pairs = (long)n_sets * P;
    // with the device guard the extent check ran first; this launch is then only the preparation of the rest
    hipLaunchKernelGGL(raster_plan_check_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       guard ? 0l : n_pairs, header, scales, scale_modifier, rotations, status, static_cast<int*>(nullptr), h,
                       1);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    g.pass = 2;
    g.last_pass = 1;
    e = blend(g, false);
    if (e != hipSuccess) return (int)e;
  }
  if (guard) {
    // the per-call pipeline, armed: every kernel of it retires at once unless the extent check fired (measured: on a
    // side stream beside the blend the four launches do not get cheaper — their workgroups wait for the persistent
    // blend's slots — so they stay in the caller's stream)
    return ocrf::raster_forward_chain(P, n_sets, vps, H, W, means3D, colors, opacities, scales, scale_modifier,
                                      rotations, nullptr, call_cameras ? call_cameras : reinterpret_cast<const float*>(cams),
                                      item_view, bg,
                                      depth_mode, out_color, out_depth, out_final_T, nullptr, radii, nullptr, nullptr,
                                      chain_workspace, chain_workspace_bytes, flag, true, true, stream);
  }
  return 0;
}

}  // extern "C"

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
// (depth bits, id), with their static per-record data.  A step then is two launches:
//   raster_plan_update_kernel   one thread per Gaussian: covariance -> conic / radius / tile rect of its kept
//                               records, written in Gaussian-major record order (coalesced; the blend reaches a
//                               list entry's record through the plan's static list -> record map), + the check
//                               that no Gaussian exceeds the plan's extent bound (status bit 4);
//   raster_blend_sorted_kernel  a tile pair filters the list by tile rect — the survivors arrive in the
//                               reference's per-tile order, so there is no sort, no depth bucket, no carry —
//                               and each of its four waves blends only the records whose alpha >= 1/255
//                               ellipse reaches the wave's own 16x8 pixel block (exact conservative test at
//                               staging, so a wave skips records by construction).
// against zero-fill -> preprocess (all P x V pairs) -> bucket scan -> scatter -> blend (with an in-LDS bitonic
// sort) of rasterize.hip.  Per-pixel arithmetic and its order are those of raster_blend_kernel: colour, depth
// and final_T are bit-identical to the per-call pipeline (tests/test_raster_plan_gpu.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "launch.h"
#include "ocrf_hip.h"
#include "raster_common.h"

namespace {

using namespace rc;

constexpr unsigned kPlanMagic = 0x4F435250u;      // "OCRP"
constexpr int kHeaderInts = 16;                   // magic, P, V, H, W, gx, gy, bound bits, total lo, total hi, ...
#ifndef OCRF_PLAN_STAGE
#define OCRF_PLAN_STAGE 128
#endif
#ifndef OCRF_PLAN_SCAN
#define OCRF_PLAN_SCAN 1      // 256 rects per scan round (2: 256, then 512 per round — 126.8 vs 122.8 us for cfg2's 12 views; 3: 133.0)
#endif
constexpr int kStageP = OCRF_PLAN_STAGE;          // records staged per batch (a tile pair saturates after ~110 at cfg2)
constexpr int kScanUnrollP = OCRF_PLAN_SCAN;      // rect batches in flight in the scan
#ifndef OCRF_PLAN_TRIP
#define OCRF_PLAN_TRIP 2
#endif
constexpr int kTrip = OCRF_PLAN_TRIP;             // records per trip of the blend loop (2 or 4)
static_assert(kTrip == 2 || kTrip == 4, "records per trip");
constexpr int kStageParts = kBlock / kStageP;     // threads per staged record: each tests 4 / kStageParts waves
constexpr int kReachPerThread = 4 / kStageParts;
constexpr int kSrcWaves = kStageP / 64;           // waves that hold one copy of the staged batch
static_assert(kStageP == 128 || kStageP == 256, "one or two threads per staged record");
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
  size_t rect, con, flag, bytes;
};

inline void dyn_layout(long T, int n_sets, DynLayout* L) {
  size_t off = 0;
  auto take = [&](size_t bts) { size_t o = off; off += align_up(bts, 256); return o; };
  const size_t n = (size_t)T * n_sets;
  L->rect = take(n * sizeof(Rect));
  L->con = take(n * 16);
  L->flag = take(256);
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
// step 1: the parameter-dependent half of preprocessCUDA (forward.cu:201-256), ONE THREAD PER GAUSSIAN for EVERY
// parameter set of the call: per set the parameters are read once (coalesced), the extent is checked against the
// plan's bound, the 3D covariance is built once, then every view the set renders that keeps the Gaussian gets its
// conic / radius / tile rect.  Everything it writes is in GAUSSIAN-MAJOR record order e (coalesced); the blend reaches
// a record of its depth-ordered list through the plan's static list position -> e map (s_e):
//   d_rect[set * set_stride + e]   tile rect ((0,0,0,0): not rendered this step), gathered by the blend's scan.  Written
//                     at the record's place in the sorted list instead, the scan read it coalesced but the writes were
//                     random 8-byte stores — a memory-side read-modify-write each: 21.5 -> 15.2 us alone in round 3;
//   d_con[set * set_stride + e]    (-0.5 conic.x, -0.5 conic.z, conic.y, opacity), read when a record is staged.
// set_stride = 0 when every plan view is rendered by at most ONE set of the call (the hot path: frames of a sample
// share a plan, frame f renders its own six views with its own parameters) — one dynamic array for all sets, every
// line of it written once, by one workgroup; else the plan's record capacity (a set per copy).
// Round 3 ran one workgroup per (256 Gaussians, set): a Gaussian's records of the two frames interleave in memory, so
// the two workgroups (on different XCDs: separate L2s) each fetched every record line and each wrote half of every
// output line — 216 MB of HBM traffic for 120 MB algorithmic (PMC), 42.8 us.  The items of one set name distinct views.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSets = 32;

__global__ __launch_bounds__(kBlock) void raster_plan_update_kernel(
    int P, int vps, int n_sets, long set_stride, long n_cap, const int* __restrict__ header,
    const unsigned* __restrict__ g_mask,
    const int* __restrict__ g_off, const float4* __restrict__ e_q0,
    const float4* __restrict__ e_q1, const int* __restrict__ view_sel, const float* __restrict__ opacities,
    const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations,
    Rect* __restrict__ d_rect, float4* __restrict__ d_con, int* __restrict__ radii, int* __restrict__ status,
    int* __restrict__ flag, int* __restrict__ queue, const unsigned* __restrict__ call_cams,
    const unsigned* __restrict__ plan_cams) {
  __shared__ int l_v2i[kMaxSets * 32];             // [set][plan view] -> item of the set that renders it, or -1
  __shared__ unsigned l_setmask[kMaxSets];         // [set] -> plan views the set renders
  __shared__ int l_owner[32];                      // disjoint mode: the set that renders a plan view
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *queue = 0;                  // ticket counter of the blend that follows
    if (flag) flag[1] = 0;       // arrival counter of the armed per-call blend (rasterize.hip)
  }
  if (call_cams && blockIdx.x == 0) {
    // the plan is keyed by its cameras ON THE DEVICE: a call that means other cameras (another pose) is not rendered
    // from this plan — status bit 4 (value 16), and the armed per-call chain (guard) takes the call over
    bool differ = false;
    for (int i = threadIdx.x; i < header[2] * 36; i += kBlock) differ |= call_cams[i] != plan_cams[i];
    if (__ballot(differ) != 0ull && (threadIdx.x & 63) == 0) {
      atomicOr(status, 16);
      if (flag) atomicOr(flag, 1);
    }
  }
  if (header[0] != (int)kPlanMagic) {      // unusable plan (capacity, key range, failed scan): the armed chain renders
    if (threadIdx.x == 0 && blockIdx.x == 0) {
      atomicOr(status, 8);
      if (flag) atomicOr(flag, 1);
    }
    return;
  }
  const int V = header[2], gx = header[5], gy = header[6];
  for (int i = threadIdx.x; i < n_sets * 32; i += kBlock) l_v2i[i] = -1;
  if (threadIdx.x < 32) l_owner[threadIdx.x] = -1;
  if ((int)threadIdx.x < n_sets) l_setmask[threadIdx.x] = 0u;
  __syncthreads();
  for (int it = threadIdx.x; it < n_sets * vps; it += kBlock) {
    const int s = it / vps, zi = it % vps;
    const int v = view_sel ? view_sel[it] : zi;
    bool ok = v >= 0 && v < V;
    if (ok) ok = atomicExch(&l_v2i[s * 32 + v], zi) == -1;            // a view named twice in one set: refused
    if (ok && set_stride == 0) ok = atomicExch(&l_owner[v], s) == -1;  // ... or by two sets that share the dynamic arrays
    if (ok) atomicOr(&l_setmask[s], 1u << v);
    if (!ok && blockIdx.x == 0) atomicOr(status, 8);
  }
  __syncthreads();
  const int id = blockIdx.x * kBlock + threadIdx.x;
  if (id >= P) return;
  const unsigned m = g_mask[id];
  const int e0 = g_off[id];
  const float bound = __int_as_float(header[7]);
  // the parameters of the next set are requested before this set's arithmetic
  float sx, sy, sz, qr, qx, qy, qz, o;
  auto load = [&](int s) {
    const long gi = (long)s * P + id;
    sx = scales[3 * gi]; sy = scales[3 * gi + 1]; sz = scales[3 * gi + 2];
    qr = rotations[4 * gi]; qx = rotations[4 * gi + 1]; qy = rotations[4 * gi + 2]; qz = rotations[4 * gi + 3];
    o = opacities[gi];
  };
  load(0);
  // The Gaussian's first kPre records (it has 2.7 on average over the twelve views of cfg2) are requested NOW, whatever
  // set they belong to, at clamped indices: one memory round trip for all of them instead of one per record inside the
  // loops below (a thread's records used to be a chain of load -> arithmetic -> store -> load ...).
  constexpr int kPre = 4;
  float4 pq0[kPre], pq1[kPre];
  int pv[kPre];
  {
    unsigned mm = m;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      pv[j] = mm ? __ffs(mm) - 1 : -1;
      mm &= mm - 1;                                            // (0 stays 0)
      const long idx = min((long)e0 + j, n_cap - 1);
      pq0[j] = e_q0[idx];
      pq1[j] = e_q1[idx];
    }
  }
  bool any_bad = false;
  for (int s = 0; s < n_sets; ++s) {
    const float csx = scale_modifier * sx, csy = scale_modifier * sy, csz = scale_modifier * sz;
    const float cqr = qr, cqx = qx, cqy = qy, cqz = qz, co = o;
    if (s + 1 < n_sets) load(s + 1);
    const float rn = extent_bound(csx, csy, csz, cqr, cqx, cqy, cqz);
    // the plan's static cull holds for extents <= bound; NaN / Inf anywhere counts as a violation (fmaxf drops NaNs)
    any_bad |= !(rn <= bound) || !(((csx + csy) + csz) * 0.f == 0.f);
    const unsigned mset = m & l_setmask[s];
    if (radii) {      // (item, Gaussian) pairs outside the lists are not visited below: their radii are 0 (no memset launch)
      for (int zi = 0; zi < vps; ++zi) {
        const int v = view_sel ? view_sel[s * vps + zi] : zi;
        const bool listed = v >= 0 && v < V && ((mset >> v) & 1u) != 0u && l_v2i[s * 32 + v] == zi;
        if (!listed) radii[((long)s * vps + zi) * P + id] = 0;
      }
    }
    if (mset == 0u) continue;
    float c3[6];
    cov3d_from_scale_rot(csx, csy, csz, cqr, cqx, cqy, cqz, c3);
    // alpha = min(0.99, o exp(power)) with power <= 0 (forward.cu:327-333): under 1/255 at EVERY pixel when o is
    // (NaN compares false: evaluated in full).  Such a Gaussian gets an empty rect — no tile pair ever scans into it —
    // and, unless its radius is asked for, no covariance work either.  Free space in a trained OcRF is mostly this.
    const bool unseen = co < 1.0f / 255.0f;
    const long dyn = (long)s * set_stride;
    auto record = [&](int v, int cur, const float4 q0, const float4 q1) {
      const int zi = l_v2i[s * 32 + v];
      const float A[2][3] = {{q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}};
      Rect rect = Rect{0, 0, 0, 0};
      int rad = 0;
      // (No conservative early-out by the Gaussian's own radius bound here, unlike the per-call preprocess: the plan's
      // static cull already removed what can never be seen, so the test rarely fired and its square root cost every
      // record; conic_radius_rect finds an empty rect itself, with the same outputs: 16.5 -> 15.6 us per six views.)
#ifdef OCRF_UPDATE_EARLY_OUT
      if ((!unseen || radii) && !surely_outside(q1.z, q1.w, radius_bound(A, rn), gx, gy)) {
#else
      if (!unseen || radii) {
#endif
        float cov_x, cov_y, cov_z, con_x, con_y, con_z;
        cov2d(A, c3, &cov_x, &cov_y, &cov_z);
        if (conic_radius_rect(cov_x, cov_y, cov_z, q1.z, q1.w, gx, gy, &con_x, &con_y, &con_z, &rad, &rect)) {
          d_con[dyn + cur] = make_float4(-0.5f * con_x, -0.5f * con_z, con_y, co);      // Gaussian-major: coalesced
        } else {
          rect = Rect{0, 0, 0, 0};
          rad = 0;
        }
      }
      d_rect[dyn + cur] = unseen ? Rect{0, 0, 0, 0} : rect;
      if (radii) radii[((long)s * vps + zi) * P + id] = rad;
    };
#pragma unroll
    for (int j = 0; j < kPre; ++j)                              // record j of the Gaussian = its j-th kept view (ascending)
      if (pv[j] >= 0 && ((mset >> pv[j]) & 1u) != 0u) record(pv[j], e0 + j, pq0[j], pq1[j]);
    if (__popc(m) > kPre) {                                     // rare: a Gaussian kept by more than kPre views
      unsigned mm = m;
#pragma unroll
      for (int j = 0; j < kPre; ++j) mm &= mm - 1;
      int cur = e0 + kPre;
      while (mm) {
        const int v = __ffs(mm) - 1;
        mm &= mm - 1;
        if ((mset >> v) & 1u) record(v, cur, e_q0[cur], e_q1[cur]);
        ++cur;
      }
    }
  }
  if (__ballot(any_bad) != 0ull && (threadIdx.x & 63) == 0) {
    atomicOr(status, 4);
    if (flag) atomicOr(flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------
// step 2: blend of a sorted list (forward.cu:261-374 + w-depth README:5-11).
// One workgroup = one vertical pair of 16x16 tiles of one rendered item; wave w owns the 16x8 pixel block of rows
// [8w, 8w+8) of the pair (waves 0-1: upper tile, 2-3: lower tile), thread (lx, r) the pixels (lx, 8w + r) and
// (lx, 8w + r + 4): same x, so dx and the dx-only part of the exponent are shared and the two pixels run as the
// halves of packed fp32 ops.
// ---------------------------------------------------------------------------------------------
#ifdef OCRF_PLAN_WAVES
#define OCRF_BLEND_BOUNDS __launch_bounds__(kBlock, OCRF_PLAN_WAVES)      // A/B build: waves per SIMD asked of the compiler
#else
#define OCRF_BLEND_BOUNDS __launch_bounds__(kBlock)
#endif
template <bool MEDIAN, bool WSKIP, bool STATS = false>
__global__ OCRF_BLEND_BOUNDS void raster_blend_sorted_kernel(
    unsigned long long* __restrict__ stats, int P, int W, int H, int gx, int gy, int n_items, int vps, long set_stride,
    const int* __restrict__ header, const int* __restrict__ view_sel, const unsigned* __restrict__ s_id,
    const unsigned* __restrict__ s_key, const float2* __restrict__ s_pix, const unsigned* __restrict__ s_e,
    const Rect* __restrict__ d_rect, const float4* __restrict__ d_con, const float* __restrict__ colors,
    const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_final_T,
    const int* __restrict__ skip_if, int* __restrict__ queue, int* __restrict__ chain_hist, int chain_hist_words,
    const int* __restrict__ yield_if, int base_grid) {
  if (skip_if && *skip_if != 0) {
    // the plan's bound does not hold this step: the armed per-call chain renders.  Its bucket histograms are cleared
    // here — by the kernel that sits in front of it anyway — instead of by a launch of their own in every call.
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < chain_hist_words; i += gridDim.x * kBlock) chain_hist[i] = 0;
    return;
  }
  // A scheduling hint, not a dependency: launched on every slot the device has, the workgroups beyond `base_grid`
  // leave at once while *yield_if says another stream's chain is still running (it wants those wave slots); the
  // ticket queue makes any number of participants render the same image.
  if (yield_if && (int)blockIdx.x >= base_grid && *yield_if != 0) return;
  __shared__ unsigned l_pos[kCapPos];
  __shared__ float4 l_a[kStageP + 1], l_b[kStageP + 1], l_c[kStageP + 1];
  __shared__ unsigned short l_list[4][kStageP + 3 * kTrip + 2];
  __shared__ int l_wtot[kScanUnrollP * 4];
  __shared__ int l_lcnt[kSrcWaves][4];        // [source wave of the batch copy][destination wave]
  __shared__ int l_work;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (header[0] != (int)kPlanMagic) return;                           // reported by the update kernel (status bit 8)
  const int V = header[2];
  const int* view_off = header + kHeaderInts;
  const int lx = lane & 15, r = lane >> 4;
  const int gyp = (gy + 1) / 2;
  const int n_work = gx * gyp * n_items;
  if (tid == 0) {      // slot kStageP: a record that changes nothing (opacity 0), pads odd list lengths
    l_a[kStageP] = make_float4(0.f, 0.f, 0.f, 0.f);
    l_b[kStageP] = make_float4(0.f, 0.f, 0.f, 0.f);
    l_c[kStageP] = make_float4(0.f, 0.f, 0.f, INFINITY);
  }
  // Persistent workgroups over a ticket queue: the grid is what the chip holds at once, every workgroup takes the next
  // tile pair until none is left (2 112 pairs on 1 792 resident workgroups would otherwise run as a full round plus
  // a nearly empty one).  Every wave reaches the exit: the ticket counter only grows.
  for (;;) {
  __syncthreads();                                                    // the previous pair's LDS traffic is over
  if (tid == 0) l_work = atomicAdd(queue, 1);
  __syncthreads();
  const int work = l_work;
  if (work >= n_work) break;
  const int z = work / (gx * gyp);
  const int tx = (work - z * gx * gyp) % gx, ty2 = (work - z * gx * gyp) / gx;
  const int v = view_sel ? view_sel[z] : z;
  if (v < 0 || v >= V) continue;                                      // reported by the update kernel (status bit 8)
  const int set = z / vps;
  const int tyA = 2 * ty2, tyB = tyA + 1;
  const int off = view_off[v], nv = view_off[v + 1] - off;
  const float4* set_con = d_con + (long)set * set_stride;
  const Rect* set_rect = d_rect + (long)set * set_stride;
  const float* set_colors = colors + 3 * (long)set * P;
  const int pxi = tx * kTileX + lx;
  const int py0 = tyA * kTileY + 8 * wave + r, py1 = py0 + 4;
  const bool tile_ok = (tyA + (wave >> 1)) < gy;
  const bool inside0 = tile_ok && pxi < W && py0 < H, inside1 = tile_ok && pxi < W && py1 < H;
  const float pixf_x = (float)pxi;
  const f2 pixf_y = f2{(float)py0, (float)py1};
  // the wave's pixel block as float bounds (pixel centres are the integer coordinates, forward.cu:283)
  const float bx0 = (float)(tx * kTileX), bx1 = bx0 + 15.f;

  // T < 0 <=> the pixel has stopped (or lies outside the image); |T| is its final transmittance
  f2 T = f2{inside0 ? 1.0f : -1.0f, inside1 ? 1.0f : -1.0f};
  f2 C0 = splat(0.f), C1 = splat(0.f), C2 = splat(0.f);
  f2 D = splat(MEDIAN ? 15.0f : 0.0f);
  f2 h = splat(0.5f);                          // T - 0.5 carried from record to record while the median test is live

  int scan = 0, npos = 0;
  bool all_done = false;
  // diagnostic build only: phase cycles and record counts of this workgroup
  unsigned long long t_prev = 0, t_acc[3] = {0, 0, 0};
  unsigned n_staged = 0, n_listed = 0, n_eval = 0, n_newstop = 0;
  auto stamp = [&](int slot) {
    if constexpr (STATS) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_prev;
      t_prev = t;
    }
  };
  stamp(-1);
  while (!all_done) {
    // ---- scan: positions of the records whose rect covers this tile pair, in list (= blend) order ----
    while (scan < nv && npos < kStageP) {
#ifdef OCRF_PLAN_SCAN_WIDE_FIRST
      const int n_u = kScanUnrollP;
#else
      const int n_u = (scan < kBlock) ? 1 : kScanUnrollP;      // a dense tile pair fills its first batch from 256 rects
#endif
      unsigned code[kScanUnrollP];
      bool hit[kScanUnrollP];
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        const int i = scan + u * kBlock + tid;
        hit[u] = false;
        code[u] = 0u;
        if (u < n_u && i < nv) {
          const Rect rc = set_rect[s_e[off + i]];      // list order -> the record's Gaussian-major slot
          const bool cA = (tyA >= rc.y0) && (tyA < rc.y1), cB = (tyB >= rc.y0) && (tyB < rc.y1);
          hit[u] = (tx >= rc.x0) && (tx < rc.x1) && (cA || cB);
          code[u] = (unsigned)i | (cA ? 0x40000000u : 0u) | (cB ? 0x80000000u : 0u);
        }
      }
      int rank[kScanUnrollP];
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        const unsigned long long m = __ballot(hit[u]);
        rank[u] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) l_wtot[u * 4 + wave] = __popcll(m);
      }
      __syncthreads();
      int o = npos;
#pragma unroll
      for (int u = 0; u < kScanUnrollP; ++u) {
        int mine = o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int c = l_wtot[u * 4 + w];
          if (w < wave) mine += c;
          o += c;
        }
        if (hit[u]) l_pos[mine + rank[u]] = code[u];
      }
      npos = o;
      scan += n_u * kBlock;
      __syncthreads();
    }
    stamp(0);
    if (npos == 0) break;                     // list exhausted
    for (int s0 = 0; s0 < npos && !all_done; s0 += kStageP) {
      const int ns = min(kStageP, npos - s0);
      if constexpr (STATS) n_staged += ns;
      // ---- stage ns records; which waves can each one reach?  kStageParts threads per record, each testing
      // kReachPerThread of the four waves (the copies of the batch live in waves [part * kSrcWaves, ...)) ----
      const int part = tid / kStageP, ri = tid % kStageP;
      bool reach[kReachPerThread];
      bool simple = false;                     // see the blend loop: no min(0.99, .) and no `power > 0` test needed
#pragma unroll
      for (int j = 0; j < kReachPerThread; ++j) reach[j] = false;
      if (ri < ns) {
        // staged record (slots chosen so that operands broadcast into packed ops sit in slots 0-2):
        //   a = (x, y, -0.5 conic.z, -0.5 conic.x)   b = (opacity, r, g, conic.y)   c = (b, depth, 0, thr)
        const unsigned code = l_pos[s0 + ri];
        const int li = (int)(code & kPosMask);
        const float4 con = set_con[s_e[off + li]];
        const float2 pix = s_pix[off + li];
        const float o = con.w;
        // the power below which alpha = o exp(power) is under 1/255 whatever the pixel (1 % margin for v_exp_f32
        // and the log2(e) multiply); o <= 0: +inf (never rendered); NaN opacity: NaN (evaluated in full)
        const float thr = (o > 0.f) ? (__logf(1.0f / (255.0f * o)) - 0.01f) : ((o <= 0.f) ? INFINITY : o);
        const float4 a = make_float4(pix.x, pix.y, con.y, con.x);
        if (part == 0) {
          const float* col = set_colors + 3 * (long)s_id[off + li];
          l_a[ri] = a;
          l_b[ri] = make_float4(o, col[0], col[1], con.z);
          l_c[ri] = make_float4(col[2], __uint_as_float(s_key[off + li]), 0.f, thr);
        }
        // alpha >= 1/255 needs power >= thr, i.e. Q(dx, dy) = 0.5 (A dx^2 + C dy^2) + B dx dy <= -thr.  The minimum of
        // the convex Q over a wave's pixel block (a box in (dx, dy)) is 0 if the centre lies inside, else it is on
        // the box's boundary: per edge a clamped 1-D minimiser.  The block is skipped only if that minimum exceeds
        // -thr by more than the rounding of both evaluations (<= 1e-6 of the sum of the terms' magnitudes; 4e-6
        // is allowed for) — so a skipped record has alpha < 1/255 at every pixel of the block, where the reference
        // skips it too (forward.cu:331-333).  Anything unusual (NaN, non-convex conic) is evaluated in full.
        const float qa = -2.f * a.w, qc = -2.f * a.z, qb = con.z;
        const bool convex = (qa > 0.f) && (qc > 0.f) && (qa * qc - qb * qb > 0.f);
        simple = (o <= 0.99f) && (qa > 0.f) && (qc > 0.f) && (qb * qb <= 0.9990234375f * (qa * qc));
        const bool never = thr >= 0.f;                         // opacity < 1/255: no pixel ever blends it
        const float lim = -thr;
        const float inv_a = 1.f / qa, inv_c = 1.f / qc;
        const float dxlo = a.x - bx1, dxhi = a.x - bx0;
        const float Dx = fmaxf(fabsf(dxlo), fabsf(dxhi));
#pragma unroll
        for (int j = 0; j < kReachPerThread; ++j) {
          const int w = part * kReachPerThread + j;
          const bool cov = (w < 2) ? ((code & 0x40000000u) != 0u) : ((code & 0x80000000u) != 0u);
          const float by0 = (float)(tyA * kTileY + 8 * w), by1 = by0 + 7.f;
          const float dylo = a.y - by1, dyhi = a.y - by0;
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
      // ordered per-wave lists of staged indices: the kSrcWaves waves of a part hold the batch in order
      const int src = wave % kSrcWaves;
      int lrank[kReachPerThread];
#pragma unroll
      for (int j = 0; j < kReachPerThread; ++j) {
        const unsigned long long m = __ballot(reach[j]);
        lrank[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) l_lcnt[src][part * kReachPerThread + j] = __popcll(m);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kReachPerThread; ++j) {
        const int w = part * kReachPerThread + j;
        int base = 0, tot = 0;
#pragma unroll
        for (int sw = 0; sw < kSrcWaves; ++sw) {
          const int c = l_lcnt[sw][w];
          if (sw < src) base += c;
          tot += c;
        }
        if (reach[j]) l_list[w][base + lrank[j]] = (unsigned short)(ri | (simple ? 0x8000 : 0));
        if (ri == 0) {                         // pad: the loop reads two entries per trip, two trips ahead
#pragma unroll
          for (int q = 0; q < 3 * kTrip; ++q) l_list[w][tot + q] = (unsigned short)kStageP;
        }
      }
      int n_mine = 0;                          // length of THIS wave's list
#pragma unroll
      for (int sw = 0; sw < kSrcWaves; ++sw) n_mine += l_lcnt[sw][wave];
      __syncthreads();
      n_mine = __builtin_amdgcn_readfirstlane(n_mine);
      if constexpr (STATS) n_listed += n_mine;
      stamp(1);

      // ---- blend this wave's records front to back.  Every decision of forward.cu:320-352 is ONE compare feeding
      // ONE select (see raster_blend_kernel in rasterize.hip for the derivation); same arithmetic, same order.
      // Measured issue costs on gfx950 (tools/ubench/valu_cost.hip; plain VALU = 1): packed f32 1.4 (for two
      // pixels), v_exp_f32 2.4, a compare + select pair 2.5 — the selects are 40 % of a record's cost, so the loop
      // leaves out every decision whose outcome is known for the whole wave:
      //   * SIMPLE records (flag in the list entry, set at staging): opacity <= 0.99, so min(0.99, alpha) is the
      //     identity (exp2 of a non-positive power is <= 1), and a conic with B^2 <= (1 - 2^-10) A C, for which the
      //     computed power is <= 0 at every pixel (its rounding error is ~1e-3 of that margin): no `power > 0` test;
      //   * the median-depth test runs only while some pixel of the wave still has T > 0.5 (T never rises).
      {
        auto blend_one = [&](auto med_tag, auto nostop_tag, auto simple_tag, const float4 a, const float4 b, const float4 c) {
          constexpr bool MED = decltype(med_tag)::value, SIMPLE = decltype(simple_tag)::value;
          constexpr bool NOSTOP = decltype(nostop_tag)::value;
          const float cr = b.y, cg = b.z, cb = c.x, dep = c.y;
          // power = -0.5 (cxx dx dx + czz dy dy) - cxy dx dy, in the reference's order (forward.cu:320-323)
          const float dx = a.x - pixf_x;
          const float qx = (a.w * dx) * dx;
          const float bx = b.w * dx;
          const f2 dy = splat(a.y) - pixf_y;
          const f2 qy = (splat(a.z) * dy) * dy;
          const f2 power = (splat(qx) + qy) - splat(bx) * dy;          // qx, qy carry the -0.5 (staging)
          if constexpr (WSKIP) {
            const float thr = c.w;
            if (__ballot(!((power.x <= thr) & (power.y <= thr))) == 0ull) return;
          }
          const f2 p2 = power * splat(1.44269504088896340736f);       // __expf(x) = v_exp_f32(log2(e) x)
          f2 G;
          G.x = __builtin_amdgcn_exp2f(p2.x);
          G.y = __builtin_amdgcn_exp2f(p2.y);
          f2 alpha = splat(b.x) * G;
          if constexpr (SIMPLE) {
            alpha.x = (alpha.x < 1.0f / 255.0f) ? 0.f : alpha.x;
            alpha.y = (alpha.y < 1.0f / 255.0f) ? 0.f : alpha.y;
          } else {
            alpha.x = fminf(0.99f, alpha.x);
            alpha.y = fminf(0.99f, alpha.y);
            alpha.x = ((power.x > 0.0f) | (alpha.x < 1.0f / 255.0f)) ? 0.f : alpha.x;
            alpha.y = ((power.y > 0.0f) | (alpha.y < 1.0f / 255.0f)) ? 0.f : alpha.y;
          }
          const f2 test_T = T * (splat(1.0f) - alpha);
          const f2 aT = alpha * T;
          // NOSTOP trips: every pixel of the wave inside the image has T > kNoStopT, so test_T >= T (1 - 0.99f) > 1.2e-4
          // and the stop test (forward.cu:341-345) is false whatever the record: no compare, no selects
          bool stop0 = false, stop1 = false;
          f2 wgt = aT;
          if constexpr (!NOSTOP) {
            stop0 = test_T.x < 0.0001f;
            stop1 = test_T.y < 0.0001f;
            if constexpr (STATS) {      // records at which some pixel of the wave stops (diagnostic)
              if (__ballot((stop0 && T.x > 0.f) || (stop1 && T.y > 0.f)) != 0ull) ++n_newstop;
            }
            wgt.x = stop0 ? 0.f : aT.x;
            wgt.y = stop1 ? 0.f : aT.y;
          }
          C0 = fma2(splat(cr), wgt, C0);
          C1 = fma2(splat(cg), wgt, C1);
          C2 = fma2(splat(cb), wgt, C2);
          if constexpr (MEDIAN) {
            if constexpr (MED) {
              const f2 h2 = test_T - splat(0.5f);
              const f2 cross = h * h2;
              D.x = (cross.x < 0.f) ? dep : D.x;
              D.y = (cross.y < 0.f) ? dep : D.y;
              h = h2;
            }
          } else {
            D = fma2(splat(dep), wgt, D);
          }
          if constexpr (NOSTOP) {
            T = test_T;
          } else {
            T.x = stop0 ? -fabsf(T.x) : test_T.x;
            T.y = stop1 ? -fabsf(T.y) : test_T.y;
          }
        };
        // The list is padded with no-op entries (a trip reads TR entries whatever the list's length).
        const unsigned short* mylist = l_list[wave];
        constexpr int TR = kTrip;                // records per trip: independent exponent / alpha chains in flight
#ifdef OCRF_PLAN_PIPELINE
        // (A/B build) software pipeline, two deep: the records of the NEXT trip and the list entries of the one after
        // it are requested before this trip's arithmetic (index -> record is two dependent LDS reads)
        float4 na[TR], nb[TR], nc[TR];
        unsigned fl_next[TR / 2];
        auto fetch = [&](const unsigned* pairs) {
#pragma unroll
          for (int u = 0; u < TR / 2; ++u) {
            const unsigned pair = pairs[u];
            const int i0 = (int)(pair & 0x1FFu), i1 = (int)((pair >> 16) & 0x1FFu);
            na[2 * u] = l_a[i0]; nb[2 * u] = l_b[i0]; nc[2 * u] = l_c[i0];
            na[2 * u + 1] = l_a[i1]; nb[2 * u + 1] = l_b[i1]; nc[2 * u + 1] = l_c[i1];
            fl_next[u] = pair;
          }
        };
        unsigned pair_next[TR / 2];
#pragma unroll
        for (int u = 0; u < TR / 2; ++u) pair_next[u] = *reinterpret_cast<const unsigned*>(mylist + 2 * u);
        fetch(pair_next);
#pragma unroll
        for (int u = 0; u < TR / 2; ++u) pair_next[u] = *reinterpret_cast<const unsigned*>(mylist + TR + 2 * u);
#endif
        auto trip = [&](auto med_tag, auto nostop_tag, int k) {
          if constexpr (STATS) n_eval += TR;
          float4 ra[TR], rb[TR], rc4[TR];
          unsigned fl[TR / 2];
#ifndef OCRF_PLAN_PIPELINE
          // No software pipeline: a trip reads its own records.  Round 3 prefetched the next trip's records and the list
          // entries of the one after it (24 more VGPRs: 118, four waves per SIMD) because two workgroups per CU left too
          // few waves to hide the two dependent LDS reads; at 95 VGPRs a SIMD holds FIVE waves, which hide them better
          // than the prefetch did: the 12 views of cfg2 alone 145 -> 136 us, cfg2 step 0.256 -> 0.246 ms (-DOCRF_PLAN_PIPELINE builds
          // the old form for the A/B; asking the compiler for 6 / 8 waves spills and loses: tools/sweep_r4.sh)
#pragma unroll
          for (int u = 0; u < TR / 2; ++u) {
            const unsigned pair = *reinterpret_cast<const unsigned*>(mylist + k + 2 * u);
            const int i0 = (int)(pair & 0x1FFu), i1 = (int)((pair >> 16) & 0x1FFu);
            ra[2 * u] = l_a[i0]; rb[2 * u] = l_b[i0]; rc4[2 * u] = l_c[i0];
            ra[2 * u + 1] = l_a[i1]; rb[2 * u + 1] = l_b[i1]; rc4[2 * u + 1] = l_c[i1];
            fl[u] = __builtin_amdgcn_readfirstlane(pair);
          }
#else
#pragma unroll
          for (int u = 0; u < TR; ++u) { ra[u] = na[u]; rb[u] = nb[u]; rc4[u] = nc[u]; }
#pragma unroll
          for (int u = 0; u < TR / 2; ++u) fl[u] = __builtin_amdgcn_readfirstlane(fl_next[u]);
          fetch(pair_next);
#pragma unroll
          for (int u = 0; u < TR / 2; ++u) pair_next[u] = *reinterpret_cast<const unsigned*>(mylist + k + 2 * TR + 2 * u);
#endif
#pragma unroll
          for (int u = 0; u < TR; ++u) {
            const bool simple_rec = (fl[u / 2] >> ((u & 1) ? 31 : 15)) & 1u;
            if (simple_rec) blend_one(med_tag, nostop_tag, std::true_type{}, ra[u], rb[u], rc4[u]);
            else blend_one(med_tag, nostop_tag, std::false_type{}, ra[u], rb[u], rc4[u]);
          }
        };
        int k = 0;
        // "some pixel of the wave could stop at the next record": a pixel inside the image at or below kNoStopT (stopped
        // pixels carry T < 0).  It only ever turns true: the loops run in the order (median, no stop) -> (median, stop) ->
        // (no median, no stop) -> (no median, stop), each leaving when its own condition ends.
        constexpr float kNoStopT = 0.0125f;     // T > 1/80 and alpha <= 0.99  =>  T (1 - alpha) > 1.2e-4 > the 1e-4 of the stop test
        auto may_stop = [&]() { return __ballot((inside0 & !(T.x > kNoStopT)) | (inside1 & !(T.y > kNoStopT))) != 0ull; };
        if constexpr (MEDIAN) {
          h = T - splat(0.5f);
          for (; k < n_mine; k += TR) {
            if (may_stop()) break;
            if (__ballot((T.x > 0.5f) | (T.y > 0.5f)) == 0ull) break;
            trip(std::true_type{}, std::true_type{}, k);
          }
          for (; k < n_mine; k += TR) {
            // live <=> sign bit of T clear; the median test is needed while some pixel is still above 0.5
            if (__ballot((__float_as_int(T.x) & __float_as_int(T.y)) >= 0) == 0ull) { k = n_mine; break; }
            if (__ballot((T.x > 0.5f) | (T.y > 0.5f)) == 0ull) break;
            trip(std::true_type{}, std::false_type{}, k);
          }
        }
        for (; k < n_mine; k += TR) {
          if (may_stop()) break;
          trip(std::false_type{}, std::true_type{}, k);
        }
        for (; k < n_mine; k += TR) {
          if (__ballot((__float_as_int(T.x) & __float_as_int(T.y)) >= 0) == 0ull) break;      // every pixel stopped
          trip(std::false_type{}, std::false_type{}, k);
        }
      }
      // every pixel saturated -> stop (forward.cu:304-307)
      all_done = __syncthreads_count((T.x < 0.f) && (T.y < 0.f)) == kBlock;
      stamp(2);
    }
    npos = 0;
    if (scan >= nv) break;
  }
  if constexpr (STATS) {
    // per wave: slot = workgroup * 4 + wave: scan, stage, blend cycles | scanned, staged, listed, evaluated records
    if (lane == 0) {
      const long w = (long)work * 4 + wave;
      stats[w * 8 + 0] = t_acc[0]; stats[w * 8 + 1] = t_acc[1]; stats[w * 8 + 2] = t_acc[2];
      stats[w * 8 + 3] = (unsigned long long)scan; stats[w * 8 + 4] = n_staged;
      stats[w * 8 + 5] = n_listed; stats[w * 8 + 6] = n_eval; stats[w * 8 + 7] = n_newstop;
    }
  }

  const long npix = (long)W * H;
  auto store = [&](bool inside, int py, float t, float c0, float c1, float c2, float d) {
    if (!inside) return;
    const long pix = (long)py * W + pxi;
    out_final_T[z * npix + pix] = t;
    out_color[((long)z * 3 + 0) * npix + pix] = c0 + t * bg[0];
    out_color[((long)z * 3 + 1) * npix + pix] = c1 + t * bg[1];
    out_color[((long)z * 3 + 2) * npix + pix] = c2 + t * bg[2];
    out_depth[z * npix + pix] = d;
  };
  store(inside0, py0, fabsf(T.x), C0.x, C1.x, C2.x, D.x);
  store(inside1, py1, fabsf(T.y), C0.y, C1.y, C2.y, D.y);
  }   // ticket loop
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
    cached[dev] = per_cu * cus;
  }
  return cached[dev];
}

int g_plan_wskip = 0;
int g_update_lds = 0;       // ocrf_tune_set(12, bytes): dynamic LDS padding of the update kernel = an occupancy cap (diagnostic)
int g_plan_grid = 0;        // ocrf_tune_set(11, n): workgroups of the persistent blend (0 = what the device holds at once)
unsigned long long* g_plan_stats = nullptr;      // ocrf_diag_plan_stats: the next planned blends run the STATS build       // ocrf_tune_set(OCRF_TUNE_PLAN_WSKIP): the pixel-exact wave skip inside the loop

}  // namespace

namespace ocrf {
void raster_plan_tune(int key, int value) {
  if (key == 10) g_plan_wskip = value != 0;
  if (key == 11) g_plan_grid = value > 0 ? value : 0;
  if (key == 12) g_update_lds = value > 0 ? value : 0;
}
}

extern "C" {

// Diagnostic: the size of the persistent blend's grid (median depth, no wave skip) as the occupancy API reports it
int ocrf_diag_plan_resident(void) { return resident_blocks(raster_blend_sorted_kernel<true, false>); }

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
  e = ocrf::radix_sort_ids(keys, (int)capacity, 32, wb + B.sort, B.scan - B.sort, &sk, &se, stream);
  if (e != hipSuccess) return (int)e;
  const unsigned long long* sort_states = nullptr;
  int n_states = 0;
  long stride = 0;
  ocrf::radix_sort_states(wb + B.sort, (int)capacity, 32, &sort_states, &n_states, &stride);
  const unsigned ggrid = (unsigned)std::min<long>((capacity + kBlock - 1) / kBlock, 4096);
  hipLaunchKernelGGL(plan_gather_kernel, dim3(ggrid), dim3(kBlock), 0, stream, P, n_views, H, W, gx, gy, extent_bound_,
                     capacity, static_cast<const int*>(counts), cameras, sk, se, static_cast<const int*>(rec_id),
                     static_cast<const float4*>(e_q1), sort_states, n_states, stride,
                     reinterpret_cast<const unsigned long long*>(wb + B.scan), header,
                     reinterpret_cast<float*>(pb + L.cams), reinterpret_cast<unsigned*>(pb + L.s_id),
                     reinterpret_cast<unsigned*>(pb + L.s_key), reinterpret_cast<float2*>(pb + L.s_pix),
                     reinterpret_cast<unsigned*>(pb + L.s_e));
  return (int)hipGetLastError();
}

size_t ocrf_rasterize_planned_workspace_bytes(long total_kept, int n_sets) {
  if (total_kept < 0 || n_sets <= 0) return 0;
  DynLayout L;
  dyn_layout(total_kept, n_sets, &L);
  return L.bytes;
}

int ocrf_rasterize_planned(const void* plan, size_t plan_bytes, int P, int n_plan_views, long total_kept, int H, int W,
                           int n_sets, int n_items, const int* item_view, const float* colors,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* bg, int depth_mode, float* out_color, float* out_depth, float* out_final_T,
                           int* radii, int* status, void* workspace, size_t workspace_bytes, int guard,
                           const float* means3D, void* chain_workspace, size_t chain_workspace_bytes,
                           int blend_workgroups, const int* yield_if, int phase, const float* call_cameras,
                           int views_disjoint, ocrf_stream_t stream_) {
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
  DynLayout D;
  dyn_layout(total_kept, views_disjoint ? 1 : n_sets, &D);
  if (plan_bytes < L.bytes || workspace_bytes < D.bytes) return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  const char* pb = static_cast<const char*>(plan);
  char* wb = static_cast<char*>(workspace);
  const int* header = reinterpret_cast<const int*>(pb + L.header);
  const Camera* cams = reinterpret_cast<const Camera*>(pb + L.cams);
  auto* d_rect = reinterpret_cast<Rect*>(wb + D.rect);
  auto* d_con = reinterpret_cast<float4*>(wb + D.con);
  int* queue = reinterpret_cast<int*>(wb + D.flag) + 16;
  int* flag = nullptr;
  int* chain_hist = nullptr;
  size_t chain_hist_words = 0;
  if (guard) {
    if (!means3D || !radii || !chain_workspace ||
        chain_workspace_bytes < ocrf_rasterize_workspace_bytes(P, n_items))
      return (int)hipErrorInvalidValue;
    // flag[0]: "the extent check fired".  No memset per call: the update kernel only RAISES it, the armed blend lowers
    // it after a call that fired, so it is zero on entry unless a fired call was cut short — then this call takes the
    // exact per-call path once more and lowers it.  (The scratch is zero-filled when it is allocated; any other first
    // value only costs one slow call.)  flag[1]: arrival counter of that blend, zeroed by the update kernel.
    flag = reinterpret_cast<int*>(wb + D.flag);
    chain_hist = ocrf::raster_chain_hist(chain_workspace, P, n_items, &chain_hist_words);
  }
  if (phase != 2) {
  ocrf::launch(OCRF_K_RASTER_PLAN_UPDATE, raster_plan_update_kernel, dim3((P + kBlock - 1) / kBlock),
               dim3(kBlock), (size_t)g_update_lds, stream, P, vps, n_sets, set_stride, total_kept, header,
               reinterpret_cast<const unsigned*>(pb + L.g_mask),
               reinterpret_cast<const int*>(pb + L.g_off),
               reinterpret_cast<const float4*>(pb + L.e_q0), reinterpret_cast<const float4*>(pb + L.e_q1), item_view,
               opacities, scales, scale_modifier, rotations, d_rect, d_con, radii, status, flag, queue,
               reinterpret_cast<const unsigned*>(call_cameras), reinterpret_cast<const unsigned*>(cams));
  }   // phase != 2
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (phase == 1) return 0;
  const int n_work = gx * ((gy + 1) / 2) * n_items;
  // yield_if: the grid is everything the device holds, `blend_workgroups` of it stay whatever the hint says
  const int base_grid = (yield_if && blend_workgroups > 0) ? blend_workgroups : (1 << 30);
  const int want_grid = g_plan_grid ? g_plan_grid : ((yield_if && blend_workgroups > 0) ? 0 : blend_workgroups);      // the diagnostic knob wins
  if (g_plan_stats) {      // diagnostic build (median depth), never used by the product path
    const dim3 sgrid((unsigned)std::min(n_work, want_grid ? want_grid : resident_blocks(raster_blend_sorted_kernel<true, false, true>)));
    hipLaunchKernelGGL((raster_blend_sorted_kernel<true, false, true>), sgrid, dim3(kBlock), 0, stream, g_plan_stats, P, W, H,
                       gx, gy, n_items, vps, set_stride, header, item_view, reinterpret_cast<const unsigned*>(pb + L.s_id),
                       reinterpret_cast<const unsigned*>(pb + L.s_key), reinterpret_cast<const float2*>(pb + L.s_pix),
                       reinterpret_cast<const unsigned*>(pb + L.s_e), static_cast<const Rect*>(d_rect),
                       static_cast<const float4*>(d_con), colors, bg, out_color, out_depth, out_final_T,
                       static_cast<const int*>(flag), queue, chain_hist, (int)chain_hist_words, yield_if, base_grid);
    return (int)hipGetLastError();
  }
#define OCRF_BLEND_SORTED(MED, WS)                                                                                   \
  ocrf::launch(OCRF_K_RASTER_BLEND_SORTED, raster_blend_sorted_kernel<MED, WS>,                                       \
               dim3((unsigned)std::min(n_work, want_grid ? want_grid                                                   \
                                                          : resident_blocks(raster_blend_sorted_kernel<MED, WS>))),     \
               dim3(kBlock), 0,                                                                                        \
               stream, (unsigned long long*)nullptr, P, W, H, gx, gy, n_items, vps, set_stride, header, item_view,     \
               reinterpret_cast<const unsigned*>(pb + L.s_id), reinterpret_cast<const unsigned*>(pb + L.s_key),        \
               reinterpret_cast<const float2*>(pb + L.s_pix), reinterpret_cast<const unsigned*>(pb + L.s_e),           \
               static_cast<const Rect*>(d_rect),                                                                       \
               static_cast<const float4*>(d_con), colors, bg, out_color, out_depth, out_final_T,                       \
               static_cast<const int*>(flag), queue, chain_hist, (int)chain_hist_words, yield_if, base_grid)
  if (depth_mode == 0 && g_plan_wskip) OCRF_BLEND_SORTED(true, true);
  else if (depth_mode == 0) OCRF_BLEND_SORTED(true, false);
  else if (g_plan_wskip) OCRF_BLEND_SORTED(false, true);
  else OCRF_BLEND_SORTED(false, false);
#undef OCRF_BLEND_SORTED
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
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

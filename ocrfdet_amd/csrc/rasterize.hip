// Tile-based Gaussian rasteriser forward for MI355X (gfx950): RGB + depth + transmittance.
//
// Replaces the forward of `diff_gaussian_rasterization` (w-depth flavour) as OcRFDet calls it
// (mmdet3d/models/necks/MVSGaussian/lib/gaussian_renderer/__init__.py:39-70).  Reference
// arithmetic: .../submodules/diff-gaussian-rasterization/cuda_rasterizer/forward.cu:74-256
// (preprocess), :261-374 (blend), auxiliary.h:41-77,139-164, rasterizer_impl.cu:198-336; the
// depth channel follows diff-gaussian-rasterization-w-depth/README.md:5-11 (median depth,
// default 15; source absent from the reference tree).
//
// This is NOT the reference's pipeline re-typed.  The reference duplicates every Gaussian once
// per touched tile (R = 0.8 M .. tens of M (tile|depth, id) pairs), radix-sorts all R 64-bit
// keys, derives tile ranges and needs a blocking device->host read of R to size its buffers.
// Here (DESIGN.md "rasteriser"):
//   1. preprocess  — one thread per (view, Gaussian); same arithmetic, fp-contract off so that
//                    radii / tile rectangles are compiler-independent; visible Gaussians also
//                    count themselves into a histogram of 8192 depth buckets per view (bucket =
//                    top 18 bits of the positive depth float: 0.2 % wide), through an LDS
//                    histogram per 2048-Gaussian workgroup;
//   2. bucket scan — exclusive scan of the histogram (one workgroup per view);
//   3. scatter     — every visible Gaussian drops its (tile rect, depth|id) record into its
//                    bucket's range: the list is now ordered by bucket, unordered inside one;
//   4. blend       — one 16x16-pixel workgroup per (tile, view) scans the bucket-ordered list,
//                    keeps the records whose tile rectangle covers its tile (wave ballot + prefix
//                    compaction into LDS), sorts what it holds EXACTLY by (depth bits, id) in LDS
//                    (bitonic), blends the records of complete buckets front to back and carries
//                    an incomplete last bucket into the next round; it stops as soon as every
//                    pixel is saturated, i.e. usually after a few hundred records.
// The sequence each pixel blends is therefore identical, element for element, to the reference's
// sorted per-tile range — (depth, id) ascending — but nothing R-sized is ever built, only the
// front of each tile's list is ever sorted, and there is no host synchronisation: the whole
// forward is hipGraph-capturable and batches any number of views over one Gaussian set.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"
#include "raster_blend_body.h"
#include "raster_blend_math.h"
#include "raster_common.h"

namespace {

using rc::kBlock;
using rc::kTileX;
using rc::kTileY;
using rc::Camera;
using rc::Rect;
using rc::ndc2pix;
using rc::f2;
using rc::splat;
using rc::fma2;

constexpr int kBuckets = 8192;               // depth buckets per view
constexpr int kBucketShift = 14;             // bucket = (depth bits >> 14) - base: 9 mantissa bits
constexpr unsigned kBucketBase = 0x3E4CCCCDu >> kBucketShift;   // depth > 0.2f always (auxiliary.h:154)
constexpr int kChunk = 2048;                 // compact records per workgroup in the scatter
constexpr int kPreChunk = 1024;              // Gaussians per workgroup in the preprocess
constexpr int kCapRec = 2048;                // LDS record capacity of the blend kernel (28 KB with the stage: 5 workgroups per CU)
constexpr int kStage = 256;                  // payload entries staged per blend batch
constexpr int kScanUnroll = 4;               // rect batches in flight in the scan
constexpr unsigned long long kPad = ~0ull;

__device__ __forceinline__ int bucket_of(unsigned key) {
  const int b = (int)(key >> kBucketShift) - (int)kBucketBase;
  return min(max(b, 0), kBuckets - 1);
}

// ---------------------------------------------------------------------------------------------
// 1. preprocess (forward.cu:155-256).  Expression order mirrors oracle/rasterize_ref.c exactly.
//    Per-Gaussian state is kept in id order: key (depth bits, 0xFFFFFFFF = not rendered), rect,
//    xy, conic_opacity.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_preprocess_kernel(
    int P, int n_views, int n_chunks, int W, int H, int gx, int gy, const float* __restrict__ means3D,
    const float* __restrict__ opacities, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const Camera* __restrict__ cams, uint4* __restrict__ vis_rec, int* __restrict__ vis_count,
    Rect* __restrict__ rects, float2* __restrict__ xy, float4* __restrict__ conic_o,
    int* __restrict__ radii, unsigned* __restrict__ tiles_touched, int* __restrict__ hist, int vps,
    const int* __restrict__ view_sel, const int* __restrict__ gate, int means_stride, int tighten) {
  if (gate && *gate == 0) return;      // armed as a fallback that is not needed (raster_plan.hip)
  // One workgroup = kPreChunk consecutive Gaussians of one view.  Their depth buckets are counted in
  // an LDS histogram first and only the non-empty bins go to the global one: scattered global
  // atomics run at ~20 G/s chip-wide, and a depth slice of a regular grid puts thousands of
  // Gaussians into ONE bucket.
  // Workgroup -> (chunk, view), XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so
  // ids L and L+8 share an L2; the V views of a chunk get ids 8 apart and run back to back on one
  // XCD, which reads the chunk's 90 KB of Gaussian parameters from HBM once instead of V times.
  // two 16-bit counters per word (a workgroup counts at most kPreChunk Gaussians: no carry):
  // 16 KB instead of 32 KB of LDS, so the 92-VGPR limit (5 workgroups per CU), not the LDS (4), sets
  // the occupancy
  static_assert(kPreChunk < 65536, "16-bit bucket counters");
  __shared__ unsigned s_hist[kBuckets / 2];
  __shared__ int s_wsum[kBlock / 64];
  __shared__ int s_base;
  const int L = blockIdx.x;
  const int kq = L >> 3;
  const int chunk = (kq / n_views) * 8 + (L & 7);
  const int v = kq % n_views;
  if (chunk >= n_chunks) return;
  for (int i = threadIdx.x; i < kBuckets / 2; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  unsigned key_of[kPreChunk / kBlock];
  Rect rect_of[kPreChunk / kBlock];
  // the view's camera (wave-uniform: scalar registers) is read ONCE per workgroup: read where it is used, every one of a
  // thread's four Gaussians re-fetched it in four to five dependent scalar round trips (view index -> row -> fields)
  float vm[16], pm[16], cam_tanx, cam_tany, cam_fx, cam_fy;
  {
    const Camera& cam = cams[view_sel ? view_sel[v] : v];
#pragma unroll
    for (int i = 0; i < 16; ++i) { vm[i] = cam.view[i]; pm[i] = cam.proj[i]; }
    cam_tanx = cam.tanfovx; cam_tany = cam.tanfovy; cam_fx = cam.focal_x; cam_fy = cam.focal_y;
  }
#pragma unroll
  for (int it = 0; it < kPreChunk / kBlock; ++it) {
  const int idx = chunk * kPreChunk + it * kBlock + threadIdx.x;
  key_of[it] = 0xFFFFFFFFu;
  rect_of[it] = Rect{0, 0, 0, 0};
  if (idx >= P) continue;
  const long o = (long)v * P + idx;
  const long gi = (long)(v / vps) * P + idx;      // Gaussian sets: view v renders set v / views_per_set
  unsigned key = 0xFFFFFFFFu;
  int my_radii = 0;
  unsigned touched = 0;

  const long mi = (long)(v / vps) * means_stride + idx;      // sets share one set of means when means_stride == 0
  const float px = means3D[3 * mi], py = means3D[3 * mi + 1], pz = means3D[3 * mi + 2];
  // transformPoint4x3 (auxiliary.h:58-66)
  const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
  const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
  const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
  if (vz > 0.2f) {                                            // auxiliary.h:154
    const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
    const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
    const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;

    // computeCov2D's Jacobian rows first (forward.cu:83-98): they are needed anyway and give a
    // conservative screen-space radius, so a Gaussian whose rect is certainly empty leaves before
    // the covariance / conic / double-precision part (~2/3 of the work).  Exactness: the reference
    // gives such a Gaussian radii = 0, tiles_touched = 0 and no key (forward.cu:236-238), which is
    // what the early exit leaves; the bound only ever over-estimates the radius:
    //   lambda_max(A Sigma A^T + 0.3 I) <= 0.3 + |A|_F^2 |Sigma|_2,  Sigma = R diag(s^2) R^T,
    //   R = (1 - |q|^2) I + |q|^2 Rot(q/|q|)  =>  |R|_2 <= |1 - |q|^2| + |q|^2.
    const float limx = 1.3f * cam_tanx, limy = 1.3f * cam_tany;
    const float txtz = vx / vz, tytz = vy / vz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    const float j00 = cam_fx / vz, j02 = -(cam_fx * tx) / (vz * vz);
    const float j11 = cam_fy / vz, j12 = -(cam_fy * ty) / (vz * vz);
    float A[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r0 = vm[4 * c + 0], r1 = vm[4 * c + 1], r2 = vm[4 * c + 2];
      A[0][c] = j00 * r0 + j02 * r2;
      A[1][c] = j11 * r1 + j12 * r2;
    }
    bool surely_empty = false;
    if (!cov3D_precomp) {
      const float s0 = scale_modifier * scales[3 * gi], s1 = scale_modifier * scales[3 * gi + 1],
                  s2 = scale_modifier * scales[3 * gi + 2];
      const float smax = fmaxf(fabsf(s0), fmaxf(fabsf(s1), fabsf(s2)));
      const float qr = rotations[4 * gi], qx = rotations[4 * gi + 1], qy = rotations[4 * gi + 2],
                  qz = rotations[4 * gi + 3];
      const float qq = qr * qr + qx * qx + qy * qy + qz * qz;
      const float rn = (fabsf(1.f - qq) + qq) * smax;
      const float af = A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][0] * A[1][0] +
                       A[1][1] * A[1][1] + A[1][2] * A[1][2];
      const float rb = 3.f * sqrtf(0.3f + af * rn * rn) * 1.001f + 3.f;      // >= the reference's radius + 2 px
      const float fxp = ((projx + 1.f) * (float)W - 1.f) * 0.5f, fyp = ((projy + 1.f) * (float)H - 1.f) * 0.5f;
      // rect empty <=> x1 <= x0 or y1 <= y0 (auxiliary.h:46-56); NaNs compare false and take the full path
      surely_empty = (fxp + rb < 0.f) || (fxp - rb > (float)(kTileX * gx) + 1.f) || (fyp + rb < 0.f) ||
                     (fyp - rb > (float)(kTileY * gy) + 1.f);
    }
    if (!surely_empty) {
    float c3[6];
    if (cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c3[i] = cov3D_precomp[6 * gi + i];
    } else {
      // computeCov3D (forward.cu:118-152): Sigma = R diag(s^2) R^T, quaternion not normalised
      const float sx = scale_modifier * scales[3 * gi], sy = scale_modifier * scales[3 * gi + 1],
                  sz = scale_modifier * scales[3 * gi + 2];
      const float r = rotations[4 * gi], x = rotations[4 * gi + 1], y = rotations[4 * gi + 2],
                  z = rotations[4 * gi + 3];
      const float R[3][3] = {
          {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
          {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
          {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
      float M[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        M[0][i] = sx * R[i][0];
        M[1][i] = sy * R[i][1];
        M[2][i] = sz * R[i][2];
      }
      float S[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
      c3[0] = S[0][0]; c3[1] = S[0][1]; c3[2] = S[0][2];
      c3[3] = S[1][1]; c3[4] = S[1][2]; c3[5] = S[2][2];
    }

    // computeCov2D (forward.cu:74-113), A from above
    const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float Bm[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) Bm[i][c] = A[i][0] * V[0][c] + A[i][1] * V[1][c] + A[i][2] * V[2][c];
    const float cov_x = (Bm[0][0] * A[0][0] + Bm[0][1] * A[0][1] + Bm[0][2] * A[0][2]) + 0.3f;
    const float cov_y = Bm[0][0] * A[1][0] + Bm[0][1] * A[1][1] + Bm[0][2] * A[1][2];
    const float cov_z = (Bm[1][0] * A[1][0] + Bm[1][1] * A[1][1] + Bm[1][2] * A[1][2]) + 0.3f;

    const float det = cov_x * cov_z - cov_y * cov_y;
    if (det != 0.0f) {
      const float det_inv = 1.f / det;
      const float con_x = cov_z * det_inv, con_y = -cov_y * det_inv, con_z = cov_x * det_inv;
      const float mid = 0.5f * (cov_x + cov_z);
      const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
      const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      const float pixx = ndc2pix(projx, W), pixy = ndc2pix(projy, H);
      const int rad = (int)my_radius;
      // getRect (auxiliary.h:46-56)
      const int x0 = min(gx, max(0, (int)((pixx - (float)rad) / (float)kTileX)));
      const int y0 = min(gy, max(0, (int)((pixy - (float)rad) / (float)kTileY)));
      const int x1 = min(gx, max(0, (int)((pixx + (float)rad + (float)(kTileX - 1)) / (float)kTileX)));
      const int y1 = min(gy, max(0, (int)((pixy + (float)rad + (float)(kTileY - 1)) / (float)kTileY)));
      if ((x1 - x0) * (y1 - y0) != 0) {
        my_radii = rad;
        touched = (unsigned)((y1 - y0) * (x1 - x0));      // (outputs: the reference's radius and tile count)
        Rect rc;
        rc.x0 = (unsigned short)x0; rc.y0 = (unsigned short)y0;
        rc.x1 = (unsigned short)x1; rc.y1 = (unsigned short)y1;
        // what the blend lists the record under: the rect tightened by the OPACITY (raster_common.h, round 6) — a tile
        // the alpha >= 1/255 ellipse does not reach holds no pixel the reference blends it at (forward.cu:331-333), and a
        // Gaussian under 1/255 everywhere is not listed at all (a trained OcRF's free space: most of the grid)
        // (inference forward only: with the contributor index — the training forward and the backward that walks the same
        // lists — a tile's list is the reference's, record for record)
        const float op = opacities[gi];
        if (tighten) {
          if (op < 1.0f / 255.0f) rc = Rect{0, 0, 0, 0};
          else tighten_rect(op, cov_x, cov_z, pixx, pixy, gx, gy, &rc);
        }
        if (((int)rc.x1 - (int)rc.x0) * ((int)rc.y1 - (int)rc.y0) != 0) {
          key = __float_as_uint(vz);        // vz > 0.2: the raw bits order like the value
          rects[o] = rc;
          rect_of[it] = rc;
          xy[o] = make_float2(pixx, pixy);
          conic_o[o] = make_float4(con_x, con_y, con_z, op);
        }
      }
    }
    }   // !surely_empty
  }
  if (key != 0xFFFFFFFFu) {
    const int bk = bucket_of(key);
    atomicAdd(&s_hist[bk >> 1], (bk & 1) ? 0x10000u : 1u);
  }
  key_of[it] = key;
  radii[o] = my_radii;
  if (tiles_touched) tiles_touched[o] = touched;
  }   // chunk loop
  // compact (depth bits, id, rect) records of the visible Gaussians: one returning global atomic
  // per workgroup reserves the range; the order is arbitrary (the scatter re-buckets, the blend sorts)
  int mine = 0;
#pragma unroll
  for (int it = 0; it < kPreChunk / kBlock; ++it) mine += key_of[it] != 0xFFFFFFFFu;
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  int inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o2 = __shfl_up(inc, off);
    if (lane >= off) inc += o2;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) before += s_wsum[w];
    total += s_wsum[w];
  }
  if (threadIdx.x == 0) s_base = total ? atomicAdd(&vis_count[v], total) : 0;
  __syncthreads();
  int pos = s_base + before + inc - mine;
#pragma unroll
  for (int it = 0; it < kPreChunk / kBlock; ++it) {
    if (key_of[it] == 0xFFFFFFFFu) continue;
    const unsigned idx = (unsigned)(chunk * kPreChunk + it * kBlock + threadIdx.x);
    const Rect rc = rect_of[it];
    vis_rec[(long)v * P + pos] = make_uint4(key_of[it], idx, (unsigned)rc.x0 | ((unsigned)rc.y0 << 16),
                                            (unsigned)rc.x1 | ((unsigned)rc.y1 << 16));
    ++pos;
  }
  for (int i = threadIdx.x; i < kBuckets / 2; i += kBlock) {
    const unsigned c = s_hist[i];
    if (c & 0xFFFFu) atomicAdd(&hist[v * kBuckets + 2 * i], (int)(c & 0xFFFFu));
    if (c >> 16) atomicAdd(&hist[v * kBuckets + 2 * i + 1], (int)(c >> 16));
  }
}

// ---------------------------------------------------------------------------------------------
// 2. exclusive scan of each view's histogram: starts[v][0..kBuckets] (last = visible count) and
//    the scatter cursors reset to the starts.  One workgroup per view, 32 buckets per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_bucket_scan_kernel(const int* __restrict__ hist,
                                                                    int* __restrict__ starts,
                                                                    int* __restrict__ cursor,
                                                                    int* __restrict__ status,
                                                                    const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  __shared__ int s_sum[kBlock];
  constexpr int per = kBuckets / kBlock;
  const int v = blockIdx.x, tid = threadIdx.x;
  if (status && v == 0 && tid == 0) *status = 0;      // the blend ORs its bits in afterwards: no memset launch
  const int* h = hist + (long)v * kBuckets + tid * per;
  int local[per];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < per; ++i) { local[i] = sum; sum += h[i]; }
  s_sum[tid] = sum;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {      // Hillis-Steele inclusive scan
    const int add = tid >= off ? s_sum[tid - off] : 0;
    __syncthreads();
    s_sum[tid] += add;
    __syncthreads();
  }
  const int base = s_sum[tid] - sum;
  int* st = starts + (long)v * (kBuckets + 1) + tid * per;
  int* cu = cursor + (long)v * kBuckets + tid * per;
#pragma unroll
  for (int i = 0; i < per; ++i) { st[i] = base + local[i]; cu[i] = base + local[i]; }
  if (tid == kBlock - 1) starts[(long)v * (kBuckets + 1) + kBuckets] = s_sum[tid];
}

// ---------------------------------------------------------------------------------------------
// 3. scatter the visible Gaussians into bucket order: b_rect, b_comp = (depth bits << 32) | id.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_scatter_kernel(
    int P, const uint4* __restrict__ vis_rec, const int* __restrict__ vis_count,
    int* __restrict__ cursor, Rect* __restrict__ b_rect, unsigned long long* __restrict__ b_comp,
    const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  // Walks the compact visible records of one view (the launch is sized for P; workgroups past the
  // view's count retire at once): count in LDS, reserve one global range per non-empty bucket (ONE
  // returning global atomic per bucket per workgroup), hand out slots with LDS atomics.
  __shared__ int s_cnt[kBuckets];
  const int v = blockIdx.y, tid = threadIdx.x;
  const int n = vis_count[v];
  if ((int)blockIdx.x * kChunk >= n) return;
  for (int i = tid; i < kBuckets; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  uint4 rec[kChunk / kBlock];
#pragma unroll
  for (int it = 0; it < kChunk / kBlock; ++it) {      // the thread's records in ONE round trip (clamped addresses)
    const int i = blockIdx.x * kChunk + it * kBlock + tid;
    rec[it] = vis_rec[(long)v * P + min(i, n - 1)];
  }
#pragma unroll
  for (int it = 0; it < kChunk / kBlock; ++it) {
    const int i = blockIdx.x * kChunk + it * kBlock + tid;
    if (i >= n) rec[it] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    if (rec[it].x != 0xFFFFFFFFu) atomicAdd(&s_cnt[bucket_of(rec[it].x)], 1);
  }
  __syncthreads();
  // all of a thread's returning atomics are issued before any result is consumed: their latency
  // (~2 us each) overlaps instead of adding up
  int cnt[kBuckets / kBlock], slot0[kBuckets / kBlock];
#pragma unroll
  for (int k = 0; k < kBuckets / kBlock; ++k) cnt[k] = s_cnt[k * kBlock + tid];
#pragma unroll
  for (int k = 0; k < kBuckets / kBlock; ++k)
    slot0[k] = cnt[k] ? atomicAdd(&cursor[v * kBuckets + k * kBlock + tid], cnt[k]) : 0;
#pragma unroll
  for (int k = 0; k < kBuckets / kBlock; ++k) s_cnt[k * kBlock + tid] = slot0[k];      // next free slot of the bucket
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kChunk / kBlock; ++it) {
    if (rec[it].x == 0xFFFFFFFFu) continue;
    const int slot = atomicAdd(&s_cnt[bucket_of(rec[it].x)], 1);
    Rect rc;
    rc.x0 = (unsigned short)(rec[it].z & 0xFFFFu); rc.y0 = (unsigned short)(rec[it].z >> 16);
    rc.x1 = (unsigned short)(rec[it].w & 0xFFFFu); rc.y1 = (unsigned short)(rec[it].w >> 16);
    b_rect[(long)v * P + slot] = rc;
    b_comp[(long)v * P + slot] = ((unsigned long long)rec[it].x << 32) | rec[it].y;
  }
}

// ---------------------------------------------------------------------------------------------
// 4. blend (forward.cu:261-374 + w-depth README:5-11).
// Dynamic LDS: u64 rec[kCapRec]; float4 l_a[kStage], l_b[kStage], l_c[kStage].
// ---------------------------------------------------------------------------------------------
// Bitonic sort of 256*R u64 records held in LDS, R consecutive records per thread in registers:
// exchange distances below R are register moves, below 64*R wave shuffles, and only the last
// log2(4)=2 distances of the last merges go through LDS with a workgroup barrier.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long x, int lane_mask) {
  const unsigned lo = __shfl_xor((unsigned)(x & 0xFFFFFFFFull), lane_mask);
  const unsigned hi = __shfl_xor((unsigned)(x >> 32), lane_mask);
  return ((unsigned long long)hi << 32) | lo;
}

template <int R>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long* rec, int tid) {
  constexpr int n = kBlock * R;
  const int lane = tid & 63;
  unsigned long long x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = rec[tid * R + r];
#pragma unroll
  for (int k = 2; k <= n; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & j) == 0) {
            const bool up = (((tid * R + r) & k) == 0);
            const unsigned long long a = x[r], b = x[r | j];
            const bool sw = (a > b) == up;
            x[r] = sw ? b : a;
            x[r | j] = sw ? a : b;
          }
        }
      } else if (j < 64 * R) {
        const int lm = j / R;
        const bool lower = (lane & lm) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned long long y = shfl_xor_u64(x[r], lm);
          const bool up = (((tid * R + r) & k) == 0);
          const bool take_min = lower == up;
          const unsigned long long mn = x[r] < y ? x[r] : y, mx = x[r] < y ? y : x[r];
          x[r] = take_min ? mn : mx;
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rec[tid * R + r] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = tid * R + r;
          const unsigned long long y = rec[i ^ j];
          const bool up = ((i & k) == 0);
          const bool take_min = ((i & j) == 0) == up;
          const unsigned long long mn = x[r] < y ? x[r] : y, mx = x[r] < y ? y : x[r];
          x[r] = take_min ? mn : mx;
        }
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rec[tid * R + r] = x[r];
  __syncthreads();
}

// generic fallback for more than 1024 records (rare: only after carries pile up)
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* rec, int n_pow2, int tid) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n_pow2 >> 1); t += kBlock) {     // one compare-exchange per thread
        const int i = 2 * t - (t & (j - 1));                  // lower index of the t-th pair
        const int ixj = i + j;
        const unsigned long long a = rec[i], b = rec[ixj];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { rec[i] = b; rec[ixj] = a; }
      }
      __syncthreads();
    }
  }
}

// DPP lane permutations inside a 16-lane row (VALU-rate, no LDS): used by the backward reduction.
constexpr int kDppQuadXor1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i within 8
constexpr int kDppRowMirror = 0x140;   // row_mirror: lane i <-> 15-i
constexpr int kDppRowRor8 = 0x128;     // row_ror:8: lane i <- lane (i+8)%16
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// Backward mode (BWD): the same traversal, front to back, of exactly the same per-tile sequence;
// each pixel stops at the contributor index the forward recorded (n_contrib) and evaluates
//   dL/dalpha_i = sum_ch dL/dC_ch * (c_i T_i - (S_ch - prefix_ch(i)) / (1 - alpha_i)) - T_final/(1-alpha_i) * (bg . dL/dC)
// with S = C_out - T_final * bg the colour the forward accumulated — algebraically the reference's
// back-to-front recurrence (backward.cu:470-517) without needing the list reversed.  The nine
// per-Gaussian partial derivatives are summed over each 16-lane row with DPP, over the workgroup
// in LDS, and leave as nine float atomics per (tile, record).
struct BwdArgs {
  const float* dL_dcolor;      // (V,3,H,W)
  const float* fwd_color;      // (V,3,H,W) forward output
  const float* fwd_final_T;    // (V,H,W)
  const unsigned* fwd_n_contrib;
  float* acc;                  // (V,P,9): mean2D x,y | conic x,y,z | opacity | colour r,g,b
};

// One workgroup = one PAIR of vertically adjacent 16x16 tiles (tile rows 2*by and 2*by+1) of one
// view; thread (lx, ly) owns pixel (lx, ly) of both tiles — same x, y 16 apart.  The two pixels
// share dx and the dx-only part of the exponent and run as the two halves of packed fp32 ops; the
// per-record LDS reads, the scan and the sort are shared by both tiles.  Each tile still sees
// exactly its own list: a record carries two coverage bits (rect covers tile A / tile B), a pixel
// ignores records that do not cover its tile, and the contributor index of each tile is a scalar
// counter of its covered records.
// (forward: five waves per SIMD asked of the compiler — 28 KB of LDS hold five workgroups per CU; left alone hipcc takes
// 110 VGPRs for the staging and the record loops keep everything in registers at 96 too)
template <bool STAMP, bool BWD, bool MEDIAN, bool CONTRIB = true>
__global__ __launch_bounds__(kBlock, (BWD || STAMP) ? 1 : (CONTRIB ? 5 : 4)) void raster_blend_kernel(
    unsigned long long* __restrict__ stamps,
    int P, int W, int H, int gy, const int* __restrict__ starts, const Rect* __restrict__ rects,
    const Rect* __restrict__ b_rect, const unsigned long long* __restrict__ b_comp,
    const float2* __restrict__ xy, const float4* __restrict__ conic_o,
    const float* __restrict__ colors, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_final_T,
    unsigned* __restrict__ out_n_contrib, int* __restrict__ status, BwdArgs bw, int vps,
    int* __restrict__ gate) {
  if (gate) {
    if (*gate == 0) return;
    // Armed AND fired (raster_plan.hip): this is the last kernel that reads the flag, and every workgroup gets here
    // having read it — the last arriver lowers it for the next call (no memset launch per call; the arrival counter
    // gate[1] is zeroed by the plan's update kernel).  The barrier keeps this workgroup's own reads ahead of its arrival.
    __syncthreads();
    if (threadIdx.x == 0) {
      const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
      if (atomicAdd(gate + 1, 1) == total - 1) gate[0] = 0;
    }
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long rec[];
  float4* l_a = reinterpret_cast<float4*>(rec + kCapRec);          // x, y, conic.x, conic.y
  float4* l_b = l_a + kStage;                                      // BWD: conic.z, opacity, depth, r; fwd: see staging
  float4* l_c = l_b + kStage;                                      // BWD: g, b, coverage bits, -
  unsigned* l_id = reinterpret_cast<unsigned*>(l_c + kStage);     // BWD only
  float* l_g = reinterpret_cast<float*>(l_id + kStage);           // BWD only: [kStage][9]
  __shared__ int l_wtot[kScanUnroll * (kBlock / 64)];
  __shared__ int l_ready;
  __shared__ int l_fmax, l_generic;      // forward: largest need factor (float bits) / a GENERIC record in the staged batch
  // LISTS — the inference forward (no contributor index): a batch is staged and blended by raster_blend_body.h, exactly
  // as the planned kernel does it: wave w owns the 16 x 8 pixel block of rows [8 w, 8 w + 8) of the tile pair, every wave
  // walks its OWN list of the records whose alpha >= 1/255 ellipse reaches that block (round 5 walked the whole staged
  // batch per wave and skipped by a per-record test on the evaluated exponent: 162 us against the planned blend's 128 for
  // the same views of cfg2).  The backward and the training forward (CONTRIB) keep thread (lx, ly) = pixel (lx, ly) of
  // both tiles: the backward's DPP row reductions and contributor counters are laid out for it.
  constexpr bool LISTS = !BWD && !CONTRIB;
  __shared__ __attribute__((aligned(16))) unsigned short l_list[LISTS ? 4 : 1][rbody::kListLen];
  __shared__ int l_lcnt[rbody::kSrcWavesB][4];
  __shared__ int l_generic4[4], l_fmax4[4];

  const int tid = threadIdx.x;
  const int wave = tid / 64, lane = tid % 64;
  const int tx = blockIdx.x, v = blockIdx.z;
  const int tyA = 2 * blockIdx.y, tyB = tyA + 1;      // tyB == gy: the pair has no second tile
  const int lx = LISTS ? (lane & 15) : tid % kTileX, ly = tid / kTileX;
  const int pxi = tx * kTileX + lx;
  // LISTS: the lane's two pixels are 4 rows apart inside the wave's block (rows 8 w + r and 8 w + r + 4 of the pair)
  const int pyA = LISTS ? tyA * kTileY + 8 * wave + (lane >> 4) : tyA * kTileY + ly;
  const int pyB = LISTS ? pyA + 4 : tyB * kTileY + ly;
  const bool tile_ok = (tyA + (wave >> 1)) < gy;       // (LISTS) the wave's tile exists
  const bool insideA = LISTS ? (tile_ok && pxi < W && pyA < H) : (pxi < W && pyA < H);
  const bool insideB = LISTS ? (tile_ok && pxi < W && pyB < H) : (tyB < gy && pxi < W && pyB < H);
  const float pixf_x = (float)pxi;
  const f2 pixf_y = f2{(float)pyA, (float)pyB};
  const long base = (long)v * P;
  const int nv = starts[(long)v * (kBuckets + 1) + kBuckets];

  bool doneA = !insideA, doneB = !insideB;      // backward only; the forward keeps "stopped" in the sign of T
  f2 T = splat(1.0f);
  if constexpr (!BWD) {
    // forward: T < 0 <=> the pixel has stopped (or lies outside the image); |T| is its final transmittance
    T.x = insideA ? 1.0f : -1.0f;
    T.y = insideB ? 1.0f : -1.0f;
  }
  int jA = 0, jB = 0;                                  // wave-uniform: covered records so far
  unsigned lastA = 0, lastB = 0;
  f2 C0 = splat(0.f), C1 = splat(0.f), C2 = splat(0.f);
  f2 D = splat(MEDIAN ? 15.0f : 0.0f);
  // forward: the two pixels' running state and the loop constants of raster_blend_math.h
  rb::Px px;
  px.T = T; px.C0 = C0; px.C1 = C1; px.C2 = C2; px.D = D; px.cnt = splat(0.f);
  rb::Consts kc = rb::consts();
  if constexpr (!BWD) asm volatile("" : "+v"(kc.neg_k255), "+v"(kc.neg_half));      // kept in VGPR pairs
  // backward-only per-pixel state
  f2 dL0 = splat(0.f), dL1 = splat(0.f), dL2 = splat(0.f), S0 = splat(0.f), S1 = splat(0.f), S2 = splat(0.f);
  f2 Tfin = splat(0.f), bgdot = splat(0.f);
  int n_lastA = 0, n_lastB = 0;
  if constexpr (BWD) {
    const long npix = (long)W * H;
    struct PixelBwd { float d0, d1, d2, s0, s1, s2, tf, bd; int nl; };
    auto load = [&](bool inside, int py) {
      PixelBwd q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
      if (!inside) return q;
      const long pix = (long)py * W + pxi;
      q.d0 = bw.dL_dcolor[(v * 3 + 0) * npix + pix];
      q.d1 = bw.dL_dcolor[(v * 3 + 1) * npix + pix];
      q.d2 = bw.dL_dcolor[(v * 3 + 2) * npix + pix];
      q.tf = bw.fwd_final_T[v * npix + pix];
      q.s0 = bw.fwd_color[(v * 3 + 0) * npix + pix] - q.tf * bg[0];
      q.s1 = bw.fwd_color[(v * 3 + 1) * npix + pix] - q.tf * bg[1];
      q.s2 = bw.fwd_color[(v * 3 + 2) * npix + pix] - q.tf * bg[2];
      q.bd = bg[0] * q.d0 + bg[1] * q.d1 + bg[2] * q.d2;
      q.nl = (int)bw.fwd_n_contrib[v * npix + pix];
      return q;
    };
    const PixelBwd qa = load(insideA, pyA), qb = load(insideB, pyB);
    dL0 = f2{qa.d0, qb.d0}; dL1 = f2{qa.d1, qb.d1}; dL2 = f2{qa.d2, qb.d2};
    S0 = f2{qa.s0, qb.s0}; S1 = f2{qa.s1, qb.s1}; S2 = f2{qa.s2, qb.s2};
    Tfin = f2{qa.tf, qb.tf}; bgdot = f2{qa.bd, qb.bd};
    n_lastA = qa.nl; n_lastB = qb.nl;
    doneA = doneA || (n_lastA == 0);
    doneB = doneB || (n_lastB == 0);
  }

  bool all_done = false;
  rbody::Tile tile;
  tile.tyA = tyA; tile.wave = wave; tile.lane = lane; tile.tid = tid;
  tile.pixf_x = pixf_x; tile.pixf_y = pixf_y; tile.inside0 = insideA; tile.inside1 = insideB;
  tile.bx0 = (float)(tx * kTileX); tile.bx1 = tile.bx0 + 15.f;
  const rbody::Lds lds{l_a, l_b, l_c, l_list, l_lcnt, l_generic4, l_fmax4};
  rbody::NoStats no_stats;
  if constexpr (LISTS) {
    if (tid == 0) {      // slot kStageB: a record that changes nothing, pads odd list lengths (read after the scan's barriers)
      const rb::Staged st = rb::stage_noop();
      l_a[rbody::kStageB] = st.a;
      l_b[rbody::kStageB] = st.b;
      l_c[rbody::kStageB] = st.c;
    }
  }
  // blend rec[0, n) (sorted) front to back, kStage at a time; sets all_done
  auto blend_records = [&](int n) {
    if constexpr (LISTS) {
      for (int s0 = 0; s0 < n && !all_done; s0 += rbody::kStageB) {
        const int ns = min(rbody::kStageB, n - s0);
        auto fetch = [&](int ri) {
          const unsigned long long c = rec[s0 + ri];
          const unsigned id = (unsigned)(c & 0xFFFFFFFFull);
          const float4 co = conic_o[base + id];
          const Rect rc = rects[base + id];
          rbody::Fetched fr;
          fr.con = make_float4(-0.5f * co.x, -0.5f * co.z, co.y, co.w);
          fr.pix = xy[base + id];
          fr.col = colors + 3 * ((long)(v / vps) * P + id);     // the view's Gaussian set
          fr.depth = __uint_as_float((unsigned)(c >> 32));
          fr.covA = tyA >= rc.y0 && tyA < rc.y1;
          fr.covB = tyB >= rc.y0 && tyB < rc.y1;
          return fr;
        };
        all_done = rbody::blend_batch<MEDIAN, false>(px, kc, tile, lds, ns, fetch, 0, no_stats);
      }
      return;
    }
    for (int s0 = 0; s0 < n && !all_done; s0 += kStage) {
      const int ns = min(kStage, n - s0);
      if constexpr (!BWD) {
        if (tid == 0) {
          l_fmax = 0;
          l_generic = 0;
        }
        __syncthreads();
      }
      if (tid < ns) {
        const unsigned long long c = rec[s0 + tid];
        const unsigned id = (unsigned)(c & 0xFFFFFFFFull);
        const float2 p = xy[base + id];
        const float4 co = conic_o[base + id];
        const Rect rc = rects[base + id];
        const unsigned cov = ((tyA >= rc.y0 && tyA < rc.y1) ? 1u : 0u) | ((tyB >= rc.y0 && tyB < rc.y1) ? 2u : 0u);
        const float* col = colors + 3 * ((long)(v / vps) * P + id);     // the view's Gaussian set
        if constexpr (BWD) {
          l_a[tid] = make_float4(p.x, p.y, co.x, co.y);
          l_b[tid] = make_float4(co.z, co.w, __uint_as_float((unsigned)(c >> 32)), col[0]);
          l_c[tid] = make_float4(col[1], col[2], __uint_as_float(cov), 0.f);
          l_id[tid] = id;
#pragma unroll
          for (int k = 0; k < 9; ++k) l_g[tid * 9 + k] = 0.f;
        } else {
          // forward: the staged form of raster_blend_math.h; the record's tile coverage is in its per-pixel constant
          // term — pixel A (tile A) / pixel B (tile B) see L = log2(opacity) (GENERIC: the opacity) only if the record
          // covers their tile, else -inf (0): alpha 0 there, below the 1/255 cut — the loop needs no coverage test
          const float4 con = make_float4(-0.5f * co.x, -0.5f * co.z, co.y, co.w);
          const bool simple = rb::is_simple(con);
          const rb::Staged st = rb::stage(con, p.x, p.y, col[0], col[1], col[2], __uint_as_float((unsigned)(c >> 32)), simple);
          const float none = simple ? -INFINITY : 0.f;
          l_a[tid] = st.a;
          l_b[tid] = st.b;
          // slot w: coverage bits | SIMPLE flag (bit 2)
          l_c[tid] = make_float4((cov & 1u) ? st.c.x : none, (cov & 2u) ? st.c.x : none, st.c.z,
                                 __uint_as_float(cov | (simple ? 4u : 0u)));
          if (!simple) l_generic = 1;
          // positive floats order like their bits; a NaN factor (NaN opacity) wins: "may stop" throughout
          if (cov != 0u) atomicMax(&l_fmax, __float_as_int(st.c.y));
        }
      } else if (!BWD && tid < ((ns + 3) & ~3)) {
        // pad the batch to a multiple of four records with no-ops (SIMPLE, alpha 0): the loop reads two per trip
        l_a[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l_b[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l_c[tid] = make_float4(-INFINITY, -INFINITY, 0.f, __uint_as_float(4u));
      }
      __syncthreads();
      if constexpr (!BWD) {
        // Forward: the per-record arithmetic of raster_blend_math.h (shared with the planned kernel: same bits for the
        // same record sequence).  Two records per trip; a record whose alpha is under 1/255 at every pixel of the WAVE is
        // skipped as a whole (about a third of a tile pair's records lie in the corners of their 3-sigma square or
        // beyond the wave's rows); NOSTOP trips: while every pixel of the wave inside the image has T above
        // `need` (from the batch's largest opacity: rb::no_stop_need) no record can trip the stop test — no compare,
        // no selects.  The condition only ever turns false: one loop per phase.
        const int ns4 = (ns + 3) & ~3;
        constexpr int kTrip = 2;
        const float fmax_b = __int_as_float(l_fmax);
        const float need = rb::no_stop_need(fmax_b, fmax_b);
        const bool generic_batch = l_generic != 0;
        auto trip = [&](auto nostop_tag, auto generic_tag, int j0) __attribute__((always_inline)) {
          constexpr bool NOSTOP = decltype(nostop_tag)::value, GEN = decltype(generic_tag)::value;
          float4 ra[kTrip], rb4[kTrip], rc4[kTrip];
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            ra[u] = l_a[j0 + u];
            rb4[u] = l_b[j0 + u];
            rc4[u] = l_c[j0 + u];
          }
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            const float4 a = ra[u];
            const float4 b = rb4[u];
            const unsigned bits = __builtin_amdgcn_readfirstlane(__float_as_uint(rc4[u].w));
            const float dx = a.w - pixf_x;
            const f2 dy = splat(a.x) - pixf_y;
            f2 alpha, s;
            bool skipped = false;
            auto fast = [&]() {
              const float t = b.z * dx;
              const float nb = b.w * dx;
              const f2 qxl = f2{__builtin_fmaf(t, dx, rc4[u].x), __builtin_fmaf(t, dx, rc4[u].y)};
              const f2 p2 = rb::p2_simple(nb, a.y, qxl, dy);
              if (__ballot(!((p2.x <= rb::kSkipP2) & (p2.y <= rb::kSkipP2))) == 0ull) {
                skipped = true;                     // alpha < 1/255 at every pixel of the wave: nothing changes
                return;
              }
              rb::alpha_of_p2(p2, kc, &alpha, &s);
            };
            if constexpr (GEN) {
              if (bits & 4u) fast();
              else rb::alpha_generic(dx, b.z, b.w, a.y, f2{rc4[u].x, rc4[u].y}, dy, &alpha, &s);
            } else {
              fast();
            }
            if constexpr (CONTRIB) {
              jA += (int)(bits & 1u);
              jB += (int)((bits >> 1) & 1u);
            }
            if (skipped) {
              if constexpr (MEDIAN) rb::count_above_half(px, kc);
              continue;
            }
            const f2 wgt = rb::chain<MEDIAN, !MEDIAN, NOSTOP>(px, alpha, s, a.z, b.x, b.y, rc4[u].z, kc);
            if constexpr (CONTRIB) {
              lastA = (wgt.x > 0.f) ? (unsigned)jA : lastA;       // alpha T > 0 <=> this Gaussian was blended
              lastB = (wgt.y > 0.f) ? (unsigned)jB : lastB;
            }
          }
        };
        const unsigned long long inA = __ballot(insideA), inB = __ballot(insideB);
        auto may_stop = [&]() {
          return ((__ballot(!(px.T.x > need)) & inA) | (__ballot(!(px.T.y > need)) & inB)) != 0ull;
        };
        // live <=> sign bit of T clear
        auto any_alive = [&]() { return __ballot((__float_as_int(px.T.x) & __float_as_int(px.T.y)) >= 0) != 0ull; };
        int j0 = 0;
        if (!generic_batch) {
          for (; j0 < ns4; j0 += kTrip) {
            if (may_stop()) break;
            trip(std::true_type{}, std::false_type{}, j0);
          }
          for (; j0 < ns4; j0 += kTrip) {
            if (!any_alive()) break;
            trip(std::false_type{}, std::false_type{}, j0);
          }
        } else {
          for (; j0 < ns4; j0 += kTrip) {
            if (!any_alive()) break;
            trip(std::false_type{}, std::true_type{}, j0);
          }
        }
        if constexpr (MEDIAN) {
          // the record at which a pixel crossed 0.5 in this batch, if it did (raster_blend_math.h)
          const int mA = rb::median_index(px.cnt.x, px.T.x), mB = rb::median_index(px.cnt.y, px.T.y);
          if (mA >= 0) px.D.x = l_c[mA].z;
          if (mB >= 0) px.D.y = l_c[mB].z;
          px.cnt = splat(0.f);
        }
        doneA = rb::dead(px.T.x);
        doneB = rb::dead(px.T.y);
      } else {
      // Branch-free per-record update so that the LDS reads of the next records can be issued
      // ahead (4 records per trip); a wave leaves the batch as soon as its 128 pixels are done.
      for (int j0 = 0; j0 < ns; j0 += 4) {
        if (__ballot(!(doneA && doneB)) == 0ull) break;
        float4 ra[4], rb[4], rc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = min(j0 + u, ns - 1);
          ra[u] = l_a[j];
          rb[u] = l_b[j];
          rc4[u] = l_c[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 a = ra[u];
          const float4 b = rb[u];
          const float cg = rc4[u].x, cb = rc4[u].y;
          const unsigned cov = __builtin_amdgcn_readfirstlane(__float_as_uint(rc4[u].z));
          const bool in = (j0 + u) < ns;
          const bool covA = in && (cov & 1u), covB = in && (cov & 2u);
          jA += covA ? 1 : 0;
          jB += covB ? 1 : 0;
          const bool liveA = covA && !doneA, liveB = covB && !doneB;
          // power = -0.5 (cxx dx dx + czz dy dy) - cxy dx dy, in the reference's order (forward.cu:320-323)
          const float dx = a.x - pixf_x;
          const float qx = (a.z * dx) * dx;
          const float bx = a.w * dx;
          const f2 dy = splat(a.y) - pixf_y;
          const f2 qy = (splat(b.x) * dy) * dy;
          f2 power;
          if constexpr (BWD) power = splat(-0.5f) * (splat(qx) + qy) - splat(bx) * dy;
          else power = (splat(qx) + qy) - splat(bx) * dy;          // qx, qy carry the -0.5 (staging above)
          // __expf(x) is v_exp_f32(log2(e) x): the two multiplies go as one packed op
          const f2 p2 = power * splat(1.44269504088896340736f);
          f2 G;
          G.x = __builtin_amdgcn_exp2f(p2.x);
          G.y = __builtin_amdgcn_exp2f(p2.y);
          f2 alpha = splat(b.y) * G;
          alpha.x = fminf(0.99f, alpha.x);
          alpha.y = fminf(0.99f, alpha.y);
          const bool validA = liveA && !(power.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
          const bool validB = liveB && !(power.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
          const f2 test_T = T * (splat(1.0f) - alpha);
          const f2 aT = alpha * T;
          if constexpr (BWD) {
            const bool contribA = validA && (jA <= n_lastA), contribB = validB && (jB <= n_lastB);
            f2 wgt;
            wgt.x = contribA ? aT.x : 0.f;
            wgt.y = contribB ? aT.y : 0.f;
            C0 = fma2(splat(b.w), wgt, C0);                // prefix including this record
            C1 = fma2(splat(cg), wgt, C1);
            C2 = fma2(splat(cb), wgt, C2);
            const f2 om = splat(1.0f) - alpha;             // alpha <= 0.99
            f2 inv;
            inv.x = __builtin_amdgcn_rcpf(om.x);
            inv.y = __builtin_amdgcn_rcpf(om.y);
            f2 dLda = dL0 * (splat(b.w) * T - (S0 - C0) * inv) + dL1 * (splat(cg) * T - (S1 - C1) * inv) +
                      dL2 * (splat(cb) * T - (S2 - C2) * inv) - Tfin * inv * bgdot;
            dLda.x = contribA ? dLda.x : 0.f;
            dLda.y = contribB ? dLda.y : 0.f;
            const f2 dG = splat(b.y) * dLda;
            const f2 gdx = G * splat(dx), gdy = G * dy;
            const f2 m0 = dG * (-gdx * splat(a.z) - gdy * splat(a.w));
            const f2 m1 = dG * (-gdy * splat(b.x) - gdx * splat(a.w));
            const f2 k2 = gdx * splat(dx) * dG, k3 = gdx * dy * dG, k4 = gdy * dy * dG;
            const f2 k5 = G * dLda;
            const f2 k6 = wgt * dL0, k7 = wgt * dL1, k8 = wgt * dL2;
            float g[9];
            g[0] = (m0.x + m0.y) * (0.5f * (float)W);
            g[1] = (m1.x + m1.y) * (0.5f * (float)H);
            g[2] = -0.5f * (k2.x + k2.y);
            g[3] = -0.5f * (k3.x + k3.y);
            g[4] = -0.5f * (k4.x + k4.y);
            g[5] = k5.x + k5.y;
            g[6] = k6.x + k6.y;
            g[7] = k7.x + k7.y;
            g[8] = k8.x + k8.y;
            if (__ballot(contribA || contribB) != 0ull) {
              // Transposed reduction on the VALU (DPP), no LDS traffic: each halving step keeps half
              // of the values per lane, so 8 values cost 4+2+1 pair-sums instead of 8 x 6; lane l
              // ends with value (l & 7) summed over its 16-lane row, and the four rows meet in LDS.
              float h[4], q[2];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float lo = g[i] + dpp_mov<kDppHalfMirror>(g[i]);
                const float hi = g[i + 4] + dpp_mov<kDppHalfMirror>(g[i + 4]);
                h[i] = (lane & 4) ? hi : lo;
              }
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const float lo = h[i] + dpp_mov<kDppQuadXor2>(h[i]);
                const float hi = h[i + 2] + dpp_mov<kDppQuadXor2>(h[i + 2]);
                q[i] = (lane & 2) ? hi : lo;
              }
              const float lo = q[0] + dpp_mov<kDppQuadXor1>(q[0]);
              const float hi = q[1] + dpp_mov<kDppQuadXor1>(q[1]);
              float r = (lane & 1) ? hi : lo;
              r += dpp_mov<kDppRowRor8>(r);
              float e = g[8];
              e += dpp_mov<kDppQuadXor1>(e);
              e += dpp_mov<kDppQuadXor2>(e);
              e += dpp_mov<kDppHalfMirror>(e);
              e += dpp_mov<kDppRowMirror>(e);
              const int l16 = lane & 15;
              if (l16 < 9) atomicAdd(&l_g[(j0 + u) * 9 + l16], l16 == 8 ? e : r);
            }
            T.x = contribA ? test_T.x : T.x;
            T.y = contribB ? test_T.y : T.y;
            doneA = doneA || (liveA && jA >= n_lastA);
            doneB = doneB || (liveB && jB >= n_lastB);
          }
        }
      }
      }
      // every pixel saturated -> stop (forward.cu:304-307)
      all_done = __syncthreads_count(doneA && doneB) == kBlock;
      if constexpr (BWD) {
        // Float atomics execute at the memory side as 64-B requests: the rate is set by the
        // number of requests, not of adds.  16 lanes per record (9 active) keep a record's 36
        // contiguous bytes in 1-2 requests; one lane per record would issue 9.
        for (int r0 = 0; r0 < ns; r0 += kBlock / 16) {
          const int r = r0 + tid / 16, k = tid % 16;
          if (r < ns && k < 9) {
            const float val = l_g[r * 9 + k];
            if (val != 0.f) atomicAdd(&bw.acc[(base + l_id[r]) * 9 + k], val);
          }
        }
        __syncthreads();
      }
    }
  };

  // does the record's tile rectangle cover one of the two tiles of this workgroup?
  auto covers = [&](const Rect rc) { return (tx >= rc.x0) && (tx < rc.x1) && (tyA < rc.y1) && (tyB >= rc.y0); };

  int scan = 0;      // next list entry to look at
  int nrec = 0;      // records held in LDS (sorted prefix left over from the previous round)
  unsigned long long t_prev = 0, t_acc[6] = {0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_prev;
      t_prev = t;
    }
  };
  stamp(-1);
  while (!all_done) {
    // ---- scan: keep the records whose rect covers this tile pair, until ~2 batches are held ----
    // stop once ~half a batch of new records is in: nrec then lands in (128, 384] on the first
    // round, so the sort pads to 256 or 512.  kScanUnroll batches of 256 rects are in flight at
    // once (the scan is latency-bound: one dependent global load per batch otherwise).
    const int target = min(nrec + kBlock / 2 + 1, kCapRec - kScanUnroll * kBlock);
    while (scan < nv && nrec < target) {
      // dense scenes fill a round from the first 256-512 entries: look at one batch at a time twice (four at
      // once after a sparse first batch overshoots into a 512-record sort of which ~120 records are consumed),
      // then four
      const int n_u = (scan < 2 * kBlock) ? 1 : kScanUnroll;
      bool hit[kScanUnroll];
#pragma unroll
      for (int u = 0; u < kScanUnroll; ++u) {
        const int i = scan + u * kBlock + tid;
        hit[u] = false;
        if (u < n_u && i < nv) hit[u] = covers(b_rect[base + i]);
      }
      int rank[kScanUnroll];
#pragma unroll
      for (int u = 0; u < kScanUnroll; ++u) {
        const unsigned long long m = __ballot(hit[u]);
        rank[u] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) l_wtot[u * (kBlock / 64) + wave] = __popcll(m);
      }
      __syncthreads();
      int off = nrec;
#pragma unroll
      for (int u = 0; u < kScanUnroll; ++u) {
        int mine = off;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
          const int c = l_wtot[u * (kBlock / 64) + w];
          if (w < wave) mine += c;
          off += c;
        }
        if (hit[u]) rec[mine + rank[u]] = b_comp[base + scan + u * kBlock + tid];
      }
      nrec = off;
      scan += n_u * kBlock;
      __syncthreads();
    }
    stamp(0);
    const bool at_end = scan >= nv;
    if (nrec == 0) {
      if (at_end) break;
      continue;
    }
    // ---- exact order of what is held: bitonic sort on (depth bits, id) ----
    int n2 = kBlock;
    while (n2 < nrec) n2 <<= 1;
    for (int i = nrec + tid; i < n2; i += kBlock) rec[i] = kPad;
    __syncthreads();
    if (n2 == kBlock) bitonic_sort_regs<1>(rec, tid);
    else if (n2 == 2 * kBlock) bitonic_sort_regs<2>(rec, tid);
    else if (n2 == 4 * kBlock) bitonic_sort_regs<4>(rec, tid);
    else bitonic_sort_lds(rec, n2, tid);
    stamp(1);
    // ---- records of buckets that cannot receive further entries are final ----
    int n_ready = nrec;
    if (!at_end) {
      // every unscanned entry lies in bucket >= bucket(entry scan-1): earlier buckets are complete
      const unsigned last_key = (unsigned)(b_comp[base + scan - 1] >> 32);
      const int b_last = bucket_of(last_key);
      if (tid == 0) l_ready = 0;
      __syncthreads();
      int cnt = 0;
      for (int i = tid; i < nrec; i += kBlock) cnt += bucket_of((unsigned)(rec[i] >> 32)) < b_last;
      for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
      if (lane == 0 && cnt) atomicAdd(&l_ready, cnt);
      __syncthreads();
      n_ready = l_ready;
      if (n_ready == 0) {
        if (nrec >= kCapRec - kScanUnroll * kBlock) {
          // ONE depth bucket holds more tile hits than the LDS can sort at once.  Exact streaming
          // selection over that bucket's range: keep the K smallest (depth bits, id) composites
          // above `last`, blend them, repeat from `last` until the bucket is exhausted.
          constexpr int K = kCapRec - kBlock;
          const int rs = starts[(long)v * (kBuckets + 1) + b_last];
          const int re = starts[(long)v * (kBuckets + 1) + b_last + 1];
          unsigned long long last = 0ull;
          if (tid == 0) atomicOr(status, 2);          // informational: the slow exact path ran
          while (!all_done) {
            int cnt = 0;
            unsigned long long thr = kPad;
            bool full = false;
            for (int p0 = rs; p0 < re; p0 += kBlock) {
              const int i = p0 + tid;
              unsigned long long comp = 0ull;
              bool take = false;
              if (i < re) {
                if (covers(b_rect[base + i])) {
                  comp = b_comp[base + i];
                  take = comp > last && comp < thr;
                }
              }
              const unsigned long long m = __ballot(take);
              const int rank = __popcll(m & ((1ull << lane) - 1ull));
              if (lane == 0) l_wtot[wave] = __popcll(m);
              __syncthreads();
              int off = cnt, tot = 0;
#pragma unroll
              for (int w = 0; w < kBlock / 64; ++w) {
                const int c = l_wtot[w];
                if (w < wave) off += c;
                tot += c;
              }
              if (take) rec[off + rank] = comp;
              cnt += tot;
              __syncthreads();
              if (cnt > K) {                           // merge: keep the K smallest seen so far
                for (int q = cnt + tid; q < kCapRec; q += kBlock) rec[q] = kPad;
                __syncthreads();
                bitonic_sort_lds(rec, kCapRec, tid);
                cnt = K;
                thr = rec[K - 1];
                full = true;
                __syncthreads();
              }
            }
            if (cnt == 0) break;
            int m2 = kBlock;
            while (m2 < cnt) m2 <<= 1;
            for (int q = cnt + tid; q < m2; q += kBlock) rec[q] = kPad;
            __syncthreads();
            bitonic_sort_lds(rec, m2, tid);
            last = rec[cnt - 1];
            __syncthreads();
            blend_records(cnt);
            if (!full) break;                          // everything above `last` fitted: bucket done
          }
          scan = re;
          nrec = 0;
          if (all_done) break;
          continue;
        } else {
          continue;      // keep scanning: the only bucket held is still open
        }
      }
    }
    stamp(2);
    blend_records(n_ready);
    stamp(3);
    if (all_done || (at_end && n_ready == nrec)) break;
    // ---- carry the records of the open bucket to the front (still sorted) ----
    const int left = nrec - n_ready;
    for (int i0 = 0; i0 < left; i0 += kBlock) {
      unsigned long long c = 0;
      if (i0 + tid < left) c = rec[n_ready + i0 + tid];
      __syncthreads();
      if (i0 + tid < left) rec[i0 + tid] = c;
      __syncthreads();
    }
    nrec = left;
    stamp(4);
  }
  if constexpr (STAMP) {
    if (tid == 0) {
      const long w = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      for (int k = 0; k < 5; ++k) stamps[w * 8 + k] = t_acc[k];
      stamps[w * 8 + 5] = (unsigned long long)scan;
      stamps[w * 8 + 6] = (unsigned long long)max(jA, jB);
    }
  }

  if constexpr (!BWD) {
    const long npix = (long)W * H;
    auto store = [&](bool inside, int py, float t, unsigned last, float c0, float c1, float c2, float d) {
      if (!inside) return;
      const long pix = (long)py * W + pxi;
      out_final_T[v * npix + pix] = t;
      if constexpr (CONTRIB) out_n_contrib[v * npix + pix] = last;
      out_color[(v * 3 + 0) * npix + pix] = c0 + t * bg[0];
      out_color[(v * 3 + 1) * npix + pix] = c1 + t * bg[1];
      out_color[(v * 3 + 2) * npix + pix] = c2 + t * bg[2];
      out_depth[v * npix + pix] = d;
    };
    store(insideA, pyA, fabsf(px.T.x), lastA, px.C0.x, px.C1.x, px.C2.x, px.D.x);
    store(insideB, pyB, fabsf(px.T.y), lastB, px.C0.y, px.C1.y, px.C2.y, px.D.y);
  }
}


// ---------------------------------------------------------------------------------------------
// Backward of the per-Gaussian stages (backward.cu:144-276 computeCov2DCUDA, :346-396 preprocessCUDA,
// :278-342 computeCov3D): one thread per Gaussian, views summed in ascending order (deterministic).
// acc (V,P,9) comes from the backward blend: mean2D x,y | conic x,y,z | opacity | colour r,g,b.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_preprocess_backward_kernel(
    int P, int n_views, int W, int H, const float* __restrict__ means3D, const float* __restrict__ scales,
    float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const Camera* __restrict__ cams,
    const int* __restrict__ radii, const float* __restrict__ acc, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drotations, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dmeans2D) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= P) return;
  const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
  // covariance handed over (backward.cu:346-396 stops at dL/dcov3D then) or built from scale and rotation
  const bool precomp = cov3D_precomp != nullptr;
  float sv[3] = {0.f, 0.f, 0.f}, r = 1.f, x = 0.f, y = 0.f, z = 0.f;
  if (!precomp) {
    sv[0] = scale_modifier * scales[3 * idx]; sv[1] = scale_modifier * scales[3 * idx + 1];
    sv[2] = scale_modifier * scales[3 * idx + 2];
    r = rotations[4 * idx]; x = rotations[4 * idx + 1]; y = rotations[4 * idx + 2]; z = rotations[4 * idx + 3];
  }
  const float Rm[3][3] = {
      {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  float Mk[3][3];                                   // M[k][i] = s_k R[i][k]
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) Mk[k][i] = sv[k] * Rm[i][k];
  float V3[3][3];                                   // Sigma = M^T M
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V3[i][j] = Mk[0][i] * Mk[0][j] + Mk[1][i] * Mk[1][j] + Mk[2][i] * Mk[2][j];
  if (precomp) {
    const float* c6 = cov3D_precomp + 6 * (long)idx;       // xx xy xz yy yz zz (forward.cu:118-121)
    V3[0][0] = c6[0]; V3[0][1] = V3[1][0] = c6[1]; V3[0][2] = V3[2][0] = c6[2];
    V3[1][1] = c6[3]; V3[1][2] = V3[2][1] = c6[4]; V3[2][2] = c6[5];
  }

  float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dcol[3] = {0.f, 0.f, 0.f}, dop = 0.f;
  for (int v = 0; v < n_views; ++v) {
    const float* a9 = acc + ((long)v * P + idx) * 9;
    dop += a9[5];
    dcol[0] += a9[6]; dcol[1] += a9[7]; dcol[2] += a9[8];
    if (dL_dmeans2D) {
      dL_dmeans2D[((long)v * P + idx) * 3 + 0] = a9[0];
      dL_dmeans2D[((long)v * P + idx) * 3 + 1] = a9[1];
      dL_dmeans2D[((long)v * P + idx) * 3 + 2] = 0.f;
    }
    if (!(radii[(long)v * P + idx] > 0)) continue;
    const Camera& cam = cams[v];
    const float* vm = cam.view;
    const float* pm = cam.proj;
    const float dcx = a9[2], dcy = a9[3], dcz = a9[4];
    float t0 = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
    float t1 = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
    const float t2 = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
    const float txtz = t0 / t2, tytz = t1 / t2;
    t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
    t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float h_x = cam.focal_x, h_y = cam.focal_y;
    const float j00 = h_x / t2, j02 = -(h_x * t0) / (t2 * t2);
    const float j11 = h_y / t2, j12 = -(h_y * t1) / (t2 * t2);
    float Tm[2][3], TV[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Tm[0][c] = j00 * vm[4 * c + 0] + j02 * vm[4 * c + 2];
      Tm[1][c] = j11 * vm[4 * c + 1] + j12 * vm[4 * c + 2];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) TV[i][c] = Tm[i][0] * V3[0][c] + Tm[i][1] * V3[1][c] + Tm[i][2] * V3[2][c];
    const float a = (TV[0][0] * Tm[0][0] + TV[0][1] * Tm[0][1] + TV[0][2] * Tm[0][2]) + 0.3f;
    const float b = TV[0][0] * Tm[1][0] + TV[0][1] * Tm[1][1] + TV[0][2] * Tm[1][2];
    const float c2 = (TV[1][0] * Tm[1][0] + TV[1][1] * Tm[1][1] + TV[1][2] * Tm[1][2]) + 0.3f;
    const float denom = a * c2 - b * b;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0.f) {
      dL_da = denom2inv * (-c2 * c2 * dcx + 2 * b * c2 * dcy + (denom - a * c2) * dcz);
      dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c2) * dcx);
      dL_db = denom2inv * 2 * (b * c2 * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
      dcov[0] += Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc;
      dcov[3] += Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc;
      dcov[5] += Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc;
      dcov[1] += 2 * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db + 2 * Tm[1][0] * Tm[1][1] * dL_dc;
      dcov[2] += 2 * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db + 2 * Tm[1][0] * Tm[1][2] * dL_dc;
      dcov[4] += 2 * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db + 2 * Tm[1][1] * Tm[1][2] * dL_dc;
    }
    float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float dT0 = 2 * TV[0][c] * dL_da + TV[1][c] * dL_db;
      const float dT1 = 2 * TV[1][c] * dL_dc + TV[0][c] * dL_db;
      dJ00 += vm[4 * c + 0] * dT0;
      dJ02 += vm[4 * c + 2] * dT0;
      dJ11 += vm[4 * c + 1] * dT1;
      dJ12 += vm[4 * c + 2] * dT1;
    }
    const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = x_grad_mul * -h_x * tz2 * dJ02;
    const float dty = y_grad_mul * -h_y * tz2 * dJ12;
    const float dtz = -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * t0) * tz3 * dJ02 + (2 * h_y * t1) * tz3 * dJ12;
    dmean[0] += vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
    dmean[1] += vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
    dmean[2] += vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
    // screen-space mean -> 3D mean through the full projection (backward.cu:365-383)
    const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = (pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12]) * m_w * m_w;
    const float mul2 = (pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13]) * m_w * m_w;
    const float g2x = a9[0], g2y = a9[1];
    dmean[0] += (pm[0] * m_w - pm[3] * mul1) * g2x + (pm[1] * m_w - pm[3] * mul2) * g2y;
    dmean[1] += (pm[4] * m_w - pm[7] * mul1) * g2x + (pm[5] * m_w - pm[7] * mul2) * g2y;
    dmean[2] += (pm[8] * m_w - pm[11] * mul1) * g2x + (pm[9] * m_w - pm[11] * mul2) * g2y;
  }
  dL_dopacity[idx] = dop;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dL_dcolors[3 * idx + i] = dcol[i];
    dL_dmeans3D[3 * idx + i] = dmean[i];
  }
  if (precomp) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dL_dcov3D[6 * (long)idx + i] = dcov[i];
    return;
  }
  // Sigma = M^T M -> scales and (un-normalised) quaternion (backward.cu:278-342)
  const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                          {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
  float dM[3][3], Q[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) dM[k][i] = 2.0f * (Mk[k][0] * dS[0][i] + Mk[k][1] * dS[1][i] + Mk[k][2] * dS[2][i]);
#pragma unroll
  for (int k = 0; k < 3; ++k) dL_dscales[3 * idx + k] = Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) Q[i][k] = sv[k] * dM[k][i];
  dL_drotations[4 * idx + 0] = 2 * z * (Q[1][0] - Q[0][1]) + 2 * y * (Q[0][2] - Q[2][0]) + 2 * x * (Q[2][1] - Q[1][2]);
  dL_drotations[4 * idx + 1] = 2 * y * (Q[0][1] + Q[1][0]) + 2 * z * (Q[0][2] + Q[2][0]) + 2 * r * (Q[2][1] - Q[1][2]) - 4 * x * (Q[2][2] + Q[1][1]);
  dL_drotations[4 * idx + 2] = 2 * x * (Q[0][1] + Q[1][0]) + 2 * r * (Q[0][2] - Q[2][0]) + 2 * z * (Q[2][1] + Q[1][2]) - 4 * y * (Q[2][2] + Q[0][0]);
  dL_drotations[4 * idx + 3] = 2 * r * (Q[1][0] - Q[0][1]) + 2 * x * (Q[0][2] + Q[2][0]) + 2 * y * (Q[2][1] + Q[1][2]) - 4 * z * (Q[1][1] + Q[0][0]);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct RasterWs {
  size_t vis_rec, rects, xy, conic_o, b_rect, b_comp, hist, starts, cursor, status, total, acc, radii, bwd_total;
};

inline void raster_layout(int P, int n_views, RasterWs* ws) {
  const size_t n = (size_t)P * n_views;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  ws->vis_rec = take(n * sizeof(uint4));
  ws->rects = take(n * sizeof(Rect));
  ws->xy = take(n * sizeof(float2));
  ws->conic_o = take(n * sizeof(float4));
  ws->b_rect = take(n * sizeof(Rect));
  ws->b_comp = take(n * 8);
  ws->hist = take((size_t)n_views * (kBuckets + 1) * sizeof(int));      // + the per-view visible counts
  ws->starts = take((size_t)n_views * (kBuckets + 1) * sizeof(int));
  ws->cursor = take((size_t)n_views * kBuckets * sizeof(int));
  ws->status = take(256);
  ws->total = off;
  ws->acc = take(n * 9 * sizeof(float));          // backward only
  ws->radii = take(n * sizeof(int));
  ws->bwd_total = off;
}

}  // namespace

unsigned long long* g_stamps = nullptr;   // diagnostic: per-tile phase cycles (ocrf_diag_raster_stamps)

namespace ocrf {

// the bucket histograms (+ per-view visible counts) inside a chain workspace: what a caller that arms the chain
// (hist_is_zero) has to clear itself before the chain's first kernel
int* raster_chain_hist(void* workspace, int P, int n_views, size_t* n_words) {
  RasterWs ws;
  raster_layout(P, n_views, &ws);
  *n_words = (size_t)n_views * (kBuckets + 1);
  return reinterpret_cast<int*>(static_cast<char*>(workspace) + ws.hist);
}

int raster_forward_chain(int P, int n_sets, int views_per_set, int H, int W, const float* means3D, const float* colors,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* cameras, const int* view_sel, const float* bg,
                         int depth_mode, float* out_color, float* out_depth, float* out_final_T,
                         uint32_t* out_n_contrib, int* radii, uint32_t* tiles_touched, int* status, void* workspace,
                         size_t workspace_bytes, int* gate, bool shared_means, bool hist_is_zero,
                         hipStream_t stream) {
  if (n_sets <= 0 || views_per_set <= 0) return (int)hipErrorInvalidValue;
  const int n_views = n_sets * views_per_set;
  if (P < 0 || n_views <= 0 || H <= 0 || W <= 0 || (depth_mode != 0 && depth_mode != 1) ||
      !out_color || !out_depth || !out_final_T || !bg || !cameras)
    return (int)hipErrorInvalidValue;
  const size_t npix = (size_t)H * W * n_views;
  if (P == 0) {   // rasterize_points.cu:68-69: zero-filled outputs, nothing launched
    hipError_t e = ocrf::zero_async(out_color, npix * 3 * sizeof(float), stream);
    if (e == hipSuccess) e = ocrf::zero_async(out_depth, npix * sizeof(float), stream);
    if (e == hipSuccess) e = ocrf::zero_async(out_final_T, npix * sizeof(float), stream);
    if (e == hipSuccess && out_n_contrib) e = ocrf::zero_async(out_n_contrib, npix * sizeof(uint32_t), stream);
    return (int)e;
  }
  if (!means3D || !colors || !opacities || !radii || (!cov3D_precomp && (!scales || !rotations)))
    return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  if (gx > 65535 || gy > 65535) return (int)hipErrorInvalidValue;
  RasterWs ws;
  raster_layout(P, n_views, &ws);
  if (!workspace || workspace_bytes < ws.total) return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(workspace);
  auto* vis_rec = reinterpret_cast<uint4*>(base + ws.vis_rec);
  auto* rects = reinterpret_cast<Rect*>(base + ws.rects);
  auto* xy = reinterpret_cast<float2*>(base + ws.xy);
  auto* conic_o = reinterpret_cast<float4*>(base + ws.conic_o);
  auto* b_rect = reinterpret_cast<Rect*>(base + ws.b_rect);
  auto* b_comp = reinterpret_cast<unsigned long long*>(base + ws.b_comp);
  auto* hist = reinterpret_cast<int*>(base + ws.hist);
  auto* starts = reinterpret_cast<int*>(base + ws.starts);
  auto* cursor = reinterpret_cast<int*>(base + ws.cursor);
  int* st = status ? status : reinterpret_cast<int*>(base + ws.status);
  const Camera* cams = reinterpret_cast<const Camera*>(cameras);

  int* vis_count = hist + (size_t)n_views * kBuckets;
  hipError_t e = hipSuccess;
  if (!hist_is_zero) e = ocrf::zero_async(hist, (size_t)n_views * (kBuckets + 1) * sizeof(int), stream);
  if (e != hipSuccess) return (int)e;
  const int n_chunks = (P + kChunk - 1) / kChunk, n_pre = (P + kPreChunk - 1) / kPreChunk;
  const dim3 pgrid(n_chunks, n_views);
  const dim3 xgrid((unsigned)((n_pre + 7) / 8 * 8 * n_views));         // (chunk, view) pairs, XCD-aware order
  ocrf::launch(OCRF_K_RASTER_PREPROCESS, raster_preprocess_kernel, xgrid, dim3(kBlock), 0, stream, P, n_views,
               n_pre, W, H, gx, gy, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp,
               cams, vis_rec, vis_count, rects, xy, conic_o, radii, tiles_touched, hist, views_per_set, view_sel, gate,
               shared_means ? 0 : P, out_n_contrib == nullptr ? 1 : 0);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_RASTER_SCAN, raster_bucket_scan_kernel, dim3(n_views), dim3(kBlock), 0, stream,
               static_cast<const int*>(hist), starts, cursor, st, gate);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_RASTER_GATHER, raster_scatter_kernel, pgrid, dim3(kBlock), 0, stream, P,
               static_cast<const uint4*>(vis_rec), static_cast<const int*>(vis_count), cursor, b_rect, b_comp, gate);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const size_t lds = (size_t)kCapRec * 8 + (size_t)kStage * (16 + 16 + 16);
  const dim3 bgrid(gx, (gy + 1) / 2, n_views);        // one workgroup per vertical pair of tiles
  if (g_stamps) {      // diagnostic build of the same kernel (median depth), with or without the contributor index
#define OCRF_BLEND_STAMPED(CON)                                                                                     \
  hipLaunchKernelGGL((raster_blend_kernel<true, false, true, CON>), bgrid, dim3(kBlock), lds, stream, g_stamps, P, \
                     W, H, gy, static_cast<const int*>(starts), static_cast<const Rect*>(rects),                    \
                     static_cast<const Rect*>(b_rect), static_cast<const unsigned long long*>(b_comp),              \
                     static_cast<const float2*>(xy), static_cast<const float4*>(conic_o), colors, bg, out_color,    \
                     out_depth, out_final_T, out_n_contrib, st, BwdArgs{}, views_per_set, gate)
    if (out_n_contrib) OCRF_BLEND_STAMPED(true);
    else OCRF_BLEND_STAMPED(false);
#undef OCRF_BLEND_STAMPED
    return (int)hipGetLastError();
  }
#define OCRF_BLEND(MED, CON)                                                                                     \
  ocrf::launch(OCRF_K_RASTER_BLEND, raster_blend_kernel<false, false, MED, CON>, bgrid, dim3(kBlock), lds, stream, \
               (unsigned long long*)nullptr, P, W, H, gy, static_cast<const int*>(starts),                          \
               static_cast<const Rect*>(rects), static_cast<const Rect*>(b_rect),                                   \
               static_cast<const unsigned long long*>(b_comp), static_cast<const float2*>(xy),                      \
               static_cast<const float4*>(conic_o), colors, bg, out_color, out_depth, out_final_T, out_n_contrib,    \
               st, BwdArgs{}, views_per_set, gate)
  if (depth_mode == 0 && out_n_contrib) OCRF_BLEND(true, true);
  else if (depth_mode == 0) OCRF_BLEND(true, false);
  else if (out_n_contrib) OCRF_BLEND(false, true);
  else OCRF_BLEND(false, false);
#undef OCRF_BLEND
  return (int)hipGetLastError();
}

}  // namespace ocrf

extern "C" {

// Diagnostic: when set (device buffer of tiles*views*8 u64), the next forwards run the stamped
// build of the blend kernel.  Never used by the product path.
int ocrf_diag_raster_stamps(unsigned long long* buf) { g_stamps = buf; return 0; }

size_t ocrf_rasterize_workspace_bytes(int P, int n_views) {
  if (P <= 0 || n_views <= 0) return 0;
  RasterWs ws;
  raster_layout(P, n_views, &ws);
  return ws.total;
}

int ocrf_rasterize_forward_sets(int P, int n_sets, int views_per_set, int H, int W, const float* means3D,
                           const float* colors, const float* opacities, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* cameras, const float* bg, int depth_mode, float* out_color,
                           float* out_depth, float* out_final_T, uint32_t* out_n_contrib, int* radii,
                           uint32_t* tiles_touched, int* status, void* workspace,
                           size_t workspace_bytes, ocrf_stream_t stream_) {
  return ocrf::raster_forward_chain(P, n_sets, views_per_set, H, W, means3D, colors, opacities, scales, scale_modifier,
                                    rotations, cov3D_precomp, cameras, nullptr, bg, depth_mode, out_color, out_depth,
                                    out_final_T, out_n_contrib, radii, tiles_touched, status, workspace,
                                    workspace_bytes, nullptr, false, false, static_cast<hipStream_t>(stream_));
}

int ocrf_rasterize_forward(int P, int n_views, int H, int W, const float* means3D,
                           const float* colors, const float* opacities, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* cameras, const float* bg, int depth_mode, float* out_color,
                           float* out_depth, float* out_final_T, uint32_t* out_n_contrib, int* radii,
                           uint32_t* tiles_touched, int* status, void* workspace,
                           size_t workspace_bytes, ocrf_stream_t stream_) {
  return ocrf_rasterize_forward_sets(P, 1, n_views, H, W, means3D, colors, opacities, scales, scale_modifier, rotations,
                                     cov3D_precomp, cameras, bg, depth_mode, out_color, out_depth, out_final_T,
                                     out_n_contrib, radii, tiles_touched, status, workspace, workspace_bytes, stream_);
}

size_t ocrf_rasterize_backward_workspace_bytes(int P, int n_views) {
  if (P <= 0 || n_views <= 0) return 0;
  RasterWs ws;
  raster_layout(P, n_views, &ws);
  return ws.bwd_total;
}

static int rasterize_backward_impl(int P, int n_views, int H, int W, const float* means3D, const float* colors,
                                   const float* opacities, const float* scales, float scale_modifier,
                                   const float* rotations, const float* cov3D_precomp, const float* cameras,
                                   const float* bg, const float* fwd_color, const float* fwd_final_T,
                                   const uint32_t* fwd_n_contrib, const float* dL_dcolor, float* dL_dmeans3D,
                                   float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                                   float* dL_dcov3D, float* dL_dmeans2D, void* workspace, size_t workspace_bytes,
                                   ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (P <= 0 || n_views <= 0 || H <= 0 || W <= 0 || !means3D || !colors || !opacities || !cameras || !bg ||
      !fwd_color || !fwd_final_T || !fwd_n_contrib || !dL_dcolor || !dL_dmeans3D || !dL_dcolors || !dL_dopacity)
    return (int)hipErrorInvalidValue;
  if (cov3D_precomp ? !dL_dcov3D : (!scales || !rotations || !dL_dscales || !dL_drotations))
    return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  if (gx > 65535 || gy > 65535) return (int)hipErrorInvalidValue;
  RasterWs ws;
  raster_layout(P, n_views, &ws);
  if (!workspace || workspace_bytes < ws.bwd_total) return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(workspace);
  auto* vis_rec = reinterpret_cast<uint4*>(base + ws.vis_rec);
  auto* rects = reinterpret_cast<Rect*>(base + ws.rects);
  auto* xy = reinterpret_cast<float2*>(base + ws.xy);
  auto* conic_o = reinterpret_cast<float4*>(base + ws.conic_o);
  auto* b_rect = reinterpret_cast<Rect*>(base + ws.b_rect);
  auto* b_comp = reinterpret_cast<unsigned long long*>(base + ws.b_comp);
  auto* hist = reinterpret_cast<int*>(base + ws.hist);
  auto* starts = reinterpret_cast<int*>(base + ws.starts);
  auto* cursor = reinterpret_cast<int*>(base + ws.cursor);
  int* st = reinterpret_cast<int*>(base + ws.status);
  auto* acc = reinterpret_cast<float*>(base + ws.acc);
  auto* radii = reinterpret_cast<int*>(base + ws.radii);
  const Camera* cams = reinterpret_cast<const Camera*>(cameras);
  // rebuild the bucket-ordered lists exactly as the forward did (same kernels, same inputs)
  int* vis_count = hist + (size_t)n_views * kBuckets;
  hipError_t e = ocrf::zero_async(hist, (size_t)n_views * (kBuckets + 1) * sizeof(int), stream);
  if (e == hipSuccess) e = ocrf::zero_async(acc, (size_t)n_views * P * 9 * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  const int n_chunks = (P + kChunk - 1) / kChunk, n_pre = (P + kPreChunk - 1) / kPreChunk;
  const dim3 pgrid(n_chunks, n_views);
  const dim3 xgrid((unsigned)((n_pre + 7) / 8 * 8 * n_views));
  hipLaunchKernelGGL(raster_preprocess_kernel, xgrid, dim3(kBlock), 0, stream, P, n_views, n_pre, W, H, gx, gy,
                     means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp, cams, vis_rec,
                     vis_count, rects, xy, conic_o, radii, (unsigned*)nullptr, hist, n_views, (const int*)nullptr,
                     (const int*)nullptr, P, 0);
  hipLaunchKernelGGL(raster_bucket_scan_kernel, dim3(n_views), dim3(kBlock), 0, stream, static_cast<const int*>(hist),
                     starts, cursor, st, (const int*)nullptr);
  hipLaunchKernelGGL(raster_scatter_kernel, pgrid, dim3(kBlock), 0, stream, P, static_cast<const uint4*>(vis_rec),
                     static_cast<const int*>(vis_count), cursor, b_rect, b_comp, (const int*)nullptr);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  BwdArgs bw;
  bw.dL_dcolor = dL_dcolor; bw.fwd_color = fwd_color; bw.fwd_final_T = fwd_final_T;
  bw.fwd_n_contrib = fwd_n_contrib; bw.acc = acc;
  const size_t lds = (size_t)kCapRec * 8 + (size_t)kStage * (16 + 16 + 16 + 4 + 36);
  ocrf::launch(OCRF_K_RASTER_BLEND_BWD, raster_blend_kernel<false, true, true>, dim3(gx, (gy + 1) / 2, n_views),
               dim3(kBlock), lds, stream, (unsigned long long*)nullptr, P, W, H, gy, static_cast<const int*>(starts),
               static_cast<const Rect*>(rects), static_cast<const Rect*>(b_rect),
               static_cast<const unsigned long long*>(b_comp),
               static_cast<const float2*>(xy), static_cast<const float4*>(conic_o), colors, bg, (float*)nullptr,
               (float*)nullptr, (float*)nullptr, (unsigned*)nullptr, st, bw, n_views, (int*)nullptr);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_RASTER_PRE_BWD, raster_preprocess_backward_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock),
               0, stream, P, n_views, W, H, means3D, scales, scale_modifier, rotations, cov3D_precomp, cams,
               static_cast<const int*>(radii), static_cast<const float*>(acc), dL_dmeans3D, dL_dcolors, dL_dopacity,
               dL_dscales, dL_drotations, dL_dcov3D, dL_dmeans2D);
  return (int)hipGetLastError();
}

int ocrf_rasterize_backward(int P, int n_views, int H, int W, const float* means3D, const float* colors,
                            const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cameras, const float* bg,
                            const float* fwd_color, const float* fwd_final_T, const uint32_t* fwd_n_contrib,
                            const float* dL_dcolor, float* dL_dmeans3D, float* dL_dcolors,
                            float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                            float* dL_dmeans2D, void* workspace, size_t workspace_bytes,
                            ocrf_stream_t stream) {
  if (!scales || !rotations) return (int)hipErrorInvalidValue;
  return rasterize_backward_impl(P, n_views, H, W, means3D, colors, opacities, scales, scale_modifier, rotations,
                                 nullptr, cameras, bg, fwd_color, fwd_final_T, fwd_n_contrib, dL_dcolor, dL_dmeans3D,
                                 dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, nullptr, dL_dmeans2D, workspace,
                                 workspace_bytes, stream);
}

int ocrf_rasterize_backward_cov3d(int P, int n_views, int H, int W, const float* means3D, const float* colors,
                                  const float* opacities, const float* cov3D_precomp, const float* cameras,
                                  const float* bg, const float* fwd_color, const float* fwd_final_T,
                                  const uint32_t* fwd_n_contrib, const float* dL_dcolor, float* dL_dmeans3D,
                                  float* dL_dcolors, float* dL_dopacity, float* dL_dcov3D, float* dL_dmeans2D,
                                  void* workspace, size_t workspace_bytes, ocrf_stream_t stream) {
  if (!cov3D_precomp) return (int)hipErrorInvalidValue;
  return rasterize_backward_impl(P, n_views, H, W, means3D, colors, opacities, nullptr, 1.0f, nullptr, cov3D_precomp,
                                 cameras, bg, fwd_color, fwd_final_T, fwd_n_contrib, dL_dcolor, dL_dmeans3D, dL_dcolors,
                                 dL_dopacity, nullptr, nullptr, dL_dcov3D, dL_dmeans2D, workspace, workspace_bytes,
                                 stream);
}

}  // extern "C"

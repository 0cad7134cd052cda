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
//                    radii / tile rectangles are compiler-independent;
//   2. depth sort  — ONE sort of the P per-view (depth bits, id) pairs (P <= 0.5 M), not of R;
//   3. gather      — per-Gaussian state re-laid out in depth order (streamed, not gathered, later);
//   4. blend       — one 16x16-pixel workgroup per (tile, view) walks the depth-ordered list,
//                    keeps the entries whose tile rectangle covers its tile (wave ballot +
//                    prefix -> order-preserving compaction into LDS) and alpha-blends them front
//                    to back, stopping as soon as every pixel is saturated.  The per-tile list
//                    is therefore identical, element for element, to the reference's sorted
//                    range (same (depth, id) order), but no R-sized buffer, no 64-bit R sort, no
//                    range pass and no host synchronisation exist; the whole forward is
//                    hipGraph-capturable and batches any number of views over one Gaussian set.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTileX = 16, kTileY = 16;     // cuda_rasterizer/config.h:15-17
constexpr int kCap = 512;                    // LDS list capacity of the blend kernel
constexpr unsigned kInvisible = 0xFFFFFFFFu;

struct Camera {            // 36 floats per view, see ocrf_hip.h
  float view[16];
  float proj[16];
  float tanfovx, tanfovy, focal_x, focal_y;
};

struct __attribute__((aligned(8))) Rect { unsigned short x0, y0, x1, y1; };

// auxiliary.h:41-44 — the reference evaluates this in double precision (its literals are double)
__device__ __forceinline__ float ndc2pix(float v, int S) {
  return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// ---------------------------------------------------------------------------------------------
// 1. preprocess (forward.cu:155-256).  Expression order mirrors oracle/rasterize_ref.c exactly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_preprocess_kernel(
    int P, int W, int H, int gx, int gy, const float* __restrict__ means3D,
    const float* __restrict__ opacities, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const Camera* __restrict__ cams, unsigned long long* __restrict__ keys,
    unsigned* __restrict__ vals, Rect* __restrict__ rects, float2* __restrict__ xy,
    float4* __restrict__ conic_o, int* __restrict__ radii, unsigned* __restrict__ tiles_touched,
    int* __restrict__ n_vis) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  const int v = blockIdx.y;
  if (idx >= P) return;
  const long o = (long)v * P + idx;
  const Camera& cam = cams[v];
  const float* vm = cam.view;
  const float* pm = cam.proj;
  unsigned key = kInvisible;
  int my_radii = 0;
  unsigned touched = 0;

  const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
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

    float c3[6];
    if (cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c3[i] = cov3D_precomp[6 * (long)idx + i];
    } else {
      // computeCov3D (forward.cu:118-152): Sigma = R diag(s^2) R^T, quaternion not normalised
      const float sx = scale_modifier * scales[3 * idx], sy = scale_modifier * scales[3 * idx + 1],
                  sz = scale_modifier * scales[3 * idx + 2];
      const float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2],
                  z = rotations[4 * idx + 3];
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

    // computeCov2D (forward.cu:74-113)
    const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
    const float txtz = vx / vz, tytz = vy / vz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    const float j00 = cam.focal_x / vz, j02 = -(cam.focal_x * tx) / (vz * vz);
    const float j11 = cam.focal_y / vz, j12 = -(cam.focal_y * ty) / (vz * vz);
    float A[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r0 = vm[4 * c + 0], r1 = vm[4 * c + 1], r2 = vm[4 * c + 2];
      A[0][c] = j00 * r0 + j02 * r2;
      A[1][c] = j11 * r1 + j12 * r2;
    }
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
        key = __float_as_uint(vz);        // vz > 0.2: the raw bits order like the value
        my_radii = rad;
        touched = (unsigned)((y1 - y0) * (x1 - x0));
        Rect rc;
        rc.x0 = (unsigned short)x0; rc.y0 = (unsigned short)y0;
        rc.x1 = (unsigned short)x1; rc.y1 = (unsigned short)y1;
        rects[o] = rc;
        xy[o] = make_float2(pixx, pixy);
        conic_o[o] = make_float4(con_x, con_y, con_z, opacities[idx]);
        atomicAdd(&n_vis[v], 1);
      }
    }
  }
  keys[o] = ((unsigned long long)v << 32) | key;
  vals[o] = (unsigned)idx;
  radii[o] = my_radii;
  if (tiles_touched) tiles_touched[o] = touched;
}

// ---------------------------------------------------------------------------------------------
// 3. gather the visible Gaussians' state into depth order.
//    sa = (x, y, conic.x, conic.y)  sb = (conic.z, opacity, depth, r)  sc = (g, b)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_gather_kernel(
    int P, const unsigned long long* __restrict__ keys_sorted, const unsigned* __restrict__ vals_sorted,
    const Rect* __restrict__ rects, const float2* __restrict__ xy, const float4* __restrict__ conic_o,
    const float* __restrict__ colors, const int* __restrict__ n_vis, Rect* __restrict__ s_rect,
    float4* __restrict__ sa, float4* __restrict__ sb, float2* __restrict__ sc) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  const int v = blockIdx.y;
  if (k >= n_vis[v]) return;
  const long o = (long)v * P + k;
  const unsigned id = vals_sorted[o];
  const long g = (long)v * P + id;
  const float2 p = xy[g];
  const float4 co = conic_o[g];
  const float depth = __uint_as_float((unsigned)(keys_sorted[o] & 0xFFFFFFFFull));
  s_rect[o] = rects[g];
  sa[o] = make_float4(p.x, p.y, co.x, co.y);
  sb[o] = make_float4(co.z, co.w, depth, colors[3 * (long)id]);
  sc[o] = make_float2(colors[3 * (long)id + 1], colors[3 * (long)id + 2]);
}

// ---------------------------------------------------------------------------------------------
// 4. blend (forward.cu:261-374 + w-depth README:5-11).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void raster_blend_kernel(
    int P, int W, int H, int gx, int depth_mode, const int* __restrict__ n_vis,
    const Rect* __restrict__ s_rect, const float4* __restrict__ sa, const float4* __restrict__ sb,
    const float2* __restrict__ sc, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_final_T,
    unsigned* __restrict__ out_n_contrib) {
  __shared__ float4 l_a[kCap];
  __shared__ float4 l_b[kCap];
  __shared__ float2 l_c[kCap];
  __shared__ int l_wtot[kBlock / 64];

  const int tid = threadIdx.x;
  const int tx = blockIdx.x, ty = blockIdx.y, v = blockIdx.z;
  const int lx = tid % kTileX, ly = tid / kTileX;
  const int pxi = tx * kTileX + lx, pyi = ty * kTileY + ly;
  const bool inside = pxi < W && pyi < H;
  const float pixf_x = (float)pxi, pixf_y = (float)pyi;
  const long base = (long)v * P;
  const int nv = n_vis[v];
  const int wave = tid / 64, lane = tid % 64;

  bool done = !inside;
  float T = 1.0f;
  unsigned contributor = 0, last_contributor = 0;
  float C0 = 0.f, C1 = 0.f, C2 = 0.f;
  float D = depth_mode == 0 ? 15.0f : 0.0f;

  int scan = 0;
  while (true) {
    // ---- fill: scan the depth-ordered list, keep the entries whose rect covers this tile ----
    int count = 0;
    while (count < kBlock && scan < nv) {
      const int i = scan + tid;
      bool hit = false;
      if (i < nv) {
        const Rect rc = s_rect[base + i];
        hit = (tx >= rc.x0) && (tx < rc.x1) && (ty >= rc.y0) && (ty < rc.y1);
      }
      const unsigned long long m = __ballot(hit);
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) l_wtot[wave] = __popcll(m);
      __syncthreads();
      int off = count, tot = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) {
        const int c = l_wtot[w];
        if (w < wave) off += c;
        tot += c;
      }
      if (hit) {
        l_a[off + rank] = sa[base + i];
        l_b[off + rank] = sb[base + i];
        l_c[off + rank] = sc[base + i];
      }
      count += tot;
      scan += kBlock;
      __syncthreads();
    }
    if (count == 0) break;     // list exhausted
    // ---- blend the `count` staged entries front to back ----
    for (int j = 0; j < count && !done; ++j) {
      contributor++;
      const float4 a = l_a[j];
      const float4 b = l_b[j];
      const float dx = a.x - pixf_x, dy = a.y - pixf_y;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      if (power > 0.0f) continue;
      const float alpha = fminf(0.99f, b.y * __expf(power));
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1 - alpha);
      if (test_T < 0.0001f) { done = true; continue; }
      const float2 c = l_c[j];
      const float wgt = alpha * T;
      C0 = fmaf(b.w, wgt, C0);
      C1 = fmaf(c.x, wgt, C1);
      C2 = fmaf(c.y, wgt, C2);
      if (depth_mode == 0) {
        if (T > 0.5f && test_T < 0.5f) D = b.z;
      } else {
        D = fmaf(b.z, wgt, D);
      }
      T = test_T;
      last_contributor = contributor;
    }
    // every pixel saturated -> stop scanning (forward.cu:304-307)
    if (__syncthreads_count(done) == kBlock) break;
  }

  if (inside) {
    const long npix = (long)W * H;
    const long pix = (long)pyi * W + pxi;
    out_final_T[v * npix + pix] = T;
    out_n_contrib[v * npix + pix] = last_contributor;
    out_color[(v * 3 + 0) * npix + pix] = C0 + T * bg[0];
    out_color[(v * 3 + 1) * npix + pix] = C1 + T * bg[1];
    out_color[(v * 3 + 2) * npix + pix] = C2 + T * bg[2];
    out_depth[v * npix + pix] = D;
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct RasterWs {
  size_t keys_in, keys_out, vals_in, vals_out, rects, xy, conic_o, s_rect, sa, sb, sc, n_vis, sort_tmp,
      total;
  size_t sort_tmp_bytes;
};

inline hipError_t raster_layout(int P, int n_views, RasterWs* ws) {
  const size_t n = (size_t)P * n_views;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  ws->keys_in = take(n * 8);
  ws->keys_out = take(n * 8);
  ws->vals_in = take(n * 4);
  ws->vals_out = take(n * 4);
  ws->rects = take(n * sizeof(Rect));
  ws->xy = take(n * sizeof(float2));
  ws->conic_o = take(n * sizeof(float4));
  ws->s_rect = take(n * sizeof(Rect));
  ws->sa = take(n * sizeof(float4));
  ws->sb = take(n * sizeof(float4));
  ws->sc = take(n * sizeof(float2));
  ws->n_vis = take((size_t)n_views * sizeof(int));
  size_t tmp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const unsigned long long*)nullptr,
                                           (unsigned long long*)nullptr, (const unsigned*)nullptr,
                                           (unsigned*)nullptr, n, 0, 64, nullptr);
  ws->sort_tmp_bytes = tmp;
  ws->sort_tmp = take(tmp);
  ws->total = off;
  return e;
}

inline int bits_for(int n) {
  int b = 0;
  while ((1 << b) < n) ++b;
  return b;
}

}  // namespace

extern "C" {

size_t ocrf_rasterize_workspace_bytes(int P, int n_views) {
  if (P <= 0 || n_views <= 0) return 0;
  RasterWs ws;
  if (raster_layout(P, n_views, &ws) != hipSuccess) return 0;
  return ws.total;
}

int ocrf_rasterize_forward(int P, int n_views, int H, int W, const float* means3D,
                           const float* colors, const float* opacities, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* cameras, const float* bg, int depth_mode, float* out_color,
                           float* out_depth, float* out_final_T, uint32_t* out_n_contrib, int* radii,
                           uint32_t* tiles_touched, void* workspace, size_t workspace_bytes,
                           ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (P < 0 || n_views <= 0 || H <= 0 || W <= 0 || (depth_mode != 0 && depth_mode != 1) ||
      !out_color || !out_depth || !out_final_T || !out_n_contrib || !bg || !cameras)
    return (int)hipErrorInvalidValue;
  const size_t npix = (size_t)H * W * n_views;
  if (P == 0) {   // rasterize_points.cu:68-69: zero-filled outputs, nothing launched
    hipError_t e = hipMemsetAsync(out_color, 0, npix * 3 * sizeof(float), stream);
    if (e == hipSuccess) e = hipMemsetAsync(out_depth, 0, npix * sizeof(float), stream);
    if (e == hipSuccess) e = hipMemsetAsync(out_final_T, 0, npix * sizeof(float), stream);
    if (e == hipSuccess) e = hipMemsetAsync(out_n_contrib, 0, npix * sizeof(uint32_t), stream);
    return (int)e;
  }
  if (!means3D || !colors || !opacities || !radii || (!cov3D_precomp && (!scales || !rotations)))
    return (int)hipErrorInvalidValue;
  const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
  if (gx > 65535 || gy > 65535) return (int)hipErrorInvalidValue;
  RasterWs ws;
  hipError_t e = raster_layout(P, n_views, &ws);
  if (e != hipSuccess) return (int)e;
  if (!workspace || workspace_bytes < ws.total) return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(workspace);
  auto* keys_in = reinterpret_cast<unsigned long long*>(base + ws.keys_in);
  auto* keys_out = reinterpret_cast<unsigned long long*>(base + ws.keys_out);
  auto* vals_in = reinterpret_cast<unsigned*>(base + ws.vals_in);
  auto* vals_out = reinterpret_cast<unsigned*>(base + ws.vals_out);
  auto* rects = reinterpret_cast<Rect*>(base + ws.rects);
  auto* xy = reinterpret_cast<float2*>(base + ws.xy);
  auto* conic_o = reinterpret_cast<float4*>(base + ws.conic_o);
  auto* s_rect = reinterpret_cast<Rect*>(base + ws.s_rect);
  auto* sa = reinterpret_cast<float4*>(base + ws.sa);
  auto* sb = reinterpret_cast<float4*>(base + ws.sb);
  auto* sc = reinterpret_cast<float2*>(base + ws.sc);
  auto* n_vis = reinterpret_cast<int*>(base + ws.n_vis);
  const Camera* cams = reinterpret_cast<const Camera*>(cameras);

  e = hipMemsetAsync(n_vis, 0, (size_t)n_views * sizeof(int), stream);
  if (e != hipSuccess) return (int)e;
  const dim3 pgrid((P + kBlock - 1) / kBlock, n_views);
  ocrf::launch(OCRF_K_RASTER_PREPROCESS, raster_preprocess_kernel, pgrid, dim3(kBlock), 0, stream, P,
               W, H, gx, gy, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp,
               cams, keys_in, vals_in, rects, xy, conic_o, radii, tiles_touched, n_vis);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  size_t tmp = ws.sort_tmp_bytes;
  e = rocprim::radix_sort_pairs(base + ws.sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out,
                                (size_t)P * n_views, 0, 32 + bits_for(n_views), stream);
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_RASTER_GATHER, raster_gather_kernel, pgrid, dim3(kBlock), 0, stream, P,
               static_cast<const unsigned long long*>(keys_out), static_cast<const unsigned*>(vals_out),
               static_cast<const Rect*>(rects), static_cast<const float2*>(xy),
               static_cast<const float4*>(conic_o), colors, static_cast<const int*>(n_vis), s_rect, sa,
               sb, sc);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_RASTER_BLEND, raster_blend_kernel, dim3(gx, gy, n_views), dim3(kBlock), 0, stream,
               P, W, H, gx, depth_mode, static_cast<const int*>(n_vis), static_cast<const Rect*>(s_rect),
               static_cast<const float4*>(sa), static_cast<const float4*>(sb),
               static_cast<const float2*>(sc), bg, out_color, out_depth, out_final_T, out_n_contrib);
  return (int)hipGetLastError();
}

}  // extern "C"

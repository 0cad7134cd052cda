// Shared by rasterize.hip (per-call pipeline) and raster_plan.hip (static render plans): the camera
// block, tile geometry and the per-Gaussian arithmetic of the rasteriser's preprocess, written once
// so that both pipelines produce the same bits (fp-contract is off for the whole library).
// Reference arithmetic: diff-gaussian-rasterization/cuda_rasterizer/forward.cu:74-256,
// auxiliary.h:41-77,139-164.
#pragma once
#include <hip/hip_runtime.h>

#include "ocrf_hip.h"

namespace rc {

constexpr int kBlock = 256;
constexpr int kTileX = 16, kTileY = 16;     // cuda_rasterizer/config.h:15-17

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

typedef float f2 __attribute__((ext_vector_type(2)));   // maps to v_pk_{mul,add,fma}_f32 on gfx950
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// ---- the view-dependent, parameter-independent half of preprocessCUDA (forward.cu:166-199) -------------
// What a (camera, mean) pair fixes: view-space depth, the projected centre and the four distinct entries of
// computeCov2D's Jacobian (forward.cu:83-98).  Same expressions, same order as raster_preprocess_kernel.
struct StaticPoint {
  float vz;                 // view-space depth (the sort key is its bit pattern)
  float projx, projy;       // NDC
  float j00, j02, j11, j12;
};

__device__ __forceinline__ bool static_point(const Camera& cam, float px, float py, float pz, StaticPoint* s) {
  const float* vm = cam.view;
  const float* pm = cam.proj;
  const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
  const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
  const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
  s->vz = vz;
  if (!(vz > 0.2f)) return false;                                // auxiliary.h:154
  const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
  const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
  const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
  const float p_w = 1.0f / (hw + 0.0000001f);
  s->projx = hx * p_w;
  s->projy = hy * p_w;
  const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
  const float txtz = vx / vz, tytz = vy / vz;
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
  s->j00 = cam.focal_x / vz;
  s->j02 = -(cam.focal_x * tx) / (vz * vz);
  s->j11 = cam.focal_y / vz;
  s->j12 = -(cam.focal_y * ty) / (vz * vz);
  return true;
}

// A = J W (2x3): rows of the Jacobian times the view rotation
__device__ __forceinline__ void jacobian_rows(const Camera& cam, float j00, float j02, float j11, float j12,
                                              float A[2][3]) {
  const float* vm = cam.view;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r0 = vm[4 * c + 0], r1 = vm[4 * c + 1], r2 = vm[4 * c + 2];
    A[0][c] = j00 * r0 + j02 * r2;
    A[1][c] = j11 * r1 + j12 * r2;
  }
}

// Conservative screen-space radius (pixels) of a Gaussian whose world-space extent is bounded by
// `rn` >= scale_modifier * max|s_k| * |R(q)|_2:  lambda_max(A Sigma A^T + 0.3 I) <= 0.3 + |A|_F^2 rn^2; the
// result is >= the reference's radius + 2 px.  Monotone in rn.
__device__ __forceinline__ float radius_bound(const float A[2][3], float rn) {
  const float af = A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][0] * A[1][0] +
                   A[1][1] * A[1][1] + A[1][2] * A[1][2];
  return 3.f * sqrtf(0.3f + af * rn * rn) * 1.001f + 3.f;
}

// |R(q)|_2 * max|s| bound for an unnormalised quaternion: R = (1 - |q|^2) I + |q|^2 Rot(q/|q|)
__device__ __forceinline__ float extent_bound(float s0, float s1, float s2, float qr, float qx, float qy, float qz) {
  const float smax = fmaxf(fabsf(s0), fmaxf(fabsf(s1), fabsf(s2)));
  const float qq = qr * qr + qx * qx + qy * qy + qz * qz;
  return (fabsf(1.f - qq) + qq) * smax;
}

// rect certainly empty (auxiliary.h:46-56: x1 <= x0 or y1 <= y0) for a centre (fxp, fyp) px and radius bound rb;
// NaNs compare false (-> not surely empty)
__device__ __forceinline__ bool surely_outside(float fxp, float fyp, float rb, int gx, int gy) {
  return (fxp + rb < 0.f) || (fxp - rb > (float)(kTileX * gx) + 1.f) || (fyp + rb < 0.f) ||
         (fyp - rb > (float)(kTileY * gy) + 1.f);
}

// computeCov3D (forward.cu:118-152): Sigma = R diag(s^2) R^T, quaternion not normalised; s already times the modifier
__device__ __forceinline__ void cov3d_from_scale_rot(float sx, float sy, float sz, float r, float x, float y, float z,
                                                     float c3[6]) {
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

// computeCov2D (forward.cu:74-113) given A = J W -> (cov_x, cov_y, cov_z) with the 0.3 dilation
__device__ __forceinline__ void cov2d(const float A[2][3], const float c3[6], float* cov_x, float* cov_y, float* cov_z) {
  const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
  float Bm[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) Bm[i][c] = A[i][0] * V[0][c] + A[i][1] * V[1][c] + A[i][2] * V[2][c];
  *cov_x = (Bm[0][0] * A[0][0] + Bm[0][1] * A[0][1] + Bm[0][2] * A[0][2]) + 0.3f;
  *cov_y = Bm[0][0] * A[1][0] + Bm[0][1] * A[1][1] + Bm[0][2] * A[1][2];
  *cov_z = (Bm[1][0] * A[1][0] + Bm[1][1] * A[1][1] + Bm[1][2] * A[1][2]) + 0.3f;
}

// conic, integer radius and tile rect from the 2D covariance and the pixel centre (forward.cu:201-238,
// auxiliary.h:46-56).  false: the reference leaves this Gaussian unrendered (det == 0 or an empty rect).
__device__ __forceinline__ bool conic_radius_rect(float cov_x, float cov_y, float cov_z, float pixx, float pixy, int gx,
                                                  int gy, float* con_x, float* con_y, float* con_z, int* rad_out,
                                                  Rect* rect) {
  const float det = cov_x * cov_z - cov_y * cov_y;
  if (!(det != 0.0f)) return false;
  const float det_inv = 1.f / det;
  *con_x = cov_z * det_inv;
  *con_y = -cov_y * det_inv;
  *con_z = cov_x * det_inv;
  const float mid = 0.5f * (cov_x + cov_z);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  const int rad = (int)my_radius;
  const int x0 = min(gx, max(0, (int)((pixx - (float)rad) / (float)kTileX)));
  const int y0 = min(gy, max(0, (int)((pixy - (float)rad) / (float)kTileY)));
  const int x1 = min(gx, max(0, (int)((pixx + (float)rad + (float)(kTileX - 1)) / (float)kTileX)));
  const int y1 = min(gy, max(0, (int)((pixy + (float)rad + (float)(kTileY - 1)) / (float)kTileY)));
  if ((x1 - x0) * (y1 - y0) == 0) return false;
  *rad_out = rad;
  rect->x0 = (unsigned short)x0; rect->y0 = (unsigned short)y0;
  rect->x1 = (unsigned short)x1; rect->y1 = (unsigned short)y1;
  return true;
}

// The tile rect of a record, tightened by its OPACITY.  alpha = min(0.99, o exp(power)) reaches 1/255 only where
// power >= -ln(255 o), i.e. inside the ellipse d^T Sigma^-1 d <= k = 2 ln(255 o) around the projected centre, whose bounding
// box has the half-widths sqrt(k cov_x), sqrt(k cov_z) (Sigma = the dilated 2D covariance whose inverse the conic is).  A
// tile of the reference's rect (a square around 3 sigma_max, whatever the opacity) that this box does not reach holds no
// pixel with alpha >= 1/255: the reference skips the record at every one of them (forward.cu:331-333: no colour, no change
// of T), so leaving the tile out changes nothing — and a faint Gaussian (a trained OcRF is mostly those) is listed in a
// fraction of the tiles.  The margin (0.02 on the logarithm, i.e. 2 % on alpha, + 1e-4 relative + 0.01 px) dominates the
// rounding of the conic, of v_exp_f32 and of the blend's folded exponent; NaN / Inf anywhere keep the reference's rect.
__device__ __forceinline__ void tighten_rect(float o, float cov_x, float cov_z, float pixx, float pixy, int gx, int gy,
                                             Rect* r) {
  if (!(o > 0.f) || !(o < 3.0e38f)) return;
  const float k = 2.f * (fmaxf(__logf(255.f * o), 0.f) + 0.02f);
  const float hx = sqrtf(k * cov_x) * 1.0001f + 0.01f, hy = sqrtf(k * cov_z) * 1.0001f + 0.01f;
  if (!(hx < 1.0e9f) || !(hy < 1.0e9f) || !(fabsf(pixx) < 1.0e9f) || !(fabsf(pixy) < 1.0e9f)) return;
  const int tx0 = min(gx, max(0, (int)floorf((pixx - hx) * (1.f / kTileX))));
  const int tx1 = min(gx, max(0, (int)floorf((pixx + hx) * (1.f / kTileX)) + 1));
  const int ty0 = min(gy, max(0, (int)floorf((pixy - hy) * (1.f / kTileY))));
  const int ty1 = min(gy, max(0, (int)floorf((pixy + hy) * (1.f / kTileY)) + 1));
  const int x0 = max((int)r->x0, tx0), x1 = min((int)r->x1, tx1), y0 = max((int)r->y0, ty0), y1 = min((int)r->y1, ty1);
  if (x1 <= x0 || y1 <= y0) {
    *r = Rect{0, 0, 0, 0};
    return;
  }
  r->x0 = (unsigned short)x0; r->y0 = (unsigned short)y0; r->x1 = (unsigned short)x1; r->y1 = (unsigned short)y1;
}

}  // namespace rc

namespace ocrf {

// rasterize.hip: the per-call pipeline (zero -> preprocess -> scan -> scatter -> blend) with two additions the
// plan path needs when it arms the pipeline as its on-device fallback:
//   gate      device int[2] or null: every kernel of the chain retires at once unless gate[0] != 0; when it is, the
//             last workgroup of the blend to arrive lowers it again (gate[1] = arrival counter, zero on entry);
//   hist_is_zero  the caller has cleared raster_chain_hist() itself (no zero launch at the head of the chain);
//   view_sel  device ints or null: item z renders camera view_sel[z] of `cameras`;
//   shared_means  means3D is (P,3) for every set instead of (n_sets,P,3).
int raster_forward_chain(int P, int n_sets, int views_per_set, int H, int W, const float* means3D, const float* colors,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* cameras, const int* view_sel, const float* bg,
                         int depth_mode, float* out_color, float* out_depth, float* out_final_T,
                         uint32_t* out_n_contrib, int* radii, uint32_t* tiles_touched, int* status, void* workspace,
                         size_t workspace_bytes, int* gate, bool shared_means, bool hist_is_zero, hipStream_t stream);
int* raster_chain_hist(void* workspace, int P, int n_views, size_t* n_words);

// index_prep.hip: stable LSD radix sort of n 32-bit keys carrying their original index (ids ascending among equal
// keys).  `keys` is clobbered; the sorted keys / ids are returned through the two pointers (they point into
// `scratch`, >= radix_sort_ids_bytes(n) bytes, or at `keys`).
size_t radix_sort_ids_bytes(int n);
// `plan_emit` (radix_emit.h): the last pass writes the render plan's per-view list arrays itself; *sorted_ids is then
// null (the values are not written).
struct RadixPlanEmit;
hipError_t radix_sort_ids(unsigned* keys, int n, int key_bits, void* scratch, size_t scratch_bytes,
                          const unsigned** sorted_keys, const int** sorted_ids, hipStream_t stream,
                          const RadixPlanEmit* plan_emit = nullptr);
// the state blocks of that call's look-back scans (bit 63 of a block's first word: the scan gave up)
void radix_sort_states(void* scratch, int n, int key_bits, const unsigned long long** states, int* n_states,
                       long* stride_words);


// index_prep.hip: in-place exclusive prefix sum of n non-negative ints (*total = their sum, may be null)
size_t exclusive_scan_bytes(long n);
hipError_t exclusive_scan_ints(int* data, long n, int* total, void* scratch, size_t scratch_bytes, hipStream_t stream);

}  // namespace ocrf

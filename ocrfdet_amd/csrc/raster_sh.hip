// Spherical-harmonics colours of the rasteriser's reference API (shs != None): per Gaussian the view-dependent RGB from
// up to 16 real SH coefficients per channel, evaluated along the unit vector from the camera centre to the mean, + 0.5,
// clamped at 0 with the clamp recorded for the backward — what preprocessCUDA does when colors_precomp is NULL
// (cuda_rasterizer/forward.cu:20-71, called at :240-247), and its backward (cuda_rasterizer/backward.cu:20-140).
// OcRFDet itself renders precomputed colours (gaussian_renderer/__init__.py:65: shs=None); this is the rest of the
// module's surface.  The colours come out as a (P,3) array that feeds the ordinary pipeline as colors_precomp, so the SH
// path is single-view by construction, as the reference's.
//
// Formulation: colour_c = 0.5 + sum_i basis_i(dir) * sh[i][c] with the 16 basis polynomials in one table (value and
// gradient per polynomial), accumulated in index order; the backward is the same table read the other way:
//   dL/dsh[i][c] = basis_i * g_c,   dL/ddir = sum_i grad(basis_i) * (sh[i] . g),   dL/dmean = (I - d d^T) dL/ddir / |v|
// with g = dL/dcolour where the forward did not clamp.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                          0.5462742152960396f};
constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                          -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// basis[i] for i < (deg + 1)^2; with GRAD also d basis[i] / d (x, y, z) (x, y, z treated as independent)
template <bool GRAD>
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float (&b)[16], float (&gx)[16], float (&gy)[16],
                                         float (&gz)[16]) {
  b[0] = kC0;
  if (GRAD) gx[0] = gy[0] = gz[0] = 0.0f;
  if (deg < 1) return;
  b[1] = -kC1 * y; b[2] = kC1 * z; b[3] = -kC1 * x;
  if (GRAD) {
    gx[1] = 0.0f; gy[1] = -kC1; gz[1] = 0.0f;
    gx[2] = 0.0f; gy[2] = 0.0f; gz[2] = kC1;
    gx[3] = -kC1; gy[3] = 0.0f; gz[3] = 0.0f;
  }
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  b[4] = kC2[0] * xy;
  b[5] = kC2[1] * yz;
  b[6] = kC2[2] * (2.0f * zz - xx - yy);
  b[7] = kC2[3] * xz;
  b[8] = kC2[4] * (xx - yy);
  if (GRAD) {
    gx[4] = kC2[0] * y;          gy[4] = kC2[0] * x;          gz[4] = 0.0f;
    gx[5] = 0.0f;                gy[5] = kC2[1] * z;          gz[5] = kC2[1] * y;
    gx[6] = kC2[2] * -2.0f * x;  gy[6] = kC2[2] * -2.0f * y;  gz[6] = kC2[2] * 4.0f * z;
    gx[7] = kC2[3] * z;          gy[7] = 0.0f;                gz[7] = kC2[3] * x;
    gx[8] = kC2[4] * 2.0f * x;   gy[8] = kC2[4] * -2.0f * y;  gz[8] = 0.0f;
  }
  if (deg < 3) return;
  b[9] = kC3[0] * y * (3.0f * xx - yy);
  b[10] = kC3[1] * xy * z;
  b[11] = kC3[2] * y * (4.0f * zz - xx - yy);
  b[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
  b[13] = kC3[4] * x * (4.0f * zz - xx - yy);
  b[14] = kC3[5] * z * (xx - yy);
  b[15] = kC3[6] * x * (xx - 3.0f * yy);
  if (GRAD) {
    gx[9] = kC3[0] * 6.0f * xy;                       gy[9] = kC3[0] * 3.0f * (xx - yy);                 gz[9] = 0.0f;
    gx[10] = kC3[1] * yz;                             gy[10] = kC3[1] * xz;                              gz[10] = kC3[1] * xy;
    gx[11] = kC3[2] * -2.0f * xy;                     gy[11] = kC3[2] * (4.0f * zz - xx - 3.0f * yy);    gz[11] = kC3[2] * 8.0f * yz;
    gx[12] = kC3[3] * -6.0f * xz;                     gy[12] = kC3[3] * -6.0f * yz;                      gz[12] = kC3[3] * 3.0f * (2.0f * zz - xx - yy);
    gx[13] = kC3[4] * (4.0f * zz - 3.0f * xx - yy);   gy[13] = kC3[4] * -2.0f * xy;                      gz[13] = kC3[4] * 8.0f * xz;
    gx[14] = kC3[5] * 2.0f * xz;                      gy[14] = kC3[5] * -2.0f * yz;                      gz[14] = kC3[5] * (xx - yy);
    gx[15] = kC3[6] * 3.0f * (xx - yy);               gy[15] = kC3[6] * -6.0f * xy;                      gz[15] = 0.0f;
  }
}

__global__ __launch_bounds__(256) void sh_colors_kernel(int P, int deg, int M, const float* __restrict__ means,
                                                        const float* __restrict__ campos, const float* __restrict__ shs,
                                                        float* __restrict__ colors, unsigned char* __restrict__ clamped) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float vx = means[3 * i] - campos[0], vy = means[3 * i + 1] - campos[1], vz = means[3 * i + 2] - campos[2];
  const float len = sqrtf(vx * vx + vy * vy + vz * vz);
  float b[16], u0[16], u1[16], u2[16];
  sh_basis<false>(deg, vx / len, vy / len, vz / len, b, u0, u1, u2);
  const int n = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * M * 3;
  float r[3] = {0.0f, 0.0f, 0.0f};
  for (int k = 0; k < n; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = k ? r[c] + b[k] * sh[3 * k + c] : b[0] * sh[c];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = r[c] + 0.5f;
    clamped[3 * i + c] = v < 0.0f;
    colors[3 * i + c] = fmaxf(v, 0.0f);
  }
}

__global__ __launch_bounds__(256) void sh_colors_backward_kernel(int P, int deg, int M, const float* __restrict__ means,
                                                                 const float* __restrict__ campos,
                                                                 const float* __restrict__ shs,
                                                                 const unsigned char* __restrict__ clamped,
                                                                 const float* __restrict__ dL_dcolors,
                                                                 float* __restrict__ dL_dmeans, float* __restrict__ dL_dshs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float vx = means[3 * i] - campos[0], vy = means[3 * i + 1] - campos[1], vz = means[3 * i + 2] - campos[2];
  const float len = sqrtf(vx * vx + vy * vy + vz * vz);
  const float x = vx / len, y = vy / len, z = vz / len;
  float b[16], gx[16], gy[16], gz[16];
  sh_basis<true>(deg, x, y, z, b, gx, gy, gz);
  const int n = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * M * 3;
  float* dsh = dL_dshs + (size_t)i * M * 3;
  float g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c] = clamped[3 * i + c] ? 0.0f : dL_dcolors[3 * i + c];
  float dx = 0.0f, dy = 0.0f, dz = 0.0f;
  for (int k = 0; k < M; ++k) {
    if (k < n) {
      const float w = sh[3 * k] * g[0] + sh[3 * k + 1] * g[1] + sh[3 * k + 2] * g[2];
      dx += gx[k] * w; dy += gy[k] * w; dz += gz[k] * w;
#pragma unroll
      for (int c = 0; c < 3; ++c) dsh[3 * k + c] = b[k] * g[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) dsh[3 * k + c] = 0.0f;      // coefficients above the active degree: no gradient
    }
  }
  // through dir = v / |v|:  (I - dir dir^T) / |v|
  const float dot = x * dx + y * dy + z * dz;
  dL_dmeans[3 * i] += (dx - x * dot) / len;
  dL_dmeans[3 * i + 1] += (dy - y * dot) / len;
  dL_dmeans[3 * i + 2] += (dz - z * dot) / len;
}

inline bool sh_shape_ok(int P, int deg, int M) { return P >= 0 && deg >= 0 && deg <= 3 && M >= (deg + 1) * (deg + 1); }

}  // namespace

extern "C" {

int ocrf_sh_to_rgb(int P, int deg, int max_coeffs, const float* means3D, const float* campos, const float* shs,
                   float* colors, unsigned char* clamped, ocrf_stream_t stream) {
  if (!sh_shape_ok(P, deg, max_coeffs)) return (int)hipErrorInvalidValue;
  if (P == 0) return 0;
  if (!means3D || !campos || !shs || !colors || !clamped) return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_RASTER_SH, sh_colors_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, deg,
               max_coeffs, means3D, campos, shs, colors, clamped);
  return (int)hipGetLastError();
}

int ocrf_sh_to_rgb_backward(int P, int deg, int max_coeffs, const float* means3D, const float* campos, const float* shs,
                            const unsigned char* clamped, const float* dL_dcolors, float* dL_dmeans3D, float* dL_dshs,
                            ocrf_stream_t stream) {
  if (!sh_shape_ok(P, deg, max_coeffs)) return (int)hipErrorInvalidValue;
  if (P == 0) return 0;
  if (!means3D || !campos || !shs || !clamped || !dL_dcolors || !dL_dmeans3D || !dL_dshs) return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_RASTER_SH_BWD, sh_colors_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream,
               P, deg, max_coeffs, means3D, campos, shs, clamped, dL_dcolors, dL_dmeans3D, dL_dshs);
  return (int)hipGetLastError();
}

}  // extern "C"

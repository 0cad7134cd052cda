// Height-aware Opacity-based Attention (HOA) reductions for MI355X (gfx950).
//
// Reference (pure torch, ~5 full passes over the BEV feature map plus ~12 tiny launches per
// HeightAttention): mmdet3d/models/necks/view_transformer_ocrf.py
//   ObatinOpacityMask.forward :236-242 and its application :1197-1199      (HOA-3)
//   HeightAttention.forward   :447-461, used as `ca(x) * x` :499-514        (inside HOA-2)
//
// HOA-3 here is two HBM-bound kernels: (1) channel mean+max in one read of x, (2) 7x7 conv of
// the two statistic planes + opacity_bev + sigmoid and the gate x*mask in one read + one write
// of x.  HeightAttention is (1) per-channel partial maxima, (2) final max + the four tiny 1x1
// MLPs + sigmoid recomputed per workgroup and applied to x in the same pass.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// (1) mean and max over channels: one thread per pixel, channel loop with coalesced rows.
__global__ __launch_bounds__(kBlock) void hoa_channel_stats_kernel(const float* __restrict__ x, int C,
                                                                   long plane, float* __restrict__ stats) {
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= plane) return;
  const float* p = x + (long)b * C * plane + pix;
  float sum = 0.f, mx = -INFINITY;
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    const float v0 = p[(long)c * plane], v1 = p[(long)(c + 1) * plane], v2 = p[(long)(c + 2) * plane],
                v3 = p[(long)(c + 3) * plane];
    sum += v0; sum += v1; sum += v2; sum += v3;         // channel order, like torch.mean's sum
    mx = fmaxf(fmaxf(mx, fmaxf(v0, v1)), fmaxf(v2, v3));
  }
  for (; c < C; ++c) {
    const float v = p[(long)c * plane];
    sum += v;
    mx = fmaxf(mx, v);
  }
  stats[((long)b * 2 + 0) * plane + pix] = sum / (float)C;
  stats[((long)b * 2 + 1) * plane + pix] = mx;
}

// (2) mask = sigmoid(conv_kxk([mean, max]) + opacity_bev); gated = x * mask.
__global__ __launch_bounds__(kBlock) void hoa_mask_gate_kernel(
    const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ opacity_bev,
    const float* __restrict__ conv_w, int k, int C, int Y, int X, float* __restrict__ mask,
    float* __restrict__ gated) {
  extern __shared__ float s_w[];          // 2*k*k conv weights
  for (int i = threadIdx.x; i < 2 * k * k; i += kBlock) s_w[i] = conv_w[i];
  __syncthreads();
  const long plane = (long)Y * X;
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= plane) return;
  const int yy = (int)(pix / X), xx = (int)(pix % X);
  const int r = k / 2;
  float acc = 0.f;
  for (int ch = 0; ch < 2; ++ch) {
    const float* sp = stats + ((long)b * 2 + ch) * plane;
    for (int i = 0; i < k; ++i) {
      const int y2 = yy + i - r;
      if (y2 < 0 || y2 >= Y) continue;
      for (int j = 0; j < k; ++j) {
        const int x2 = xx + j - r;
        if (x2 < 0 || x2 >= X) continue;
        acc = fmaf(sp[(long)y2 * X + x2], s_w[(ch * k + i) * k + j], acc);
      }
    }
  }
  const float m = sigmoidf_(acc + opacity_bev[(long)b * plane + pix]);
  mask[(long)b * plane + pix] = m;
  if (gated) {
    const float* p = x + (long)b * C * plane + pix;
    float* q = gated + (long)b * C * plane + pix;
    for (int c = 0; c < C; ++c) q[(long)c * plane] = p[(long)c * plane] * m;
  }
}

// HeightAttention (1): partial maxima, grid (n_split, B*C).
__global__ __launch_bounds__(kBlock) void hoa_height_max_kernel(const float* __restrict__ x, long plane,
                                                                int n_split, float* __restrict__ partial) {
  __shared__ float s_m[kBlock / 64];
  const long bc = blockIdx.y;
  const float* p = x + bc * plane;
  const long chunk = (plane + n_split - 1) / n_split;
  const long lo = (long)blockIdx.x * chunk, hi = min(lo + chunk, plane);
  float m = -INFINITY;
  for (long i = lo + threadIdx.x; i < hi; i += kBlock) m = fmaxf(m, p[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_m[w]);
    partial[bc * n_split + blockIdx.x] = m;
  }
}

// HeightAttention (2): final max per channel, the four quarter MLPs (1x1 -> ReLU -> 1x1, no
// bias) + sigmoid -> gate[b][c]; optionally gated = gate * x for this workgroup's pixels.
//   w1: [4][hid][q]   w2: [4][q][hid]   (q = C/4 channels per height quarter)
__global__ __launch_bounds__(kBlock) void hoa_height_gate_kernel(
    const float* __restrict__ x, int C, int hid, long plane, int n_split, const float* __restrict__ partial,
    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ gate,
    float* __restrict__ gated) {
  __shared__ float s_max[64];
  __shared__ float s_hid[64];
  __shared__ float s_gate[64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int q = C / 4;
  if (tid < C) {
    float m = -INFINITY;
    for (int s = 0; s < n_split; ++s) m = fmaxf(m, partial[((long)b * C + tid) * n_split + s]);
    s_max[tid] = m;
  }
  __syncthreads();
  if (tid < 4 * hid) {                       // hidden unit `h` of quarter `g`
    const int g = tid / hid, h = tid % hid;
    float a = 0.f;
    for (int i = 0; i < q; ++i) a = fmaf(w1[(g * hid + h) * q + i], s_max[g * q + i], a);
    s_hid[tid] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < C) {
    const int g = tid / q, o = tid % q;
    float a = 0.f;
    for (int h = 0; h < hid; ++h) a = fmaf(w2[(g * q + o) * hid + h], s_hid[g * hid + h], a);
    const float gt = sigmoidf_(a);
    s_gate[tid] = gt;
    if (blockIdx.x == 0) gate[(long)b * C + tid] = gt;
  }
  __syncthreads();
  if (!gated) return;
  const long pix = (long)blockIdx.x * kBlock + tid;
  if (pix >= plane) return;
  const float* p = x + (long)b * C * plane + pix;
  float* o = gated + (long)b * C * plane + pix;
  for (int c = 0; c < C; ++c) o[(long)c * plane] = p[(long)c * plane] * s_gate[c];
}

}  // namespace

extern "C" {

int ocrf_hoa_channel_stats(const float* x, int B, int C, int Y, int X, float* stats, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !stats || B <= 0 || C <= 0 || Y <= 0 || X <= 0) return (int)hipErrorInvalidValue;
  const long plane = (long)Y * X;
  ocrf::launch(OCRF_K_HOA_STATS, hoa_channel_stats_kernel, dim3((unsigned)((plane + kBlock - 1) / kBlock), B),
               dim3(kBlock), 0, stream, x, C, plane, stats);
  return (int)hipGetLastError();
}

int ocrf_hoa_opacity_mask_gate(const float* x, const float* stats, const float* opacity_bev,
                               const float* conv_w, int k, int B, int C, int Y, int X, float* mask,
                               float* gated, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !stats || !opacity_bev || !conv_w || !mask || k <= 0 || (k & 1) == 0 || k > 15 || B <= 0 ||
      C <= 0 || Y <= 0 || X <= 0)
    return (int)hipErrorInvalidValue;
  const long plane = (long)Y * X;
  ocrf::launch(OCRF_K_HOA_MASK_GATE, hoa_mask_gate_kernel, dim3((unsigned)((plane + kBlock - 1) / kBlock), B),
               dim3(kBlock), (size_t)2 * k * k * sizeof(float), stream, x, stats, opacity_bev, conv_w, k, C, Y,
               X, mask, gated);
  return (int)hipGetLastError();
}

size_t ocrf_hoa_height_attention_workspace_bytes(int B, int C) {
  return (size_t)B * C * 64 * sizeof(float);
}

int ocrf_hoa_height_attention(const float* x, int B, int C, int hid, int Y, int X, const float* w1,
                              const float* w2, float* gate, float* gated, void* workspace,
                              size_t workspace_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !w1 || !w2 || !gate || B <= 0 || C <= 0 || (C % 4) != 0 || C > 64 || hid <= 0 || 4 * hid > 64 ||
      Y <= 0 || X <= 0 || !workspace || workspace_bytes < ocrf_hoa_height_attention_workspace_bytes(B, C))
    return (int)hipErrorInvalidValue;
  const long plane = (long)Y * X;
  int n_split = (int)((plane + 4095) / 4096);
  if (n_split > 64) n_split = 64;
  float* partial = static_cast<float*>(workspace);
  ocrf::launch(OCRF_K_HOA_HEIGHT_MAX, hoa_height_max_kernel, dim3(n_split, B * C), dim3(kBlock), 0, stream, x,
               plane, n_split, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const unsigned gx = gated ? (unsigned)((plane + kBlock - 1) / kBlock) : 1u;
  ocrf::launch(OCRF_K_HOA_HEIGHT_GATE, hoa_height_gate_kernel, dim3(gx, B), dim3(kBlock), 0, stream, x, C, hid,
               plane, n_split, static_cast<const float*>(partial), w1, w2, gate, gated);
  return (int)hipGetLastError();
}

}  // extern "C"

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

#include <algorithm>

#include "hoa_gate.h"
#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;
int g_mg_groups = 0, g_mg_threads = kBlock, g_stats_threads = kBlock;       // ocrf::hoa_tune

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// (1) mean and max over channels: one thread per pixel, channel loop with coalesced rows.
__global__ __launch_bounds__(kBlock) void hoa_channel_stats_kernel(const float* __restrict__ x, int C,
                                                                   long plane, float* __restrict__ stats) {
  OCRF_MAIN_CHAIN_PRIO();
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= plane) return;
  const float* p = x + (long)b * C * plane + pix;
  float sum = 0.f, mx = -INFINITY;
  int c = 0;
  for (; c + 16 <= C; c += 16) {            // 16 plane reads in flight per lane (few workgroups: depth, not occupancy)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(long)(c + u) * plane];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      sum += v[u];                              // channel order, like torch.mean's sum
      mx = fmaxf(mx, v[u]);
    }
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
// One workgroup = 256 * VEC CONSECUTIVE pixels of the flattened (Y*X) plane (whole runs of every channel plane
// whatever X is: a 64-wide 2-D tile left a quarter of the workgroups 8 pixels wide at X = 200) and ONE GROUP of
// the channels (grid.z = batch x groups; every group recomputes the 2 x k x k taps of its pixels from the
// statistic rows staged in LDS, group 0 writes the mask).  VEC = 4 (X % 4 == 0): a thread owns 4 adjacent pixels
// of one row — 16-byte loads and stores of x / gated, and the k-wide windows of adjacent pixels share their reads.
// KT = k when it is known at compile time (7, the only size OcRFDet uses: the tap loops unroll and their LDS reads
// pipeline instead of paying one LDS latency per tap), 0 = runtime k.
// FAST (VEC = 4, X <= 256): the statistic rows are staged as 16-byte words, one row = 64 word slots (no index
// divisions: the generic staging spends ~ 80 integer divisions per thread, more than the convolution), at clamped
// addresses without a branch; a staged row carries 4 zero columns on the left (3 would do: 4 keeps the words aligned).
// D1 (with FAST): opacity_bev does not exist yet — it is HOA-2's output conv (view_transformer_ocrf.py:516) of the gated
// decoder1 activations, folded in here: `d1` (B,4,Y,X) + decoder1's per-tile channel maxima `d1_pm` -> the
// HeightAttention gate in the prologue (hoa_gate.h), opacity = out_b + sum_c fma(d1_c g_c, out_w_c, .) per pixel in the
// arithmetic of hoa_v2b_out_kernel; channel group 0 also writes it to `opacity_out`.  One launch (and its ~ 4.5 us floor
// plus the gap in front of it) less on the step's main chain; same bits.  d1_w: out_w[4], out_b, gate w1[4], w2[4].
struct MaskGateD1 {
  const float* d1; const float* pm; const float* out_w; const float* out_b; const float* g_w1; const float* g_w2;
  float* opacity_out; int tiles;
};
template <int VEC, int KT, bool FAST = false, bool D1 = false>
__global__ __launch_bounds__(kBlock) void hoa_mask_gate_kernel(
    const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ opacity_bev,
    const float* __restrict__ conv_w, int k_rt, int C, int Y, int X, int groups, int n_rows,
    float* __restrict__ mask, float* __restrict__ gated, MaskGateD1 dd) {
  OCRF_MAIN_CHAIN_PRIO();
  const int k = KT ? KT : k_rt;
  const int nt = (int)blockDim.x;              // 256, or fewer for more and shorter workgroups (ocrf_hoa_opacity_mask_gate)
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];      // weights, then 2 planes of n_rows x tw, zero padded
  // VEC = 4: rows padded to a multiple of 4 floats and the planes 16-byte aligned behind the weights — a thread's
  // (k + 3)-wide window is then three ds_read_b128 (lanes 16 B apart: conflict-free) instead of ten ds_read_b32 whose
  // lanes, 4 floats apart, hit 8 banks (5.1 bank conflicts per LDS instruction by PMC)
  const int r = k / 2, tw = FAST ? X + 8 : ((VEC == 4) ? ((X + k - 1 + 3) & ~3) : (X + k - 1));
  constexpr int PL = FAST ? 1 : 0;             // extra zero columns on the left of a staged row
  float* s_w = s_dyn;
  float* s_s = s_dyn + ((VEC == 4) ? ((2 * k * k + 3) & ~3) : 2 * k * k);
  const long plane = (long)Y * X;
  const int b = blockIdx.z / groups, g = blockIdx.z % groups;
  const long p0 = (long)blockIdx.x * nt * VEC;
  const int row0 = (int)(p0 / X) - r;                      // first staged row (may be negative: zeros)
  for (int i = threadIdx.x; i < 2 * k * k; i += nt) s_w[i] = conv_w[i];
  // the first batch of this thread's x values is on its way while the mask is computed
  const long pix_e = p0 + (long)threadIdx.x * VEC;
  const int cpg_e = (C + groups - 1) / groups;
  const int c0_e = g * cpg_e, c1_e = min(C, c0_e + cpg_e);
  float4 first[5];
  const bool pre = (VEC == 4) && gated && pix_e < plane && c0_e + 5 <= c1_e;
  if constexpr (VEC == 4) {
    if (pre) {
      const float* pe = x + (long)b * C * plane + pix_e;
#pragma unroll
      for (int u = 0; u < 5; ++u) first[u] = *reinterpret_cast<const float4*>(pe + (long)(c0_e + u) * plane);
    }
  }
  // D1: decoder1's activations of this thread's four pixels, its per-tile maxima and the 13 weights, all in flight now
  __shared__ __attribute__((aligned(16))) float s_red[16 * 4];
  __shared__ float s_g[4], s_ow[16];
  float4 d1v[D1 ? 4 : 1];
  float4 pmv[hoa_gate::kPmRounds];
  float owv = 0.f;
  if constexpr (D1) {
    const long pc = pix_e < plane ? pix_e : plane - VEC;
#pragma unroll
    for (int c = 0; c < 4; ++c) d1v[c] = *reinterpret_cast<const float4*>(dd.d1 + ((long)b * 4 + c) * plane + pc);
    hoa_gate::pm_issue<4>(dd.pm, b, dd.tiles, (int)threadIdx.x, pmv);
    const int t = (int)threadIdx.x & 15;                       // out_w[0..3] | out_b | w1[0..3] | w2[0..3]
    const float* src = t < 4 ? dd.out_w + t : (t == 4 ? dd.out_b : (t < 9 ? dd.g_w1 + (t - 5) : dd.g_w2 + (min(t, 12) - 9)));
    owv = *src;
  }
  if constexpr (FAST) {
    const int n_items = 2 * n_rows * 64, x4 = X >> 2;
    float4* s_s4 = reinterpret_cast<float4*>(s_s);
    const int tw4 = tw >> 2;
    for (int i0 = 0; i0 < n_items; i0 += 8 * nt) {       // one trip at 200 x 200 (26 rows)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u * nt + (int)threadIdx.x, n_items - 1);
        const int rp = i >> 6, c4 = min(i & 63, x4 - 1);
        const int ch = rp >= n_rows ? 1 : 0, y = row0 + rp - ch * n_rows;
        v[u] = *reinterpret_cast<const float4*>(stats + ((long)b * 2 + ch) * plane + (long)min(max(y, 0), Y - 1) * X + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nt + (int)threadIdx.x;
        const int rp = i >> 6, c4 = i & 63;
        const int ch = rp >= n_rows ? 1 : 0, y = row0 + rp - ch * n_rows;
        const bool in = y >= 0 && y < Y;
        if (i < n_items && c4 < x4) s_s4[rp * tw4 + 1 + c4] = in ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (int i = threadIdx.x; i < 2 * n_rows * 2; i += nt)      // the zero words left and right of every row
      s_s4[(i >> 1) * tw4 + ((i & 1) ? 1 + x4 : 0)] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
  // the statistic rows: all of a thread's loads are issued before any is stored
  const int n_stage = 2 * n_rows * tw;
  for (int i0 = 0; i0 < n_stage; i0 += 8 * nt) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nt + threadIdx.x;
      const int ch = i / (n_rows * tw), rem = i % (n_rows * tw);
      const int y = row0 + rem / tw, xx = rem % tw - r;
      v[u] = (i < n_stage && y >= 0 && y < Y && xx >= 0 && xx < X)
                 ? stats[((long)b * 2 + ch) * plane + (long)y * X + xx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nt + threadIdx.x;
      if (i < n_stage) s_s[i] = v[u];
    }
  }
  }
  if constexpr (D1) {
    if (threadIdx.x < 16) s_ow[threadIdx.x] = owv;
    hoa_gate::pm_reduce_rows<4>(dd.pm, b, dd.tiles, (int)threadIdx.x, pmv, s_red);
  }
  __syncthreads();
  if constexpr (D1) {
    if (threadIdx.x < 64) {                                   // decoder1's HeightAttention gate (q = hid = 1)
      const float gt = hoa_gate::gate_of_lane<4>(s_red, s_ow + 5, s_ow + 9, (int)threadIdx.x);
      if (threadIdx.x < 4) s_g[threadIdx.x] = gt;
    }
    __syncthreads();
  }
  const long pix = p0 + (long)threadIdx.x * VEC;
  if (pix >= plane) return;
  const int yy = (int)(pix / X), xx = (int)(pix % X);
  const int ly = yy - row0 - r;                            // staged row of (yy - r) is ly
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if constexpr (KT > 0) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const float* row = s_s + (ch * n_rows + ly + i) * tw + xx;
        const float* wr = s_w + (ch * KT + i) * KT;
        float win[(KT + VEC - 1 + 3) & ~3];               // the k-wide windows of adjacent pixels overlap
        if constexpr (VEC == 4) {
          const float4* row4 = reinterpret_cast<const float4*>(row);      // xx, tw and the plane base: multiples of 4 floats
#pragma unroll
          for (int j = 0; j < (KT + VEC - 1 + 3) / 4; ++j) {
            const float4 q4 = row4[j];
            win[4 * j] = q4.x; win[4 * j + 1] = q4.y; win[4 * j + 2] = q4.z; win[4 * j + 3] = q4.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < KT + VEC - 1; ++j) win[j] = row[j];
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const float w = wr[j];
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = fmaf(win[j + e + PL], w, acc[e]);
        }
      }
    }
  } else {
    for (int ch = 0; ch < 2; ++ch)
      for (int i = 0; i < k; ++i) {
        const float* row = s_s + (ch * n_rows + ly + i) * tw + xx;
        const float* wr = s_w + (ch * k + i) * k;
        for (int j = 0; j < k; ++j) {
          const float w = wr[j];
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = fmaf(row[j + e], w, acc[e]);
        }
      }
  }
  float m[VEC];
  if constexpr (D1) {
    const float dv[4][4] = {{d1v[0].x, d1v[0].y, d1v[0].z, d1v[0].w}, {d1v[1].x, d1v[1].y, d1v[1].z, d1v[1].w},
                            {d1v[2].x, d1v[2].y, d1v[2].z, d1v[2].w}, {d1v[3].x, d1v[3].y, d1v[3].z, d1v[3].w}};
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float o = s_ow[4];                                       // hoa_v2b_out_kernel's arithmetic
#pragma unroll
      for (int c = 0; c < 4; ++c) o = fmaf(dv[c][e] * s_g[c], s_ow[c], o);
      if (g == 0) dd.opacity_out[(long)b * plane + pix + e] = o;
      m[e] = sigmoidf_(acc[e] + o);
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = sigmoidf_(acc[e] + opacity_bev[(long)b * plane + pix + e]);
  }
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) mask[(long)b * plane + pix + e] = m[e];
  }
  if (gated) {
    const int cpg = (C + groups - 1) / groups;
    const int c0 = g * cpg, c1 = min(C, c0 + cpg);
    const float* p = x + (long)b * C * plane + pix;
    float* q = gated + (long)b * C * plane + pix;
    if constexpr (VEC == 4) {
      int c = c0;
      if (pre) {
#pragma unroll
        for (int u = 0; u < 5; ++u)
          *reinterpret_cast<float4*>(q + (long)(c + u) * plane) =
              make_float4(first[u].x * m[0], first[u].y * m[1], first[u].z * m[2], first[u].w * m[3]);
        c += 5;
      }
      for (; c + 5 <= c1; c += 5) {                        // 5 x 16 B in flight per lane
        float4 v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(c + u) * plane);
#pragma unroll
        for (int u = 0; u < 5; ++u)
          *reinterpret_cast<float4*>(q + (long)(c + u) * plane) =
              make_float4(v[u].x * m[0], v[u].y * m[1], v[u].z * m[2], v[u].w * m[3]);
      }
      for (; c < c1; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long)c * plane);
        *reinterpret_cast<float4*>(q + (long)c * plane) = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
      }
    } else {
      int c = c0;
      for (; c + 10 <= c1; c += 10) {                      // 10 plane reads in flight per lane
        float v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = p[(long)(c + u) * plane];
#pragma unroll
        for (int u = 0; u < 10; ++u) q[(long)(c + u) * plane] = v[u] * m[0];
      }
      for (; c < c1; ++c) q[(long)c * plane] = p[(long)c * plane] * m[0];
    }
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

// Depthwise 3x3 convolution (padding 1, stride 1) for the TRAINING path of HOA-2's UNet blocks
// (view_transformer_ocrf.py:483-489 conv_block's first layer): MIOpen's depthwise backward-weight runs
// 2-10 ms per call on these 4..16-channel maps (26 ms of an 84 ms neck forward + backward at 200x200).
// One kernel serves the forward and the input gradient (the caller passes the flipped weights, no bias);
// the weight / bias gradient is a band-wise partial reduction the caller sums (deterministic).
constexpr int kDwBand = 16;
__global__ __launch_bounds__(kBlock) void hoa_dw3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int C, int Y, int X,
                                                           float* __restrict__ y) {
  const int bc = blockIdx.z, c = bc % C;
  const int xx = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (xx >= X || yy >= Y) return;
  const float* p = x + (long)bc * Y * X;
  float acc = bias ? bias[c] : 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sy = yy + i - 1;
    if (sy < 0 || sy >= Y) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sx = xx + j - 1;
      if (sx < 0 || sx >= X) continue;
      acc = fmaf(w[c * 9 + i * 3 + j], p[(long)sy * X + sx], acc);
    }
  }
  y[(long)bc * Y * X + (long)yy * X + xx] = acc;
}

// partial[(bc * n_bands + band) * 10 + t]: t < 9 -> sum over the band of dy[y][x] * x[y+i-1][x+j-1], t = 9 -> sum dy
__global__ __launch_bounds__(kBlock) void hoa_dw3x3_wgrad_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy, int Y, int X,
                                                                 float* __restrict__ partial) {
  __shared__ float s_acc[kBlock / 64][10];
  const int bc = blockIdx.y, band = blockIdx.x;
  const float* px = x + (long)bc * Y * X;
  const float* pg = dy + (long)bc * Y * X;
  const int y0 = band * kDwBand, y1 = min(y0 + kDwBand, Y);
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  for (int idx = threadIdx.x; idx < (y1 - y0) * X; idx += kBlock) {
    const int yy = y0 + idx / X, xx = idx % X;
    const float g = pg[(long)yy * X + xx];
    acc[9] += g;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int sy = yy + i - 1;
      if (sy < 0 || sy >= Y) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int sx = xx + j - 1;
        if (sx < 0 || sx >= X) continue;
        acc[i * 3 + j] = fmaf(g, px[(long)sy * X + sx], acc[i * 3 + j]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float v = acc[t];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    float v = 0.f;
    for (int wv = 0; wv < kBlock / 64; ++wv) v += s_acc[wv][threadIdx.x];
    partial[((long)bc * gridDim.x + band) * 10 + threadIdx.x] = v;
  }
}

}  // namespace

namespace ocrf {
// diagnostic knobs reached through ocrf_tune_set (keys 20-29): 20 = channel groups of the mask-gate launch (0 = by C),
// 21 = threads per workgroup of it (64 | 128 | 256), 22 = threads per workgroup of the channel statistics
void hoa_tune(int key, int value) {
  if (key == 20 && value >= 0) g_mg_groups = value;
  if (key == 21 && (value == 64 || value == 128 || value == 256)) g_mg_threads = value;
  if (key == 22 && (value == 64 || value == 128 || value == 256)) g_stats_threads = value;
}
}  // namespace ocrf

extern "C" {

int ocrf_hoa_dw3x3(const float* x, const float* w, const float* bias, int B, int C, int Y, int X, float* y,
                   ocrf_stream_t stream_) {
  if (!x || !w || !y || B <= 0 || C <= 0 || Y <= 0 || X <= 0 || (long)B * C > 65535) return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_HOA_DW3X3, hoa_dw3x3_kernel, dim3((X + 63) / 64, (Y + 3) / 4, B * C), dim3(kBlock), 0,
               static_cast<hipStream_t>(stream_), x, w, bias, C, Y, X, y);
  return (int)hipGetLastError();
}

int ocrf_hoa_dw3x3_wgrad_bands(int Y) { return (Y + kDwBand - 1) / kDwBand; }

int ocrf_hoa_dw3x3_wgrad(const float* x, const float* dy, int B, int C, int Y, int X, float* partial,
                         ocrf_stream_t stream_) {
  if (!x || !dy || !partial || B <= 0 || C <= 0 || Y <= 0 || X <= 0 || (long)B * C > 65535)
    return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_HOA_DW3X3_WGRAD, hoa_dw3x3_wgrad_kernel, dim3(ocrf_hoa_dw3x3_wgrad_bands(Y), B * C),
               dim3(kBlock), 0, static_cast<hipStream_t>(stream_), x, dy, Y, X, partial);
  return (int)hipGetLastError();
}

int ocrf_hoa_channel_stats(const float* x, int B, int C, int Y, int X, float* stats, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !stats || B <= 0 || C <= 0 || Y <= 0 || X <= 0) return (int)hipErrorInvalidValue;
  const long plane = (long)Y * X;
  const int nt = g_stats_threads;
  ocrf::launch(OCRF_K_HOA_STATS, hoa_channel_stats_kernel, dim3((unsigned)((plane + nt - 1) / nt), B), dim3(nt), 0, stream,
               x, C, plane, stats);
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
  const bool vec4 = (X % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gated)) & 15u) == 0 &&
                    (X + 6) * 14 * 2 * 4 <= 60 * 1024;
  const int threads = g_mg_threads;
  const int per_wg = threads * (vec4 ? 4 : 1);
  // rows a run of per_wg consecutive pixels can touch (it may start mid-row) plus the k - 1 halo rows
  const int n_rows = (per_wg + X - 2) / X + 1 + (k - 1);
  const bool fast = vec4 && k == 7 && X <= 256 && (reinterpret_cast<uintptr_t>(stats) & 15u) == 0;
  const int tw = fast ? X + 8 : (vec4 ? ((X + k - 1 + 3) & ~3) : (X + k - 1));
  const size_t lds = (size_t)((vec4 ? ((2 * k * k + 3) & ~3) : 2 * k * k) + 2 * n_rows * tw) * sizeof(float);
  if (lds > 64 * 1024) return (int)hipErrorInvalidValue;             // X beyond ~1 000: not a BEV plane
  int groups = gated ? ((C >= 40) ? 4 : (C >= 16 ? 2 : 1)) : 1;
  if (gated && g_mg_groups > 0) groups = std::min(g_mg_groups, C);
  const dim3 grid((unsigned)((plane + per_wg - 1) / per_wg), 1, (unsigned)(B * groups));
#define OCRF_MASK_GATE(V, K, F)                                                                                       \
  ocrf::launch(OCRF_K_HOA_MASK_GATE, hoa_mask_gate_kernel<V, K, F, false>, grid, dim3(threads), lds, stream, x, stats, opacity_bev, \
               conv_w, k, C, Y, X, groups, n_rows, mask, gated, MaskGateD1{})
  if (fast) OCRF_MASK_GATE(4, 7, true);
  else if (vec4 && k == 7) OCRF_MASK_GATE(4, 7, false);
  else if (vec4) OCRF_MASK_GATE(4, 0, false);
  else if (k == 7) OCRF_MASK_GATE(1, 7, false);
  else OCRF_MASK_GATE(1, 0, false);
#undef OCRF_MASK_GATE
  return (int)hipGetLastError();
}

// (Rounds 4-5 also exported ocrf_hoa_opacity_mask_gate_v2b: HOA-2's output conv folded into this gate through the kernel's
// D1 form — one launch less on the chain, but 288 VGPRs and not faster (DESIGN, round 4); nothing selected it, so the entry
// point and its instantiation are gone.  The D1 branches of hoa_mask_gate_kernel are not instantiated.)

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

// =================================================================================================
// HOA-2: OpacityVoxelToBEVConverter (view_transformer_ocrf.py:463-518) as fused UNet blocks.
//
// One kernel = one `conv_block` (depthwise 3x3 -> 1x1 -> BatchNorm(eval) -> ReLU, :485-491) with
// everything around it folded in:
//   * input transform: identity | 2x2 max-pool of the producer (:501,504) | ConvTranspose2d(k=2,s=2)
//     of the producer concatenated with a skip tensor (:507-509, :512-514);
//   * the HeightAttention gate of each producer (`ca(x) * x`, :500-515) applied while reading it;
//   * optional addend after ReLU (the positional encoding, :498);
//   * per-channel partial maxima of the result for the block's own HeightAttention.
// The virtual input tile (16x16 + halo) is built once in LDS; channel counts are <= 16.
// The reference runs this as ~60 MIOpen/elementwise launches on (B,<=16,Y,X) tensors.
// =================================================================================================
namespace {

constexpr int kUT = 16;                  // tile edge
constexpr int kUH = kUT + 2;             // with halo
constexpr int kUMaxC = 16;

struct UnetBlockArgs {
  const float* src0; const float* gate0; int C0, H0, W0; int mode;   // 0 identity, 1 maxpool2, 2 upconv k2s2
  const float* up_w; const float* up_b; int Cup;                      // mode 2: (C0,Cup,2,2), (Cup)
  const float* src1; const float* gate1; int C1;                      // skip tensor (B,C1,H,W) or null
  const float* dw_w; const float* dw_b;                               // (Cin,3,3), (Cin)
  const float* pw_w; const float* pw_b; int Cout;                     // BN-folded (Cout,Cin), (Cout)
  const float* addend;                                                // (B,Cout,H,W) or null
  float* out; float* partial_max; int H, W, tiles_x, tiles_y;         // partial_max (B*Cout, tiles) or null
  // gate of src0 / src1 computed IN this kernel from the producer's per-tile channel maxima (pm != null; else the
  // gate0 / gate1 vectors above, else 1): HeightAttention = sigmoid(w2 relu(w1 maxpool)) per channel quarter
  const float* pm0; const float* g0w1; const float* g0w2; int g0hid, g0tiles;
  const float* pm1; const float* g1w1; const float* g1w2; int g1hid, g1tiles;
};

// HeightAttention gate of one batch entry from per-tile maxima, by the whole workgroup, into s_gate[C]
// (view_transformer_ocrf.py:421-461: global max-pool per channel, per quarter q -> hid -> q, sigmoid).  Every
// consumer workgroup redoes this tiny reduction (C <= 16 channels x <= a few hundred tiles from L2) instead of
// waiting for a separate launch between every two blocks of the UNet.  Contains barriers: call uniformly.
__device__ __forceinline__ void gate_from_tiles_dev(int b, int C, int hid, int n_tiles, const float* __restrict__ partial,
                                                     const float* __restrict__ w1, const float* __restrict__ w2,
                                                     float* s_gate, float* s_max, float* s_hid) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = C / 4;
  for (int c = wave; c < C; c += kBlock / 64) {
    float m = -INFINITY;
    for (int t = lane; t < n_tiles; t += 64) m = fmaxf(m, partial[((long)b * C + c) * n_tiles + t]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    if (lane == 0) s_max[c] = m;
  }
  __syncthreads();
  if (tid < 4 * hid) {
    const int g = tid / hid, h = tid % hid;
    float acc = 0.f;
    for (int i = 0; i < q; ++i) acc = fmaf(w1[(g * hid + h) * q + i], s_max[g * q + i], acc);
    s_hid[tid] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  if (tid < C) {
    const int g = tid / q, o = tid % q;
    float acc = 0.f;
    for (int h = 0; h < hid; ++h) acc = fmaf(w2[(g * q + o) * hid + h], s_hid[g * hid + h], acc);
    s_gate[tid] = sigmoidf_(acc);
  }
  __syncthreads();
}

// s_gate[C] = the gate the kernel multiplies a source with: from tiles, from a vector, or 1
__device__ __forceinline__ void load_gate_dev(int b, int C, const float* gate_vec, const float* pm, const float* w1,
                                              const float* w2, int hid, int n_tiles, float* s_gate, float* s_max,
                                              float* s_hid) {
  if (pm) {
    gate_from_tiles_dev(b, C, hid, n_tiles, pm, w1, w2, s_gate, s_max, s_hid);
  } else {
    if ((int)threadIdx.x < C) s_gate[threadIdx.x] = gate_vec ? gate_vec[b * C + threadIdx.x] : 1.f;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void hoa_unet_block_kernel(UnetBlockArgs a) {
  __shared__ float s_v[kUMaxC][kUH][kUH + 1];
  __shared__ float s_red[kBlock / 64][kUMaxC];
  __shared__ float s_upw[kUMaxC * kUMaxC * 4], s_upb[kUMaxC], s_dww[kUMaxC * 9], s_dwb[kUMaxC],
      s_pww[kUMaxC * kUMaxC], s_pwb[kUMaxC], s_g0[kUMaxC], s_g1[kUMaxC];
  const int tid = threadIdx.x;
  {
    const int Cf = (a.mode == 2) ? a.Cup : a.C0, Ci = Cf + a.C1;
    if (a.mode == 2) {
      for (int i = tid; i < a.C0 * a.Cup * 4; i += kBlock) s_upw[i] = a.up_w[i];
      if (tid < a.Cup) s_upb[tid] = a.up_b[tid];
    }
    for (int i = tid; i < Ci * 9; i += kBlock) s_dww[i] = a.dw_w[i];
    for (int i = tid; i < a.Cout * Ci; i += kBlock) s_pww[i] = a.pw_w[i];
    if (tid < Ci) s_dwb[tid] = a.dw_b[tid];
    if (tid < a.Cout) s_pwb[tid] = a.pw_b[tid];
  }
  __syncthreads();
  {
    __shared__ float s_gm[64], s_gh[64];
    load_gate_dev(blockIdx.z, a.C0, a.gate0, a.pm0, a.g0w1, a.g0w2, a.g0hid, a.g0tiles, s_g0, s_gm, s_gh);
    if (a.C1 > 0) load_gate_dev(blockIdx.z, a.C1, a.gate1, a.pm1, a.g1w1, a.g1w2, a.g1hid, a.g1tiles, s_g1, s_gm, s_gh);
  }
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * kUT, tx0 = blockIdx.x * kUT;
  const int Cfirst = (a.mode == 2) ? a.Cup : a.C0;
  const int Cin = Cfirst + a.C1;
  const int H = a.H, W = a.W;

  // (1) virtual input tile with halo, zero outside the image (conv padding = 1)
  for (int i = tid; i < kUH * kUH; i += kBlock) {
    const int hy = i / kUH, hx = i % kUH;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool in = (y >= 0) && (y < H) && (x >= 0) && (x < W);
    // all channel loads of a position are issued before any is used (fully unrolled, predicated):
    // a runtime-bounded loop would pay one L2 latency per channel
    float sv[kUMaxC];
    if (a.mode == 2) {
      const int sy = in ? (y >> 1) : 0, sx = in ? (x >> 1) : 0, ki = (y & 1) * 2 + (x & 1);
#pragma unroll
      for (int ci = 0; ci < kUMaxC; ++ci)
        sv[ci] = (in && ci < a.C0) ? a.src0[(((long)b * a.C0 + ci) * a.H0 + sy) * a.W0 + sx] * s_g0[ci] : 0.f;
      float acc[kUMaxC];
#pragma unroll
      for (int co = 0; co < kUMaxC; ++co) acc[co] = (co < a.Cup) ? s_upb[co] : 0.f;
#pragma unroll
      for (int ci = 0; ci < kUMaxC; ++ci) {
        if (ci < a.C0) {
#pragma unroll
          for (int co = 0; co < kUMaxC; ++co)
            if (co < a.Cup) acc[co] = fmaf(sv[ci], s_upw[(ci * a.Cup + co) * 4 + ki], acc[co]);
        }
      }
#pragma unroll
      for (int co = 0; co < kUMaxC; ++co)
        if (co < a.Cup) s_v[co][hy][hx] = in ? acc[co] : 0.f;
    } else if (a.mode == 1) {
#pragma unroll
      for (int c = 0; c < kUMaxC; ++c) {
        float v = 0.f;
        if (in && c < a.C0) {
          const float* q = a.src0 + ((long)b * a.C0 + c) * a.H0 * a.W0 + (long)(2 * y) * a.W0 + 2 * x;
          // gates are sigmoids (> 0), so max-pooling before or after the multiply is the same
          v = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[a.W0], q[a.W0 + 1])) * s_g0[c];
        }
        sv[c] = v;
      }
#pragma unroll
      for (int c = 0; c < kUMaxC; ++c)
        if (c < a.C0) s_v[c][hy][hx] = sv[c];
    } else {
#pragma unroll
      for (int c = 0; c < kUMaxC; ++c)
        sv[c] = (in && c < a.C0) ? a.src0[((long)b * a.C0 + c) * a.H0 * a.W0 + (long)y * a.W0 + x] * s_g0[c] : 0.f;
#pragma unroll
      for (int c = 0; c < kUMaxC; ++c)
        if (c < a.C0) s_v[c][hy][hx] = sv[c];
    }
    for (int c = 0; c < a.C1; ++c) {
      float v = 0.f;
      if (in) {
        v = a.src1[(((long)b * a.C1 + c) * H + y) * W + x] * s_g1[c];
      }
      s_v[Cfirst + c][hy][hx] = v;
    }
  }
  __syncthreads();

  // (2) depthwise 3x3 -> 1x1 (BN folded) -> ReLU (+ addend), one pixel per thread
  const int ly = tid / kUT, lx = tid % kUT;
  const int y = ty0 + ly, x = tx0 + lx;
  const bool valid = (y < H) && (x < W);
  float outv[kUMaxC];
#pragma unroll
  for (int co = 0; co < kUMaxC; ++co) outv[co] = (co < a.Cout) ? s_pwb[co] : 0.f;
  for (int c = 0; c < Cin; ++c) {
    float d = s_dwb[c];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) d = fmaf(s_v[c][ly + i][lx + j], s_dww[c * 9 + i * 3 + j], d);
#pragma unroll
    for (int co = 0; co < kUMaxC; ++co)
      if (co < a.Cout) outv[co] = fmaf(d, s_pww[co * Cin + c], outv[co]);
  }
  const long plane = (long)H * W;
  const long pix = (long)y * W + x;
#pragma unroll
  for (int co = 0; co < kUMaxC; ++co) {
    if (co < a.Cout) {
      float v = fmaxf(outv[co], 0.f);
      if (valid) {
        if (a.addend) v += a.addend[((long)b * a.Cout + co) * plane + pix];
        a.out[((long)b * a.Cout + co) * plane + pix] = v;
      } else {
        v = -INFINITY;
      }
      outv[co] = v;
    }
  }
  // (3) per-channel maximum of this tile for the block's HeightAttention
  if (a.partial_max) {
#pragma unroll
    for (int co = 0; co < kUMaxC; ++co) {
      if (co < a.Cout) {
        float m = outv[co];
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
        if ((tid & 63) == 0) s_red[tid >> 6][co] = m;
      }
    }
    __syncthreads();
    if (tid < a.Cout) {
      float m = s_red[0][tid];
      for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_red[w][tid]);
      const int n_tiles = a.tiles_x * a.tiles_y;
      a.partial_max[((long)b * a.Cout + tid) * n_tiles + blockIdx.y * a.tiles_x + blockIdx.x] = m;
    }
  }
}

// Compile-time-shaped variant of hoa_unet_block_kernel for the five blocks of the reference UNet:
// channel loops unroll completely, so every weight index is a constant and the weights arrive through
// the scalar cache into SGPRs (no LDS traffic, no per-channel LDS latency chain) — the generic
// kernel above spends its time waiting on serialized LDS reads inside runtime-bounded loops.
template <int MODE, int C0, int CUP, int C1, int COUT>
__global__ __launch_bounds__(kBlock) void hoa_unet_block_fixed_kernel(UnetBlockArgs a) {
  constexpr int CF = (MODE == 2) ? CUP : C0;
  constexpr int CIN = CF + C1;
  __shared__ float s_v[CIN][kUH][kUH + 1];
  __shared__ float s_red[kBlock / 64][COUT];
  __shared__ float s_upw[(MODE == 2) ? C0 * CUP * 4 : 1];   // the 2x2 tap is per lane: vector (LDS) reads
  __shared__ float s_g0[C0], s_g1[C1 > 0 ? C1 : 1], s_gm[64], s_gh[64];
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * kUT, tx0 = blockIdx.x * kUT;
  const int H = a.H, W = a.W;
  if (MODE == 2) {
    for (int i = tid; i < C0 * CUP * 4; i += kBlock) s_upw[i] = a.up_w[i];
  }
  // The raw values of this thread's (at most two) halo pixels are requested BEFORE the producers' gates are rebuilt
  // from their per-tile maxima: both are global round trips and a block is a latency chain (16-338 workgroups).
  constexpr int kIt = (kUH * kUH + kBlock - 1) / kBlock;
  float raw0[kIt][C0], raw1[kIt][C1 > 0 ? C1 : 1];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = tid + it * kBlock;
    const int hy = i / kUH, hx = i % kUH;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool in = (i < kUH * kUH) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      float v = 0.f;
      if (in) {
        const float* p = a.src0 + ((long)b * C0 + c) * a.H0 * a.W0;
        if (MODE == 2) {
          v = p[(long)(y >> 1) * a.W0 + (x >> 1)];
        } else if (MODE == 1) {
          const float* q = p + (long)(2 * y) * a.W0 + 2 * x;
          v = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[a.W0], q[a.W0 + 1]));
        } else {
          v = p[(long)y * a.W0 + x];
        }
      }
      raw0[it][c] = v;
    }
#pragma unroll
    for (int c = 0; c < C1; ++c) raw1[it][c] = in ? a.src1[(((long)b * C1 + c) * H + y) * W + x] : 0.f;
  }
  load_gate_dev(b, C0, a.gate0, a.pm0, a.g0w1, a.g0w2, a.g0hid, a.g0tiles, s_g0, s_gm, s_gh);
  if (C1 > 0) load_gate_dev(b, C1, a.gate1, a.pm1, a.g1w1, a.g1w2, a.g1hid, a.g1tiles, s_g1, s_gm, s_gh);

#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = tid + it * kBlock;
    if (i >= kUH * kUH) continue;
    const int hy = i / kUH, hx = i % kUH;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool in = (y >= 0) && (y < H) && (x >= 0) && (x < W);
    if (MODE == 2) {
      const int ki = (y & 1) * 2 + (x & 1);
      float sv[C0];
#pragma unroll
      for (int ci = 0; ci < C0; ++ci) sv[ci] = raw0[it][ci] * s_g0[ci];
#pragma unroll
      for (int co = 0; co < CUP; ++co) {
        float acc = a.up_b[co];
#pragma unroll
        for (int ci = 0; ci < C0; ++ci) acc = fmaf(sv[ci], s_upw[(ci * CUP + co) * 4 + ki], acc);
        s_v[co][hy][hx] = in ? acc : 0.f;
      }
    } else {
#pragma unroll
      for (int c = 0; c < C0; ++c) s_v[c][hy][hx] = in ? raw0[it][c] * s_g0[c] : 0.f;      // gates > 0: commute with the max-pool
    }
#pragma unroll
    for (int c = 0; c < C1; ++c) s_v[CF + c][hy][hx] = in ? raw1[it][c] * s_g1[c] : 0.f;
  }
  __syncthreads();

  const int ly = tid / kUT, lx = tid % kUT;
  const int y = ty0 + ly, x = tx0 + lx;
  const bool valid = (y < H) && (x < W);
  float outv[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) outv[co] = a.pw_b[co];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    float d = a.dw_b[c];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) d = fmaf(s_v[c][ly + i][lx + j], a.dw_w[c * 9 + i * 3 + j], d);
#pragma unroll
    for (int co = 0; co < COUT; ++co) outv[co] = fmaf(d, a.pw_w[co * CIN + c], outv[co]);
  }
  const long plane = (long)H * W;
  const long pix = (long)y * W + x;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float v = fmaxf(outv[co], 0.f);
    if (valid) {
      if (a.addend) v += a.addend[((long)b * COUT + co) * plane + pix];
      a.out[((long)b * COUT + co) * plane + pix] = v;
    } else {
      v = -INFINITY;
    }
    outv[co] = v;
  }
  if (a.partial_max) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float m = outv[co];
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
      if ((tid & 63) == 0) s_red[tid >> 6][co] = m;
    }
    __syncthreads();
    if (tid < COUT) {
      float m = s_red[0][tid];
      for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_red[w][tid]);
      const int n_tiles = a.tiles_x * a.tiles_y;
      a.partial_max[((long)b * COUT + tid) * n_tiles + blockIdx.y * a.tiles_x + blockIdx.x] = m;
    }
  }
}

// HeightAttention gate from per-tile maxima (any number of tiles): one workgroup per batch entry,
// one wave per channel for the max (no barrier per channel).
__global__ __launch_bounds__(kBlock) void hoa_height_gate_from_tiles_kernel(
    int C, int hid, int n_tiles, const float* __restrict__ partial, const float* __restrict__ w1,
    const float* __restrict__ w2, float* __restrict__ gate) {
  __shared__ float s_max[64];
  __shared__ float s_hid[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = C / 4;
  for (int c = wave; c < C; c += kBlock / 64) {
    float m = -INFINITY;
    for (int t = lane; t < n_tiles; t += 64) m = fmaxf(m, partial[((long)b * C + c) * n_tiles + t]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    if (lane == 0) s_max[c] = m;
  }
  __syncthreads();
  if (tid < 4 * hid) {
    const int g = tid / hid, h = tid % hid;
    float acc = 0.f;
    for (int i = 0; i < q; ++i) acc = fmaf(w1[(g * hid + h) * q + i], s_max[g * q + i], acc);
    s_hid[tid] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  if (tid < C) {
    const int g = tid / q, o = tid % q;
    float acc = 0.f;
    for (int h = 0; h < hid; ++h) acc = fmaf(w2[(g * q + o) * hid + h], s_hid[g * hid + h], acc);
    gate[(long)b * C + tid] = sigmoidf_(acc);
  }
}

// Final 1x1 output conv (:516) over the gated decoder output; the gate as a vector or from per-tile maxima.
__global__ __launch_bounds__(kBlock) void hoa_gated_conv1x1_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gate, int C,
                                                                   long plane, const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ out, const float* pm,
                                                                   const float* gw1, const float* gw2, int ghid,
                                                                   int gtiles) {
  __shared__ float s_g[64], s_gm[64], s_gh[64];
  const int b = blockIdx.y;
  load_gate_dev(b, C, gate, pm, gw1, gw2, ghid, gtiles, s_g, s_gm, s_gh);
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  if (pix >= plane) return;
  float acc = bias[0];
  for (int c = 0; c < C; ++c) acc = fmaf(x[((long)b * C + c) * plane + pix] * s_g[c], w[c], acc);
  out[(long)b * plane + pix] = acc;
}

int launch_unet_block(const UnetBlockArgs& a, int B, hipStream_t stream);

}  // namespace

extern "C" {

int ocrf_hoa_unet_block(const float* src0, const float* gate0, int C0, int H0, int W0, int mode,
                        const float* up_w, const float* up_b, int Cup, const float* src1,
                        const float* gate1, int C1, const float* dw_w, const float* dw_b,
                        const float* pw_w, const float* pw_b, int Cout, const float* addend, float* out,
                        float* partial_max, int B, int H, int W, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int Cfirst = (mode == 2) ? Cup : C0;
  if (!src0 || !dw_w || !dw_b || !pw_w || !pw_b || !out || B <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2 ||
      C0 <= 0 || C0 > kUMaxC || C1 < 0 || Cfirst + C1 > kUMaxC || Cout <= 0 || Cout > kUMaxC ||
      (mode == 2 && (!up_w || !up_b || Cup <= 0)) || (C1 > 0 && !src1))
    return (int)hipErrorInvalidValue;
  if ((mode == 0 && (H0 != H || W0 != W)) || (mode == 1 && (H0 / 2 != H || W0 / 2 != W)) ||
      (mode == 2 && (H0 * 2 != H || W0 * 2 != W)))
    return (int)hipErrorInvalidValue;
  UnetBlockArgs a;
  a.src0 = src0; a.gate0 = gate0; a.C0 = C0; a.H0 = H0; a.W0 = W0; a.mode = mode;
  a.up_w = up_w; a.up_b = up_b; a.Cup = Cup; a.src1 = src1; a.gate1 = gate1; a.C1 = C1;
  a.dw_w = dw_w; a.dw_b = dw_b; a.pw_w = pw_w; a.pw_b = pw_b; a.Cout = Cout; a.addend = addend;
  a.out = out; a.partial_max = partial_max; a.H = H; a.W = W;
  a.tiles_x = (W + kUT - 1) / kUT; a.tiles_y = (H + kUT - 1) / kUT;
  a.pm0 = a.g0w1 = a.g0w2 = a.pm1 = a.g1w1 = a.g1w2 = nullptr;
  a.g0hid = a.g0tiles = a.g1hid = a.g1tiles = 0;
  return launch_unet_block(a, B, stream);
}

}  // extern "C"

namespace {
int launch_unet_block(const UnetBlockArgs& a, int B, hipStream_t stream) {
  const int mode = a.mode, C0 = a.C0, Cup = a.Cup, C1 = a.C1, Cout = a.Cout;
  const dim3 grid(a.tiles_x, a.tiles_y, B);
#define OCRF_UNET_FIXED(M, c0, cup, c1, co)                                                                   \
  if (mode == M && C0 == c0 && (M != 2 || Cup == cup) && C1 == c1 && Cout == co) {                            \
    ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_unet_block_fixed_kernel<M, c0, cup, c1, co>, grid, dim3(kBlock), 0, \
                 stream, a);                                                                                    \
    return (int)hipGetLastError();                                                                              \
  }
  OCRF_UNET_FIXED(0, 13, 0, 0, 4)     // encoder1   (:498)
  OCRF_UNET_FIXED(1, 4, 0, 0, 8)      // encoder2   (:501)
  OCRF_UNET_FIXED(1, 8, 0, 0, 16)     // bottleneck (:504)
  OCRF_UNET_FIXED(2, 16, 8, 8, 8)     // decoder2   (:507-509)
  OCRF_UNET_FIXED(2, 8, 4, 4, 4)      // decoder1   (:512-514)
#undef OCRF_UNET_FIXED
  ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_unet_block_kernel, grid, dim3(kBlock), 0, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// ocrf_hoa_v2b_weights_len / _workspace_bytes / _forward (the whole converter as six launches): csrc/hoa_v2b.hip

int ocrf_hoa_unet_tiles(int H, int W) { return ((W + kUT - 1) / kUT) * ((H + kUT - 1) / kUT); }

int ocrf_hoa_height_gate_from_tiles(int B, int C, int hid, int n_tiles, const float* partial_max,
                                    const float* w1, const float* w2, float* gate, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!partial_max || !w1 || !w2 || !gate || B <= 0 || C <= 0 || (C % 4) || C > 64 || hid <= 0 || 4 * hid > 64 ||
      n_tiles <= 0)
    return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_HOA_HEIGHT_GATE, hoa_height_gate_from_tiles_kernel, dim3(B), dim3(kBlock), 0, stream, C, hid,
               n_tiles, partial_max, w1, w2, gate);
  return (int)hipGetLastError();
}

int ocrf_hoa_gated_conv1x1(const float* x, const float* gate, int B, int C, int H, int W, const float* w,
                           const float* bias, float* out, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !gate || !w || !bias || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  const long plane = (long)H * W;
  ocrf::launch(OCRF_K_HOA_OUT_CONV, hoa_gated_conv1x1_kernel, dim3((unsigned)((plane + kBlock - 1) / kBlock), B),
               dim3(kBlock), 0, stream, x, gate, C, plane, w, bias, out, (const float*)nullptr, (const float*)nullptr,
               (const float*)nullptr, 0, 0);
  return (int)hipGetLastError();
}

}  // extern "C"

// =================================================================================================
// HOA-1 (view_transformer_ocrf.py:1159-1161 + mmdet3d/ops/cross_attention_2d.py:142-220), eval mode,
// for the configuration OcRFDet instantiates (:639-648): dim = 13, heads = 1, dim_head = 8,
// offset_groups = 1, downsample_factor = 4, offset_kernel_size = 6, CPB dim = 3, depth = 2.
//   att  = to_out(softmax(q k^T * scale + CPB(grid_q - vgrid)) v)  on the (hq,wq) = (Y/6,X/6) maps
//   out  = upsample_bilinear(att, (Y,X), align_corners) + opacity
// Kernel 1: one workgroup per 256 query tokens; every workgroup rebuilds q for the whole (hq,wq)
// map in LDS (the offset conv needs it) straight from the full-resolution opacity volume, the
// kv tokens (offsets -> sampling grid -> bilinear-of-bilinear sample of alpha -> k, v), then runs
// the attention of its tokens with an online softmax.  Kernel 2: upsample + residual.
// The reference does this with ~50 tiny launches per sample.
// Packed weights (floats): to_q[8][13] | off_dw[8][36] | off_db[8] | off_pw[2][8] | to_k[8][13] |
//   to_v[8][13] | to_out[13][8] | to_out_b[13] | cpb0_w[3][2] | cpb0_b[3] | cpb1_w[3][3] | cpb1_b[3]
//   | cpb2_w[3] | cpb2_b[1]
// =================================================================================================
namespace {

constexpr int kHD = 13, kHI = 8, kHMaxTok = 1600, kHMaxKV = 128;
constexpr int oQ = 0, oDW = oQ + 8 * 13, oDB = oDW + 8 * 36, oPW = oDB + 8, oK = oPW + 16, oV = oK + 104,
              oO = oV + 104, oOB = oO + 104, oC0W = oOB + 13, oC0B = oC0W + 6, oC1W = oC0B + 3, oC1B = oC1W + 9,
              oC2W = oC1B + 3, oC2B = oC2W + 3, kHoaWeights = oC2B + 1;

// F.interpolate(..., mode='bilinear', align_corners=True) sample of one channel plane at (sy, sx): the four corner taps
// and their mix, separately — corner q of a sample is one load, the mix happens once they are all back
struct AcTaps { int y0, y1, x0, x1; float ly, lx; };
__device__ __forceinline__ AcTaps ac_taps(int H, int W, float sy, float sx) {
  AcTaps a;
  a.y0 = min((int)sy, H - 1); a.x0 = min((int)sx, W - 1);
  a.y1 = min(a.y0 + 1, H - 1); a.x1 = min(a.x0 + 1, W - 1);
  a.ly = sy - (float)a.y0; a.lx = sx - (float)a.x0;
  return a;
}
__device__ __forceinline__ float ac_mix(float p00, float p01, float p10, float p11, float ly, float lx) {
  const float top = p00 * (1.f - lx) + p01 * lx;
  const float bot = p10 * (1.f - lx) + p11 * lx;
  return top * (1.f - ly) + bot * ly;
}

constexpr int kHoaWLoads = (kHoaWeights + kBlock - 1) / kBlock;      // 3
constexpr int kHoaWLds = kHoaWLoads * kBlock;    // LDS copy of the packed weights: every thread stores every word it loaded

// ONE kv token per workgroup (grid (nkv, B): 128 workgroups at 2 x 200 x 200, where one workgroup per 8 tokens left
// the chip to 16): offsets (depthwise 6x6 stride 4 pad 1 -> GELU -> 1x1 -> tanh -> * scale), sampling grid, bilinear
// sample (zeros padding, align_corners=False) of downsample(alpha), k and v.  kvbuf (B, nkv, 18) = k[8] | v[8] | grid[2].
// The token's offset conv reads the 6 x 6 window of the query map (hq, wq) at rows 4 ky - 1 .., columns 4 kx - 1 ..:
// q is rebuilt for exactly those 36 tokens straight from the full-resolution opacity volume (no q launch, no q buffer).
// A latency chain with two memory round trips: (1) weights + the 36 x 13 x 4 corner taps of the window, one
// (token, channel) sample per thread and trip, all issued before the first wait; (2) the 13 x 4 x 4 corner taps of the
// sampled alpha, ONE load per thread.  Arithmetic per value as in the eight-tokens-per-workgroup form it replaces.
__global__ __launch_bounds__(kBlock) void hoa1_kv_kernel(const float* __restrict__ opacity,
                                                         const float* __restrict__ alpha,
                                                         const float* __restrict__ wts, int Y, int X, int hq, int wq,
                                                         int hk, int wk, float offset_scale,
                                                         float* __restrict__ kvbuf) {
  OCRF_MAIN_CHAIN_PRIO();
  __shared__ float s_w[kHoaWLds];
  __shared__ float s_tok[36 * kHD];                 // window token (u, v): bilinear sample of the 13 opacity planes
  __shared__ float s_q[36 * kHI];
  __shared__ float s_gl[kHI];
  __shared__ float s_f[2];
  __shared__ float s_tap[kHD * 16];                 // [channel][tap][corner]
  __shared__ float s_kv[kHD];
  const int tid = threadIdx.x, j = blockIdx.x, b = blockIdx.y;
  const int ky = j / wk, kx = j % wk, nkv = hk * wk;
  const long plane = (long)Y * X;
  const float* op = opacity + (long)b * kHD * plane;
  const float* al = alpha + (long)b * kHD * plane;
  float* out = kvbuf + ((long)b * nkv + j) * 18;
  const float ry = hq > 1 ? (float)(Y - 1) / (float)(hq - 1) : 0.f, rx = wq > 1 ? (float)(X - 1) / (float)(wq - 1) : 0.f;
  // ---- round trip 1: weights + window samples (tokens outside the map are read at the clamped position: their
  // conv weight is 0, as in the reference's zero padding) ----
  float wv[kHoaWLoads];
#pragma unroll
  for (int i = 0; i < kHoaWLoads; ++i) wv[i] = wts[min(tid + i * kBlock, kHoaWeights - 1)];
  float p00[2], p01[2], p10[2], p11[2], sly[2], slx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = min(tid + i * kBlock, 36 * kHD - 1);
    const int wi = it / kHD, c = it % kHD;
    const int qy = min(max(ky * 4 - 1 + wi / 6, 0), hq - 1), qx = min(max(kx * 4 - 1 + wi % 6, 0), wq - 1);
    const AcTaps a = ac_taps(Y, X, ry * (float)qy, rx * (float)qx);
    const float* p = op + c * plane;
    p00[i] = p[(long)a.y0 * X + a.x0]; p01[i] = p[(long)a.y0 * X + a.x1];
    p10[i] = p[(long)a.y1 * X + a.x0]; p11[i] = p[(long)a.y1 * X + a.x1];
    sly[i] = a.ly; slx[i] = a.lx;
  }
#pragma unroll
  for (int i = 0; i < kHoaWLoads; ++i) s_w[tid + i * kBlock] = wv[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = min(tid + i * kBlock, 36 * kHD - 1);            // (the clamped duplicates store the same value)
    s_tok[it] = ac_mix(p00[i], p01[i], p10[i], p11[i], sly[i], slx[i]);
  }
  __syncthreads();
  // ---- to_q of the 36 window tokens ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = tid + i * kBlock;
    if (it < 36 * kHI) {
      const int wi = it / kHI, d = it % kHI;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < kHD; ++c) q = fmaf(s_w[oQ + d * kHD + c], s_tok[wi * kHD + c], q);
      s_q[it] = q;
    }
  }
  __syncthreads();
  // ---- offset conv (36 taps in order, weight 0 outside the map) -> GELU ----
  if (tid < kHI) {
    const int d = tid;
    float a = s_w[oDB + d];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int yy = ky * 4 - 1 + u;
      const bool yin = yy >= 0 && yy < hq;
#pragma unroll
      for (int v = 0; v < 6; ++v) {
        const int xx = kx * 4 - 1 + v;
        const bool in = yin && xx >= 0 && xx < wq;
        const float w = in ? s_w[oDW + d * 36 + u * 6 + v] : 0.f;
        a = fmaf(w, s_q[(u * 6 + v) * kHI + d], a);
      }
    }
    s_gl[d] = 0.5f * a * (1.f + erff(a * 0.70710678118654752f));            // nn.GELU (exact)
  }
  __syncthreads();
  if (tid == 0) {
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int d = 0; d < kHI; ++d) {
      ox = fmaf(s_w[oPW + d], s_gl[d], ox);
      oy = fmaf(s_w[oPW + 8 + d], s_gl[d], oy);
    }
    const float vx = (float)kx + tanhf(ox) * offset_scale, vy = (float)ky + tanhf(oy) * offset_scale;
    // normalize_grid as written (cross_attention_2d.py:30-38): channel 0 over (h-1), 1 over (w-1)
    const float gx = 2.0f * vx / (float)max(hk - 1, 1) - 1.0f, gy = 2.0f * vy / (float)max(wk - 1, 1) - 1.0f;
    out[16] = gx;
    out[17] = gy;
    s_f[0] = ((gx + 1.f) * (float)wq - 1.f) * 0.5f;
    s_f[1] = ((gy + 1.f) * (float)hq - 1.f) * 0.5f;
  }
  __syncthreads();
  // ---- round trip 2: one corner of one tap of one alpha channel per thread (taps outside the map: clamped position,
  // weight 0 below) ----
  const float fx = s_f[0], fy = s_f[1];
  const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  {
    const int it = min(tid, kHD * 16 - 1);
    const int c = it >> 4, tp = (it >> 2) & 3, cr = it & 3;
    const int xi = min(max(x0 + (tp & 1), 0), wq - 1), yi = min(max(y0 + (tp >> 1), 0), hq - 1);
    const AcTaps a = ac_taps(Y, X, ry * (float)yi, rx * (float)xi);
    const float v = al[c * plane + (long)((cr >> 1) ? a.y1 : a.y0) * X + ((cr & 1) ? a.x1 : a.x0)];
    if (tid < kHD * 16) s_tap[tid] = v;
  }
  __syncthreads();
  if (tid < kHD) {
    const int c = tid;
    float val[4], wt[4];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      const int xi = x0 + (tp & 1), yi = y0 + (tp >> 1);
      const bool in = xi >= 0 && xi < wq && yi >= 0 && yi < hq;
      wt[tp] = in ? (1.f - fabsf(fx - (float)xi)) * (1.f - fabsf(fy - (float)yi)) : 0.f;
      const AcTaps a = ac_taps(Y, X, ry * (float)min(max(yi, 0), hq - 1), rx * (float)min(max(xi, 0), wq - 1));
      const float* q = s_tap + c * 16 + tp * 4;
      val[tp] = ac_mix(q[0], q[1], q[2], q[3], a.ly, a.lx);
    }
    float acc = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) acc = fmaf(val[tp], wt[tp], acc);
    s_kv[c] = acc;
  }
  __syncthreads();
  if (tid < 2 * kHI) {
    const int d = tid % kHI, o = tid < kHI ? oK : oV;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < kHD; ++c) a = fmaf(s_w[o + d * kHD + c], s_kv[c], a);
    out[tid] = a;                                   // k[8] | v[8]
  }
}

// attention + to_out + bilinear upsample + residual of one 16x16 output tile.  The tile's pixels interpolate between
// at most kTT x kTT tokens of the (hq, wq) attention map ((hq - 1) / (Y - 1) < 1/6: 15 pixels span < 2.5 tokens); the
// workgroup rebuilds q for those tokens from the opacity volume, runs their attention over the sample's kv tokens
// (LDS, online softmax, kAttSplit lanes per token merged by a fixed DPP butterfly) and keeps the 13 output channels
// in LDS for the interpolation: no q / attention buffers and no separate upsample launch.  Tokens shared by
// neighbouring tiles are recomputed (a token's attention is ~6 k flops).  The kernel is a latency chain on a chip
// it cannot fill (338 workgroups at 2 x 200 x 200), so every global load — the token's 52 bilinear taps, the pixel's
// 13 residual values, weights and kv tokens — is issued before the first barrier.
constexpr int kAttSplit = 8, kTP = 16, kTT = 5;
constexpr int kKvLoads = (kHMaxKV * 9 + kBlock - 1) / kBlock;       // 8-byte words of the kv tokens per thread (5)
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppMirror8 = 0x141;   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
template <int CTRL>
__device__ __forceinline__ float hoa_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

__global__ __launch_bounds__(kBlock) void hoa1_attention_upsample_kernel(
    const float* __restrict__ opacity, const float* __restrict__ kvbuf, const float* __restrict__ wts, int Y, int X,
    int hq, int wq, int nkv, float* __restrict__ out) {
  OCRF_MAIN_CHAIN_PRIO();
  __shared__ float s_w[kHoaWLds];
  __shared__ __attribute__((aligned(8))) float s_kv[kKvLoads * kBlock * 2];      // [nkv][18], padded to whole trips
  __shared__ float s_tok[kTT * kTT * kHD];
  __shared__ float s_att[kTT * kTT * kHD];
  const int tid = threadIdx.x, b = blockIdx.z;
  const long plane = (long)Y * X;
  const float* op = opacity + (long)b * kHD * plane;
  // token window of the tile: the bilinear_ac taps of its first and last pixel
  const float dy = Y > 1 ? (float)(hq - 1) / (float)(Y - 1) : 0.f, dx = X > 1 ? (float)(wq - 1) / (float)(X - 1) : 0.f;
  const int py0 = blockIdx.y * kTP, px0 = blockIdx.x * kTP;
  const int py1 = min(py0 + kTP, Y) - 1, px1 = min(px0 + kTP, X) - 1;
  const int ty0 = min((int)(dy * (float)py0), hq - 1), tx0 = min((int)(dx * (float)px0), wq - 1);
  const int ty1 = min(min((int)(dy * (float)py1), hq - 1) + 1, hq - 1), tx1 = min(min((int)(dx * (float)px1), wq - 1) + 1, wq - 1);
  const int nty = ty1 - ty0 + 1, ntx = tx1 - tx0 + 1;       // <= kTT each
  const int tl = tid / kAttSplit, part = tid % kAttSplit;
  const bool has = tl < nty * ntx;
  const int lt = has ? tl : 0;                               // idle lanes of a live wave run token 0
  const int ty = ty0 + lt / ntx, tx = tx0 + lt % ntx;
  const float ry = hq > 1 ? (float)(Y - 1) / (float)(hq - 1) : 0.f, rx = wq > 1 ? (float)(X - 1) / (float)(wq - 1) : 0.f;
  // ---- all global loads, unconditional (clamped), before the first wait: weights, kv tokens, the window tokens'
  // bilinear samples — one (token, channel) sample of four corner taps per thread and trip instead of every one of
  // a token's eight lanes loading all 52 — and the pixel's residual ----
  float wv[kHoaWLoads];
#pragma unroll
  for (int i = 0; i < kHoaWLoads; ++i) wv[i] = wts[min(tid + i * kBlock, kHoaWeights - 1)];
  float2 kvv[kKvLoads];
  {
    const float2* kv2 = reinterpret_cast<const float2*>(kvbuf + (long)b * nkv * 18);      // 18 floats per token: 8-byte aligned
#pragma unroll
    for (int i = 0; i < kKvLoads; ++i) kvv[i] = kv2[min(tid + i * kBlock, nkv * 9 - 1)];
  }
  float p00[2], p01[2], p10[2], p11[2], sly[2], slx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = min(tid + i * kBlock, nty * ntx * kHD - 1);
    const int wi = it / kHD, c = it % kHD;
    const AcTaps a = ac_taps(Y, X, ry * (float)(ty0 + wi / ntx), rx * (float)(tx0 + wi % ntx));
    const float* p = op + c * plane;
    p00[i] = p[(long)a.y0 * X + a.x0]; p01[i] = p[(long)a.y0 * X + a.x1];
    p10[i] = p[(long)a.y1 * X + a.x0]; p11[i] = p[(long)a.y1 * X + a.x1];
    sly[i] = a.ly; slx[i] = a.lx;
  }
  const int y = py0 + tid / kTP, x = px0 + tid % kTP;
  const bool inside = y < Y && x < X;
  const long pix = (long)min(y, Y - 1) * X + min(x, X - 1);
  float res[kHD];
#pragma unroll
  for (int c = 0; c < kHD; ++c) res[c] = op[c * plane + pix];
#pragma unroll
  for (int i = 0; i < kHoaWLoads; ++i) s_w[tid + i * kBlock] = wv[i];
#pragma unroll
  for (int i = 0; i < kKvLoads; ++i) reinterpret_cast<float2*>(s_kv)[tid + i * kBlock] = kvv[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = min(tid + i * kBlock, nty * ntx * kHD - 1);    // (clamped duplicates store the same value)
    s_tok[it] = ac_mix(p00[i], p01[i], p10[i], p11[i], sly[i], slx[i]);
  }
  __syncthreads();
  float tok[kHD];
#pragma unroll
  for (int c = 0; c < kHD; ++c) tok[c] = s_tok[lt * kHD + c];
  // ---- attention of the window's tokens: kAttSplit adjacent lanes per token ----
  if (tl < kTT * kTT) {          // wave-uniform up to the last wave
    const float scale = 0.35355339059327373f;               // dim_head ** -0.5
    float q[kHI];
#pragma unroll
    for (int d = 0; d < kHI; ++d) q[d] = 0.f;
#pragma unroll
    for (int c = 0; c < kHD; ++c) {
#pragma unroll
      for (int d = 0; d < kHI; ++d) q[d] = fmaf(s_w[oQ + d * kHD + c], tok[c], q[d]);
    }
#pragma unroll
    for (int d = 0; d < kHI; ++d) q[d] *= scale;
    // query grid: create_grid_like(x_kv) normalised with dim=0 -> x over (h-1), y over (w-1)
    const float qx = 2.0f * (float)tx / (float)max(hq - 1, 1) - 1.0f, qy = 2.0f * (float)ty / (float)max(wq - 1, 1) - 1.0f;
    float m = -INFINITY, l = 0.f, acc[kHI];
#pragma unroll
    for (int d = 0; d < kHI; ++d) acc[d] = 0.f;
#pragma unroll 2
    for (int j = part; j < nkv; j += kAttSplit) {
      const float* kv = s_kv + j * 18;
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < kHI; ++d) sc = fmaf(q[d], kv[d], sc);
      // CPB (cross_attention_2d.py:74-89)
      const float px = qx - kv[16], py = qy - kv[17];
      const float bx = copysignf(__logf(fabsf(px) + 1.f), px), by = copysignf(__logf(fabsf(py) + 1.f), py);
      float h0[3], h1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) h0[i] = fmaxf(fmaf(s_w[oC0W + i * 2], bx, fmaf(s_w[oC0W + i * 2 + 1], by, s_w[oC0B + i])), 0.f);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        h1[i] = fmaxf(s_w[oC1B + i] + s_w[oC1W + i * 3] * h0[0] + s_w[oC1W + i * 3 + 1] * h0[1] + s_w[oC1W + i * 3 + 2] * h0[2], 0.f);
      sc += s_w[oC2B] + s_w[oC2W] * h1[0] + s_w[oC2W + 1] * h1[1] + s_w[oC2W + 2] * h1[2];
      const float mn = fmaxf(m, sc);
      const float corr = __expf(m - mn), p = __expf(sc - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < kHI; ++d) acc[d] = fmaf(p, kv[8 + d], acc[d] * corr);
      m = mn;
    }
    // merge the kAttSplit partial softmax states: fixed butterfly over the 8 lanes (lane ^ 1, lane ^ 2, 7 - lane)
    auto merge = [&](auto perm) {
      const float mo = perm(m), lo = perm(l);
      const float mn = fmaxf(m, mo);
      const float c1 = (m == -INFINITY) ? 0.f : __expf(m - mn), c2 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
      l = l * c1 + lo * c2;
#pragma unroll
      for (int d = 0; d < kHI; ++d) acc[d] = acc[d] * c1 + perm(acc[d]) * c2;
      m = mn;
    };
    merge([](float v) { return hoa_dpp<kDppXor1>(v); });
    merge([](float v) { return hoa_dpp<kDppXor2>(v); });
    merge([](float v) { return hoa_dpp<kDppMirror8>(v); });
    if (has) {
      const float inv = 1.f / l;
      for (int c = part; c < kHD; c += kAttSplit) {          // the output projection's channels are shared out too
        float o = s_w[oOB + c];
#pragma unroll
        for (int d = 0; d < kHI; ++d) o = fmaf(s_w[oO + c * kHI + d], acc[d] * inv, o);
        s_att[lt * kHD + c] = o;
      }
    }
  }
  __syncthreads();
  // ---- upsample (bilinear, align_corners) + residual: one pixel per thread, 13 channels ----
  if (!inside) return;
  const float sy = dy * (float)y, sx = dx * (float)x;
  const int y0 = min((int)sy, hq - 1), x0 = min((int)sx, wq - 1);
  const int y1 = min(y0 + 1, hq - 1), x1 = min(x0 + 1, wq - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const int i00 = ((y0 - ty0) * ntx + (x0 - tx0)) * kHD, i01 = ((y0 - ty0) * ntx + (x1 - tx0)) * kHD;
  const int i10 = ((y1 - ty0) * ntx + (x0 - tx0)) * kHD, i11 = ((y1 - ty0) * ntx + (x1 - tx0)) * kHD;
#pragma unroll
  for (int c = 0; c < kHD; ++c) {
    const float top = s_att[i00 + c] * (1.f - lx) + s_att[i01 + c] * lx;
    const float bot = s_att[i10 + c] * (1.f - lx) + s_att[i11 + c] * lx;
    out[((long)b * kHD + c) * plane + pix] = (top * (1.f - ly) + bot * ly) + res[c];
  }
}

}  // namespace

extern "C" {

int ocrf_hoa1_weights_len(void) { return kHoaWeights; }

int ocrf_hoa1_forward(const float* opacity, const float* alpha, const float* weights, int B, int Y, int X,
                      float offset_scale, float* att_workspace, float* out, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int hq = Y / 6, wq = X / 6;                      // int(Width / 6), int(Length / 6) (:1159)
  if (!opacity || !alpha || !weights || !att_workspace || !out || B <= 0 || hq < 6 || wq < 6)
    return (int)hipErrorInvalidValue;
  const int hk = (hq + 2 - 6) / 4 + 1, wk = (wq + 2 - 6) / 4 + 1;
  const int ntok = hq * wq, nkv = hk * wk;
  if (ntok > kHMaxTok || nkv > kHMaxKV) return (int)hipErrorInvalidValue;
  // (the workspace keeps round 1's layout att | q | kv; only the kv part is used now)
  float* kvbuf = att_workspace + (size_t)B * kHD * ntok + (size_t)B * ntok * kHI;
  ocrf::launch(OCRF_K_HOA1_KV, hoa1_kv_kernel, dim3(nkv, B), dim3(kBlock), 0, stream, opacity, alpha, weights, Y, X, hq, wq,
               hk, wk, offset_scale, kvbuf);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  ocrf::launch(OCRF_K_HOA1_ATTN, hoa1_attention_upsample_kernel, dim3((X + kTP - 1) / kTP, (Y + kTP - 1) / kTP, B),
               dim3(kBlock), 0, stream, opacity, static_cast<const float*>(kvbuf), weights, Y, X, hq, wq, nkv, out);
  return (int)hipGetLastError();
}

}  // extern "C"

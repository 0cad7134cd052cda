// Training side of the voxel -> Gaussian heads (SURVEY.md 8 rows f1 + f3): the backward of ocrf_gauss_heads.
//
// The reference trains VoxelFeatureExtractor (view_transformer_ocrf.py:520-531) + S/R/A/C_MLP (:272-320, :1130-1133) through
// autograd on the (B,13,Y,X,80) voxel feature: 333 MB at cfg2 that every one of ~100 torch kernels of the forward and the
// backward reads or writes again.  Here the backward is ONE kernel over the same register tile as the forward's
// (csrc/neck.hip: one lane = one BEV pillar x a group of heights) + a two-stage sum of per-wave partials:
//
//   phase 1   the lift and the 16 hidden units of every height of the group are recomputed from the (B,C,Y,X) BEV map —
//             nothing of the forward is kept but its inputs;
//   epilogue  the four heads' output gradients go back through softplus / L2-normalise / sigmoid and the second layers to
//             the hidden pre-activations dz1[h][16] (in the registers that held the hidden units); the 83 small parameter
//             gradients (second layers, first-layer bias, the colour head's rgb columns) are summed over the wave;
//   phase 2   per channel c:  x_h = relu(a_h v + b_h),   g_h = sum_k W1[k][c] dz1[h][k]  (-> d v, d a_h, d b_h),
//                             dW1[c][k] = sum_{h, pillars} dz1[h][k] x_h
//             — the 16 per-lane sums of dW1[c][.] meet across the wave by recursive halving (each exchange step halves the
//             values a lane carries: 8 + 4 + 2 + 1 exchanges instead of 16 x 6), and 16 lanes store one 64-byte row of the
//             wave's partial.
//
// d(bev): one plain store per (channel, pillar) into a map of the height group (four groups at 13 heights: 4 + 3 + 3 + 3),
// summed afterwards in group order.  The parameter gradient is the sum of per-wave partial rows, taken by
// a fixed-order two-stage reduction: the whole backward is deterministic.
// The lift's coefficients (a_h, b_h) are INPUTS here (training: batch statistics folded by the caller, with autograd through
// that folding; the gradient w.r.t. them leaves in the same packed layout as the parameters).
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf(float x) { return rcp_fast(1.0f + __expf(-x)); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
constexpr int kQuadXor1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;     // quad_perm:[2,3,0,1]
// ds_swizzle, bit-mask mode: lane <- lane ^ X inside a group of 32 (offset = xor << 10 | or << 5 | and)
template <int X>
__device__ __forceinline__ float swz_xor(float x) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (X << 10) | 0x1F));
}

__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_mov<kQuadXor1>(x);
  x += dpp_mov<kQuadXor2>(x);
  x += swz_xor<4>(x);
  x += swz_xor<8>(x);
  x += swz_xor<16>(x);
  x += __shfl_xor(x, 32);
  return x;
}

// The wave-wide sums of p[0..15], one per lane: lane l returns sum over the wave of p[halving_slot(l)], where
// halving_slot(l) = 8 (l & 1) + 4 (l >> 1 & 1) + 2 (l >> 2 & 1) + (l >> 3 & 1)   (each step keeps the half its bit names).
__device__ __forceinline__ float halving_sum16(const float (&p)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float r8[8], r4[4], r2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) r8[i] = (b0 ? p[8 + i] : p[i]) + dpp_mov<kQuadXor1>(b0 ? p[i] : p[8 + i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) r4[i] = (b1 ? r8[4 + i] : r8[i]) + dpp_mov<kQuadXor2>(b1 ? r8[i] : r8[4 + i]);
#pragma unroll
  for (int i = 0; i < 2; ++i) r2[i] = (b2 ? r4[2 + i] : r4[i]) + swz_xor<4>(b2 ? r4[i] : r4[2 + i]);
  float r = (b3 ? r2[1] : r2[0]) + swz_xor<8>(b3 ? r2[0] : r2[1]);
  r += swz_xor<16>(r);
  r += __shfl_xor(r, 32);
  return r;
}
__device__ __forceinline__ int halving_slot(int lane) {
  return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
}

#ifndef OCRF_HEADS_BWD_G13
#define OCRF_HEADS_BWD_G13 4
#endif
constexpr int kSmall = 12 + 16 + 15 + 20 + 5 + 15;      // W1rgb | b1 | S | R | A | Col: everything after W1t

template <int ZH, int HG>
__global__ __launch_bounds__(64, HG <= 5 ? 3 : 2) void neck_gauss_heads_backward_kernel(
    const float* __restrict__ bev, const float* __restrict__ rgb_avg, const float* __restrict__ prm, int C, int YX,
    const float* __restrict__ g_op, const float* __restrict__ g_sc, const float* __restrict__ g_rot,
    const float* __restrict__ g_col, float* __restrict__ d_bev, float* __restrict__ partial, int L) {
  constexpr int kGroups = (ZH + HG - 1) / HG;
  const int lane = threadIdx.x;
  const int q_raw = blockIdx.x * 64 + lane;
  const bool live = q_raw < YX;
  const int q = min(q_raw, YX - 1);                 // lanes past the end redo the last pillar with zero output gradients
  // heights over the groups as evenly as they go (13 over 4 groups: 4 + 3 + 3 + 3): a group has HG or HG - 1 heights, and a
  // wave of the shorter kind skips the last slot of its register tile (wave-uniform branches) instead of computing it for
  // nothing
  constexpr int kBase = ZH / kGroups, kRem = ZH % kGroups;
  const int grp = blockIdx.y % kGroups;
  const int b = blockIdx.y / kGroups, h0 = grp * kBase + min(grp, kRem), nh = kBase + (grp < kRem ? 1 : 0);
  static_assert(kBase + (kRem ? 1 : 0) == HG, "groups of HG or HG - 1 heights");
  float* part = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * L;
  float la[HG], lb[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    la[h] = prm[min(h0 + h, ZH - 1)];
    lb[h] = prm[ZH + min(h0 + h, ZH - 1)];
  }
  const f32x2* W1t = reinterpret_cast<const f32x2*>(prm + 2 * ZH);
  const float* W1rgb = prm + 2 * ZH + 16 * C;
  const float* b1 = W1rgb + 12;
  const float* S2 = b1 + 16;
  const float* R2 = S2 + 15;
  const float* A2 = R2 + 20;
  const float* C2 = A2 + 5;
  const float* bp = bev + (size_t)b * C * YX + q;

  // ---------------------------------------------------------------- phase 1: the hidden units again
  f32x2 hid[HG][8];            // (after the epilogue: dz1[h][2k], dz1[h][2k+1])
#pragma unroll
  for (int h = 0; h < HG; ++h)
#pragma unroll
    for (int k = 0; k < 8; ++k) hid[h][k] = f32x2{0.0f, 0.0f};
  constexpr int kAhead = 8;
  {
    float vv[kAhead], nx[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(j, C - 1) * YX];
    // a channel's 16 first-layer weights are one wave-uniform 64-byte scalar load, requested ONE CHANNEL AHEAD of their
    // use (as in the forward kernel: loaded where they are used, every channel waits ~0.3 us for its own load)
    f32x2 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = W1t[k];
    for (int c0 = 0; c0 < C; c0 += kAhead) {
#pragma unroll
      for (int j = 0; j < kAhead; ++j) vv[j] = nx[j];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(c0 + kAhead + j, C - 1) * YX];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        f32x2 wn[8];
        asm volatile("" ::"s"(w[0].x));          // this channel's weights have arrived before the next are requested
        const int cn = min(c0 + j + 1, C - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) wn[k] = W1t[cn * 8 + k];
        __builtin_amdgcn_sched_barrier(0);
        if (c0 + j < C) {
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            if (h == HG - 1 && nh < HG) break;
            const float f = fmaxf(fmaf(la[h], vv[j], lb[h]), 0.0f);
            const f32x2 ff = {f, f};
#pragma unroll
            for (int k = 0; k < 8; ++k) hid[h][k] = __builtin_elementwise_fma(w[k], ff, hid[h][k]);
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = wn[k];
      }
    }
  }

  // ---------------------------------------------------------------- epilogue: output gradients -> dz1, small parameters
  float small[kSmall];
#pragma unroll
  for (int i = 0; i < kSmall; ++i) small[i] = 0.0f;
  float* aW1rgb = small;
  float* ab1 = small + 12;
  float* aS = ab1 + 16;
  float* aR = aS + 15;
  float* aA = aR + 20;
  float* aC = aA + 5;
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    if (h == HG - 1 && nh < HG) break;             // (its dz1 slot stays zero and is never read)
    const bool on = live;
    const size_t g = ((size_t)b * ZH + h0 + h) * YX + q;
    constexpr float k255 = 1.0f / 255.0f;
    const float r01[3] = {rgb_avg[g * 3] * k255, rgb_avg[g * 3 + 1] * k255, rgb_avg[g * 3 + 2] * k255};
    float a[16], da[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[2 * k] = hid[h][k].x; a[2 * k + 1] = hid[h][k].y; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) a[12 + k] = fmaf(W1rgb[k * 3 + j], r01[j], a[12 + k]);
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = fmaxf(a[k] + b1[k], 0.0f); da[k] = 0.0f; }
    // scales: softplus (beta 1, threshold 20)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = S2[12 + o];
#pragma unroll
      for (int k = 0; k < 4; ++k) v = fmaf(S2[o * 4 + k], a[k], v);
      const float go = (on && g_sc) ? g_sc[g * 3 + o] : 0.0f;
      const float dv = go * (v > 20.0f ? 1.0f : sigmoidf(v));
      aS[12 + o] += dv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        aS[o * 4 + k] = fmaf(dv, a[k], aS[o * 4 + k]);
        da[k] = fmaf(S2[o * 4 + k], dv, da[k]);
      }
    }
    // rotation: v / max(|v|, 1e-12)
    {
      float v[4], nn = 0.0f, gr[4], dot = 0.0f;
      const float4 g4 = (on && g_rot) ? *reinterpret_cast<const float4*>(g_rot + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      gr[0] = g4.x; gr[1] = g4.y; gr[2] = g4.z; gr[3] = g4.w;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float t = R2[16 + o];
#pragma unroll
        for (int k = 0; k < 4; ++k) t = fmaf(R2[o * 4 + k], a[4 + k], t);
        v[o] = t;
        nn = fmaf(t, t, nn);
      }
      const float n = sqrtf(nn);
      const float inv = rcp_fast(fmaxf(n, 1e-12f));
#pragma unroll
      for (int o = 0; o < 4; ++o) dot = fmaf(v[o] * inv, gr[o], dot);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float dv = n > 1e-12f ? (gr[o] - v[o] * inv * dot) * inv : gr[o] * inv;
        aR[16 + o] += dv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          aR[o * 4 + k] = fmaf(dv, a[4 + k], aR[o * 4 + k]);
          da[4 + k] = fmaf(R2[o * 4 + k], dv, da[4 + k]);
        }
      }
    }
    // opacity: sigmoid
    {
      float v = A2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v = fmaf(A2[k], a[8 + k], v);
      const float s = sigmoidf(v);
      const float dv = ((on && g_op) ? g_op[g] : 0.0f) * s * (1.0f - s);
      aA[4] += dv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        aA[k] = fmaf(dv, a[8 + k], aA[k]);
        da[8 + k] = fmaf(A2[k], dv, da[8 + k]);
      }
    }
    // colour: sigmoid
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = C2[12 + o];
#pragma unroll
      for (int k = 0; k < 4; ++k) v = fmaf(C2[o * 4 + k], a[12 + k], v);
      const float s = sigmoidf(v);
      const float dv = ((on && g_col) ? g_col[g * 3 + o] : 0.0f) * s * (1.0f - s);
      aC[12 + o] += dv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        aC[o * 4 + k] = fmaf(dv, a[12 + k], aC[o * 4 + k]);
        da[12 + k] = fmaf(C2[o * 4 + k], dv, da[12 + k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      da[k] = a[k] > 0.0f ? da[k] : 0.0f;              // through the first layer's ReLU: dz1
      ab1[k] += da[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) aW1rgb[k * 3 + j] = fmaf(da[12 + k], r01[j], aW1rgb[k * 3 + j]);
#pragma unroll
    for (int k = 0; k < 8; ++k) hid[h][k] = f32x2{da[2 * k], da[2 * k + 1]};
    __builtin_amdgcn_sched_barrier(0);          // (one height at a time)
  }
  {
    float* ps = part + 2 * ZH + 16 * C;
#pragma unroll
    for (int i = 0; i < kSmall; ++i) {
      const float s = wave_sum(small[i]);
      if (lane == 0) ps[i] = s;
    }
  }

  // ---------------------------------------------------------------- phase 2: d bev, d lift, dW1
  float dla[HG], dlb[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) dla[h] = dlb[h] = 0.0f;
  // (one plain store per (channel, pillar): each height group owns a map of its own, summed afterwards — float atomics on
  // one shared map ran at the memory side's atomic rate and took more time than all the arithmetic of this kernel)
  float* dbp = d_bev + (((size_t)(blockIdx.y % kGroups) * (gridDim.y / kGroups) + b) * C) * YX + q;
  const int slot = halving_slot(lane);
  {
    float vv[kAhead], nx[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(j, C - 1) * YX];
    f32x2 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = W1t[k];
    for (int c0 = 0; c0 < C; c0 += kAhead) {
#pragma unroll
      for (int j = 0; j < kAhead; ++j) vv[j] = nx[j];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(c0 + kAhead + j, C - 1) * YX];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        const int c = c0 + j;
        f32x2 wn[8];
        asm volatile("" ::"s"(w[0].x));
        const int cn = min(c + 1, C - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) wn[k] = W1t[cn * 8 + k];
        __builtin_amdgcn_sched_barrier(0);
        if (c < C) {
          f32x2 pp[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) pp[k] = f32x2{0.0f, 0.0f};
          const float v = vv[j];
          float dv = 0.0f;
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            if (h == HG - 1 && nh < HG) break;
            const float pre = fmaf(la[h], v, lb[h]);
            const float x = fmaxf(pre, 0.0f);
            const f32x2 xx = {x, x};
            f32x2 gg = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              gg = __builtin_elementwise_fma(w[k], hid[h][k], gg);
              pp[k] = __builtin_elementwise_fma(hid[h][k], xx, pp[k]);
            }
            const float gx = pre > 0.0f ? gg.x + gg.y : 0.0f;
            dla[h] = fmaf(gx, v, dla[h]);
            dlb[h] += gx;
            dv = fmaf(gx, la[h], dv);
          }
          if (live) dbp[(size_t)c * YX] = dv;
          float p[16];
#pragma unroll
          for (int k = 0; k < 8; ++k) { p[2 * k] = pp[k].x; p[2 * k + 1] = pp[k].y; }
          const float s = halving_sum16(p, lane);
          if (lane < 16) part[2 * ZH + c * 16 + slot] = s;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = wn[k];
      }
    }
  }
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float sa = wave_sum(dla[h]), sb = wave_sum(dlb[h]);
    if (lane == 0 && h < nh) { part[h0 + h] = sa; part[ZH + h0 + h] = sb; }
  }
  // the heights of the other groups: zero in this wave's row (the rows are summed whole)
  if (lane < ZH && (lane < h0 || lane >= h0 + nh)) { part[lane] = 0.0f; part[ZH + lane] = 0.0f; }
}

// out[chunk][l] = sum of in[t][l] over the chunk's rows t, in order (fixed order: the same bits every run)
__global__ __launch_bounds__(256) void neck_partial_rows_sum_kernel(const float* __restrict__ in, int T, int L, int rows,
                                                                    float* __restrict__ out) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const int t0 = blockIdx.y * rows, t1 = min(T, t0 + rows);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int t = t0;
  for (; t + 4 <= t1; t += 4) {
    s0 += in[(size_t)t * L + l];
    s1 += in[(size_t)(t + 1) * L + l];
    s2 += in[(size_t)(t + 2) * L + l];
    s3 += in[(size_t)(t + 3) * L + l];
  }
  for (; t < t1; ++t) s0 += in[(size_t)t * L + l];
  out[(size_t)blockIdx.y * L + l] = (s0 + s1) + (s2 + s3);
}

// out = in[0] + in[1] (+ ...): the per-height-group maps of d(bev), n floats each (maps start 16-byte aligned, n apart)
template <typename V>
__global__ __launch_bounds__(256) void neck_sum_maps_kernel(const float* __restrict__ in, size_t n, int maps,
                                                            float* __restrict__ out) {
  constexpr int kPer = sizeof(V) / sizeof(float);
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * kPer;
  if (i >= n) return;
  float s[kPer];
  *reinterpret_cast<V*>(s) = *reinterpret_cast<const V*>(in + i);
  for (int m = 1; m < maps; ++m) {
    float t[kPer];
    *reinterpret_cast<V*>(t) = *reinterpret_cast<const V*>(in + (size_t)m * n + i);
#pragma unroll
    for (int k = 0; k < kPer; ++k) s[k] += t[k];
  }
  *reinterpret_cast<V*>(out + i) = *reinterpret_cast<V*>(s);
}

inline int heads_group(int Zh) {
  switch (Zh) {
    case 13: return OCRF_HEADS_BWD_G13;
    case 8: return 4;
    case 6: return 3;
    case 4: return 4;
    case 2: return 2;
    case 1: return 1;
    default: return 0;
  }
}
constexpr int kStageRows = 64;       // rows of partials one thread of the first reduction stage sums

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

size_t ocrf_gauss_heads_backward_workspace_bytes(int B, int C, int Zh, int YX) {
  const int G = heads_group(Zh);
  if (B <= 0 || C <= 0 || YX <= 0 || !G) return 0;
  const size_t L = (size_t)ocrf_gauss_heads_params_len(C, Zh);
  const size_t T = (size_t)((YX + 63) / 64) * B * ((Zh + G - 1) / G);
  const size_t groups = (size_t)((Zh + G - 1) / G);
  const size_t maps = groups > 1 ? groups * B * C * ((size_t)YX + 3) : 0;      // (+ 3: room to start 16-byte aligned)
  return ((T + (T + kStageRows - 1) / kStageRows) * L + maps + 4) * sizeof(float);
}

int ocrf_gauss_heads_backward(const float* bev, const float* rgb_avg, const float* params, int B, int C, int Zh, int YX,
                              const float* g_opacity, const float* g_scales, const float* g_rotations,
                              const float* g_color, float* d_bev, float* d_params, void* workspace,
                              size_t workspace_bytes, ocrf_stream_t stream) {
  const int G = heads_group(Zh);
  if (B <= 0 || C <= 0 || YX <= 0 || !G || !bev || !rgb_avg || !params || !d_bev || !d_params || !workspace)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < ocrf_gauss_heads_backward_workspace_bytes(B, C, Zh, YX)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(g_rotations) & 15) || ((reinterpret_cast<uintptr_t>(params) + 2 * Zh * sizeof(float)) & 7))
    return (int)hipErrorInvalidValue;
  const int L = ocrf_gauss_heads_params_len(C, Zh);
  const dim3 grid((YX + 63) / 64, B * ((Zh + G - 1) / G));
  const int T = (int)(grid.x * grid.y), stage = (T + kStageRows - 1) / kStageRows;
  float* partial = static_cast<float*>(workspace);
  float* mid = partial + (size_t)T * L;
  const int groups = (Zh + G - 1) / G;
  const size_t n = (size_t)B * C * YX;
  float* maps = d_bev;
  if (groups > 1)
    maps = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(mid + (size_t)stage * L) + 15) & ~(uintptr_t)15);
  switch (Zh) {
#define OCRF_HEADS_BWD_CASE(Z, HGRP)                                                                                  \
  case Z:                                                                                                             \
    ocrf::launch(OCRF_K_NECK_HEADS_BWD, neck_gauss_heads_backward_kernel<Z, HGRP>, grid, dim3(64), 0,                 \
                 (hipStream_t)stream, bev, rgb_avg, params, C, YX, g_opacity, g_scales, g_rotations, g_color, maps,  \
                 partial, L);                                                                                         \
    break;
    OCRF_HEADS_BWD_CASE(13, OCRF_HEADS_BWD_G13)
    OCRF_HEADS_BWD_CASE(8, 4)
    OCRF_HEADS_BWD_CASE(6, 3)
    OCRF_HEADS_BWD_CASE(4, 4)
    OCRF_HEADS_BWD_CASE(2, 2)
    OCRF_HEADS_BWD_CASE(1, 1)
#undef OCRF_HEADS_BWD_CASE
    default: return (int)hipErrorInvalidValue;
  }
  if (groups > 1) {
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(d_bev) & 15) == 0)
      ocrf::launch(OCRF_K_NECK_HEADS_BWD_SUM, neck_sum_maps_kernel<float4>, dim3((unsigned)((n / 4 + 255) / 256)),
                   dim3(256), 0, (hipStream_t)stream, (const float*)maps, n, groups, d_bev);
    else
      ocrf::launch(OCRF_K_NECK_HEADS_BWD_SUM, neck_sum_maps_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256),
                   0, (hipStream_t)stream, (const float*)maps, n, groups, d_bev);
  }
  ocrf::launch(OCRF_K_NECK_HEADS_BWD_SUM, neck_partial_rows_sum_kernel, dim3((L + 255) / 256, stage), dim3(256), 0,
               (hipStream_t)stream, (const float*)partial, T, L, kStageRows, mid);
  ocrf::launch(OCRF_K_NECK_HEADS_BWD_SUM, neck_partial_rows_sum_kernel, dim3((L + 255) / 256, 1), dim3(256), 0,
               (hipStream_t)stream, (const float*)mid, stage, L, stage, d_params);
  return last_error();
}

}  // extern "C"

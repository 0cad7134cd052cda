// Shared by hoa_v2b.hip (HOA-2 block kernels) and hoa.hip (the HOA-3 gate kernel that folds HOA-2's output conv in):
// the HeightAttention gate of a producer rebuilt from its per-tile channel maxima, with DPP row reductions.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace hoa_gate {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float x) {          // lanes without a source keep their own value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_max(float x) { return fmaxf(x, dpp_mov<CTRL, ROW_MASK>(x)); }
constexpr int kRowShr = 0x110, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

// maximum over the 64 lanes of a wave; valid in lane 63
__device__ __forceinline__ float wave_max_to_lane63(float v) {
  v = dpp_max<kRowShr + 1>(v);
  v = dpp_max<kRowShr + 2>(v);
  v = dpp_max<kRowShr + 4>(v);
  v = dpp_max<kRowShr + 8>(v);                    // lane 15 of every row: the row's maximum
  v = dpp_max<kRowBcast15, 0xa>(v);               // rows 1, 3 take in lane 15 of rows 0, 2
  v = dpp_max<kRowBcast31, 0xc>(v);               // rows 2, 3 take in lane 31
  return v;
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// Per-tile channel maxima of a producer, (tiles, C) floats per batch entry, as 16-byte words: thread t keeps words
// t, t + 256, ... — its channel quad (word index mod C / 4) is the same in every round because 256 is a multiple of it.
// kPmRounds words are in flight per thread; maps beyond 256 x 256 take further trips.
constexpr int kPmRounds = 4;
template <int C>
__device__ __forceinline__ void pm_issue(const float* __restrict__ pm, int b, int tiles, int t, float4 (&v)[kPmRounds]) {
  const float4* p4 = reinterpret_cast<const float4*>(pm) + (long)b * tiles * (C / 4);
  const int n4 = tiles * (C / 4);
#pragma unroll
  for (int u = 0; u < kPmRounds; ++u) {
    const int idx = u * kBlock + t;
    v[u] = p4[idx < n4 ? idx : t % (C / 4)];      // a repeated word of the same quad does not change a maximum
  }
}
// ... reduced over the workgroup's threads as far as a 16-lane row goes, then left in s_red[16 rows][C]
template <int C>
__device__ __forceinline__ void pm_reduce_rows(const float* __restrict__ pm, int b, int tiles, int t,
                                               const float4 (&v)[kPmRounds], float* s_red) {
  float4 m = v[0];
#pragma unroll
  for (int u = 1; u < kPmRounds; ++u) m = max4(m, v[u]);
  const int n4 = tiles * (C / 4);
  if (n4 > kPmRounds * kBlock) {                  // not at any size OcRFDet uses
    const float4* p4 = reinterpret_cast<const float4*>(pm) + (long)b * tiles * (C / 4);
    for (int idx = kPmRounds * kBlock + t; idx < n4; idx += kBlock) m = max4(m, p4[idx]);
  }
  constexpr int S = C / 4;                        // lanes S apart hold the same channel quad
  auto step = [&](auto tag) {
    constexpr int N = decltype(tag)::value;
    if constexpr (N >= S && N % S == 0) {
      m.x = dpp_max<kRowShr + N>(m.x); m.y = dpp_max<kRowShr + N>(m.y);
      m.z = dpp_max<kRowShr + N>(m.z); m.w = dpp_max<kRowShr + N>(m.w);
    }
  };
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 8>{});
  const int l16 = t & 15;
  if (l16 >= 16 - S)                              // the last S lanes of a row: one per quad
    *reinterpret_cast<float4*>(s_red + (t >> 4) * C + (l16 - (16 - S)) * 4) = m;
}

// HeightAttention gate (view_transformer_ocrf.py:447-461: global max-pool per channel, per height quarter
// q -> hid = q -> q without bias, sigmoid) of channel c = lane (clamped) from the row maxima; every lane of the wave
// runs it (the quarter's maxima come from the neighbouring lanes by quad_perm).  Same arithmetic order as
// hoa_height_gate_from_tiles_kernel.
template <int C>
__device__ __forceinline__ float gate_of_lane(const float* s_red, const float* s_w1, const float* s_w2, int lane) {
  constexpr int Q = C / 4;
  const int c = lane < C ? lane : C - 1;
  float m = s_red[c];
#pragma unroll
  for (int r = 1; r < 16; ++r) m = fmaxf(m, s_red[r * C + c]);
  float mx[Q];
  if constexpr (Q == 1) {
    mx[0] = m;
  } else if constexpr (Q == 2) {
    mx[0] = dpp_mov<0xA0>(m);                     // quad_perm [0,0,2,2]
    mx[1] = dpp_mov<0xF5>(m);                     // quad_perm [1,1,3,3]
  } else {
    static_assert(Q == 4, "channel counts of the reference converter");
    mx[0] = dpp_mov<0x00>(m); mx[1] = dpp_mov<0x55>(m); mx[2] = dpp_mov<0xAA>(m); mx[3] = dpp_mov<0xFF>(m);
  }
  const int gq = c / Q, o = c % Q;
  float hid[Q];
#pragma unroll
  for (int h = 0; h < Q; ++h) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < Q; ++i) acc = fmaf(s_w1[(gq * Q + h) * Q + i], mx[i], acc);
    hid[h] = fmaxf(acc, 0.f);
  }
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < Q; ++h) acc = fmaf(s_w2[(gq * Q + o) * Q + h], hid[h], acc);
  return sigmoidf_(acc);
}


}  // namespace hoa_gate

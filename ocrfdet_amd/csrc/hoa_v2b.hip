// HOA-2 of OcRFDet — OpacityVoxelToBEVConverter.forward (mmdet3d/models/necks/view_transformer_ocrf.py:463-518, with
// HeightAttention :421-461) for the architecture the reference instantiates (13 -> 4 -> 8 -> 16 -> 8 -> 4 -> 1), eval
// mode, as SIX launches shaped for LATENCY on MI355X (gfx950).
//
// The maps are tiny for this chip (13 x 200 x 200 floats at most): a block's time is the length of its dependent
// chain, not its bytes.  Round 3's block kernels lost their time in chains the compiler built from innocent source:
// predicated `if (in) load` per channel became branch -> load -> s_waitcnt vmcnt(0) sixteen times in a row, the
// constant-indexed weights became thirteen dependent s_load / s_waitcnt rounds, and the per-tile channel maxima 96
// serial ds_bpermute -> s_waitcnt pairs.  Here every kernel has ONE memory round trip:
//   * every global load of a workgroup — the whole packed weight vector (7 KB, two 16-byte loads per thread), the
//     producers' per-tile channel maxima, the raw halo values, the skip tensor, the addend — is issued at the top,
//     unconditionally, at CLAMPED addresses (out-of-image values are selected away afterwards: no branch, no wait);
//   * weights live in LDS and are read as wave-uniform (broadcast) words;
//   * reductions are DPP row shifts (v_max_f32 row_shr / row_bcast), no LDS crossbar traffic, no waits;
//   * a workgroup owns a 16 x 4 pixel tile, wave g of it a quarter of the channels: 4 x the workgroups of a
//     16 x 16 tile (1 300 / 350 / 104 instead of 338 / 98 / 32 at 2 x 200 x 200) and a quarter of the serial
//     arithmetic per thread; a wave's 64 lanes are the tile's 64 pixels, so the per-tile channel maxima are pure
//     wave reductions.
// The HeightAttention gate of a producer needs the GLOBAL channel maxima of its output, so each of the five blocks
// stays a launch of its own (a grid barrier costs more than a launch on this chip); the consumer
// rebuilds the gate from the producer's per-tile maxima in its prologue.
// Arithmetic order per output element is the one of csrc/hoa.hip's block-wise kernels (ocrf_hoa_unet_block +
// ocrf_hoa_height_gate_from_tiles + ocrf_hoa_gated_conv1x1): results are bit-identical (max is exact in any order).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "hoa_gate.h"
#include "launch.h"
#include "ocrf_hip.h"

namespace {

using namespace hoa_gate;

constexpr int kTW = 16, kTH = 4;                 // output tile: 64 pixels = the lanes of a wave
constexpr int kHW = kTW + 2, kHH = kTH + 2;      // with the 3x3 halo
constexpr int kHaloN = kHW * kHH;                // 108
// LDS row pitch of the halo tile.  A 32-lane group of a ds_read_b32 holds tile rows {0, 2} or {1, 3} (lane_row()):
// two rows apart = 48 words = 16 (mod 32 banks): the 9 taps are conflict-free.
constexpr int kPitch = 24;
constexpr int kLoW = kTW / 2 + 2, kLoH = kTH / 2 + 2;      // low-resolution window under the halo (MODE 2): 10 x 4
constexpr int kLoN = kLoW * kLoH;

constexpr int kCin[5] = {13, 4, 8, 16, 8}, kCout[5] = {4, 8, 16, 8, 4};     // e1, e2, bottleneck, d2, d1

// packed weights of the whole converter (floats), in this order (ocrfdet_amd/hoa.py: _packed_v2b)
struct V2bOffsets {
  int dw_w[5], dw_b[5], pw_w[5], pw_b[5], g_w1[5], g_w2[5], up_w[2], up_b[2], out_w, out_b, total, padded;
};
constexpr V2bOffsets v2b_offsets() {
  V2bOffsets o{};
  int p = 0;
  for (int k = 0; k < 5; ++k) {
    const int ci = kCin[k], co = kCout[k], q = co / 4;
    o.dw_w[k] = p; p += ci * 9;
    o.dw_b[k] = p; p += ci;
    o.pw_w[k] = p; p += co * ci;
    o.pw_b[k] = p; p += co;
    o.g_w1[k] = p; p += 4 * q * q;       // HeightAttention(co, co, ratio = 1): hid = q
    o.g_w2[k] = p; p += 4 * q * q;
  }
  o.up_w[0] = p; p += 16 * 8 * 4;  o.up_b[0] = p; p += 8;      // upconv2
  o.up_w[1] = p; p += 8 * 4 * 4;   o.up_b[1] = p; p += 4;      // upconv1
  o.out_w = p; p += 4;
  o.out_b = p; p += 1;
  o.total = p;
  o.padded = (p + 3) & ~3;               // the vector is handed over padded to whole 16-byte words
  return o;
}
constexpr V2bOffsets kOff = v2b_offsets();
constexpr int kW4 = kOff.padded / 4;                         // float4 words of the weight vector
constexpr int kW4PerThread = (kW4 + kBlock - 1) / kBlock;    // 2
constexpr int kWLds = kW4PerThread * kBlock * 4;              // LDS copy, padded so that every thread stores every word it loaded

struct V2bArgs {
  const float* src0; const float* src1; const float* pm0; const float* pm1;
  const float* weights; const float* addend;
  float* out; float* pm_out;
  int H, W, H0, W0, tiles0, tiles1;
};

// tile row of a lane: rows {0, 2} in lanes 0-31, rows {1, 3} in lanes 32-63 (see kPitch)
__device__ __forceinline__ int lane_row(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }

// One conv_block (depthwise 3x3 -> 1x1 with BatchNorm folded -> ReLU, view_transformer_ocrf.py:485-491) of the
// converter with everything around it folded in (:497-515).  K = 0 encoder1 (+ positional addend), 1 encoder2 and
// 2 bottleneck (2x2 max-pool of the gated producer), 3 decoder2 and 4 decoder1 (ConvTranspose2d k2 s2 of the gated
// producer, concatenated with the gated skip tensor).
template <int K>
__global__ __launch_bounds__(kBlock) void hoa_v2b_block_kernel(V2bArgs a) {
  OCRF_MAIN_CHAIN_PRIO();
  constexpr int MODE = K == 0 ? 0 : (K <= 2 ? 1 : 2);
  constexpr int S0 = K >= 1 ? K - 1 : 0;                         // producer of src0
  constexpr int S1 = K == 3 ? 1 : 0;                             // producer of the skip tensor (K >= 3)
  constexpr int C0 = K == 0 ? 13 : kCout[S0];
  constexpr int C1 = K >= 3 ? kCout[S1] : 0;
  constexpr int CUP = MODE == 2 ? kCin[K] - C1 : 0;
  constexpr int CF = MODE == 2 ? CUP : C0;
  constexpr int CIN = CF + C1, COUT = kCout[K], CPW = COUT / 4;  // CPW output channels per wave
  static_assert(CIN == kCin[K], "channel bookkeeping");
  constexpr int UP = K - 3;
  constexpr int NV0 = (MODE == 2) ? (C0 * kLoN + kBlock - 1) / kBlock : (C0 * kHaloN + kBlock - 1) / kBlock;
  constexpr int NV1 = (C1 * kHaloN + kBlock - 1) / kBlock;
  constexpr int NUP = (CUP * kHaloN + kBlock - 1) / kBlock;

  __shared__ __attribute__((aligned(16))) float s_w[kWLds];
  __shared__ float s_v[CIN * kHH * kPitch];
  __shared__ float s_lo[MODE == 2 ? C0 * kLoN : 1];
  __shared__ float s_d[CIN * 64];
  __shared__ __attribute__((aligned(16))) float s_red0[16 * 16], s_red1[16 * 16];
  __shared__ float s_g0[16], s_g1[16];

  const int t = threadIdx.x, lane = t & 63, g = t >> 6, b = blockIdx.z;
  const int ty0 = blockIdx.y * kTH, tx0 = blockIdx.x * kTW;
  const int H = a.H, W = a.W;

  // ---------------------------------------------------------------- every global load of the workgroup
  float4 wq[kW4PerThread];
  {
    const float4* w4 = reinterpret_cast<const float4*>(a.weights);
#pragma unroll
    for (int i = 0; i < kW4PerThread; ++i) wq[i] = w4[min(t + i * kBlock, kW4 - 1)];
  }
  float4 pmv0[kPmRounds], pmv1[kPmRounds];
  if constexpr (K >= 1) pm_issue<C0>(a.pm0, b, a.tiles0, t, pmv0);
  if constexpr (C1 > 0) pm_issue<C1>(a.pm1, b, a.tiles1, t, pmv1);

  float raw0[NV0];                                               // MODE 1: already the 2x2 maximum
  float2 rawa[MODE == 1 ? NV0 : 1], rawb[MODE == 1 ? NV0 : 1];
#pragma unroll
  for (int v = 0; v < NV0; ++v) {
    const int idx = t + v * kBlock;
    if constexpr (MODE == 2) {
      const int c = min(idx / kLoN, C0 - 1), pos = idx % kLoN;
      const int y = min(max((ty0 >> 1) - 1 + pos / kLoW, 0), a.H0 - 1), x = min(max((tx0 >> 1) - 1 + pos % kLoW, 0), a.W0 - 1);
      raw0[v] = a.src0[(((long)b * C0 + c) * a.H0 + y) * a.W0 + x];
    } else {
      const int c = min(idx / kHaloN, C0 - 1), pos = idx % kHaloN;
      const int y = min(max(ty0 - 1 + pos / kHW, 0), H - 1), x = min(max(tx0 - 1 + pos % kHW, 0), W - 1);
      if constexpr (MODE == 1) {
        const float* p = a.src0 + (((long)b * C0 + c) * a.H0 + 2 * y) * a.W0 + 2 * x;      // W0 = 2 W: 8-byte aligned
        rawa[v] = *reinterpret_cast<const float2*>(p);
        rawb[v] = *reinterpret_cast<const float2*>(p + a.W0);
      } else {
        raw0[v] = a.src0[(((long)b * C0 + c) * a.H0 + y) * a.W0 + x];
      }
    }
  }
  float raw1[NV1 > 0 ? NV1 : 1];
#pragma unroll
  for (int v = 0; v < NV1; ++v) {
    const int idx = t + v * kBlock;
    const int c = min(idx / kHaloN, C1 - 1), pos = idx % kHaloN;
    const int y = min(max(ty0 - 1 + pos / kHW, 0), H - 1), x = min(max(tx0 - 1 + pos % kHW, 0), W - 1);
    raw1[v] = a.src1[(((long)b * C1 + c) * H + y) * W + x];
  }
  // this thread's output pixel and channels [g CPW, (g + 1) CPW)
  const int ly = lane_row(lane), lx = lane & 15;
  const int oy = ty0 + ly, ox = tx0 + lx;
  const bool valid = oy < H && ox < W;
  const long plane = (long)H * W;
  const long opix = (long)min(oy, H - 1) * W + min(ox, W - 1);
  float add[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k) add[k] = 0.f;
  if constexpr (K == 0) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) add[k] = a.addend[((long)b * COUT + g * CPW + k) * plane + opix];
  }

  // ---------------------------------------------------------------- weights and row maxima into LDS
#pragma unroll
  for (int i = 0; i < kW4PerThread; ++i) reinterpret_cast<float4*>(s_w)[t + i * kBlock] = wq[i];   // no branch: a branch
  // here makes hipcc sink the load into it and wait for it there
  if constexpr (K >= 1) pm_reduce_rows<C0>(a.pm0, b, a.tiles0, t, pmv0, s_red0);
  if constexpr (C1 > 0) pm_reduce_rows<C1>(a.pm1, b, a.tiles1, t, pmv1, s_red1);
  if constexpr (K >= 1) {
    __syncthreads();
    // gates of the producers: wave 0 the source, wave 1 the skip tensor
    if (g == 0) {
      const float gt = gate_of_lane<C0>(s_red0, s_w + kOff.g_w1[S0], s_w + kOff.g_w2[S0], lane);
      if (lane < C0) s_g0[lane] = gt;
    }
    if constexpr (C1 > 0) {
      if (g == 1) {
        const float gt = gate_of_lane<C1>(s_red1, s_w + kOff.g_w1[S1], s_w + kOff.g_w2[S1], lane);
        if (lane < C1) s_g1[lane] = gt;
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- the virtual input tile (zero outside the image)
#pragma unroll
  for (int v = 0; v < NV0; ++v) {
    const int idx = t + v * kBlock;
    if constexpr (MODE == 2) {
      if (idx < C0 * kLoN) s_lo[idx] = raw0[v] * s_g0[idx / kLoN];
    } else {
      const int c = idx / kHaloN, pos = idx % kHaloN;
      const int hy = pos / kHW, hx = pos % kHW;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      const bool in = y >= 0 && y < H && x >= 0 && x < W;
      if (idx < C0 * kHaloN) {
        float val;
        if constexpr (MODE == 1) {
          // gates are sigmoids (> 0): max-pooling before or after the multiply is the same
          val = fmaxf(fmaxf(rawa[v].x, rawa[v].y), fmaxf(rawb[v].x, rawb[v].y)) * s_g0[c];
        } else {
          val = raw0[v];                                          // encoder1 reads the caller's tensor: no gate
        }
        s_v[(c * kHH + hy) * kPitch + hx] = in ? val : 0.f;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV1; ++v) {
    const int idx = t + v * kBlock;
    const int c = idx / kHaloN, pos = idx % kHaloN;
    const int hy = pos / kHW, hx = pos % kHW;
    const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    if (idx < C1 * kHaloN) s_v[((CF + c) * kHH + hy) * kPitch + hx] = in ? raw1[v] * s_g1[c] : 0.f;
  }
  if constexpr (MODE == 2) {
    __syncthreads();
    // ConvTranspose2d(k = 2, s = 2) of the gated low-resolution window: one (channel, halo position) per item
    const float* upw = s_w + kOff.up_w[UP];
    const float* upb = s_w + kOff.up_b[UP];
#pragma unroll
    for (int v = 0; v < NUP; ++v) {
      const int idx = t + v * kBlock;
      const int co = min(idx / kHaloN, CUP - 1), pos = idx % kHaloN;
      const int hy = pos / kHW, hx = pos % kHW;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      const bool in = y >= 0 && y < H && x >= 0 && x < W;
      // ty0, tx0 are even: row (y >> 1) of the window is ((hy - 1) >> 1) + 1, the 2x2 tap is the parity of (y, x)
      const int lo = (((hy + 1) >> 1)) * kLoW + ((hx + 1) >> 1);
      const int ki = ((hy + 1) & 1) * 2 + ((hx + 1) & 1);
      float acc = upb[co];
#pragma unroll
      for (int ci = 0; ci < C0; ++ci) acc = fmaf(s_lo[ci * kLoN + lo], upw[(ci * CUP + co) * 4 + ki], acc);
      if (idx < CUP * kHaloN) s_v[(co * kHH + hy) * kPitch + hx] = in ? acc : 0.f;
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- depthwise 3x3: wave g takes channels g, g + 4, ...
  {
    const float* dww = s_w + kOff.dw_w[K];
    const float* dwb = s_w + kOff.dw_b[K];
#pragma unroll
    for (int cc = 0; cc < (CIN + 3) / 4; ++cc) {
      const int c = g + 4 * cc;
      if (c < CIN) {                                              // wave-uniform
        float d = dwb[c];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) d = fmaf(s_v[(c * kHH + ly + i) * kPitch + lx + j], dww[c * 9 + i * 3 + j], d);
        s_d[c * 64 + lane] = d;
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- 1x1 (BatchNorm folded) -> ReLU (+ addend), CPW channels
  float outv[CPW];
  {
    const float* pww = s_w + kOff.pw_w[K];
    const float* pwb = s_w + kOff.pw_b[K];
    float d[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) d[c] = s_d[c * 64 + lane];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int co = g * CPW + k;
      float acc = pwb[co];
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc = fmaf(d[c], pww[co * CIN + c], acc);
      float v = fmaxf(acc, 0.f);
      if constexpr (K == 0) v += add[k];
      // computed by every lane, OUTSIDE the `valid` branch: hipcc otherwise sinks each channel's arithmetic into a
      // branch of its own with a vmcnt(0) wait (= the previous channel's store) at its top
      asm volatile("" : "+v"(v));
      outv[k] = v;
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < CPW; ++k) a.out[((long)b * COUT + g * CPW + k) * plane + opix] = outv[k];
    }
#pragma unroll
    for (int k = 0; k < CPW; ++k) outv[k] = valid ? outv[k] : -INFINITY;
  }
  // per-tile channel maxima for the block's own HeightAttention: (tile, channel) floats per batch entry
#pragma unroll
  for (int k = 0; k < CPW; ++k) outv[k] = wave_max_to_lane63(outv[k]);
  if (lane == 63) {
    const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
    float* q = a.pm_out + (((long)b * gridDim.x * gridDim.y + tile) * COUT + g * CPW);
#pragma unroll
    for (int k = 0; k < CPW; ++k) q[k] = outv[k];
  }
}

// Final 1x1 output conv (:516) over the gated decoder1 output (B, 4, plane); the gate from decoder1's per-tile maxima.
__global__ __launch_bounds__(kBlock) void hoa_v2b_out_kernel(const float* __restrict__ x, const float* __restrict__ pm,
                                                             int tiles, const float* __restrict__ weights, long plane,
                                                             float* __restrict__ out) {
  OCRF_MAIN_CHAIN_PRIO();
  constexpr int C = 4;
  __shared__ __attribute__((aligned(16))) float s_w[kWLds];
  __shared__ __attribute__((aligned(16))) float s_red[16 * C];
  __shared__ float s_g[C];
  const int t = threadIdx.x, lane = t & 63, b = blockIdx.y;
  const long pix = (long)blockIdx.x * kBlock + t;
  const long pc = pix < plane ? pix : plane - 1;
  float4 wq[kW4PerThread];
  {
    const float4* w4 = reinterpret_cast<const float4*>(weights);
#pragma unroll
    for (int i = 0; i < kW4PerThread; ++i) wq[i] = w4[min(t + i * kBlock, kW4 - 1)];
  }
  float4 pmv[kPmRounds];
  pm_issue<C>(pm, b, tiles, t, pmv);
  float xv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) xv[c] = x[((long)b * C + c) * plane + pc];
#pragma unroll
  for (int i = 0; i < kW4PerThread; ++i) reinterpret_cast<float4*>(s_w)[t + i * kBlock] = wq[i];   // no branch: a branch
  // here makes hipcc sink the load into it and wait for it there
  pm_reduce_rows<C>(pm, b, tiles, t, pmv, s_red);
  __syncthreads();
  if (t < 64) {
    const float gt = gate_of_lane<C>(s_red, s_w + kOff.g_w1[4], s_w + kOff.g_w2[4], lane);
    if (lane < C) s_g[lane] = gt;
  }
  __syncthreads();
  float acc = s_w[kOff.out_b];
#pragma unroll
  for (int c = 0; c < C; ++c) acc = fmaf(xv[c] * s_g[c], s_w[kOff.out_w + c], acc);
  asm volatile("" : "+v"(acc));
  if (pix < plane) out[(long)b * plane + pix] = acc;
}

inline int tiles_of(int h, int w) { return ((w + kTW - 1) / kTW) * ((h + kTH - 1) / kTH); }

}  // namespace

namespace ocrf {
void v2b_deferred_pointers(const void* workspace, const float* weights, int B, int H, int W, const float** d1,
                           const float** pm, int* tiles, const float** out_w, const float** out_b, const float** g_w1,
                           const float** g_w2) {
  const int hs[5] = {H, H / 2, H / 4, H / 2, H}, wsz[5] = {W, W / 2, W / 4, W / 2, W};
  const float* p = static_cast<const float*>(workspace);
  for (int k = 0; k < 4; ++k) p += (size_t)B * kCout[k] * hs[k] * wsz[k] + (size_t)B * kCout[k] * tiles_of(hs[k], wsz[k]);
  *d1 = p;
  *pm = p + (size_t)B * kCout[4] * H * W;
  *tiles = tiles_of(H, W);
  *out_w = weights + kOff.out_w; *out_b = weights + kOff.out_b;
  *g_w1 = weights + kOff.g_w1[4]; *g_w2 = weights + kOff.g_w2[4];
}
}  // namespace ocrf

extern "C" {

// length of the packed weight vector ocrf_hoa_v2b_forward reads: the converter's weights in v2b_offsets() order,
// zero-padded to a whole number of 16-byte words
int ocrf_hoa_v2b_weights_len(void) { return kOff.padded; }

size_t ocrf_hoa_v2b_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H < 4 || W < 4 || (H % 4) || (W % 4)) return 0;
  size_t n = 0;
  const int hs[5] = {H, H / 2, H / 4, H / 2, H}, wsz[5] = {W, W / 2, W / 4, W / 2, W};
  for (int k = 0; k < 5; ++k)
    n += (size_t)B * kCout[k] * hs[k] * wsz[k] + (size_t)B * kCout[k] * tiles_of(hs[k], wsz[k]);
  return (n * sizeof(float) + 255) / 256 * 256;
}

// The whole OpacityVoxelToBEVConverter.forward (view_transformer_ocrf.py:497-518) of the architecture OcRFDet
// instantiates as SIX launches in one call (FIVE with out == NULL: the output conv is then left to
// ocrf_hoa_opacity_mask_gate_v2b).  x (B,13,H,W), position (B,4,H,W), out (B,1,H,W); H, W multiples of 4;
// weights: ocrf_hoa_v2b_weights_len() floats, 16-byte aligned; workspace 16-byte aligned.
int ocrf_hoa_v2b_forward(const float* x, const float* position, const float* weights, int B, int H, int W,
                         void* workspace, size_t workspace_bytes, float* out, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !position || !weights || !workspace || B <= 0 || B > 65535 || H < 4 || W < 4 || (H % 4) || (W % 4) ||
      workspace_bytes < ocrf_hoa_v2b_workspace_bytes(B, H, W) ||
      ((reinterpret_cast<uintptr_t>(weights) | reinterpret_cast<uintptr_t>(workspace)) & 15u))
    return (int)hipErrorInvalidValue;
  const int hs[5] = {H, H / 2, H / 4, H / 2, H}, wsz[5] = {W, W / 2, W / 4, W / 2, W};
  float* act[5];
  float* pm[5];
  int tiles[5];
  float* p = static_cast<float*>(workspace);
  for (int k = 0; k < 5; ++k) {               // every size below is a multiple of 4 floats: 16-byte alignment holds
    tiles[k] = tiles_of(hs[k], wsz[k]);
    act[k] = p; p += (size_t)B * kCout[k] * hs[k] * wsz[k];
    pm[k] = p; p += (size_t)B * kCout[k] * tiles[k];
  }
  const int src0[5] = {-1, 0, 1, 2, 3}, skip[5] = {-1, -1, -1, 1, 0};
  for (int k = 0; k < 5; ++k) {
    V2bArgs a;
    const int s0 = src0[k], s1 = skip[k];
    a.src0 = s0 < 0 ? x : act[s0];
    a.src1 = s1 < 0 ? nullptr : act[s1];
    a.pm0 = s0 < 0 ? nullptr : pm[s0];
    a.pm1 = s1 < 0 ? nullptr : pm[s1];
    a.weights = weights;
    a.addend = k == 0 ? position : nullptr;
    a.out = act[k]; a.pm_out = pm[k];
    a.H = hs[k]; a.W = wsz[k];
    a.H0 = s0 < 0 ? H : hs[s0]; a.W0 = s0 < 0 ? W : wsz[s0];
    a.tiles0 = s0 < 0 ? 0 : tiles[s0]; a.tiles1 = s1 < 0 ? 0 : tiles[s1];
    const dim3 grid((wsz[k] + kTW - 1) / kTW, (hs[k] + kTH - 1) / kTH, B);
    switch (k) {
      case 0: ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_v2b_block_kernel<0>, grid, dim3(kBlock), 0, stream, a); break;
      case 1: ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_v2b_block_kernel<1>, grid, dim3(kBlock), 0, stream, a); break;
      case 2: ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_v2b_block_kernel<2>, grid, dim3(kBlock), 0, stream, a); break;
      case 3: ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_v2b_block_kernel<3>, grid, dim3(kBlock), 0, stream, a); break;
      default: ocrf::launch(OCRF_K_HOA_UNET_BLOCK, hoa_v2b_block_kernel<4>, grid, dim3(kBlock), 0, stream, a); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (!out) return 0;        // deferred: the output conv runs inside ocrf_hoa_opacity_mask_gate_v2b (csrc/hoa.hip)
  const long plane = (long)H * W;
  ocrf::launch(OCRF_K_HOA_OUT_CONV, hoa_v2b_out_kernel, dim3((unsigned)((plane + kBlock - 1) / kBlock), B), dim3(kBlock), 0,
               stream, static_cast<const float*>(act[4]), static_cast<const float*>(pm[4]), tiles[4], weights, plane, out);
  return (int)hipGetLastError();
}

}  // extern "C"

// The stages of OcRFViewTransformerFull.view_transform_core that sit between the two poolings, the
// render and HOA, eval mode, fused for MI355X:
//
//   ocrf_prefilter            depth / semantic softmax + thresholds, feature written channels-last
//                             (view_transformer_ocrf.py:1323-1331 + the permute of :875/:901)
//   ocrf_pillar_sample_mean   lidar_points_to_image_values + color_voxels (:924-971): masked
//                             bilinear taps of every camera averaged per pillar point
//   ocrf_retain_valid_pixels  retain_valid_pixels (:1004-1024) without the B*6*13 Python loop
//   ocrf_gauss_heads          VoxelFeatureExtractor (:520-531) + the four Gaussian heads
//                             (:272-320, :1130-1133): the (B,13,Y,X,80) voxel feature never exists
//   ocrf_nerf_alpha/_render   the NeRF branch (:1094-1121).  ResizeNetwork (:534-554) has no
//                             non-linearity, and every consumer of its 80-channel full-resolution
//                             output starts with a Linear, so upsample2 -> upsample3 -> Linear is
//                             composed by the caller into one 32 -> (8x8 positions) map per
//                             consumer: the 6 x (80,H,W) feature images are never materialised.
//
// All arithmetic is fp32 with -ffp-contract=off.  The pre-filter's softmax and the sampling taps use
// IEEE division / ocml expf (their results feed threshold tests and index-like decisions); the MLP
// activations use the hardware transcendentals (see sigmoidf / softplusf).
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

// Activations on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): the
// ocml expf / log1pf / IEEE-division sequences were as expensive as the whole first layer of the heads
// kernel.  Absolute error <= ~1e-6 on outputs of O(1), two orders below the 1e-4 parity bar.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf(float x) { return rcp_fast(1.0f + __expf(-x)); }
__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : __logf(1.0f + __expf(x)); }

// ------------------------------------------------------------------------------------------
// prefilter: one workgroup = PXW consecutive pixels of one image x (256 / PXW) depth slices.  The map
// is small (12 images x 704 pixels at the reference shape), so the tile is narrow (16 pixels: 528
// workgroups) to give every CU work; each thread loads its <= PER depth logits ONCE, all in flight.
// ------------------------------------------------------------------------------------------
template <int PXW, int PER>
__global__ __launch_bounds__(256) void neck_prefilter_kernel(
    const float* __restrict__ x, int D, int C, int HW, float depth_thr, float sem_thr,
    float* __restrict__ depth, float* __restrict__ filter_depth, float* __restrict__ semantic,
    float* __restrict__ feat_cl) {
  constexpr int NSL = 256 / PXW;
  __shared__ float red[NSL][PXW];
  __shared__ float keep[PXW];
  extern __shared__ float tile[];            // [PXW][C + 1] transposition buffer
  const int lane = threadIdx.x % PXW, sl = threadIdx.x / PXW;
  const int bn = blockIdx.y, p0 = blockIdx.x * PXW, p = p0 + lane;
  const bool live = p < HW;
  const float* xi = x + (size_t)bn * (D + 2 + C) * HW;
  float v[PER];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = sl + NSL * i;
    v[i] = (live && d < D) ? xi[(size_t)d * HW + p] : -INFINITY;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) m = fmaxf(m, v[i]);
  red[sl][lane] = m;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NSL; ++k) m = fmaxf(m, red[k][lane]);
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = sl + NSL * i;
    v[i] = (live && d < D) ? expf(v[i] - m) : 0.0f;
    s += v[i];
  }
  red[sl][lane] = s;
  __syncthreads();
  s = 0.0f;
#pragma unroll
  for (int k = 0; k < NSL; ++k) s += red[k][lane];        // same order in every slice: one value per pixel
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = sl + NSL * i;
    if (live && d < D) {
      const float pr = v[i] / s;
      const size_t o = ((size_t)bn * D + d) * HW + p;
      depth[o] = pr;
      filter_depth[o] = pr < depth_thr ? 0.0f : pr;
    }
  }
  if (sl == 0) {
    float k = 0.0f;
    if (live) {
      const float a = xi[(size_t)D * HW + p], b = xi[(size_t)(D + 1) * HW + p];
      const float mm = fmaxf(a, b), ea = expf(a - mm), eb = expf(b - mm), den = ea + eb;
      const float s0 = ea / den, s1 = eb / den;
      semantic[((size_t)bn * 2 + 0) * HW + p] = s0;
      semantic[((size_t)bn * 2 + 1) * HW + p] = s1;
      k = s1 >= sem_thr ? 1.0f : 0.0f;
    }
    keep[lane] = k;
  }
  __syncthreads();
  // feature: read (c, pixel) coalesced along pixels, write the (pixel, c) tile as one contiguous run
  const int stride = C + 1;
  for (int c = sl; c < C; c += NSL)
    tile[lane * stride + c] = live ? xi[(size_t)(D + 2 + c) * HW + p] * keep[lane] : 0.0f;
  __syncthreads();
  const int npx = min(PXW, HW - p0);
  float* dst = feat_cl + ((size_t)bn * HW + p0) * C;
  for (int i = threadIdx.x; i < npx * C; i += 256) dst[i] = tile[(i / C) * stride + (i % C)];
}

// ------------------------------------------------------------------------------------------
// bilinear tap exactly as F.grid_sample(align_corners=True, padding zeros) evaluates it after the
// reference's own normalisation (view_transformer_ocrf.py:929-931)
// ------------------------------------------------------------------------------------------
struct Tap {
  int x0, y0;
  float w[4];       // nw, ne, sw, se (0 for out-of-range corners)
  bool in[4];
};

__device__ __forceinline__ Tap make_tap(float px, float py, int H, int W) {
  const float wm = (float)(W - 1), hm = (float)(H - 1);
  const float xn = (px / wm) * 2.0f - 1.0f, yn = (py / hm) * 2.0f - 1.0f;
  const float ix = ((xn + 1.0f) / 2.0f) * wm, iy = ((yn + 1.0f) / 2.0f) * hm;
  const float fx = floorf(ix), fy = floorf(iy);
  const float x1 = fx + 1.0f, y1 = fy + 1.0f;
  Tap t;
  t.w[0] = (x1 - ix) * (y1 - iy);
  t.w[1] = (ix - fx) * (y1 - iy);
  t.w[2] = (x1 - ix) * (iy - fy);
  t.w[3] = (ix - fx) * (iy - fy);
  // the float -> int conversion saturates, so far-away (or non-finite) coordinates stay out of range
  const bool fin = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  t.x0 = fin ? (int)fx : -4;
  t.y0 = fin ? (int)fy : -4;
  const bool xin0 = t.x0 >= 0 && t.x0 < W, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  const bool yin0 = t.y0 >= 0 && t.y0 < H, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  t.in[0] = xin0 && yin0;
  t.in[1] = xin1 && yin0;
  t.in[2] = xin0 && yin1;
  t.in[3] = xin1 && yin1;
  return t;
}

// A latency kernel (one thread per voxel, <= 2 of the N cameras see it): THREE memory round trips per thread instead of
// one per tap.  Round 4's form — `if (mask) load` per camera, `if (inside) load` per tap — compiled to a branch and an
// s_waitcnt vmcnt(0) behind every one of its 112 loads (C = 3): up to 12 dependent round trips per valid camera.  Here
// (1) the mask bytes of eight cameras, (2) the coordinates of the lane's next TWO valid cameras, (3) their 2 x 4 x C
// taps at clamped addresses, every load of a round issued before the first use; what a tap outside the image (or a
// lane without a second camera) contributes is selected away.  Same operations in the same order per output value.
template <int C>
__global__ __launch_bounds__(256) void neck_pillar_sample_mean_kernel(
    const float* __restrict__ imgs, const float2* __restrict__ pix, const unsigned char* __restrict__ mask,
    float* __restrict__ avg, int N, int Hi, int Wi, int ZQ) {
  const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const bool live = q < ZQ;
  const int qc = live ? q : ZQ - 1;
  const size_t plane = (size_t)Hi * Wi;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.0f;
  int cnt = 0;
  for (int n0 = 0; n0 < N; n0 += 8) {
    unsigned char m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = mask[((size_t)b * N + min(n0 + j, N - 1)) * ZQ + qc];
    unsigned rem = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) rem |= (live && n0 + j < N && m[j]) ? (1u << j) : 0u;
    cnt += __popc(rem);
    while (__ballot(rem != 0u) != 0ull) {          // (wave-uniform trip count: at most 4, one for most waves)
      bool ok[2];
      const float* img[2];
      float2 uv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        ok[k] = rem != 0u;
        const int cam = n0 + (ok[k] ? __ffs((int)rem) - 1 : 0);       // cameras in ascending order, as the loop had them
        rem &= rem - 1u;
        uv[k] = pix[((size_t)b * N + cam) * ZQ + qc];
        img[k] = imgs + ((size_t)b * N + cam) * C * plane;
      }
      Tap t[2];
      float v[2][C][4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        t[k] = make_tap(uv[k].x, uv[k].y, Hi, Wi);
        const int ya = min(max(t[k].y0, 0), Hi - 1), yb = min(max(t[k].y0 + 1, 0), Hi - 1);
        const int xa = min(max(t[k].x0, 0), Wi - 1), xb = min(max(t[k].x0 + 1, 0), Wi - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* pl = img[k] + (size_t)c * plane;
          v[k][c][0] = pl[(size_t)ya * Wi + xa];
          v[k][c][1] = pl[(size_t)ya * Wi + xb];
          v[k][c][2] = pl[(size_t)yb * Wi + xa];
          v[k][c][3] = pl[(size_t)yb * Wi + xb];
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float a = 0.0f;                              // one camera's bilinear sample, term by term
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float s = a + v[k][c][i] * t[k].w[i];
            a = t[k].in[i] ? s : a;
          }
          const float s = acc[c] + a;
          acc[c] = ok[k] ? s : acc[c];
        }
      }
    }
  }
  if (!live) return;
  const float den = cnt ? (float)cnt : 1.0f;
  float* o = avg + ((size_t)b * ZQ + q) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = acc[c] / den;
}

// ------------------------------------------------------------------------------------------
// retain_valid_pixels: 255-fill, then every valid projection copies its pixel (all writers of a
// pixel write the same value)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void neck_fill_kernel(float4* __restrict__ out, size_t n4, float* __restrict__ tail,
                                                        int ntail, float v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) out[i] = make_float4(v, v, v, v);
  if (i < (size_t)ntail) tail[i] = v;
}

__global__ __launch_bounds__(256) void neck_retain_scatter_kernel(
    const float* __restrict__ imgs, const float2* __restrict__ pix, const unsigned char* __restrict__ mask,
    const int* __restrict__ cam_sel, float* __restrict__ out, int N, int C, int H, int W, int ZQ) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.z;
  const int n = cam_sel ? cam_sel[b] : (int)blockIdx.y;
  if (q >= ZQ || n < 0 || n >= N) return;
  const size_t pi = ((size_t)b * N + n) * ZQ + q;
  if (!mask[pi]) return;
  const float2 uv = pix[pi];
  if (uv.x == -1.0f) return;                       // the reference's sentinel test (:1017)
  const int hi = max(W, H) - 1;
  // .long() truncates toward zero; clamp to [0, max(W,H)-1] as the reference does (:1018), then to
  // the image for memory safety (the reference would raise an index error there)
  int xi = (int)uv.x, yi = (int)uv.y;
  xi = min(max(xi, 0), min(hi, W - 1));
  yi = min(max(yi, 0), min(hi, H - 1));
  const float* src = imgs + ((size_t)b * N + n) * C * H * W + (size_t)yi * W + xi;
  float* dst = out + (cam_sel ? (size_t)b : ((size_t)b * N + n)) * C * H * W + (size_t)yi * W + xi;
  for (int c = 0; c < C; ++c) dst[(size_t)c * H * W] = src[(size_t)c * H * W];
}

// ------------------------------------------------------------------------------------------
// Gaussian heads.  One thread = one pillar x a group of HG heights: a channel of the NCHW BEV is read
// once per group (lanes = consecutive pillars: coalesced), its 16 first-layer weights arrive as ONE
// wave-uniform 64-byte scalar load (W1 is stored channel-major for that), and the 16 hidden units of
// every height of the group are accumulated as 8 packed pairs (v_pk_fma_f32).  HG x 16 accumulators
// live in registers; two groups (7 + 6 of the 13 heights) give 2 500 waves at 200x200 instead of
// 1 250 — with one wave per SIMD and a quarter of the SIMDs holding two, the finer grain is what
// shortens the critical path.
// params: lift_a[Zh] lift_b[Zh] | W1t[C][16] (columns S,R,A,Col: 4 each) | W1rgb[4][3] | b1[16] |
//         S: W2[3][4] b2[3] | R: W2[4][4] b2[4] | A: W2[1][4] b2[1] | Col: W2[3][4] b2[3]
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ZH, int HG>
__global__ __launch_bounds__(64) void neck_gauss_heads_kernel(
    const float* __restrict__ bev, const float* __restrict__ rgb_avg, const float* __restrict__ prm, int C, int YX,
    float* __restrict__ opacity, float* __restrict__ scales, float* __restrict__ rot, float* __restrict__ color) {
  constexpr int kGroups = (ZH + HG - 1) / HG;
  const int q = min(blockIdx.x * 64 + (int)threadIdx.x, YX - 1), b = blockIdx.y / kGroups, h0 = (blockIdx.y % kGroups) * HG;
  float la[HG], lb[HG];      // lanes past the end redo the last pillar and store nothing
#pragma unroll
  for (int h = 0; h < HG; ++h) {        // the last group may be short: its spare slots repeat the last height
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
  f32x2 hid[HG][8];
#pragma unroll
  for (int h = 0; h < HG; ++h)
#pragma unroll
    for (int k = 0; k < 8; ++k) hid[h][k] = f32x2{0.0f, 0.0f};
  const float* bp = bev + (size_t)b * C * YX + q;
  // 8 channel reads in flight per lane, requested one batch AHEAD of the batch being accumulated (with two or three waves
  // per SIMD nothing else hides their latency: a batch's 560 VALU instructions take about as long as its loads)
  constexpr int kAhead = 8;
  float vv[kAhead], nx[kAhead];
  f32x2 w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = W1t[k];
#pragma unroll
  for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(j, C - 1) * YX];
  for (int c0 = 0; c0 < C; c0 += kAhead) {
#pragma unroll
    for (int j = 0; j < kAhead; ++j) vv[j] = nx[j];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) nx[j] = bp[(size_t)min(c0 + kAhead + j, C - 1) * YX];      // (clamped: no branch)
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      // the NEXT channel's 16 weights (one wave-uniform 64-byte scalar load) are requested before this channel's
      // arithmetic: loaded where they are used, every channel waited ~ 0.3 us for its own load — 80 times per wave
      f32x2 wn[8];
      // (scalar loads return out of order: any wait for one is a wait for all.  This channel's weights are therefore
      // waited for HERE, before the next channel's are requested — else the first use below waits for both)
      asm volatile("" ::"s"(w[0].x));
      const int cn = min(c0 + j + 1, C - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) wn[k] = W1t[cn * 8 + k];
      __builtin_amdgcn_sched_barrier(0);       // (left alone the scheduler sinks the load back to its first use)
      if (c0 + j < C) {
#pragma unroll
        for (int h = 0; h < HG; ++h) {
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
  // epilogue.  The 3-float rows of rgb / scales / colour go through an LDS transposition (one wave =
  // 64 consecutive Gaussians = 768 contiguous bytes per array): per-lane 12-byte-stride accesses cost
  // three partial-line transactions per cache line.
  __shared__ float stg[2][64 * 3];
  const int lane = threadIdx.x;
  const int q0 = blockIdx.x * 64, nq = min(64, YX - q0);
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    if (h0 + h >= ZH) break;
    const size_t g0 = ((size_t)b * ZH + h0 + h) * YX + q0, g = g0 + lane;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (lane + 64 * k < nq * 3) stg[0][lane + 64 * k] = rgb_avg[g0 * 3 + lane + 64 * k];
    __syncthreads();
    constexpr float k255 = 1.0f / 255.0f;
    const float r01[3] = {stg[0][lane * 3] * k255, stg[0][lane * 3 + 1] * k255, stg[0][lane * 3 + 2] * k255};
    float hv[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { hv[2 * k] = hid[h][k].x; hv[2 * k + 1] = hid[h][k].y; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int j = 0; j < 3; ++j) hv[12 + k] = fmaf(W1rgb[k * 3 + j], r01[j], hv[12 + k]);
#pragma unroll
    for (int k = 0; k < 16; ++k) hv[k] = fmaxf(hv[k] + b1[k], 0.0f);
    __syncthreads();
    for (int o = 0; o < 3; ++o) {                    // scales: softplus
      float v = S2[12 + o];
      for (int k = 0; k < 4; ++k) v = fmaf(S2[o * 4 + k], hv[k], v);
      stg[0][lane * 3 + o] = softplusf(v);
    }
    for (int o = 0; o < 3; ++o) {                    // colour: sigmoid
      float v = C2[12 + o];
      for (int k = 0; k < 4; ++k) v = fmaf(C2[o * 4 + k], hv[12 + k], v);
      stg[1][lane * 3 + o] = sigmoidf(v);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (lane + 64 * k < nq * 3) {
        scales[g0 * 3 + lane + 64 * k] = stg[0][lane + 64 * k];
        color[g0 * 3 + lane + 64 * k] = stg[1][lane + 64 * k];
      }
    float r[4], nn = 0.0f;                           // rotation: L2-normalised (F.normalize eps 1e-12)
    for (int o = 0; o < 4; ++o) {
      float v = R2[16 + o];
      for (int k = 0; k < 4; ++k) v = fmaf(R2[o * 4 + k], hv[4 + k], v);
      r[o] = v;
      nn = fmaf(v, v, nn);
    }
    const float inv = rcp_fast(fmaxf(sqrtf(nn), 1e-12f));
    float v = A2[4];
    for (int k = 0; k < 4; ++k) v = fmaf(A2[k], hv[8 + k], v);
    if (lane < nq) {
      *reinterpret_cast<float4*>(rot + g * 4) = make_float4(r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv);
      opacity[g] = sigmoidf(v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// NeRF branch.  z (M,32,h2,w2) is the output of ResizeNetwork.conv2; an output pixel (y, x) of the
// 8x up-sampled image reads z[:, y/8, x/8] and the composed map of its sub-position (y%8, x%8).
// ------------------------------------------------------------------------------------------
// One thread = 8 horizontally consecutive output pixels (one coarse pixel j, one sub-row a): the 32
// channels of z[:, i, j] are read once for the 8 pixels, and since the sub-row is uniform over the
// workgroup (blockIdx.y = output row) the 32 x 8 weights are wave-uniform scalar loads.  4 loads and
// 16 packed FMAs per pixel instead of 64 loads and 32 FMAs; the 8 results leave as two float4 stores.
__global__ __launch_bounds__(64) void neck_nerf_alpha_kernel(
    const float* __restrict__ z, const float* __restrict__ Wm /*[32][64]*/, const float* __restrict__ cm /*[64]*/,
    float* __restrict__ alpha, int h2, int w2) {
  const int j = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, m = blockIdx.z;
  if (j >= w2) return;
  const int a = y & 7;
  const float* zp = z + ((size_t)m * 32 * h2 + (y >> 3)) * w2 + j;
  const f32x2* wrow = reinterpret_cast<const f32x2*>(Wm + a * 8);
  f32x2 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = f32x2{cm[a * 8 + 2 * k], cm[a * 8 + 2 * k + 1]};
#pragma unroll 8
  for (int ci = 0; ci < 32; ++ci) {
    const float zv = zp[(size_t)ci * h2 * w2];
    const f32x2 zz = {zv, zv};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_elementwise_fma(wrow[ci * 32 + k], zz, acc[k]);
  }
  float o[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = 1.0f - __expf(-softplusf(acc[k].x));
    o[2 * k + 1] = 1.0f - __expf(-softplusf(acc[k].y));
  }
  float4* dst = reinterpret_cast<float4*>(alpha + ((size_t)m * 8 * h2 + y) * (8 * w2) + 8 * j);
  dst[0] = make_float4(o[0], o[1], o[2], o[3]);
  dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// prm: W12[12][32][64] | c12[12][64] | W1rgb[12][3] | Cn: W2[3][4] b2[3] | R1: W2[3][4] b2[3] | R2: W2[1][4] b2[1]
// hidden order: C_MLP_nerf (0-3), img_feat_resize1 (4-7), img_feat_resize2 (8-11).  The depth weight is a
// softmax over a size-1 axis (:1108), i.e. 1, so D_MLP_nerf never reaches an output and is not evaluated.
__global__ __launch_bounds__(256) void neck_nerf_render_kernel(
    const float* __restrict__ z, const int* __restrict__ cam_sel, const float* __restrict__ alpha,
    const float* __restrict__ sparse_rgb, const float* __restrict__ prm, int N, int h2, int w2,
    float* __restrict__ img_n, float* __restrict__ dep_n) {
  const int Wo = 8 * w2, Ho = 8 * h2;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= Wo) return;
  const int cs = cam_sel[b];
  if (cs < 0 || cs >= N) return;
  const int m = b * N + cs;
  const int pos = (y & 7) * 8 + (x & 7);
  const float* zp = z + ((size_t)m * 32 * h2 + (y >> 3)) * w2 + (x >> 3);
  const float* W12 = prm;
  const float* c12 = W12 + 12 * 32 * 64;
  const float* W1rgb = c12 + 12 * 64;
  const float* Cn = W1rgb + 36;
  const float* R1 = Cn + 15;
  const float* R2 = R1 + 15;
  float hid[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) hid[k] = c12[k * 64 + pos];
  for (int ci = 0; ci < 32; ++ci) {
    const float zv = zp[(size_t)ci * h2 * w2];
#pragma unroll
    for (int k = 0; k < 12; ++k) hid[k] = fmaf(zv, W12[(k * 32 + ci) * 64 + pos], hid[k]);
  }
  const size_t pix = (size_t)y * Wo + x, plane = (size_t)Ho * Wo;
  const float* sp = sparse_rgb + (size_t)b * 3 * plane + pix;
  constexpr float k255 = 1.0f / 255.0f;
  const float r01[3] = {sp[0] * k255, sp[plane] * k255, sp[2 * plane] * k255};
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float v = hid[k];
    for (int j = 0; j < 3; ++j) v = fmaf(W1rgb[k * 3 + j], r01[j], v);
    hid[k] = fmaxf(v, 0.0f);
  }
  float cw[3], rad[3];
  for (int o = 0; o < 3; ++o) {
    float v = Cn[12 + o], u = R1[12 + o];
    for (int k = 0; k < 4; ++k) {
      v = fmaf(Cn[o * 4 + k], hid[k], v);
      u = fmaf(R1[o * 4 + k], hid[4 + k], u);
    }
    cw[o] = sigmoidf(v);
    rad[o] = fmaxf(u, 0.0f);
  }
  const float mx = fmaxf(cw[0], fmaxf(cw[1], cw[2]));
  const float e0 = __expf(cw[0] - mx), e1 = __expf(cw[1] - mx), e2 = __expf(cw[2] - mx), ies = rcp_fast(e0 + e1 + e2);
  float d = R2[4];
  for (int k = 0; k < 4; ++k) d = fmaf(R2[k], hid[8 + k], d);
  d = fmaxf(d, 0.0f);
  const float a = alpha[(size_t)m * plane + pix];
  float* io = img_n + (size_t)b * 3 * plane + pix;
  io[0] = a * (rad[0] * (e0 * ies));
  io[plane] = a * (rad[1] * (e1 * ies));
  io[2 * plane] = a * (rad[2] * (e2 * ies));
  dep_n[(size_t)b * plane + pix] = a * d;
}

// ------------------------------------------------------------------------------------------
// DualFeatFusion (view_transformer_ocrf.py:203-213 with MS_CAM :36-66), eval mode: per pixel
//   h = relu(W1 [x1; x2] + b1)   (2C -> M, BatchNorm folded by the caller)
//   l = W2 h + b2                (M -> C, BatchNorm folded)
//   cf = sigmoid(l + g[b])       g = the global-attention branch of the pooled map, (B,C)
//   out = cf * x1 + (1 - cf) * x2
// Lanes = consecutive pixels (every plane read / write is coalesced), the weights are wave-uniform
// scalar loads, both layers accumulate in packed pairs (v_pk_fma_f32).  Replaces the
// reference's cat + 2 GEMMs + 2 BatchNorm + ReLU + 7 elementwise passes over (B,C..2C,Y,X).
// prm: W1t[2C][M] | b1[M] | W2[C][M] | b2[C]
// ------------------------------------------------------------------------------------------
template <int C, int M>
__global__ __launch_bounds__(256) void neck_dual_fusion_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ prm,
    const float* __restrict__ gvec, float* __restrict__ out, int YX, const float* __restrict__ addend,
    float* __restrict__ out_plus) {
  // workgroup = 64 pixels x 4 waves; wave g owns hidden units [g*M/4, (g+1)*M/4) in the first layer
  // and output channels [g*C/4, (g+1)*C/4) in the second, the hidden vector crosses through LDS.
  // (One wave per 64 pixels left ~1 wave per SIMD, each streaming all 38 KB of weights through the
  // 16 KB scalar cache: 120 us.  Four waves per tile quarter both the serial chain and the weights
  // a wave touches.)
  constexpr int MQ = M / 4, CQ = C / 4;
  static_assert(M % 8 == 0 && C % 16 == 0, "quarter sizes must be even / a multiple of the read-ahead");
  __shared__ float s_h[M][64];
  // the wave index is wave-uniform, but only readfirstlane tells the compiler so (scalar weight loads)
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q = blockIdx.x * 64 + lane, b = blockIdx.y;
  const bool live = q < YX;
  const int qc = live ? q : YX - 1;
  const float* b1 = prm + 2 * C * M;
  const float* W2 = b1 + M;
  const float* b2 = W2 + C * M;
  const float* p1 = x1 + (size_t)b * C * YX + qc;
  const float* p2 = x2 + (size_t)b * C * YX + qc;
  f32x2 h[MQ / 2];
#pragma unroll
  for (int m = 0; m < MQ / 2; ++m) h[m] = f32x2{b1[grp * MQ + 2 * m], b1[grp * MQ + 2 * m + 1]};
  constexpr int kAhead = 8;                  // plane reads in flight per lane
  for (int half = 0; half < 2; ++half) {
    const float* p = half ? p2 : p1;
    for (int c0 = 0; c0 < C; c0 += kAhead) {
      float vv[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) vv[j] = p[(size_t)(c0 + j) * YX];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        const f32x2 v = {vv[j], vv[j]};
        const f32x2* w = reinterpret_cast<const f32x2*>(prm + (size_t)(half * C + c0 + j) * M + grp * MQ);
#pragma unroll
        for (int m = 0; m < MQ / 2; ++m) h[m] = __builtin_elementwise_fma(w[m], v, h[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MQ / 2; ++m) {
    s_h[grp * MQ + 2 * m][lane] = fmaxf(h[m].x, 0.0f);
    s_h[grp * MQ + 2 * m + 1][lane] = fmaxf(h[m].y, 0.0f);
  }
  __syncthreads();
  f32x2 hh[M / 2];
#pragma unroll
  for (int m = 0; m < M / 2; ++m) hh[m] = f32x2{s_h[2 * m][lane], s_h[2 * m + 1][lane]};
  float* po = out + (size_t)b * C * YX + qc;
  // out_plus = out + addend (the positional encoding ProbNet's first convolution reads the fused map with,
  // view_transformer_ocrf.py:1188: one more store here instead of an elementwise launch over the map)
  const float* pa = addend ? addend + (size_t)b * C * YX + qc : nullptr;
  float* pp = out_plus ? out_plus + (size_t)b * C * YX + qc : nullptr;
  for (int c0 = grp * CQ; c0 < (grp + 1) * CQ; c0 += kAhead / 2) {
    float va[kAhead / 2], vb[kAhead / 2], vc[kAhead / 2];
#pragma unroll
    for (int j = 0; j < kAhead / 2; ++j) {
      va[j] = p1[(size_t)(c0 + j) * YX];
      vb[j] = p2[(size_t)(c0 + j) * YX];
      vc[j] = pa ? pa[(size_t)(c0 + j) * YX] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < kAhead / 2; ++j) {
      const int c = c0 + j;
      f32x2 acc = {b2[c] + gvec[b * C + c], 0.0f};
      const f32x2* w = reinterpret_cast<const f32x2*>(W2 + (size_t)c * M);
#pragma unroll
      for (int m = 0; m < M / 2; ++m) acc = __builtin_elementwise_fma(w[m], hh[m], acc);
      const float cf = sigmoidf(acc.x + acc.y);
      const float o = cf * va[j] + (1.0f - cf) * vb[j];
      if (live) po[(size_t)c * YX] = o;
      if (live && pp) pp[(size_t)c * YX] = vc[j] + o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// CBAM / ProbNet tail (view_transformer_ocrf.py:68-137 ChannelAttention / SpatialAttention /
// ResCBAMBlock, :139-201 ProbNet; MS_CAM's global branch :50-58).  The convolutions themselves
// stay with MIOpen; everything between them is here: ~30 elementwise / reduction launches of the
// torch graph become 6.
// ---------------------------------------------------------------------------------------------
// One pass over every (b, c) plane: y += bias[c] (BatchNorm folded by the caller), optional ReLU,
// written back in place (WRITE), and / or partial sum + max of the result per plane chunk (STATS):
// grid (S, B*C); partials land at [(b * out_C + c_off + c) * S + chunk].
// `y2` (statistics only): the channels [C1, C) of the pass are the planes of a SECOND tensor (B, C - C1, plane) — the pooled
// vector of cat(x1, x2) in one launch without the cat (ocrf_plane_stats_pair); y2 == null: C1 = C.
template <bool WRITE, bool RELU, bool STATS>
__global__ __launch_bounds__(256) void neck_plane_pass_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                              int C, long plane, int S, int out_C, int c_off,
                                                              float* __restrict__ psum, float* __restrict__ pmax,
                                                              float* __restrict__ y2, int C1) {
  __shared__ float s_sum[4], s_max[4];
  const long bc = blockIdx.y;
  const int b = (int)(bc / C), c = (int)(bc % C);
  const long chunk = (((plane + S - 1) / S) + 3) & ~3L;
  const long lo = (long)blockIdx.x * chunk, hi = min(lo + chunk, plane);
  float* p = (y2 != nullptr && c >= C1) ? y2 + ((long)b * (C - C1) + (c - C1)) * plane
                                        : y + ((long)b * C1 + c) * plane;
  const float bv = bias ? bias[c] : 0.f;
  float sum = 0.f, mx = -INFINITY;
  if ((plane & 3) == 0) {
    for (long i = lo + 4 * (long)threadIdx.x; i < hi; i += 4 * 256) {
      float4 v = *reinterpret_cast<const float4*>(p + i);
      v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (WRITE) *reinterpret_cast<float4*>(p + i) = v;
      if (STATS) {
        sum += (v.x + v.y) + (v.z + v.w);
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      }
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      float v = p[i] + bv;
      if (RELU) v = fmaxf(v, 0.f);
      if (WRITE) p[i] = v;
      if (STATS) { sum += v; mx = fmaxf(mx, v); }
    }
  }
  if (STATS) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sum += __shfl_down(sum, off);
      mx = fmaxf(mx, __shfl_down(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const long o = ((long)b * out_C + c_off + c) * S + blockIdx.x;
      psum[o] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
      pmax[o] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
  }
}

// The pooled vector(s) through a two-layer MLP, one workgroup per sample:
//   v_mean[k] = sum_s psum[b][k][s] * inv_n,  v_max[k] = max_s pmax[b][k][s]
//   out[b][n] = act( W2 . relu(W1 . v_mean + b1) + b2  [+ W2 . relu(W1 . v_max + b1) + b2] )
// ChannelAttention: fc(avg) + fc(max) -> sigmoid (no biases); MS_CAM.global_att: mean only, BatchNorms
// folded into (W1, b1), (W2, b2), no activation.   K <= 256, M <= 64.
__global__ __launch_bounds__(256) void neck_channel_mlp_kernel(const float* __restrict__ psum,
                                                               const float* __restrict__ pmax, int K, int S,
                                                               float inv_n, const float* __restrict__ W1,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ W2,
                                                               const float* __restrict__ b2, int M, int N,
                                                               int use_max, int do_sigmoid, float* __restrict__ out) {
  __shared__ float s_in[2][256];
  __shared__ float s_h[2][64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < K) {
    float sum = 0.f, mx = -INFINITY;
    for (int s2 = 0; s2 < S; ++s2) {
      sum += psum[((long)b * K + tid) * S + s2];
      if (use_max) mx = fmaxf(mx, pmax[((long)b * K + tid) * S + s2]);
    }
    s_in[0][tid] = sum * inv_n;
    s_in[1][tid] = mx;
  }
  __syncthreads();
  const int nv = use_max ? 2 : 1;
  for (int h = tid; h < nv * M; h += 256) {
    const int which = h / M, m = h % M;
    float a = b1 ? b1[m] : 0.f;
    for (int k = 0; k < K; ++k) a = fmaf(W1[m * K + k], s_in[which][k], a);
    s_h[which][m] = fmaxf(a, 0.f);
  }
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    float a = b2 ? b2[n] : 0.f;
    for (int m = 0; m < M; ++m) a = fmaf(W2[n * M + m], s_h[0][m], a);
    if (use_max) {
      float a2 = b2 ? b2[n] : 0.f;
      for (int m = 0; m < M; ++m) a2 = fmaf(W2[n * M + m], s_h[1][m], a2);
      a += a2;
    }
    out[(long)b * N + n] = do_sigmoid ? 1.0f / (1.0f + __expf(-a)) : a;
  }
}

// SpatialAttention's input statistics of the channel-gated map: mean and max over c of scale[b][c] * x[b][c]
// (the gated map itself is never written).  One thread per pixel, 8 plane reads in flight.
__global__ __launch_bounds__(256) void neck_scaled_channel_stats_kernel(const float* __restrict__ x,
                                                                        const float* __restrict__ scale, int C,
                                                                        long plane, float* __restrict__ stats) {
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= plane) return;
  const float* p = x + (long)b * C * plane + pix;
  const float* sc = scale + (long)b * C;
  float sum = 0.f, mx = -INFINITY;
  int c = 0;
  for (; c + 8 <= C; c += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(c + u) * plane];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = sc[c + u] * v[u];
      sum += w;
      mx = fmaxf(mx, w);
    }
  }
  for (; c < C; ++c) {
    const float w = sc[c] * p[(long)c * plane];
    sum += w;
    mx = fmaxf(mx, w);
  }
  stats[((long)b * 2 + 0) * plane + pix] = sum / (float)C;
  stats[((long)b * 2 + 1) * plane + pix] = mx;
}

// ResCBAMBlock's tail + ProbNet.mask_net in one read of the block's conv output y and residual res:
//   m = sigmoid(conv_kxk(stats))                       SpatialAttention   (:68-81)
//   o_c = relu(m * (scale_c * y_c) + res_c)            ca * out, sa * out, += residual, ReLU (:128-136)
//   logit = sum_c wm_c * o_c + bm                      mask_net 1x1       (:160,172)
// 64x4 pixel tiles (whole 256-B row segments per wave), statistic planes + halo in LDS.
constexpr int kCTX = 64, kCTY = 4;
__global__ __launch_bounds__(256) void neck_cbam_tail_kernel(const float* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ conv_w, int k,
                                                             const float* __restrict__ res,
                                                             const float* __restrict__ wm, float bm, int C, int Y,
                                                             int X, float* __restrict__ logit,
                                                             float* __restrict__ block_out) {
  extern __shared__ float s_dyn[];
  const int r = k / 2, tw = kCTX + k - 1, th = kCTY + k - 1;
  float* s_w = s_dyn;
  float* s_s = s_dyn + 2 * k * k;
  const long plane = (long)Y * X;
  const int b = blockIdx.z, ty0 = blockIdx.y * kCTY, tx0 = blockIdx.x * kCTX;
  for (int i = threadIdx.x; i < 2 * k * k; i += 256) s_w[i] = conv_w[i];
  for (int i = threadIdx.x; i < 2 * tw * th; i += 256) {
    const int ch = i / (tw * th), rem = i % (tw * th);
    const int yy = ty0 + rem / tw - r, xx = tx0 + rem % tw - r;
    s_s[i] = (yy >= 0 && yy < Y && xx >= 0 && xx < X) ? stats[((long)b * 2 + ch) * plane + (long)yy * X + xx] : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x / kCTX, lx = threadIdx.x % kCTX;
  const int yy = ty0 + ly, xx = tx0 + lx;
  if (yy >= Y || xx >= X) return;
  float acc = 0.f;
  for (int ch = 0; ch < 2; ++ch)
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
        acc = fmaf(s_s[(ch * th + ly + i) * tw + lx + j], s_w[(ch * k + i) * k + j], acc);
  const float m = 1.0f / (1.0f + __expf(-acc));
  const long pix = (long)yy * X + xx;
  const float* py = y + (long)b * C * plane + pix;
  const float* pr = res + (long)b * C * plane + pix;
  const float* sc = scale + (long)b * C;
  float* po = block_out ? block_out + (long)b * C * plane + pix : nullptr;
  float lg = 0.f;
  int c = 0;
  for (; c + 8 <= C; c += 8) {
    float v[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = py[(long)(c + u) * plane]; q[u] = pr[(long)(c + u) * plane]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float o = fmaxf(m * (sc[c + u] * v[u]) + q[u], 0.f);
      if (po) po[(long)(c + u) * plane] = o;
      lg = fmaf(wm[c + u], o, lg);
    }
  }
  for (; c < C; ++c) {
    const float o = fmaxf(m * (sc[c] * py[(long)c * plane]) + pr[(long)c * plane], 0.f);
    if (po) po[(long)c * plane] = o;
    lg = fmaf(wm[c], o, lg);
  }
  logit[(long)b * plane + pix] = lg + bm;
}

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int ocrf_prefilter(const float* x, int BN, int D, int C, int HW, float depth_threshold, float semantic_threshold,
                   float* depth, float* filter_depth, float* semantic, float* feat_channels_last,
                   ocrf_stream_t stream) {
  if (BN <= 0 || D <= 0 || C <= 0 || HW <= 0) return (int)hipErrorInvalidValue;
  if (!x || !depth || !filter_depth || !semantic || !feat_channels_last) return (int)hipErrorInvalidValue;
  constexpr int kPx = 16;                     // pixels per workgroup
  const size_t lds = (size_t)kPx * (C + 1) * sizeof(float);
  if (lds > 60000) return (int)hipErrorInvalidValue;
  const dim3 grid((HW + kPx - 1) / kPx, BN), block(256);
  if (D <= 32)
    ocrf::launch(OCRF_K_NECK_PREFILTER, neck_prefilter_kernel<kPx, 2>, grid, block, lds, (hipStream_t)stream, x, D, C,
                 HW, depth_threshold, semantic_threshold, depth, filter_depth, semantic, feat_channels_last);
  else if (D <= 128)
    ocrf::launch(OCRF_K_NECK_PREFILTER, neck_prefilter_kernel<kPx, 8>, grid, block, lds, (hipStream_t)stream, x, D, C,
                 HW, depth_threshold, semantic_threshold, depth, filter_depth, semantic, feat_channels_last);
  else if (D <= 512)
    ocrf::launch(OCRF_K_NECK_PREFILTER, neck_prefilter_kernel<kPx, 32>, grid, block, lds, (hipStream_t)stream, x, D, C,
                 HW, depth_threshold, semantic_threshold, depth, filter_depth, semantic, feat_channels_last);
  else
    return (int)hipErrorInvalidValue;
  return last_error();
}

int ocrf_pillar_sample_mean(const float* imgs, const float* pix, const unsigned char* mask, float* avg, int B, int N,
                            int C, int Hi, int Wi, int ZQ, ocrf_stream_t stream) {
  if (B <= 0 || N <= 0 || Hi < 2 || Wi < 2 || ZQ <= 0 || !imgs || !pix || !mask || !avg)
    return (int)hipErrorInvalidValue;
  const dim3 grid((ZQ + 255) / 256, B), block(256);
  const float2* p2 = reinterpret_cast<const float2*>(pix);
  if (C == 1)
    ocrf::launch(OCRF_K_NECK_SAMPLE, neck_pillar_sample_mean_kernel<1>, grid, block, 0, (hipStream_t)stream, imgs, p2,
                 mask, avg, N, Hi, Wi, ZQ);
  else if (C == 3)
    ocrf::launch(OCRF_K_NECK_SAMPLE, neck_pillar_sample_mean_kernel<3>, grid, block, 0, (hipStream_t)stream, imgs, p2,
                 mask, avg, N, Hi, Wi, ZQ);
  else
    return (int)hipErrorInvalidValue;
  return last_error();
}

int ocrf_retain_valid_pixels(const float* imgs, const float* pix, const unsigned char* mask, const int* cam_sel,
                             float* out, int B, int N, int C, int H, int W, int ZQ, ocrf_stream_t stream) {
  if (B <= 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ZQ <= 0 || !imgs || !pix || !mask || !out)
    return (int)hipErrorInvalidValue;
  const size_t total = (size_t)B * (cam_sel ? 1 : N) * C * H * W;
  const size_t n4 = (reinterpret_cast<uintptr_t>(out) & 15) ? 0 : total / 4;
  const int ntail = (int)(total - 4 * n4 > 0x7fffffff ? 0 : total - 4 * n4);
  if (total - 4 * n4 > 0x7fffffff) return (int)hipErrorInvalidValue;   // misaligned giant buffer: not supported
  const size_t nthreads = n4 > (size_t)ntail ? n4 : (size_t)ntail;
  ocrf::launch(OCRF_K_NECK_RETAIN, neck_fill_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, reinterpret_cast<float4*>(out), n4, out + 4 * n4, ntail, 255.0f);
  ocrf::launch(OCRF_K_NECK_RETAIN, neck_retain_scatter_kernel, dim3((ZQ + 255) / 256, cam_sel ? 1 : N, B), dim3(256),
               0, (hipStream_t)stream, imgs, reinterpret_cast<const float2*>(pix), mask, cam_sel, out, N, C, H, W, ZQ);
  return last_error();
}

int ocrf_gauss_heads_params_len(int C, int Zh) { return 2 * Zh + 16 * C + 12 + 16 + 15 + 20 + 5 + 15; }

int ocrf_gauss_heads(const float* bev, const float* rgb_avg, const float* params, int B, int C, int Zh, int YX,
                     float* opacity, float* scales, float* rotations, float* color, ocrf_stream_t stream) {
  if (B <= 0 || C <= 0 || Zh <= 0 || YX <= 0 || !bev || !rgb_avg || !params || !opacity || !scales || !rotations ||
      !color)
    return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(rotations) & 15) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(params) + 2 * Zh * sizeof(float)) & 7) return (int)hipErrorInvalidValue;
  const dim3 block(64);
  switch (Zh) {       // heights are a compile-time constant of the register tile; the reference uses 13 (:578)
#define OCRF_HEADS_CASE(Z, G)                                                                                    \
  case Z:                                                                                                        \
    ocrf::launch(OCRF_K_NECK_HEADS, neck_gauss_heads_kernel<Z, G>, dim3((YX + 63) / 64, B * ((Z + G - 1) / G)),   \
                 block, 0, (hipStream_t)stream, bev, rgb_avg, params, C, YX, opacity, scales, rotations, color); \
    break;
    OCRF_HEADS_CASE(13, 7)
    OCRF_HEADS_CASE(8, 4)
    OCRF_HEADS_CASE(6, 3)
    OCRF_HEADS_CASE(4, 4)
    OCRF_HEADS_CASE(2, 2)
    OCRF_HEADS_CASE(1, 1)
#undef OCRF_HEADS_CASE
    default: return (int)hipErrorInvalidValue;
  }
  return last_error();
}

int ocrf_nerf_alpha(const float* z, const float* w_sigma, const float* c_sigma, float* alpha, int M, int h2, int w2,
                    ocrf_stream_t stream) {
  if (M <= 0 || h2 <= 0 || w2 <= 0 || !z || !w_sigma || !c_sigma || !alpha) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(alpha) & 15) || (reinterpret_cast<uintptr_t>(w_sigma) & 7)) return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_NECK_NERF_ALPHA, neck_nerf_alpha_kernel, dim3((w2 + 63) / 64, 8 * h2, M), dim3(64), 0,
               (hipStream_t)stream, z, w_sigma, c_sigma, alpha, h2, w2);
  return last_error();
}

int ocrf_nerf_render_params_len(void) { return 12 * 32 * 64 + 12 * 64 + 36 + 15 + 15 + 5; }

int ocrf_nerf_render(const float* z, const int* cam_sel, const float* alpha, const float* sparse_rgb,
                     const float* params, int B, int N, int h2, int w2, float* render_image_n, float* render_depth_n,
                     ocrf_stream_t stream) {
  if (B <= 0 || N <= 0 || h2 <= 0 || w2 <= 0 || !z || !cam_sel || !alpha || !sparse_rgb || !params ||
      !render_image_n || !render_depth_n)
    return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_NECK_NERF_RENDER, neck_nerf_render_kernel, dim3((8 * w2 + 255) / 256, 8 * h2, B), dim3(256), 0,
               (hipStream_t)stream, z, cam_sel, alpha, sparse_rgb, params, N, h2, w2, render_image_n, render_depth_n);
  return last_error();
}

static int dual_feat_fusion(const float* x1, const float* x2, const float* params, const float* global_vec, float* out,
                            int B, int C, int M, int YX, const float* addend, float* out_plus, ocrf_stream_t stream) {
  if (B <= 0 || YX <= 0 || !x1 || !x2 || !params || !global_vec || !out || ((addend == nullptr) != (out_plus == nullptr)))
    return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(params) & 7) return (int)hipErrorInvalidValue;
  if (C == 80 && M == 40)        // the reference's numC_Trans = 80 (configs/ocrfdet/ocrfdet.py:41)
    ocrf::launch(OCRF_K_NECK_FUSION, neck_dual_fusion_kernel<80, 40>, dim3((YX + 63) / 64, B), dim3(256), 0,
                 (hipStream_t)stream, x1, x2, params, global_vec, out, YX, addend, out_plus);
  else if (C == 64 && M == 32)   // LSSViewTransformer's default out_channels
    ocrf::launch(OCRF_K_NECK_FUSION, neck_dual_fusion_kernel<64, 32>, dim3((YX + 63) / 64, B), dim3(256), 0,
                 (hipStream_t)stream, x1, x2, params, global_vec, out, YX, addend, out_plus);
  else
    return (int)hipErrorInvalidValue;
  return last_error();
}

int ocrf_dual_feat_fusion(const float* x1, const float* x2, const float* params, const float* global_vec, float* out,
                          int B, int C, int M, int YX, ocrf_stream_t stream) {
  return dual_feat_fusion(x1, x2, params, global_vec, out, B, C, M, YX, nullptr, nullptr, stream);
}

// ... and out_plus = addend + out in the same pass (addend, out_plus (B, C, YX)): the fused map and the map ProbNet reads
int ocrf_dual_feat_fusion_plus(const float* x1, const float* x2, const float* params, const float* global_vec, float* out,
                               const float* addend, float* out_plus, int B, int C, int M, int YX, ocrf_stream_t stream) {
  if (!addend || !out_plus) return (int)hipErrorInvalidValue;
  return dual_feat_fusion(x1, x2, params, global_vec, out, B, C, M, YX, addend, out_plus, stream);
}

int ocrf_plane_bias_act_stats(float* y, const float* bias, int B, int C, int YX, int relu, int write, int S,
                              int out_C, int c_off, float* psum, float* pmax, ocrf_stream_t stream) {
  if (!y || B <= 0 || C <= 0 || YX <= 0 || S <= 0 || S > 65535 || (long)B * C > 65535) return (int)hipErrorInvalidValue;
  const bool stats = psum != nullptr;
  if (stats && (!pmax || out_C < c_off + C || c_off < 0)) return (int)hipErrorInvalidValue;
  if (!stats && !write) return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(y) & 15) return (int)hipErrorInvalidValue;
  const dim3 grid(S, (unsigned)(B * C)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const long plane = YX;
#define OCRF_PLANE_PASS(W, R, T)                                                                                  \
  ocrf::launch(OCRF_K_NECK_PLANE_PASS, neck_plane_pass_kernel<W, R, T>, grid, block, 0, st, y, bias, C, plane, S, \
               out_C, c_off, psum, pmax, static_cast<float*>(nullptr), C)
  if (write && relu && stats) OCRF_PLANE_PASS(true, true, true);
  else if (write && relu) OCRF_PLANE_PASS(true, true, false);
  else if (write && stats) OCRF_PLANE_PASS(true, false, true);
  else if (write) OCRF_PLANE_PASS(true, false, false);
  else if (relu) OCRF_PLANE_PASS(false, true, true);
  else OCRF_PLANE_PASS(false, false, true);
#undef OCRF_PLANE_PASS
  return last_error();
}

// Partial sums / maxima of every plane of TWO tensors y1 (B, C1, YX), y2 (B, C2, YX) as if they were concatenated along the
// channels: psum / pmax (B, C1 + C2, S).  One launch (MS_CAM's global branch pools cat(x1, x2): view_transformer_ocrf.py:50-58).
int ocrf_plane_stats_pair(const float* y1, const float* y2, int B, int C1, int C2, int YX, int S, float* psum, float* pmax,
                          ocrf_stream_t stream) {
  if (!y1 || !y2 || !psum || !pmax || B <= 0 || C1 <= 0 || C2 <= 0 || YX <= 0 || S <= 0 || S > 65535 ||
      (long)B * (C1 + C2) > 65535)
    return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(y1) | reinterpret_cast<uintptr_t>(y2)) & 15) return (int)hipErrorInvalidValue;
  const int C = C1 + C2;
  ocrf::launch(OCRF_K_NECK_PLANE_PASS, neck_plane_pass_kernel<false, false, true>, dim3(S, (unsigned)(B * C)), dim3(256), 0,
               (hipStream_t)stream, const_cast<float*>(y1), static_cast<const float*>(nullptr), C, (long)YX, S, C, 0, psum,
               pmax, const_cast<float*>(y2), C1);
  return last_error();
}

int ocrf_channel_mlp(const float* psum, const float* pmax, int B, int K, int S, float inv_n, const float* W1,
                     const float* b1, const float* W2, const float* b2, int M, int N, int use_max, int do_sigmoid,
                     float* out, ocrf_stream_t stream) {
  if (!psum || !W1 || !W2 || !out || B <= 0 || K <= 0 || K > 256 || M <= 0 || M > 64 || N <= 0 || S <= 0 ||
      (use_max && !pmax))
    return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_NECK_CHANNEL_MLP, neck_channel_mlp_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, psum,
               pmax, K, S, inv_n, W1, b1, W2, b2, M, N, use_max, do_sigmoid, out);
  return last_error();
}

int ocrf_scaled_channel_stats(const float* x, const float* scale, int B, int C, int YX, float* stats,
                              ocrf_stream_t stream) {
  if (!x || !scale || !stats || B <= 0 || C <= 0 || YX <= 0) return (int)hipErrorInvalidValue;
  ocrf::launch(OCRF_K_NECK_SCALED_STATS, neck_scaled_channel_stats_kernel, dim3((YX + 255) / 256, B), dim3(256), 0,
               (hipStream_t)stream, x, scale, C, (long)YX, stats);
  return last_error();
}

int ocrf_cbam_tail(const float* y, const float* scale, const float* stats, const float* conv_w, int k,
                   const float* res, const float* wm, float bm, int B, int C, int Y, int X, float* logit,
                   float* block_out, ocrf_stream_t stream) {
  if (!y || !scale || !stats || !conv_w || !res || !wm || !logit || k <= 0 || (k & 1) == 0 || k > 15 || B <= 0 ||
      C <= 0 || Y <= 0 || X <= 0)
    return (int)hipErrorInvalidValue;
  const int tw = kCTX + k - 1, th = kCTY + k - 1;
  ocrf::launch(OCRF_K_NECK_CBAM_TAIL, neck_cbam_tail_kernel, dim3((X + kCTX - 1) / kCTX, (Y + kCTY - 1) / kCTY, B),
               dim3(256), (size_t)(2 * k * k + 2 * tw * th) * sizeof(float), (hipStream_t)stream, y, scale, stats,
               conv_w, k, res, wm, bm, C, Y, X, logit, block_out);
  return last_error();
}

}  // extern "C"

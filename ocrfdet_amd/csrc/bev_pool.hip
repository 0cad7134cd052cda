// BEVPoolv2 voxel pooling for MI355X (gfx950) — forward + backward.
//
// Replaces mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu (reference: one thread per
// (interval, channel) looping the whole interval; interval lengths at the reference shape run
// from 1 to 5 216 with mean 51, so that kernel is bound by its longest thread).
//
// Design (see DESIGN.md "bev_pool_v2"):
//   * the POINT list, not the interval list, is cut into equal sub-chunks of S points; one lane
//     group (C/4 lanes, one float4 of channels per lane -> a feat row is one coalesced 16 B/lane
//     read) sums one sub-chunk, so every wave does the same work whatever the interval skew;
//   * a workgroup finds the intervals that overlap its points with a cooperative 256-ary search
//     on interval_starts and stages them, and the rank triples of its points, in LDS;
//   * an interval that lies inside one sub-chunk is summed in list order and stored directly;
//     an interval cut by a sub-chunk border leaves per-sub-chunk partial rows in a workspace and
//     a second, tiny kernel adds them in ascending sub-chunk order — no float atomics, results
//     are bitwise reproducible;
//   * the arithmetic is fp32 fmaf, like the reference's contracted `psum += f*d`.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kSub = 64;  // S: points per lane group

__device__ __forceinline__ float4 fma4(float4 f, float d, float4 a) {
  a.x = fmaf(f.x, d, a.x);
  a.y = fmaf(f.y, d, a.y);
  a.z = fmaf(f.z, d, a.z);
  a.w = fmaf(f.w, d, a.w);
  return a;
}

// ---------------------------------------------------------------------------------------------
// Compatibility kernel: one thread per (interval, channel), exactly the reference's mapping
// (bev_pool_cuda.cu:21-48).  Accepts any interval layout.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bev_pool_interval_kernel(
    int c, int n_intervals, const float* __restrict__ depth, const float* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int index = (int)(idx / c);
  const int cur_c = (int)(idx % c);
  if (index >= n_intervals) return;
  const int start = interval_starts[index];
  const int len = interval_lengths[index];
  float psum = 0.f;
  for (int i = 0; i < len; ++i) {
    const float d = depth[ranks_depth[start + i]];
    const float f = feat[(long)ranks_feat[start + i] * c + cur_c];
    psum = fmaf(f, d, psum);
  }
  out[(long)ranks_bev[start] * c + cur_c] = psum;
}

// ---------------------------------------------------------------------------------------------
// Load-balanced forward, pass 1.
//   c4   = C/4 (lanes per group), gpw = 64/c4 groups per wave, gpb = 4*gpw groups per block.
//   part = workspace rows [2*n_groups][c4] float4: row 2g = head partial, 2g+1 = tail partial.
//   meta = workspace [n_groups] int4 {head: 0 none / 1 ends here / 2 runs through,
//                                     tail: 0/1, tail_row: ranks_bev of the tail interval, 0}.
// Dynamic LDS: s_rd[BP] s_rf[BP] s_st[BP+256] s_ln[BP+256], BP = gpb*kSub.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bev_pool_fwd_chunked_kernel(
    int c4, int gpw, int n_intervals, int n_points, int n_groups, const float* __restrict__ depth,
    const float4* __restrict__ feat4, const int* __restrict__ ranks_depth,
    const int* __restrict__ ranks_feat, const int* __restrict__ ranks_bev,
    const int* __restrict__ interval_starts, const int* __restrict__ interval_lengths,
    float4* __restrict__ out4, float4* __restrict__ part, int4* __restrict__ meta) {
  extern __shared__ int smem[];
  const int tid = threadIdx.x;
  const int gpb = gpw * (kBlock / kWave);
  const int BP = gpb * kSub;
  int* s_rd = smem;
  int* s_rf = s_rd + BP;
  int* s_st = s_rf + BP;
  int* s_ln = s_st + BP + kBlock;

  const int bs = blockIdx.x * BP;             // first point of this block
  const int be = min(bs + BP, n_points);      // one past its last point

  // (1) stage the rank pairs of the block's points (coalesced).
  for (int i = tid; i < be - bs; i += kBlock) {
    s_rd[i] = ranks_depth[bs + i];
    s_rf[i] = ranks_feat[bs + i];
  }

  // (2) cooperative 256-ary search: k_lo = last interval with start <= bs (0 if none).
  int lo = 0, hi = n_intervals;
  while (hi - lo > 1) {
    const int span = hi - lo;
    const int stride = (span + kBlock - 1) / kBlock;
    const int idx = lo + tid * stride;
    const bool ok = (idx < hi) && (interval_starts[idx] <= bs);
    const int cnt = __syncthreads_count(ok);
    if (cnt == 0) { hi = lo + 1; break; }
    const int nlo = lo + (cnt - 1) * stride;
    hi = min(nlo + stride, hi);
    lo = nlo;
  }
  const int k_lo = lo;

  // (3) stage the intervals that can overlap [bs, be): at most BP+1 of them (lengths >= 1).
  const int ni_max = min(BP + 1, n_intervals - k_lo);
  int n_loaded = 0;
  for (int base = 0; base < ni_max; base += kBlock) {
    const int t = base + tid;
    if (t < ni_max) {
      s_st[t] = interval_starts[k_lo + t];
      s_ln[t] = interval_lengths[k_lo + t];
    }
    __syncthreads();
    n_loaded = min(base + kBlock, ni_max);
    if (s_st[n_loaded - 1] >= be) break;   // uniform: everything after starts past the block
  }
  __syncthreads();

  // (4) one lane group per sub-chunk.
  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  if (gi >= gpw) return;
  const int gb = wave * gpw + gi;
  const int gid = blockIdx.x * gpb + gb;
  if (gid >= n_groups) return;
  const int s = gid * kSub;
  const int e = min(s + kSub, n_points);

  // last staged interval with start <= s (or 0)
  int j0 = 0;
  {
    int a = 0, b = n_loaded;   // invariant: answer in [a, b)
    while (b - a > 1) {
      const int m = (a + b) >> 1;
      if (s_st[m] <= s) a = m; else b = m;
    }
    j0 = a;
  }

  int head = 0, tail = 0, tail_row = 0;
  for (int j = j0; j < n_loaded; ++j) {
    const int is = s_st[j];
    if (is >= e) break;
    const int ie = is + s_ln[j];
    const int a = max(is, s), b = min(ie, e);
    if (a >= b) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = a - bs;
    const int pe = b - bs;
    for (; p + 4 <= pe; p += 4) {
      const int d0 = s_rd[p], d1 = s_rd[p + 1], d2 = s_rd[p + 2], d3 = s_rd[p + 3];
      const int f0 = s_rf[p], f1 = s_rf[p + 1], f2 = s_rf[p + 2], f3 = s_rf[p + 3];
      const float w0 = depth[d0], w1 = depth[d1], w2 = depth[d2], w3 = depth[d3];
      const float4 v0 = feat4[(long)f0 * c4 + lg];
      const float4 v1 = feat4[(long)f1 * c4 + lg];
      const float4 v2 = feat4[(long)f2 * c4 + lg];
      const float4 v3 = feat4[(long)f3 * c4 + lg];
      acc = fma4(v0, w0, acc);
      acc = fma4(v1, w1, acc);
      acc = fma4(v2, w2, acc);
      acc = fma4(v3, w3, acc);
    }
    for (; p < pe; ++p) {
      const float w = depth[s_rd[p]];
      const float4 v = feat4[(long)s_rf[p] * c4 + lg];
      acc = fma4(v, w, acc);
    }
    const bool before = is < s, after = ie > e;
    if (!before && !after) {
      out4[(long)ranks_bev[is] * c4 + lg] = acc;
    } else if (before) {
      part[(long)(2 * gid) * c4 + lg] = acc;
      head = after ? 2 : 1;
    } else {
      part[(long)(2 * gid + 1) * c4 + lg] = acc;
      tail = 1;
      tail_row = ranks_bev[is];
    }
  }
  if (lg == 0) meta[gid] = make_int4(head, tail, tail_row, 0);
}

// Pass 2: the sub-chunk in which a cut interval starts owns it: tail + head + head + ...
__global__ __launch_bounds__(kBlock) void bev_pool_fwd_fixup_kernel(
    int c4, int gpw, int n_groups, float4* __restrict__ out4, const float4* __restrict__ part,
    const int4* __restrict__ meta) {
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  if (gi >= gpw) return;
  const int gid = (blockIdx.x * (kBlock / kWave) + wave) * gpw + gi;
  if (gid >= n_groups) return;
  const int4 m = meta[gid];
  if (!m.y) return;
  float4 acc = part[(long)(2 * gid + 1) * c4 + lg];
  for (int j = gid + 1; j < n_groups; ++j) {
    const int h = meta[j].x;
    if (h == 0) break;
    const float4 v = part[(long)(2 * j) * c4 + lg];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    if (h == 1) break;
  }
  out4[(long)m.z * c4 + lg] = acc;
}

__global__ __launch_bounds__(kBlock) void bev_pool_check_intervals_kernel(
    int n_intervals, int n_points, const int* __restrict__ starts, const int* __restrict__ lengths,
    int* __restrict__ flag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_intervals) return;
  const int s = starts[k], l = lengths[k];
  int bad = 0;
  if (s < 0 || l <= 0) bad |= 4;
  if ((long)s + l > n_points) bad |= 2;
  if (k + 1 < n_intervals && (long)s + l > starts[k + 1]) bad |= 1;
  if (bad) atomicOr(flag, bad);
}

// ---------------------------------------------------------------------------------------------
// Backward (bev_pool_cuda.cu:67-121): one lane group per ranks_feat-run.
//   depth_grad[rd[p]]       = sum_c out_grad[rb[p],c] * feat[rf[p],c]         (plain store)
//   feat_grad[rf[start], c] = sum_p out_grad[rb[p],c] * depth[rd[p]]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bev_pool_grad_vec_kernel(
    int c4, int gpw, int n_intervals, const float4* __restrict__ out_grad4,
    const float* __restrict__ depth, const float4* __restrict__ feat4,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, float* __restrict__ depth_grad,
    float4* __restrict__ feat_grad4) {
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  const int k = (blockIdx.x * (kBlock / kWave) + wave) * gpw + gi;
  const bool active = (gi < gpw) && (k < n_intervals);
  const int start = active ? interval_starts[k] : 0;
  const int len = active ? interval_lengths[k] : 0;
  // all lanes of a wave run the same trip count so the shuffles below stay convergent
  int maxlen = len;
  for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
  float4 facc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < maxlen; ++i) {
    const bool on = i < len;
    float dot = 0.f;
    int rdp = 0;
    if (on) {
      const int p = start + i;
      rdp = ranks_depth[p];
      const float4 og = out_grad4[(long)ranks_bev[p] * c4 + lg];
      const float4 f = feat4[(long)ranks_feat[p] * c4 + lg];
      const float d = depth[rdp];
      facc = fma4(og, d, facc);
      dot = fmaf(og.w, f.w, fmaf(og.z, f.z, fmaf(og.y, f.y, og.x * f.x)));
    }
    // fixed-shape tree over the c4 lanes of the group (deterministic)
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(dot, off);
      if (lg + off < c4) dot += o;
    }
    if (on && lg == 0) depth_grad[rdp] = dot;
  }
  if (active && len > 0) feat_grad4[(long)ranks_feat[start] * c4 + lg] = facc;
}

// scalar fallback for C not a multiple of 4 (or > 256): the reference's mapping.
__global__ __launch_bounds__(kBlock) void bev_pool_grad_scalar_kernel(
    int c, int n_intervals, const float* __restrict__ out_grad, const float* __restrict__ depth,
    const float* __restrict__ feat, const int* __restrict__ ranks_depth,
    const int* __restrict__ ranks_feat, const int* __restrict__ ranks_bev,
    const int* __restrict__ interval_starts, const int* __restrict__ interval_lengths,
    float* __restrict__ depth_grad, float* __restrict__ feat_grad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_intervals) return;
  const int start = interval_starts[idx], len = interval_lengths[idx];
  for (int i = 0; i < len; ++i) {
    const float* og = out_grad + (long)ranks_bev[start + i] * c;
    const float* f = feat + (long)ranks_feat[start + i] * c;
    float g = 0.f;
    for (int cc = 0; cc < c; ++cc) g = fmaf(og[cc], f[cc], g);
    depth_grad[ranks_depth[start + i]] = g;
  }
  for (int cc = 0; cc < c; ++cc) {
    float g = 0.f;
    for (int i = 0; i < len; ++i)
      g = fmaf(out_grad[(long)ranks_bev[start + i] * c + cc], depth[ranks_depth[start + i]], g);
    feat_grad[(long)ranks_feat[start] * c + cc] = g;
  }
}

inline bool vec_ok(int c) { return c >= 32 && c <= 256 && (c % 4) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int n_groups_for(int n_points) { return (n_points + kSub - 1) / kSub; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

void bev_pool_v2(int c, int n_intervals, const float* depth, const float* feat,
                 const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                 const int* interval_starts, const int* interval_lengths, float* out) {
  if (c <= 0 || n_intervals <= 0) return;
  const long total = (long)n_intervals * c;
  const unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(bev_pool_interval_kernel, dim3(grid), dim3(kBlock), 0, nullptr, c,
                     n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                     interval_lengths, out);
}

size_t ocrf_bev_pool_v2_workspace_bytes(int c, int n_points) {
  if (!vec_ok(c) || n_points <= 0) return 0;
  const size_t ng = (size_t)n_groups_for(n_points);
  return align_up(ng * 2 * (size_t)c * sizeof(float), 256) + ng * sizeof(int4);
}

int ocrf_bev_pool_v2(int c, int n_intervals, int n_points, const float* depth, const float* feat,
                     const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                     const int* interval_starts, const int* interval_lengths, float* out,
                     void* workspace, size_t workspace_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (c <= 0 || n_intervals < 0 || n_points < 0) return (int)hipErrorInvalidValue;
  if (n_intervals == 0 || n_points == 0) return 0;
  if (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_starts ||
      !interval_lengths || !out)
    return (int)hipErrorInvalidValue;
  if (!vec_ok(c) || !aligned16(feat) || !aligned16(out)) {
    // scalar mapping of the reference; correct for every C and alignment
    const long total = (long)n_intervals * c;
    const unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
    ocrf::launch(OCRF_K_BEV_POOL_INTERVAL, bev_pool_interval_kernel, dim3(grid), dim3(kBlock), 0,
                 stream, c, n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                 interval_starts, interval_lengths, out);
    return (int)hipGetLastError();
  }
  const size_t need = ocrf_bev_pool_v2_workspace_bytes(c, n_points);
  if (!workspace || workspace_bytes < need || !aligned16(workspace))
    return (int)hipErrorInvalidValue;
  const int c4 = c / 4;
  const int gpw = kWave / c4;
  const int gpb = gpw * (kBlock / kWave);
  const int BP = gpb * kSub;
  const int ng = n_groups_for(n_points);
  float4* part = static_cast<float4*>(workspace);
  int4* meta = reinterpret_cast<int4*>(static_cast<char*>(workspace) +
                                       align_up((size_t)ng * 2 * c * sizeof(float), 256));
  const unsigned grid1 = (unsigned)((n_points + BP - 1) / BP);
  const size_t lds = (size_t)(4 * BP + 2 * kBlock) * sizeof(int);
  ocrf::launch(OCRF_K_BEV_POOL_FWD, bev_pool_fwd_chunked_kernel, dim3(grid1), dim3(kBlock), lds,
               stream, c4, gpw, n_intervals, n_points, ng, depth,
               reinterpret_cast<const float4*>(feat), ranks_depth, ranks_feat, ranks_bev,
               interval_starts, interval_lengths, reinterpret_cast<float4*>(out), part, meta);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  const unsigned grid2 = (unsigned)((ng + gpb - 1) / gpb);
  ocrf::launch(OCRF_K_BEV_POOL_FIXUP, bev_pool_fwd_fixup_kernel, dim3(grid2), dim3(kBlock), 0, stream,
               c4, gpw, ng, reinterpret_cast<float4*>(out), static_cast<const float4*>(part),
               static_cast<const int4*>(meta));
  return (int)hipGetLastError();
}

int ocrf_bev_pool_v2_check_intervals(int n_intervals, int n_points, const int* interval_starts,
                                     const int* interval_lengths, int* flag,
                                     ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_intervals < 0 || n_points < 0 || !flag) return (int)hipErrorInvalidValue;
  hipError_t err = hipMemsetAsync(flag, 0, sizeof(int), stream);
  if (err != hipSuccess || n_intervals == 0) return (int)err;
  if (!interval_starts || !interval_lengths) return (int)hipErrorInvalidValue;
  const unsigned grid = (unsigned)((n_intervals + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(bev_pool_check_intervals_kernel, dim3(grid), dim3(kBlock), 0, stream,
                     n_intervals, n_points, interval_starts, interval_lengths, flag);
  return (int)hipGetLastError();
}

int ocrf_bev_pool_v2_grad(int c, int n_intervals, const float* out_grad, const float* depth,
                          const float* feat, const int* ranks_depth, const int* ranks_feat,
                          const int* ranks_bev, const int* interval_starts,
                          const int* interval_lengths, float* depth_grad, float* feat_grad,
                          ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (c <= 0 || n_intervals < 0) return (int)hipErrorInvalidValue;
  if (n_intervals == 0) return 0;
  if (!out_grad || !depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
      !interval_starts || !interval_lengths || !depth_grad || !feat_grad)
    return (int)hipErrorInvalidValue;
  if (vec_ok(c) && aligned16(out_grad) && aligned16(feat) && aligned16(feat_grad)) {
    const int c4 = c / 4, gpw = kWave / c4, gpb = gpw * (kBlock / kWave);
    const unsigned grid = (unsigned)((n_intervals + gpb - 1) / gpb);
    hipLaunchKernelGGL(bev_pool_grad_vec_kernel, dim3(grid), dim3(kBlock), 0, stream, c4, gpw,
                       n_intervals, reinterpret_cast<const float4*>(out_grad), depth,
                       reinterpret_cast<const float4*>(feat), ranks_depth, ranks_feat, ranks_bev,
                       interval_starts, interval_lengths, depth_grad,
                       reinterpret_cast<float4*>(feat_grad));
  } else {
    const unsigned grid = (unsigned)((n_intervals + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(bev_pool_grad_scalar_kernel, dim3(grid), dim3(kBlock), 0, stream, c,
                       n_intervals, out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                       interval_starts, interval_lengths, depth_grad, feat_grad);
  }
  return (int)hipGetLastError();
}

void bev_pool_v2_grad(int c, int n_intervals, const float* out_grad, const float* depth,
                      const float* feat, const int* ranks_depth, const int* ranks_feat,
                      const int* ranks_bev, const int* interval_starts,
                      const int* interval_lengths, float* depth_grad, float* feat_grad) {
  (void)ocrf_bev_pool_v2_grad(c, n_intervals, out_grad, depth, feat, ranks_depth, ranks_feat,
                              ranks_bev, interval_starts, interval_lengths, depth_grad, feat_grad,
                              nullptr);
}

}  // extern "C"

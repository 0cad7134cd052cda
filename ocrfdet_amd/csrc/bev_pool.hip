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
constexpr int kSubDefault = 32;  // S: points per lane group (tunable: ocrf_tune_set(0, 32|64))
int g_sub = kSubDefault;

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
//   c4  = C/4 lanes per group, gpw = 64/c4 groups per wave, gpb = 4*gpw groups per block,
//   BP  = gpb*kSub points per block.
//   dst = row-addressed float4 output: row r, lane lg -> dst[r*c4 + lg].  The row of interval j
//         is ranks_bev[start_j] (scatter into the (B,Z,Y,X,C) tensor, the reference's contract)
//         or, with compact_rows, the interval's own index (dense [n_intervals][C] table that the
//         NCHW epilogue consumes).
//   part = workspace rows [2*n_blocks][c4] float4: 2b = block head partial, 2b+1 = tail partial.
//   bmeta = workspace [n_blocks] int4 {head: 0 none / 1 ends in this block / 2 runs through,
//                                      tail: 0/1, tail_row, 0}.
// Phases: (1) stage per-point {depth weight, code} and the feat row index in LDS; (2) 256-ary
// cooperative search for the first interval of the block; (3) stage the block's intervals;
// (3b) one thread per interval marks the LAST point of each of its pieces (a piece = interval
// /\ sub-chunk) with an emit code, zeroes the weight of points outside every interval and fills
// the per-sub-chunk head/tail table; (4) each lane group walks its kSub points branch-free —
// acc = fma(feat_row, w, acc); on a marked point it emits acc and restarts — with the row gathers
// of the next 8 points in flight while 8 are consumed; (5) pieces of intervals cut by sub-chunk
// borders are combined through LDS in ascending order, pieces cut by BLOCK borders go to `part`.
// Dynamic LDS: int2 s_wc[BP] {w bits, code}; int s_rf[BP] s_st[BP+256] s_ln[BP+256]
//              s_row[BP+256] s_meta[4*gpb]; float4 s_part[2*gpb][c4].
// code: 1 | (local interval index << 2) = last point of a whole interval -> store to its dst
// row; 2 = last point of a head piece -> keep in a register; tail pieces are the leftover.
// ---------------------------------------------------------------------------------------------
// Row gathers in flight per lane: 2 x kBatch (double-buffered).  8 + 8 cost 202 VGPRs = 2 waves per SIMD =
// 2 workgroups per CU, and the latency-type phases around the loop (staging, interval search, combine:
// 57 % of a workgroup's cycles) found nothing to overlap with; 4 + 4 fit 4 waves per SIMD.
constexpr int kBatch = 4;
struct Batch {
  float4 v[kBatch];
  int2 wc[kBatch];
};

__device__ __forceinline__ void load_batch(Batch& q, int l0, const int* s_rf, const int2* s_wc,
                                           const float4* __restrict__ feat4, int c4, int lg) {
#pragma unroll
  for (int k = 0; k < kBatch; ++k) q.v[k] = feat4[(long)s_rf[l0 + k] * c4 + lg];
#pragma unroll
  for (int k = 0; k < kBatch; ++k) q.wc[k] = s_wc[l0 + k];
}

// MODE 0: everything from the rank vectors, every call.
// MODE 1: BUILD a plan — run the rank-only phases (2)-(3c) once and keep their result: per point an
//         emit code (bit 31 = the point lies in no interval: weight 0), per sub-chunk the piece table;
//         nothing is pooled.
// MODE 2: PLANNED — (1') stage depth weights + rf + the plan's codes and piece table, then (4), (5).
//         For rank vectors that are cached across calls (``accelerate``): the interval search, interval
//         staging and marking phases (35 % of a workgroup's cycles) are gone and the workgroup needs
//         half the LDS.
template <bool STAMP, int kSub, int MODE>
__global__ __launch_bounds__(kBlock) void bev_pool_fwd_chunked_kernel(
    unsigned long long* __restrict__ stamps,
    int c4, int gpw, int n_intervals, int n_points, const int* __restrict__ counts, int compact_rows,
    const float* __restrict__ depth, const float4* __restrict__ feat4,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, float4* __restrict__ dst, float4* __restrict__ part,
    int4* __restrict__ bmeta, int* __restrict__ row_of_vox, int* __restrict__ plan_codes,
    int* __restrict__ plan_meta) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  const int tid = threadIdx.x;
  const int gpb = gpw * (kBlock / kWave);
  const int BP = gpb * kSub;
  int2* s_wc = reinterpret_cast<int2*>(smem);              // {float bits of depth weight, code}
  int* s_rf = smem + 2 * BP;
  int* s_st = s_rf + BP;                                   // MODE 2 keeps none of s_st / s_ln / s_row
  int* s_ln = s_st + BP + kBlock;
  int* s_row = s_ln + BP + kBlock;
  int* s_meta = (MODE == 2) ? (s_rf + BP) : (s_row + BP + kBlock);   // [gpb][4]: head, tail, tail row, tail runs past the block
  float4* s_part = reinterpret_cast<float4*>(s_meta + 4 * gpb);   // [2*gpb][c4]

  const int bs = blockIdx.x * BP;             // first point of this block
  if (counts) {
    // sizes produced on the device (ocrf_lss_prepare / ocrf_ht_prepare): the launch was sized for
    // the capacities, workgroups past the real end leave a neutral record for the fix-up and go
    n_points = counts[0];
    n_intervals = counts[1];
    if (bs >= n_points) {
      if (tid == 0) bmeta[blockIdx.x] = make_int4(0, 0, 0, 0);
      return;
    }
  }
  const int be = min(bs + BP, n_points);      // one past its last point
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      if (tid == 0) stamps[(long)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp(0);

  // (1) stage the block's points (coalesced index reads, one depth gather per point).
  for (int i = tid; i < BP; i += kBlock) {
    float w = 0.f;
    int rf = 0, code = 0;
    if (bs + i < be) {
      if constexpr (MODE == 1) {
        w = 1.f;                                   // only "zeroed or not" matters when building
      } else {
        w = depth[ranks_depth[bs + i]];
        rf = ranks_feat[bs + i];
      }
      if constexpr (MODE == 2) {
        code = plan_codes[bs + i];
        if (code < 0) { w = 0.f; code &= 0x7fffffff; }
      }
    }
    s_wc[i] = make_int2(__float_as_int(w), code);
    s_rf[i] = rf;
  }
  if constexpr (MODE == 2) {
    if (tid < 4 * gpb) s_meta[tid] = plan_meta[(long)blockIdx.x * 4 * gpb + tid];
  } else {
    if (tid < 4 * gpb) s_meta[tid] = 0;
  }
  if constexpr (MODE != 2) {

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
  stamp(1);

  // (3) stage the intervals that can overlap [bs, be): at most BP+1 of them (lengths >= 1).
  const int ni_max = min(BP + 1, n_intervals - k_lo);
  int n_loaded = 0;
  for (int base = 0; base < ni_max; base += kBlock) {
    const int t = base + tid;
    if (t < ni_max) {
      const int st = interval_starts[k_lo + t];
      s_st[t] = st;
      s_ln[t] = interval_lengths[k_lo + t];
      const int vox = ranks_bev[st];
      s_row[t] = compact_rows ? (k_lo + t) : vox;
      // the block in which an interval starts publishes voxel -> row (+1; 0 = empty voxel)
      if (compact_rows && st >= bs && st < be) row_of_vox[vox] = k_lo + t + 1;
    }
    __syncthreads();
    n_loaded = min(base + kBlock, ni_max);
    if (s_st[n_loaded - 1] >= be) break;   // uniform: everything after starts past the block
  }
  __syncthreads();
  stamp(2);

  // (3b) mark piece ends, zero the weights of uncovered points, fill the sub-chunk table.
  const bool exhausted = (k_lo + n_loaded >= n_intervals);
  for (int t = tid; t < n_loaded; t += kBlock) {
    const int is = s_st[t];
    const int ie = is + s_ln[t];
    if (t == 0) {
      for (int q = bs; q < min(is, be); ++q) s_wc[q - bs].x = 0;          // before the first interval
    }
    const int nis = (t + 1 < n_loaded) ? s_st[t + 1] : (exhausted ? 0x7fffffff : be);
    for (int q = max(ie, bs); q < min(nis, be); ++q) s_wc[q - bs].x = 0;   // gap after this one
    const int plo = max(is, bs), phi = min(ie, be);
    if (plo >= phi) continue;
    const int g0 = (plo - bs) / kSub, g1 = (phi - 1 - bs) / kSub;
    for (int g = g0; g <= g1; ++g) {
      const int sg = bs + g * kSub;
      const int eg = min(sg + kSub, be);
      const int pb = min(ie, eg);                  // one past the piece's last point
      const bool before = is < sg, after = ie > eg;
      // whole interval -> 1, head piece -> 2; a tail piece is whatever is left in the
      // accumulator when the sub-chunk ends, so it needs no mark
      if (!before && !after) s_wc[pb - 1 - bs].y = 1 | (t << 2);
      else if (before) s_wc[pb - 1 - bs].y = 2;
      if (before) {
        s_meta[4 * g + 0] = after ? 2 : 1;
      } else if (after) {
        s_meta[4 * g + 1] = 1;
        s_meta[4 * g + 2] = t;
      }
    }
  }
  __syncthreads();
  // (3c) fold the rows into the codes and the piece table, so that (4) and (5) need neither s_row nor
  // s_st / s_ln (and a plan can stand in for all of (2)-(3c))
  for (int i = tid; i < BP; i += kBlock) {
    const int code = s_wc[i].y;
    if (code & 1) s_wc[i].y = 1 | (s_row[code >> 2] << 2);
  }
  for (int g = tid; g < gpb; g += kBlock) {
    if (s_meta[4 * g + 1]) {
      const int t = s_meta[4 * g + 2];
      s_meta[4 * g + 2] = s_row[t];
      s_meta[4 * g + 3] = (s_st[t] + s_ln[t] > be) ? 1 : 0;
    }
  }
  __syncthreads();
  if constexpr (MODE == 1) {
    for (int i = tid; i < BP; i += kBlock)
      if (bs + i < be) plan_codes[bs + i] = s_wc[i].y | (s_wc[i].x == 0 ? (int)0x80000000 : 0);
    if (tid < 4 * gpb) plan_meta[(long)blockIdx.x * 4 * gpb + tid] = s_meta[tid];
    return;
  }
  }   // MODE != 2
  else {
    __syncthreads();
  }

  // (4) one lane group per sub-chunk, branch-free accumulate with marked emits.
  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  const int gb = wave * gpw + gi;
  const int s = bs + gb * kSub;
  const bool active = (gi < gpw) && (s < be);

  if (active) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 head_acc = acc;
    auto consume = [&](const Batch& q, int l0) {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int2 wc = q.wc[k];
        acc = fma4(q.v[k], __int_as_float(wc.x), acc);
        const int code = wc.y;
        // the only memory operation in the loop is this global store: an LDS store here would
        // make hipcc merge both into one flat store, whose wait drains the gathers in flight
        if (code & 1) dst[(long)(code >> 2) * c4 + lg] = acc;
        const bool is_head = code == 2;
        head_acc.x = is_head ? acc.x : head_acc.x;
        head_acc.y = is_head ? acc.y : head_acc.y;
        head_acc.z = is_head ? acc.z : head_acc.z;
        head_acc.w = is_head ? acc.w : head_acc.w;
        acc.x = code ? 0.f : acc.x;
        acc.y = code ? 0.f : acc.y;
        acc.z = code ? 0.f : acc.z;
        acc.w = code ? 0.f : acc.w;
      }
    };
    Batch A, B;
    const int l0 = gb * kSub;
    load_batch(A, l0, s_rf, s_wc, feat4, c4, lg);
#pragma nounroll      // unrolled, the scheduler hoists every gather to the top: 202 VGPRs, 2 waves per SIMD
    for (int b = 0; b < kSub / kBatch; b += 2) {
      load_batch(B, l0 + kBatch * (b + 1), s_rf, s_wc, feat4, c4, lg);
      consume(A, l0 + kBatch * b);
      if (b + 2 < kSub / kBatch) load_batch(A, l0 + kBatch * (b + 2), s_rf, s_wc, feat4, c4, lg);
      consume(B, l0 + kBatch * (b + 1));
    }
    s_part[(2 * gb) * c4 + lg] = head_acc;       // head piece (meaningful iff s_meta says so)
    s_part[(2 * gb + 1) * c4 + lg] = acc;        // tail piece: what the last interval left
  }
  stamp(3);
  __syncthreads();
  stamp(4);

  // (5) combine the pieces of intervals cut by sub-chunk borders, in ascending sub-chunk order.
  if (!active) return;
  const int n_gb = (be - bs + kSub - 1) / kSub;    // sub-chunks in use in this block
  const int head = s_meta[4 * gb + 0], tail = s_meta[4 * gb + 1], tail_j = s_meta[4 * gb + 2];
  if (tail) {
    float4 acc = s_part[(2 * gb + 1) * c4 + lg];
    for (int g2 = gb + 1; g2 < n_gb; ++g2) {
      const int h = s_meta[4 * g2];
      if (h == 0) break;
      const float4 v = s_part[(2 * g2) * c4 + lg];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (h == 1) break;
    }
    const int row = tail_j;                          // (3c): the tail interval's row
    if (s_meta[4 * gb + 3]) {                        // runs past the block: leave a block tail
      part[(long)(2 * blockIdx.x + 1) * c4 + lg] = acc;
      if (lg == 0) { bmeta[blockIdx.x].y = 1; bmeta[blockIdx.x].z = row; }
    } else {
      dst[(long)row * c4 + lg] = acc;
    }
  }
  if (gb == 0) {
    int btype = 0;
    if (head) {
      float4 acc = s_part[lg];
      btype = head;
      if (head == 2) {
        for (int g2 = 1; g2 < n_gb; ++g2) {
          const int h = s_meta[4 * g2];
          if (h == 0) { btype = 1; break; }
          const float4 v = s_part[(2 * g2) * c4 + lg];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          if (h == 1) { btype = 1; break; }
        }
      }
      part[(long)(2 * blockIdx.x) * c4 + lg] = acc;
    }
    if (lg == 0) {
      bmeta[blockIdx.x].x = btype;
      // exactly one writer of the tail flag: the owner of an interval running past `be` (above)
      // or, when there is none, this lane
      bool open = false;
      for (int g2 = 0; g2 < n_gb; ++g2) {
        open = open || (s_meta[4 * g2 + 1] && s_meta[4 * g2 + 3]);
      }
      if (!open) bmeta[blockIdx.x].y = 0;
    }
  }
  stamp(5);
}

// Pass 2: the block in which an interval cut by block borders starts owns it.
__global__ __launch_bounds__(kBlock) void bev_pool_fwd_fixup_kernel(
    int c4, int gpw, int n_blocks, float4* __restrict__ dst, const float4* __restrict__ part,
    const int4* __restrict__ bmeta) {
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  if (gi >= gpw) return;
  const int b = (blockIdx.x * (kBlock / kWave) + wave) * gpw + gi;
  if (b >= n_blocks) return;
  const int4 m = bmeta[b];
  if (!m.y) return;
  float4 acc = part[(long)(2 * b + 1) * c4 + lg];
  for (int j = b + 1; j < n_blocks; ++j) {
    const int h = bmeta[j].x;
    if (h == 0) break;
    const float4 v = part[(long)(2 * j) * c4 + lg];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    if (h == 1) break;
  }
  dst[(long)m.z * c4 + lg] = acc;
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the fused forward: dense channel-major output from the compact row table.
// One workgroup per tile of kTile consecutive x of one (b, z, y) row: the rows of the non-empty
// voxels go through LDS ([kTile][C+1] floats) and every channel leaves as one contiguous run of
// kTile floats, zeros included — the output is written exactly once, never pre-zeroed, and the
// reference's permute(0,4,1,2,3).contiguous() (bev_pool.py:91) and cat(unbind(2),1)
// (view_transformer.py:194) passes disappear.
//   layout 0: out[((b*C + c)*Z + z)*Y*X + y*X + x]      (B,C,Z,Y,X)
//   layout 1: out[((b*Z + z)*C + c)*Y*X + y*X + x]      (B,Z*C,Y,X)
// ---------------------------------------------------------------------------------------------
constexpr int kTile = 64;

__global__ __launch_bounds__(kBlock) void bev_pool_rows_to_nchw_kernel(
    int c4, int B, int Z, int Y, int X, int tiles_x, int layout, const float4* __restrict__ rows,
    const int* __restrict__ row_of_vox, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_tile[];
  const int C = 4 * c4;
  const int ld = C + 1;
  int* s_rowid = reinterpret_cast<int*>(s_tile + kTile * ld);
  const int tid = threadIdx.x;
  long t = blockIdx.x;
  const int tx = (int)(t % tiles_x); t /= tiles_x;
  const int y = (int)(t % Y); t /= Y;
  const int z = (int)(t % Z);
  const int b = (int)(t / Z);
  const int x0 = tx * kTile;
  const int nx = min(kTile, X - x0);
  const long vox0 = (((long)b * Z + z) * Y + y) * X + x0;
  int any = 0;
  if (tid < kTile) {
    const int r = (tid < nx) ? row_of_vox[vox0 + tid] : 0;
    s_rowid[tid] = r;
    any = r != 0;
  }
  const int n_any = __syncthreads_count(any);
  const long plane = (long)Y * X;
  const long base0 = (layout == 0) ? ((long)b * C * Z + z) * plane : (((long)b * Z + z) * C) * plane;
  const long cstride = (layout == 0) ? (long)Z * plane : plane;
  const long rowoff = (long)y * X + x0;
  if (n_any == 0) {
    for (int i = tid; i < C * kTile; i += kBlock) {
      const int ch = i / kTile, xx = i % kTile;
      if (xx < nx) out[base0 + ch * cstride + rowoff + xx] = 0.f;
    }
    return;
  }
  // gather rows (c4 lanes per row, float4 each) into the padded tile
  for (int i = tid; i < kTile * c4; i += kBlock) {
    const int v = i / c4, q = i % c4;
    const int r = s_rowid[v];
    const float4 val = r ? rows[(long)(r - 1) * c4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = s_tile + v * ld + 4 * q;
    d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
  }
  __syncthreads();
  for (int i = tid; i < C * kTile; i += kBlock) {
    const int ch = i / kTile, xx = i % kTile;
    if (xx < nx) out[base0 + ch * cstride + rowoff + xx] = s_tile[xx * ld + ch];
  }
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
  if (len <= 0) return;                       // empty runs (dense per-pixel run tables) touch nothing
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

namespace {
struct FwdGeom {
  int c4, gpw, gpb, BP, n_blocks;
  size_t lds, part_bytes, total_bytes;
};
inline FwdGeom fwd_geom(int c, int n_points) {
  FwdGeom g;
  const int kSub = g_sub;
  g.c4 = c / 4;
  g.gpw = kWave / g.c4;
  g.gpb = g.gpw * (kBlock / kWave);
  g.BP = g.gpb * kSub;
  g.n_blocks = (n_points + g.BP - 1) / g.BP;
  g.lds = (size_t)(6 * g.BP + 3 * kBlock + 4 * g.gpb) * sizeof(int) + (size_t)2 * g.gpb * g.c4 * sizeof(float4);
  g.part_bytes = align_up((size_t)g.n_blocks * 2 * c * sizeof(float), 256);
  g.total_bytes = g.part_bytes + (size_t)g.n_blocks * sizeof(int4);
  return g;
}

// Shared by the scatter (reference contract) and compact-row (NCHW epilogue) forms.
// mode 0: from the ranks; 1: build `plan_codes` / `plan_meta` (+ row_of_vox) only; 2: pooled from the plan
int launch_fwd(int c, int n_intervals, int n_points, const int* counts, int compact_rows, const float* depth,
               const float* feat, const int* ranks_depth, const int* ranks_feat,
               const int* ranks_bev, const int* interval_starts, const int* interval_lengths,
               float* dst, void* workspace, int* row_of_vox, hipStream_t stream, int mode = 0,
               int* plan_codes = nullptr, int* plan_meta = nullptr) {
  const FwdGeom g = fwd_geom(c, n_points);
  float4* part = static_cast<float4*>(workspace);
  int4* bmeta = workspace ? reinterpret_cast<int4*>(static_cast<char*>(workspace) + g.part_bytes) : nullptr;
  using K = decltype(&bev_pool_fwd_chunked_kernel<false, 32, 0>);
  K kern;
  size_t lds = g.lds;
  if (mode == 1) {
    kern = (g_sub == 64) ? bev_pool_fwd_chunked_kernel<false, 64, 1> : bev_pool_fwd_chunked_kernel<false, 32, 1>;
  } else if (mode == 2) {
    kern = (g_sub == 64) ? bev_pool_fwd_chunked_kernel<false, 64, 2> : bev_pool_fwd_chunked_kernel<false, 32, 2>;
    lds = (size_t)(3 * g.BP + 4 * g.gpb) * sizeof(int) + (size_t)2 * g.gpb * g.c4 * sizeof(float4);
  } else {
    kern = (g_sub == 64) ? bev_pool_fwd_chunked_kernel<false, 64, 0> : bev_pool_fwd_chunked_kernel<false, 32, 0>;
  }
  ocrf::launch(OCRF_K_BEV_POOL_FWD, kern, dim3(g.n_blocks),
               dim3(kBlock), lds, stream, (unsigned long long*)nullptr, g.c4, g.gpw, n_intervals,
               n_points, counts, compact_rows, depth,
               reinterpret_cast<const float4*>(feat), ranks_depth, ranks_feat, ranks_bev,
               interval_starts, interval_lengths, reinterpret_cast<float4*>(dst), part, bmeta,
               row_of_vox, plan_codes, plan_meta);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  if (mode == 1) return 0;
  const unsigned grid2 = (unsigned)((g.n_blocks + g.gpb - 1) / g.gpb);
  ocrf::launch(OCRF_K_BEV_POOL_FIXUP, bev_pool_fwd_fixup_kernel, dim3(grid2), dim3(kBlock), 0, stream,
               g.c4, g.gpw, g.n_blocks, reinterpret_cast<float4*>(dst),
               static_cast<const float4*>(part), static_cast<const int4*>(bmeta));
  return (int)hipGetLastError();
}
}  // namespace

// Diagnostic knob (A/B timing of kernel variants in one process): key 0 = points per lane group.
int ocrf_tune_set(int key, int value) {
  if (key == 0 && (value == 32 || value == 64)) { g_sub = value; return 0; }
  return (int)hipErrorInvalidValue;
}

// Diagnostic build of pass 1 that records s_memtime at six phase boundaries per workgroup into
// stamps[n_blocks][8] (never used by the product path; see DESIGN.md "In-kernel stamps").
int ocrf_diag_bev_pool_v2_stamps(int c, int n_intervals, int n_points, const float* depth,
                                 const float* feat, const int* ranks_depth, const int* ranks_feat,
                                 const int* ranks_bev, const int* interval_starts,
                                 const int* interval_lengths, float* out, void* workspace,
                                 unsigned long long* stamps, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!vec_ok(c) || n_points <= 0) return (int)hipErrorInvalidValue;
  const FwdGeom g = fwd_geom(c, n_points);
  float4* part = static_cast<float4*>(workspace);
  int4* bmeta = reinterpret_cast<int4*>(static_cast<char*>(workspace) + g.part_bytes);
  auto kern = (g_sub == 64) ? bev_pool_fwd_chunked_kernel<true, 64, 0>
                             : bev_pool_fwd_chunked_kernel<true, 32, 0>;
  hipLaunchKernelGGL(kern, dim3(g.n_blocks), dim3(kBlock), g.lds,
                     stream, stamps, g.c4, g.gpw, n_intervals, n_points, (const int*)nullptr, 0, depth,
                     reinterpret_cast<const float4*>(feat), ranks_depth, ranks_feat, ranks_bev,
                     interval_starts, interval_lengths, reinterpret_cast<float4*>(out), part, bmeta,
                     (int*)nullptr, (int*)nullptr, (int*)nullptr);
  return (int)hipGetLastError();
}

size_t ocrf_bev_pool_v2_workspace_bytes(int c, int n_points) {
  if (!vec_ok(c) || n_points <= 0) return 0;
  return fwd_geom(c, n_points).total_bytes;
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
  return launch_fwd(c, n_intervals, n_points, nullptr, 0, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    interval_starts, interval_lengths, out, workspace, nullptr, stream);
}

namespace {
struct NchwGeom {
  size_t fwd_bytes, rows_off, map_off, total;
};
inline NchwGeom nchw_geom(int c, int n_intervals, int n_points, long n_voxels) {
  NchwGeom g;
  g.fwd_bytes = n_points > 0 ? align_up(fwd_geom(c, n_points).total_bytes, 256) : 0;
  g.rows_off = g.fwd_bytes;
  g.map_off = g.rows_off + align_up((size_t)n_intervals * c * sizeof(float), 256);
  g.total = g.map_off + align_up((size_t)n_voxels * sizeof(int), 256);
  return g;
}
}  // namespace

size_t ocrf_bev_pool_v2_nchw_workspace_bytes(int c, int n_intervals, int n_points, long n_voxels) {
  if (!vec_ok(c) || n_intervals < 0 || n_points < 0 || n_voxels <= 0) return 0;
  return nchw_geom(c, n_intervals, n_points, n_voxels).total;
}

namespace {
int nchw_impl(int c, int n_intervals, int n_points, const int* counts, const float* depth,
              const float* feat, const int* ranks_depth, const int* ranks_feat,
              const int* ranks_bev, const int* interval_starts,
              const int* interval_lengths, float* out, int B, int Z, int Y, int X,
              int layout, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!vec_ok(c) || n_intervals < 0 || n_points < 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 ||
      (layout != 0 && layout != 1) || !out)
    return (int)hipErrorInvalidValue;
  const long n_vox = (long)B * Z * Y * X;
  if (n_vox > 0x1fffffffL) return (int)hipErrorInvalidValue;      // row ids travel in 29 bits of the emit codes
  const NchwGeom g = nchw_geom(c, n_intervals, n_points, n_vox);
  if (!workspace || workspace_bytes < g.total || !aligned16(workspace) || !aligned16(feat))
    return (int)hipErrorInvalidValue;
  char* ws = static_cast<char*>(workspace);
  float* rows = reinterpret_cast<float*>(ws + g.rows_off);
  int* row_of_vox = reinterpret_cast<int*>(ws + g.map_off);
  hipError_t err = ocrf::zero_async(row_of_vox, (size_t)n_vox * sizeof(int), stream);
  if (err != hipSuccess) return (int)err;
  if (n_intervals > 0 && n_points > 0) {
    if (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_starts ||
        !interval_lengths)
      return (int)hipErrorInvalidValue;
    const int rc = launch_fwd(c, n_intervals, n_points, counts, 1, depth, feat, ranks_depth, ranks_feat,
                              ranks_bev, interval_starts, interval_lengths, rows, workspace,
                              row_of_vox, stream);
    if (rc != 0) return rc;
  }
  const int c4 = c / 4;
  const int tiles_x = (X + kTile - 1) / kTile;
  const long n_tiles = (long)B * Z * Y * tiles_x;
  const size_t lds = (size_t)kTile * (c + 1) * sizeof(float) + kTile * sizeof(int);
  ocrf::launch(OCRF_K_BEV_POOL_NCHW, bev_pool_rows_to_nchw_kernel, dim3((unsigned)n_tiles),
               dim3(kBlock), lds, stream, c4, B, Z, Y, X, tiles_x, layout,
               reinterpret_cast<const float4*>(rows), static_cast<const int*>(row_of_vox), out);
  return (int)hipGetLastError();
}
}  // namespace

int ocrf_bev_pool_v2_nchw(int c, int n_intervals, int n_points, const float* depth,
                          const float* feat, const int* ranks_depth, const int* ranks_feat,
                          const int* ranks_bev, const int* interval_starts,
                          const int* interval_lengths, float* out, int B, int Z, int Y, int X,
                          int layout, void* workspace, size_t workspace_bytes,
                          ocrf_stream_t stream_) {
  return nchw_impl(c, n_intervals, n_points, nullptr, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                   interval_starts, interval_lengths, out, B, Z, Y, X, layout, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream_));
}

int ocrf_bev_pool_v2_nchw_dyn(int c, int cap_intervals, int cap_points, const int* counts,
                              const float* depth, const float* feat, const int* ranks_depth,
                              const int* ranks_feat, const int* ranks_bev, const int* interval_starts,
                              const int* interval_lengths, float* out, int B, int Z, int Y, int X,
                              int layout, void* workspace, size_t workspace_bytes,
                              ocrf_stream_t stream_) {
  if (!counts || cap_intervals <= 0 || cap_points <= 0) return (int)hipErrorInvalidValue;
  return nchw_impl(c, cap_intervals, cap_points, counts, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                   interval_starts, interval_lengths, out, B, Z, Y, X, layout, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream_));
}

// ---------------------------------------------------------------------------------------------
// Plans: for rank vectors that stay the same across calls (static calibration, ``accelerate``)
// ---------------------------------------------------------------------------------------------
namespace {
struct PlanGeom {
  size_t codes_off, meta_off, map_off, total;
};
inline PlanGeom plan_geom(int c, int n_points, long n_vox) {
  const FwdGeom f = fwd_geom(c, n_points);
  PlanGeom g;
  g.codes_off = 0;
  g.meta_off = align_up((size_t)f.n_blocks * f.BP * sizeof(int), 256);
  g.map_off = g.meta_off + align_up((size_t)f.n_blocks * 4 * f.gpb * sizeof(int), 256);
  g.total = g.map_off + align_up((size_t)n_vox * sizeof(int), 256);
  return g;
}
}  // namespace

size_t ocrf_bev_pool_plan_bytes(int c, int n_points, long n_voxels) {
  if (!vec_ok(c) || n_points <= 0 || n_voxels <= 0) return 0;
  return plan_geom(c, n_points, n_voxels).total;
}

int ocrf_bev_pool_plan_build(int c, int n_intervals, int n_points, const int* ranks_bev,
                             const int* interval_starts, const int* interval_lengths, long n_voxels,
                             void* plan, size_t plan_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!vec_ok(c) || n_intervals <= 0 || n_points <= 0 || n_voxels <= 0 || n_voxels > 0x1fffffffL || !ranks_bev ||
      !interval_starts || !interval_lengths || !plan)
    return (int)hipErrorInvalidValue;
  const PlanGeom g = plan_geom(c, n_points, n_voxels);
  if (plan_bytes < g.total || !aligned16(plan)) return (int)hipErrorInvalidValue;
  char* p = static_cast<char*>(plan);
  int* row_of_vox = reinterpret_cast<int*>(p + g.map_off);
  hipError_t err = ocrf::zero_async(row_of_vox, (size_t)n_voxels * sizeof(int), stream);
  if (err != hipSuccess) return (int)err;
  return launch_fwd(c, n_intervals, n_points, nullptr, 1, nullptr, nullptr, nullptr, nullptr, ranks_bev, interval_starts,
                    interval_lengths, nullptr, nullptr, row_of_vox, stream, 1, reinterpret_cast<int*>(p + g.codes_off),
                    reinterpret_cast<int*>(p + g.meta_off));
}

int ocrf_bev_pool_v2_nchw_planned(int c, int n_intervals, int n_points, const float* depth, const float* feat,
                                  const int* ranks_depth, const int* ranks_feat, const void* plan, float* out,
                                  int B, int Z, int Y, int X, int layout, void* workspace,
                                  size_t workspace_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!vec_ok(c) || n_intervals <= 0 || n_points <= 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 ||
      (layout != 0 && layout != 1) || !out || !plan || !depth || !feat || !ranks_depth || !ranks_feat)
    return (int)hipErrorInvalidValue;
  const long n_vox = (long)B * Z * Y * X;
  if (n_vox > 0x1fffffffL) return (int)hipErrorInvalidValue;
  const NchwGeom g = nchw_geom(c, n_intervals, n_points, n_vox);
  if (!workspace || workspace_bytes < g.total || !aligned16(workspace) || !aligned16(feat) || !aligned16(plan))
    return (int)hipErrorInvalidValue;
  const PlanGeom pg = plan_geom(c, n_points, n_vox);
  char* ws = static_cast<char*>(workspace);
  float* rows = reinterpret_cast<float*>(ws + g.rows_off);
  char* p = const_cast<char*>(static_cast<const char*>(plan));
  const int rc = launch_fwd(c, n_intervals, n_points, nullptr, 1, depth, feat, ranks_depth, ranks_feat, nullptr, nullptr,
                            nullptr, rows, workspace, nullptr, stream, 2, reinterpret_cast<int*>(p + pg.codes_off),
                            reinterpret_cast<int*>(p + pg.meta_off));
  if (rc != 0) return rc;
  const int c4 = c / 4;
  const int tiles_x = (X + kTile - 1) / kTile;
  const long n_tiles = (long)B * Z * Y * tiles_x;
  const size_t lds = (size_t)kTile * (c + 1) * sizeof(float) + kTile * sizeof(int);
  ocrf::launch(OCRF_K_BEV_POOL_NCHW, bev_pool_rows_to_nchw_kernel, dim3((unsigned)n_tiles), dim3(kBlock), lds, stream,
               c4, B, Z, Y, X, tiles_x, layout, reinterpret_cast<const float4*>(rows),
               reinterpret_cast<const int*>(p + pg.map_off), out);
  return (int)hipGetLastError();
}

int ocrf_bev_pool_v2_check_intervals(int n_intervals, int n_points, const int* interval_starts,
                                     const int* interval_lengths, int* flag,
                                     ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_intervals < 0 || n_points < 0 || !flag) return (int)hipErrorInvalidValue;
  hipError_t err = ocrf::zero_async(flag, sizeof(int), stream);
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

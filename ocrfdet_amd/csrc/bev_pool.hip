// BEVPoolv2 voxel pooling for MI355X (gfx950) — forward + backward.
//
// Replaces mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu (reference: one thread per
// (interval, channel) looping the whole interval — interval lengths at the reference shape run
// from 1 to 5 216 with mean 51, so that kernel is bound by its longest thread — followed by a
// zero-fill, a strided permute copy (bev_pool.py:91) and a cat (view_transformer.py:194)).
//
// Design (DESIGN.md "bev_pool_v2"): ONE output-stationary pass.
//   * The unit of work is a TILE of 64 consecutive voxels of one (b, z) plane: its pooled rows are
//     built in LDS and leave the chip exactly once, already in the layout the caller asked for
//     (channel-major runs of 64 floats = 256 B per channel, or channel-last rows), zeros included.
//     There is no row buffer, no second pass and no pre-zeroing of the output.
//   * A dense voxel table (start, length per voxel) built from the interval list replaces every
//     search: a tile reads its 64 entries with one coalesced load.  Any interval layout is legal
//     (unsorted, overlapping, empty, gaps) exactly as for the reference's kernel.
//   * Load balance: tile point counts are as skewed as the interval lengths (at the headline shape
//     1 % of the tiles hold 16 % of the points), so a tile with more than Q points is cut into
//     SLICES of Q points that run as separate workgroups; a slice leaves a partial tile (slab) and
//     the LAST slice to arrive (agent-scope release / ticket / acquire) adds the slabs in ascending
//     slice order and writes the tile.  The order is fixed whoever arrives last: no float atomics,
//     results are bitwise reproducible.
//   * Inside a workgroup the tile's points are walked in rounds of BP = 12 x 32 points (C = 80):
//     one lane group (C/4 lanes, a float4 of channels per lane -> a feat row is one coalesced
//     16 B/lane read) per 32-point sub-chunk, branch-free `acc = fma(row, w, acc)` with marked
//     emits; a voxel that lies inside one sub-chunk is summed in exactly the reference's order
//     (bit-exact), pieces of voxels cut by sub-chunk borders are added in ascending order.
//   * Workgroups are dealt to the 8 XCDs in contiguous, cost-balanced unit ranges, so that the
//     tiles of one BEV region — which gather the same feat rows and depth cells — share one L2.
//   * fp32 fmaf, like the reference's contracted `psum += f*d`.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kTVmax = 64;       // voxels per tile: 64 or 32 (g_tv)
int g_tv = 64;                   // ocrf_tune_set(3, 32|64)
constexpr int kSub = 32;         // points per lane group per round
constexpr int kUnitCost = 192;   // fixed cost of a unit in point equivalents (table read, zero-fill, tile write)
int g_rounds = 4;                // rounds per slice: Q = g_rounds * BP points (ocrf_tune_set(0, 1..64))
int g_xcd = 1;                   // XCD-contiguous unit ranges (ocrf_tune_set(1, 0|1))
int g_grid = 0;                  // workgroups of the pooling launch: 0 = one per unit, else this many (ocrf_tune_set(2, n))

__device__ __forceinline__ float4 fma4(float4 f, float d, float4 a) {
  a.x = fmaf(f.x, d, a.x);
  a.y = fmaf(f.y, d, a.y);
  a.z = fmaf(f.z, d, a.z);
  a.w = fmaf(f.w, d, a.w);
  return a;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// ---------------------------------------------------------------------------------------------
// Compatibility kernel: one thread per (interval, channel), exactly the reference's mapping
// (bev_pool_cuda.cu:21-48).  Behind the exact-signature entry point and for channel counts the
// tile kernel does not take (C % 4 != 0, C < 32, C > 256).
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
// Preparation (rank-only; cached in a plan when the rank vectors are):
//   prep block = Header | tab int2[n_vox] | tile_cnt int[n_tiles] | arrive int[n_tiles] |
//                tinfo int2[n_tiles] | units 2 x int4[max_units] {tile, j0, j1, slice} {slices, first slab, -, -}
//   (A) zero tab + tile_cnt + arrive; (B) interval k -> tab[ranks_bev[start_k]] = (start_k, len_k),
//   tile_cnt[tile] += len_k; (C) one workgroup turns the tile counts into the unit list.
// ---------------------------------------------------------------------------------------------
struct Header {
  int n_units;
  int xs[9];        // unit range of XCD x: [xs[x], xs[x+1])
  int status;       // bit 0: the unit list was clamped (more units than the workspace was sized for)
  int pad[5];
};
static_assert(sizeof(Header) == 64, "Header is one 64-byte line");

struct Geometry {    // of the voxel space and its tiling
  int planes, YX, tpp, n_tiles;   // tpp = tiles per plane
  long n_vox;
};
inline Geometry make_geometry(long planes, long YX) {
  Geometry g;
  g.planes = (int)planes;
  g.YX = (int)YX;
  g.tpp = (int)((YX + g_tv - 1) / g_tv);
  g.n_tiles = (int)(planes * g.tpp);
  g.n_vox = planes * YX;
  return g;
}

__global__ __launch_bounds__(kBlock) void bev_pool_table_kernel(
    int n_intervals, int n_points, const int* __restrict__ counts, int YX, int tpp, int tv, long n_vox,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, int2* __restrict__ tab, int* __restrict__ tile_cnt) {
  if (counts) { n_points = counts[0]; n_intervals = counts[1]; }
  const int k = blockIdx.x * kBlock + threadIdx.x;
  int tile = -1, l = 0;
  if (k < n_intervals) {
    const int s = interval_starts[k];
    l = interval_lengths[k];
    if (s >= 0 && s < n_points && l > 0) {            // an empty interval pools to 0: the table's default
      l = min(l, n_points - s);
      const int vox = ranks_bev[s];
      if (vox >= 0 && vox < n_vox) {
        tab[vox] = make_int2(s, l);
        tile = (vox / YX) * tpp + (vox % YX) / tv;
      }
    }
  }
  if (tile < 0) l = 0;
  // one atomic per RUN of equal tiles inside the wave (the producers emit intervals in voxel order, so a wave
  // holds a handful of runs): 64 same-address atomics per wave serialise at the L2
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(tile, 1);
  const bool head = (lane == 0) || (tile != prev);
  int inc = l;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  const unsigned long long heads = __ballot(head);
  const unsigned long long later = heads & ~((2ull << lane) - 1ull);      // heads after this lane
  const int end_lane = later ? (__ffsll((long long)later) - 2) : 63;       // last lane of this lane's run
  const int run_total = __shfl(inc, end_lane) - (inc - l);
  if (head && tile >= 0 && run_total > 0) atomicAdd(&tile_cnt[tile], run_total);
}

// exclusive scan of one int per thread over a 1024-thread workgroup; returns the total in `total`
__device__ __forceinline__ int block_scan_1024(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();                 // s_w may still be read by the previous call
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    const int c = s_w[w];
    if (w < wave) base += c;
    tot += c;
  }
  total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(1024) void bev_pool_units_kernel(
    int n_tiles, int Q, int max_units, int max_slabs, const int* __restrict__ tile_cnt, Header* __restrict__ hdr,
    int2* __restrict__ tinfo, int4* __restrict__ units) {
  // Unit list: XCD x owns units [xs[x], xs[x+1]) = a contiguous range of tiles (cost-balanced: the tiles of one
  // BEV region gather the same feat rows and share that XCD's L2); inside a range the slices of cut tiles come
  // FIRST — they are the longest units, started last they would be the tail of the launch — then the whole
  // tiles.  The order inside a class is whatever the slot atomics give: results do not depend on it.
  __shared__ int s_w[16];
  __shared__ int s_split[8], s_whole[8], s_cur_split[8], s_cur_whole[8], s_base[9];
  const int tid = threadIdx.x;
  auto slices_of = [&](int cnt) { return max(1, (cnt + Q - 1) / Q); };
  auto cost_of = [&](int cnt, int ns) { return (cnt + ns * kUnitCost + 63) >> 6; };   // in units of 64 points
  long cost_total = 0;
  {
    int part = 0;
    for (int t = tid; t < n_tiles; t += 1024) {
      const int cnt = tile_cnt[t];
      part += cost_of(cnt, slices_of(cnt));
    }
    int tot;
    (void)block_scan_1024(part, s_w, tot);
    cost_total = max(tot, 1);
  }
  if (tid < 8) s_split[tid] = s_whole[tid] = s_cur_split[tid] = s_cur_whole[tid] = 0;
  __syncthreads();
  // pass 2: XCD of every tile (by its exclusive cost prefix), units per XCD and class; slab bases
  int slab_base = 0;
  long cost_base = 0;
  for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
    const int t = t0 + tid;
    const int cnt = (t < n_tiles) ? tile_cnt[t] : 0;
    const int ns = (t < n_tiles) ? slices_of(cnt) : 0;
    const int cost = (t < n_tiles) ? cost_of(cnt, ns) : 0;
    int tot_s, tot_c;
    const int sb = slab_base + block_scan_1024(ns > 1 ? ns : 0, s_w, tot_s);
    const long cb = cost_base + block_scan_1024(cost, s_w, tot_c);
    if (t < n_tiles) {
      const int x = (int)min(7L, cb * 8 / cost_total);
      // The slab buffer is sized for tile counts that sum to n_points; duplicate / overlapping intervals can inflate
      // them.  A cut tile whose slabs would not fit is NOT cut (-2: one long unit, still correct) and the status says so.
      const bool fits = sb + ns <= max_slabs;
      if (ns > 1 && !fits) atomicOr(&hdr->status, 1);
      tinfo[t] = make_int2(ns > 1 ? (fits ? sb : -2) : -1, x);
      if (ns > 1 && fits) atomicAdd(&s_split[x], ns);
      else atomicAdd(&s_whole[x], 1);
    }
    slab_base += tot_s;
    cost_base += tot_c;
  }
  __syncthreads();
  if (tid == 0) {
    int b = 0;
    for (int x = 0; x < 8; ++x) { s_base[x] = b; b += s_split[x] + s_whole[x]; }
    s_base[8] = b;
  }
  __syncthreads();
  const int n_all = s_base[8];
  // pass 3: hand out the slots
  for (int t = tid; t < n_tiles; t += 1024) {
    const int cnt = tile_cnt[t];
    const int2 ti = tinfo[t];
    const int ns = (ti.x == -2) ? 1 : slices_of(cnt);
    const int x = ti.y;
    const int slot = (ns > 1) ? s_base[x] + atomicAdd(&s_cur_split[x], ns)
                              : s_base[x] + s_split[x] + atomicAdd(&s_cur_whole[x], 1);
    for (int s = 0; s < ns; ++s) {
      if (slot + s >= max_units) { atomicOr(&hdr->status, 1); break; }   // only inconsistent tile counts get here
      units[2 * (slot + s)] = make_int4(t, s * Q, ns == 1 ? cnt : min(cnt, (s + 1) * Q), s);
      units[2 * (slot + s) + 1] = make_int4(ns, ti.x, 0, 0);
    }
  }
  if (tid == 0) {
    hdr->n_units = min(n_all, max_units);
    for (int x = 0; x <= 8; ++x) hdr->xs[x] = min(s_base[x], max_units);
  }
}

// ---------------------------------------------------------------------------------------------
// The pooling kernel.
//   c4 = C/4 lanes per lane group, gpw = 64/c4 groups per wave, gpb = 4*gpw groups per workgroup,
//   BP = gpb*kSub points per round, ldq = tile row pitch in float4 (odd: conflict-free b128 column reads).
// Per unit (tile, slice [j0, j1) of the tile's flattened point list):
//   (1) table entries of the 64 voxels -> s_start, exclusive prefix of the lengths -> s_pre; zero the tile;
//   per round of BP points:
//   (2) stage: flattened index j -> (voxel slot v by binary search in s_pre, point p) -> depth weight,
//       feat row, emit code; per sub-chunk the piece table s_meta {head, tail, tail voxel, head voxel};
//       code: 1 | (v << 2) = last point of a voxel that lies inside this sub-chunk -> store the row;
//             2 = last point of a HEAD piece (the voxel began in an earlier sub-chunk) -> keep in a register;
//       a TAIL piece (the voxel goes on in the next sub-chunk) is whatever is left in the accumulator.
//       head: 0 none / 1 the head voxel ends in this sub-chunk / 2 it runs through;
//   (3) each lane group walks its kSub points branch-free with 4 + 4 row gathers in flight per lane;
//   (4) pieces are combined in ascending order through LDS: the owner of a tail adds the heads that follow,
//       tile[v] += sum (the tile is zero at the start, a voxel that crosses a ROUND border is continued by
//       the next round's sub-chunk 0 — rounds are separated by barriers, so the order stays ascending);
//   (5) whole tile: the rows leave in the caller's layout.  Slice of a cut tile: the rows leave as a slab,
//       then release -> ticket; the last arriver acquires, adds the slabs in slice order and writes the tile.
// ---------------------------------------------------------------------------------------------
constexpr int kBatch = 4;
struct Batch {
  float4 v[kBatch];
  int2 wc[kBatch];
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// feat rows are gathered through a buffer descriptor: a 32-bit byte offset per lane (row offset staged in LDS +
// 16 * lane-in-group) instead of a 64-bit address multiply-add per load
__device__ __forceinline__ void load_batch(Batch& q, int l0, const int* s_rf, const int2* s_wc,
                                           const __amdgpu_buffer_rsrc_t& feat_rsrc, int lg16) {
#pragma unroll
  for (int k = 0; k < kBatch; ++k) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, s_rf[l0 + k] + lg16, 0, 0);
    q.v[k] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  }
#pragma unroll
  for (int k = 0; k < kBatch; ++k) q.wc[k] = s_wc[l0 + k];
}

struct TileArgs {
  int c4, gpw, ldq;
  int YX, tpp, Z, layout;        // layout 0: (B,C,Z,Y,X)  1: (B,Z*C,Y,X)  2: rows (n_vox, C)
  int n_points;
  const int* counts;             // device [n_points, n_intervals] or null
  const Header* hdr;
  const int2* tab;
  const int4* units;
  int* arrive;
  const float* depth;
  const float4* feat4;
  int feat_records;              // bytes the feature tensor holds (the buffer resource's bound; 0x7fffffff: not stated)
  const int* ranks_depth;
  const int* ranks_feat;
  float* out;
  float4* slabs;
  int xcd;
};

template <bool STAMP, int kTV>
__global__ __launch_bounds__(kBlock, 5) void bev_pool_tile_kernel(TileArgs a, unsigned long long* __restrict__ stamps) {
  OCRF_POOL_PRIO();
  extern __shared__ __attribute__((aligned(16))) int smem[];
  const int tid = threadIdx.x;
  const int c4 = a.c4, gpw = a.gpw, ldq = a.ldq;
  const int gpb = gpw * (kBlock / kWave);
  const int BP = gpb * kSub;
  float4* tile = reinterpret_cast<float4*>(smem);                  // [kTV][ldq]
  float4* s_part = tile + kTV * ldq;                               // [gpb][c4] head pieces
  int2* s_wc = reinterpret_cast<int2*>(s_part + gpb * c4);         // [BP] {weight bits, code}
  int* s_rf = reinterpret_cast<int*>(s_wc + BP);                   // [BP]
  int* s_meta = s_rf + BP;                                         // [gpb][4]
  int* s_start = s_meta + 4 * gpb;                                 // [kTV]
  int* s_pre = s_start + kTV;                                      // [kTV + 1]
  int* s_flag = s_pre + kTV + 1;

  const int wave = tid / kWave, lane = tid % kWave;
  const int gi = lane / c4, lg = lane % c4;
  const int gb = wave * gpw + gi;
  const int n_points = a.counts ? a.counts[0] : a.n_points;
  // whole 4 GB window from the feat base: row offsets are 32-bit byte offsets (the host checks nothing larger is needed)
  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.feat4), 0, a.feat_records, 0x00020000);
  const int lg16 = lg * 16;
  const int row_bytes = c4 * 16;

  // units of this workgroup: a strided walk over the unit range of "its" XCD (blocks b and b + 8 share one)
  int u, u_end, u_step;
  if (a.xcd) {
    const int x = blockIdx.x & 7;
    u = a.hdr->xs[x] + (blockIdx.x >> 3);
    u_end = a.hdr->xs[x + 1];
    u_step = gridDim.x >> 3;
  } else {
    u = blockIdx.x;
    u_end = a.hdr->n_units;
    u_step = gridDim.x;
  }
  unsigned long long t_prev = 0, t_acc[6] = {0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_prev;
      t_prev = t;
    }
  };

  for (; u < u_end; u += u_step) {
    stamp(-1);
    const int4 unit = a.units[2 * u];
    const int4 uinfo = a.units[2 * u + 1];           // {slices of the tile, first slab}: no dependent read later
    const int tileid = unit.x;
    const int plane = tileid / a.tpp, kt = tileid % a.tpp;
    const int v0 = kt * kTV;
    const int nv = min(kTV, a.YX - v0);
    const long vox0 = (long)plane * a.YX + v0;

    // (1) the tile's voxel table; zero the tile
    if (tid < kTV) {
      int2 e = make_int2(0, 0);
      if (tid < nv) e = a.tab[vox0 + tid];
      if (e.x < 0 || e.x >= n_points) e = make_int2(0, 0);
      e.y = max(0, min(e.y, n_points - e.x));
      s_start[tid] = e.x;
      int inc = e.y;
      for (int off = 1; off < kTV; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (tid >= off) inc += o;
      }
      s_pre[tid + 1] = inc;
      if (tid == 0) s_pre[0] = 0;
    }
    for (int i = tid; i < kTV * ldq; i += kBlock) tile[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int j0 = unit.y, j1 = min(unit.z, s_pre[kTV]);
    stamp(0);

    for (int jr = j0; jr < j1; jr += BP) {
      const int je = min(jr + BP, j1);
      // (2) stage the round
#pragma unroll 1
      for (int i = tid; i < BP; i += kBlock) {
        const int j = jr + i;
        const int g = i / kSub;
        const int sc0 = jr + g * kSub;
        const int sc_end = min(sc0 + kSub, je);
        const int jc = min(j, je - 1);
        int v = 0;
#pragma unroll
        for (int step = kTV / 2; step >= 1; step >>= 1)
          if (s_pre[v + step] <= jc) v += step;
        const int pv = s_pre[v], pn = s_pre[v + 1];
        const int p = s_start[v] + (jc - pv);
        const int rf = a.ranks_feat[p];
        float w = 0.f;
        int code = 0;
        if (j < je) {
          w = a.depth[a.ranks_depth[p]];
          const bool lastv = (j + 1 == pn), end_sc = (j + 1 == sc_end), here = pv >= sc0;
          if (lastv && here) code = 1 | (v << 2);
          else if ((lastv || end_sc) && !here) code = 2;
          if (end_sc) {                                    // last point of its sub-chunk: the tail entry
            s_meta[4 * g + 1] = (here && pn > sc_end) ? 1 : 0;
            s_meta[4 * g + 2] = v;
          }
        }
        // padding points repeat the round's last row with weight 0 (no foreign row is touched)
        s_wc[i] = make_int2(__float_as_int(w), code);
        s_rf[i] = rf * row_bytes;
        if (i % kSub == 0) {                               // first point of its sub-chunk: the head entry
          int head = 0;
          if (sc0 < je) head = (pv < sc0) ? ((pn > sc_end) ? 2 : 1) : 0;
          else s_meta[4 * g + 1] = 0;
          s_meta[4 * g + 0] = head;
          s_meta[4 * g + 3] = v;
        }
      }
      __syncthreads();
      stamp(1);

      // (3) one lane group per sub-chunk
      const bool active = (gi < gpw) && (jr + gb * kSub < je);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 head_acc = acc;
      if (active) {
        auto consume = [&](const Batch& q) {
#pragma unroll
          for (int k = 0; k < kBatch; ++k) {
            const int2 wc = q.wc[k];
            acc = fma4(q.v[k], __int_as_float(wc.x), acc);
            const int code = wc.y;
            // emits are rare (one per voxel and lane group): a skipped branch costs less than the selects
            if (code) {
              // ONE kind of store in the loop (LDS) and a register select for the head piece: an `else` here makes
              // hipcc spill head_acc and merge both sides into a flat_store, whose vmcnt(0) drains the gathers
              if (code & 1) tile[(code >> 2) * ldq + lg] = acc;
              const bool is_head = (code & 1) == 0;
              head_acc.x = is_head ? acc.x : head_acc.x;
              head_acc.y = is_head ? acc.y : head_acc.y;
              head_acc.z = is_head ? acc.z : head_acc.z;
              head_acc.w = is_head ? acc.w : head_acc.w;
              acc = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        };
        Batch A, B;
        const int l0 = gb * kSub;
        load_batch(A, l0, s_rf, s_wc, feat_rsrc, lg16);
#pragma nounroll      // unrolled, the scheduler hoists every gather to the top: > 200 VGPRs, 2 waves per SIMD
        for (int b = 0; b < kSub / kBatch; b += 2) {
          load_batch(B, l0 + kBatch * (b + 1), s_rf, s_wc, feat_rsrc, lg16);
          consume(A);
          if (b + 2 < kSub / kBatch) load_batch(A, l0 + kBatch * (b + 2), s_rf, s_wc, feat_rsrc, lg16);
          consume(B);
        }
        s_part[gb * c4 + lg] = head_acc;             // head piece (meaningful iff s_meta says so)
      }
      __syncthreads();
      stamp(2);

      // (4) pieces of voxels cut by sub-chunk borders, ascending; a tail piece (what the last voxel of the
      // sub-chunk left in `acc`) never leaves its owner's registers
      if (active) {
        const int n_gb = (je - jr + kSub - 1) / kSub;
        if (s_meta[4 * gb + 1]) {
          for (int g2 = gb + 1; g2 < n_gb; ++g2) {
            const int h = s_meta[4 * g2];
            if (h == 0) break;
            acc = add4(acc, s_part[g2 * c4 + lg]);
            if (h == 1) break;
          }
          float4* dst = tile + s_meta[4 * gb + 2] * ldq + lg;
          *dst = add4(*dst, acc);
        }
        if (gb == 0 && s_meta[0]) {
          if (s_meta[0] == 2) {
            for (int g2 = 1; g2 < n_gb; ++g2) {
              const int h = s_meta[4 * g2];
              if (h == 0) break;
              head_acc = add4(head_acc, s_part[g2 * c4 + lg]);
              if (h == 1) break;
            }
          }
          float4* dst = tile + s_meta[3] * ldq + lg;
          *dst = add4(*dst, head_acc);
        }
      }
      __syncthreads();
      stamp(3);
    }

    // (5) the tile leaves the chip
    const int2 ti = make_int2(uinfo.x, uinfo.y);
    bool write_tile = true;
    if (ti.x > 1) {
      // the slab leaves WRITE-THROUGH (sc1 stores: the bytes are in memory once the wave's vmcnt drains, no L2
      // write-back fence needed — cdna_hip_programming.md 'In-launch split-K reduction'), then the ticket
      float4* slab = a.slabs + (long)(ti.y + unit.w) * kTV * c4;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, kTV * c4 * (int)sizeof(float4), 0x00020000);
      if (gi < gpw) {
        for (int v = gb; v < nv; v += gpb) {
          const float4 x = tile[v * ldq + lg];
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          const u4 bits = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
          __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (v * c4 + lg) * (int)sizeof(float4), 0, 16);   // aux 16 = sc1
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&a.arrive[tileid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (old == ti.x - 1) ? 1 : 0;
      }
      __syncthreads();
      write_tile = *s_flag != 0;
      if (write_tile) {
        // the ticket counter is back at 0 for the next call (plans keep it across calls)
        if (tid == 0) __hip_atomic_store(&a.arrive[tileid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The slabs are read with system-scope loads (sc0 sc1: served past this XCD's L2, where the other slices'
        // write-through stores are) instead of an agent-scope acquire fence: the fence is a buffer_inv that throws
        // away the L2 contents every other workgroup of the XCD is gathering feat rows from.
        float4* s0 = a.slabs + (long)ti.y * kTV * c4;
        const auto srs = __builtin_amdgcn_make_buffer_rsrc(s0, 0, ti.x * kTV * c4 * (int)sizeof(float4), 0x00020000);
        if (gi < gpw) {
          // slice order for the sums, but four slab loads in flight per lane (they do not depend on each other)
          constexpr int kSB = 4;
          for (int v = gb; v < nv; v += gpb) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s0 = 0; s0 < ti.x; s0 += kSB) {
              u32x4 r[kSB];
#pragma unroll
              for (int k = 0; k < kSB; ++k) {
                const int s = min(s0 + k, ti.x - 1);
                r[k] = __builtin_amdgcn_raw_buffer_load_b128(srs, ((s * kTV + v) * c4 + lg) * (int)sizeof(float4), 0, 17);
              }
#pragma unroll
              for (int k = 0; k < kSB; ++k) {
                if (s0 + k < ti.x) {
                  const float4 x = make_float4(__uint_as_float(r[k].x), __uint_as_float(r[k].y), __uint_as_float(r[k].z),
                                               __uint_as_float(r[k].w));
                  acc = (s0 + k == 0) ? x : add4(acc, x);
                }
              }
            }
            tile[v * ldq + lg] = acc;
          }
        }
        __syncthreads();
      }
    }
    if (write_tile) {
      if (a.layout == 2) {
        float4* o = reinterpret_cast<float4*>(a.out) + vox0 * c4;
        if (gi < gpw)
          for (int v = gb; v < nv; v += gpb) o[v * c4 + lg] = tile[v * ldq + lg];
      } else {
        const int C = 4 * c4;
        const long plane_sz = (long)a.YX;
        const int b = plane / a.Z, z = plane % a.Z;
        const long base0 = (a.layout == 0) ? (((long)b * C * a.Z + z) * plane_sz) : ((long)plane * C * plane_sz);
        const long cstride = (a.layout == 0) ? (long)a.Z * plane_sz : plane_sz;
        // a wave instruction writes 64 / kTV channel runs of kTV floats (256 or 2 x 128 contiguous bytes)
        constexpr int kRuns = kWave / kTV;
        const int vl = lane % kTV, sub = lane / kTV;
        float* o = a.out + base0 + v0 + vl;
        if (vl < nv) {
          for (int q = wave * kRuns + sub; q < c4; q += (kBlock / kWave) * kRuns) {
            const float4 x = tile[vl * ldq + q];
            float* oc = o + (long)(4 * q) * cstride;
            oc[0] = x.x;
            oc[cstride] = x.y;
            oc[2 * cstride] = x.z;
            oc[3 * cstride] = x.w;
          }
        }
      }
    }
    stamp(4);
    __syncthreads();            // the next unit reuses the LDS
    if constexpr (STAMP) {
      if (tid == 0) {
        unsigned long long* s = stamps + (long)u * 8;
        for (int k = 0; k < 5; ++k) s[k] = t_acc[k];
        s[5] = (unsigned long long)(j1 - j0);
        for (int k = 0; k < 5; ++k) t_acc[k] = 0;
      }
    }
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

// ---- host-side geometry of the prep block, the slabs and one launch ------------------------------
struct PoolGeom {
  int c4, gpw, gpb, BP, ldq, Q;
  size_t lds;
};
inline PoolGeom pool_geom(int c) {
  PoolGeom g;
  g.c4 = c / 4;
  g.gpw = kWave / g.c4;
  g.gpb = g.gpw * (kBlock / kWave);
  g.BP = g.gpb * kSub;
  g.ldq = g.c4 | 1;                 // odd pitch in float4: b128 column reads are conflict-free
  g.Q = g_rounds * g.BP;
  g.lds = (size_t)(g_tv * g.ldq + g.gpb * g.c4) * sizeof(float4) + (size_t)g.BP * (sizeof(int2) + sizeof(int)) +
          (size_t)(4 * g.gpb + g_tv + g_tv + 1 + 1) * sizeof(int);
  g.lds = align_up(g.lds, 16);
  return g;
}
inline int max_units_of(const Geometry& vg, const PoolGeom& pg, long n_points) {
  return (int)(vg.n_tiles + n_points / pg.Q + 8);
}
struct PrepLayout {
  size_t tab_off, cnt_off, arrive_off, tinfo_off, units_off, total;
  size_t zero_bytes;       // tab + tile_cnt + arrive are contiguous and zeroed together
  int max_units;
};
inline PrepLayout prep_layout(const Geometry& vg, const PoolGeom& pg, long n_points) {
  PrepLayout L;
  L.max_units = max_units_of(vg, pg, n_points);
  L.tab_off = sizeof(Header);
  L.cnt_off = L.tab_off + align_up((size_t)vg.n_vox * sizeof(int2), 16);
  L.arrive_off = L.cnt_off + align_up((size_t)vg.n_tiles * sizeof(int), 16);
  const size_t zero_end = L.arrive_off + align_up((size_t)vg.n_tiles * sizeof(int), 16);
  L.zero_bytes = zero_end - L.tab_off;
  L.tinfo_off = zero_end;
  L.units_off = L.tinfo_off + align_up((size_t)vg.n_tiles * sizeof(int2), 16);
  L.total = align_up(L.units_off + (size_t)L.max_units * 2 * sizeof(int4), 256);
  return L;
}
// slabs: every slice of a cut tile; cut tiles hold more than Q points each, so at most 2 n_points / Q slices
inline size_t slab_bytes_of(int c, const PoolGeom& pg, long n_points) {
  return align_up((size_t)(2 * (n_points / pg.Q) + 2) * kTVmax * c * sizeof(float), 256);
}

int launch_prep(int c, int n_intervals, int n_points, const int* counts, const Geometry& vg, const int* ranks_bev,
                const int* interval_starts, const int* interval_lengths, void* prep, hipStream_t stream) {
  const PoolGeom pg = pool_geom(c);
  const PrepLayout L = prep_layout(vg, pg, n_points);
  char* p = static_cast<char*>(prep);
  Header* hdr = reinterpret_cast<Header*>(p);
  // header + table + counters in one zero-fill (they are contiguous)
  hipError_t err = ocrf::zero_async(p, sizeof(Header) + L.zero_bytes, stream);
  if (err != hipSuccess) return (int)err;
  if (n_intervals > 0 && n_points > 0) {
    hipLaunchKernelGGL(bev_pool_table_kernel, dim3((unsigned)((n_intervals + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       stream, n_intervals, n_points, counts, vg.YX, vg.tpp, g_tv, vg.n_vox, ranks_bev, interval_starts,
                       interval_lengths, reinterpret_cast<int2*>(p + L.tab_off), reinterpret_cast<int*>(p + L.cnt_off));
    err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  hipLaunchKernelGGL(bev_pool_units_kernel, dim3(1), dim3(1024), 0, stream, vg.n_tiles, pg.Q, L.max_units,
                     (int)(2 * ((long)n_points / pg.Q) + 2), reinterpret_cast<const int*>(p + L.cnt_off), hdr, reinterpret_cast<int2*>(p + L.tinfo_off),
                     reinterpret_cast<int4*>(p + L.units_off));
  return (int)hipGetLastError();
}

int launch_tiles(int c, int n_points, const int* counts, const Geometry& vg, int Z, int layout, const float* depth,
                 const float* feat, const int* ranks_depth, const int* ranks_feat, const void* prep, float* out,
                 void* slabs, hipStream_t stream, unsigned long long* stamps = nullptr, size_t feat_bytes = 0) {
  const PoolGeom pg = pool_geom(c);
  const PrepLayout L = prep_layout(vg, pg, n_points);
  const char* p = static_cast<const char*>(prep);
  TileArgs a;
  a.c4 = pg.c4; a.gpw = pg.gpw; a.ldq = pg.ldq;
  a.YX = vg.YX; a.tpp = vg.tpp; a.Z = Z; a.layout = layout;
  a.n_points = n_points;
  a.counts = counts;
  a.hdr = reinterpret_cast<const Header*>(p);
  a.tab = reinterpret_cast<const int2*>(p + L.tab_off);
  a.units = reinterpret_cast<const int4*>(p + L.units_off);
  a.arrive = reinterpret_cast<int*>(const_cast<char*>(p) + L.arrive_off);
  a.depth = depth;
  a.feat4 = reinterpret_cast<const float4*>(feat);
  // the kernel reads features through a buffer resource (32-bit byte offsets): its bound is the tensor's real size when
  // the caller states it — a rank beyond it then reads 0 instead of someone else's memory
  a.feat_records = feat_bytes ? (int)std::min<size_t>(feat_bytes, 0x7fffffffu) : 0x7fffffff;
  a.ranks_depth = ranks_depth;
  a.ranks_feat = ranks_feat;
  a.out = out;
  a.slabs = static_cast<float4*>(slabs);
  a.xcd = g_xcd;
  // one workgroup per unit of the longest XCD range in the common case; a workgroup walks on (stride grid / 8)
  // when a range is longer, so any split of the units over the XCDs is covered
  int per_xcd = (L.max_units + 7) / 8;
  per_xcd += per_xcd / 4;
  if (g_grid > 0) per_xcd = min(per_xcd, max(1, g_grid / 8));
  const unsigned grid = (unsigned)(8 * per_xcd);
  if (stamps) {
    auto k = (g_tv == 32) ? bev_pool_tile_kernel<true, 32> : bev_pool_tile_kernel<true, 64>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), pg.lds, stream, a, stamps);
  } else {
    auto k = (g_tv == 32) ? bev_pool_tile_kernel<false, 32> : bev_pool_tile_kernel<false, 64>;
    ocrf::launch(OCRF_K_BEV_POOL_FWD, k, dim3(grid), dim3(kBlock), pg.lds, stream, a, (unsigned long long*)nullptr);
  }
  return (int)hipGetLastError();
}

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

// Diagnostic knobs (A/B timing of kernel variants in one process): key 0 = rounds per slice (1..64),
// key 1 = XCD-contiguous unit ranges on / off.  Plans and workspaces are sized for the value in force when
// they were built: change a knob only between independent runs.
int ocrf_tune_set(int key, int value) {
  if (key >= 10 && key < 20) { ocrf::raster_plan_tune(key, value); return 0; }      // raster_plan.hip
  if (key >= 20 && key < 30) { ocrf::hoa_tune(key, value); return 0; }              // hoa.hip
  if (key == 0 && value >= 1 && value <= 64) { g_rounds = value; return 0; }
  if (key == 1 && (value == 0 || value == 1)) { g_xcd = value; return 0; }
  if (key == 2 && value >= 0) { g_grid = value; return 0; }
  if (key == 3 && (value == 32 || value == 64)) { g_tv = value; return 0; }
  return (int)hipErrorInvalidValue;
}

size_t ocrf_bev_pool_v2_workspace_bytes(int c, int n_points, long n_voxels) {
  if (!vec_ok(c) || n_points < 0 || n_voxels <= 0) return 0;
  const PoolGeom pg = pool_geom(c);
  return prep_layout(make_geometry(1, n_voxels), pg, n_points).total + slab_bytes_of(c, pg, n_points);
}

int ocrf_bev_pool_v2(int c, int n_intervals, int n_points, long n_voxels, const float* depth, const float* feat,
                     const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                     const int* interval_starts, const int* interval_lengths, float* out,
                     void* workspace, size_t workspace_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (c <= 0 || n_intervals < 0 || n_points < 0 || n_voxels <= 0 || !out) return (int)hipErrorInvalidValue;
  if (n_intervals > 0 && n_points > 0 &&
      (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_starts || !interval_lengths))
    return (int)hipErrorInvalidValue;
  if (!vec_ok(c) || !aligned16(feat) || !aligned16(out)) {
    // scalar mapping of the reference; correct for every C and alignment (out pre-zeroed by the caller)
    if (n_intervals == 0 || n_points == 0) return 0;
    const long total = (long)n_intervals * c;
    const unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
    ocrf::launch(OCRF_K_BEV_POOL_INTERVAL, bev_pool_interval_kernel, dim3(grid), dim3(kBlock), 0,
                 stream, c, n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                 interval_starts, interval_lengths, out);
    return (int)hipGetLastError();
  }
  if (n_voxels > 0x3fffffffL) return (int)hipErrorInvalidValue;
  const size_t need = ocrf_bev_pool_v2_workspace_bytes(c, n_points, n_voxels);
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) return (int)hipErrorInvalidValue;
  const Geometry vg = make_geometry(1, n_voxels);
  const PoolGeom pg = pool_geom(c);
  const size_t prep_bytes = prep_layout(vg, pg, n_points).total;
  int rc = launch_prep(c, n_intervals, n_points, nullptr, vg, ranks_bev, interval_starts, interval_lengths, workspace,
                       stream);
  if (rc != 0) return rc;
  return launch_tiles(c, n_points, nullptr, vg, 1, 2, depth, feat, ranks_depth, ranks_feat, workspace, out,
                      static_cast<char*>(workspace) + prep_bytes, stream);
}

size_t ocrf_bev_pool_v2_nchw_workspace_bytes(int c, int n_intervals, int n_points, int B, int Z, int Y, int X) {
  if (!vec_ok(c) || n_intervals < 0 || n_points < 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 0;
  const PoolGeom pg = pool_geom(c);
  return prep_layout(make_geometry((long)B * Z, (long)Y * X), pg, n_points).total + slab_bytes_of(c, pg, n_points);
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
  const Geometry vg = make_geometry((long)B * Z, (long)Y * X);
  if (vg.n_vox > 0x3fffffffL) return (int)hipErrorInvalidValue;
  const size_t need = ocrf_bev_pool_v2_nchw_workspace_bytes(c, n_intervals, n_points, B, Z, Y, X);
  if (!workspace || workspace_bytes < need || !aligned16(workspace) || !aligned16(feat)) return (int)hipErrorInvalidValue;
  if (n_intervals > 0 && n_points > 0 &&
      (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_starts || !interval_lengths))
    return (int)hipErrorInvalidValue;
  const size_t prep_bytes = prep_layout(vg, pool_geom(c), n_points).total;
  int rc = launch_prep(c, n_intervals, n_points, counts, vg, ranks_bev, interval_starts, interval_lengths, workspace,
                       stream);
  if (rc != 0) return rc;
  return launch_tiles(c, n_points, counts, vg, Z, layout, depth, feat, ranks_depth, ranks_feat, workspace, out,
                      static_cast<char*>(workspace) + prep_bytes, stream);
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
// Plans: for rank vectors that stay the same across calls (static calibration, ``accelerate``) the whole
// preparation — voxel table, unit list, XCD ranges — is kept; a call is then ONE launch.
// ---------------------------------------------------------------------------------------------
size_t ocrf_bev_pool_plan_bytes(int c, int n_points, int B, int Z, int Y, int X) {
  if (!vec_ok(c) || n_points <= 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 0;
  return prep_layout(make_geometry((long)B * Z, (long)Y * X), pool_geom(c), n_points).total;
}

int ocrf_bev_pool_plan_build(int c, int n_intervals, int n_points, const int* ranks_bev,
                             const int* interval_starts, const int* interval_lengths, int B, int Z, int Y, int X,
                             void* plan, size_t plan_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!vec_ok(c) || n_intervals <= 0 || n_points <= 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || !ranks_bev ||
      !interval_starts || !interval_lengths || !plan)
    return (int)hipErrorInvalidValue;
  const Geometry vg = make_geometry((long)B * Z, (long)Y * X);
  if (vg.n_vox > 0x3fffffffL) return (int)hipErrorInvalidValue;
  if (plan_bytes < ocrf_bev_pool_plan_bytes(c, n_points, B, Z, Y, X) || !aligned16(plan)) return (int)hipErrorInvalidValue;
  return launch_prep(c, n_intervals, n_points, nullptr, vg, ranks_bev, interval_starts, interval_lengths, plan, stream);
}

size_t ocrf_bev_pool_planned_workspace_bytes(int c, int n_points) {
  if (!vec_ok(c) || n_points <= 0) return 0;
  return slab_bytes_of(c, pool_geom(c), n_points);
}

int ocrf_bev_pool_v2_nchw_planned(int c, int n_points, const float* depth, const float* feat,
                                  const int* ranks_depth, const int* ranks_feat, void* plan, float* out,
                                  int B, int Z, int Y, int X, int layout, void* workspace,
                                  size_t workspace_bytes, size_t depth_bytes, size_t feat_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!vec_ok(c) || n_points <= 0 || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 ||
      (layout != 0 && layout != 1) || !out || !plan || !depth || !feat || !ranks_depth || !ranks_feat)
    return (int)hipErrorInvalidValue;
  // 32-bit byte offsets into the operands (bev_pool_cuda.cu:39-47 indexes with int): 2 GiB or more is refused HERE, at
  // the boundary a caller links against, not only in the Python wrapper
  if (depth_bytes >= (1ull << 31) || feat_bytes >= (1ull << 31) || (size_t)B * Z * Y * X * c * 4 >= (1ull << 31))
    return (int)hipErrorInvalidValue;
  const Geometry vg = make_geometry((long)B * Z, (long)Y * X);
  if (vg.n_vox > 0x3fffffffL) return (int)hipErrorInvalidValue;
  if (!workspace || workspace_bytes < ocrf_bev_pool_planned_workspace_bytes(c, n_points) || !aligned16(workspace) ||
      !aligned16(feat) || !aligned16(plan))
    return (int)hipErrorInvalidValue;
  return launch_tiles(c, n_points, nullptr, vg, Z, layout, depth, feat, ranks_depth, ranks_feat, plan, out, workspace,
                      stream, nullptr, feat_bytes);
}

// Diagnostic build of the pooling kernel that accumulates s_memtime per phase and unit into stamps[max_units][8]
// = {table + zero, staging, gather, combine, write-out, points} (never used by the product path).
int ocrf_diag_bev_pool_stamps(int c, int n_points, const float* depth, const float* feat, const int* ranks_depth,
                              const int* ranks_feat, void* plan, float* out, int B, int Z, int Y, int X, int layout,
                              void* workspace, unsigned long long* stamps, ocrf_stream_t stream_) {
  if (!vec_ok(c) || n_points <= 0 || !plan || !stamps) return (int)hipErrorInvalidValue;
  const Geometry vg = make_geometry((long)B * Z, (long)Y * X);
  return launch_tiles(c, n_points, nullptr, vg, Z, layout, depth, feat, ranks_depth, ranks_feat, plan, out, workspace,
                      static_cast<hipStream_t>(stream_), stamps);
}
int ocrf_bev_pool_max_units(int c, int n_points, int B, int Z, int Y, int X) {
  if (!vec_ok(c)) return 0;
  return max_units_of(make_geometry((long)B * Z, (long)Y * X), pool_geom(c), n_points);
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

// Index preparation of the two bev_pool_v2 calls on the device.
//
// LSS branch: frustum template -> ego frame -> voxel id per point -> points grouped by voxel
//   (mmdet3d/models/necks/view_transformer.py:108-147 get_lidar_coor, :197-255
//   voxel_pooling_prepare_v2).
// HT branch: pillar sample points -> every camera -> image cell per (camera, height, pillar)
//   (mmdet3d/models/necks/view_transformer_ocrf.py:687-740 get_sampling_point, :785-852
//   fast_sample_prepare).
//
// The reference spends ~20 elementwise passes, a boolean compaction and an argsort per branch and
// forward.  Here one kernel evaluates the per-point arithmetic — written as the same separate
// float32 multiplies and adds, k ascending, fp-contract off, IEEE division, so the voxel indices
// are bit-exact against vectors dumped from the reference — and
//   * LSS: the points are ordered by voxel with an LSD radix sort written for this key shape
//     (keys are voxel ids: 17-20 bits -> 2 passes of 9 bits up to 262 144 voxels, 3 beyond; stable,
//     so the order inside an interval is ascending point index — deterministic where the
//     reference's argsort is not); interval starts / lengths come from one binary search per voxel;
//   * HT: the key is the pillar itself and a pillar's candidates are its (camera, height) pairs, so
//     no sort is needed at all: validity bits per (camera, pillar), totals per pillar, one scan
//     over the pillars, emit in (camera, height) order.
// Every prefix sum is a single launch (decoupled look-back over ticket-ordered tiles): 18 launches per LSS
// preparation became 10, 7 per HT preparation 5.
// Tiny per-camera algebra (3x3 inverses and products) stays on the host, as the same torch calls the
// reference makes; the kernels take the resulting per-camera blocks.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ocrf_hip.h"
#include "radix_emit.h"

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 8;                     // keys per thread and radix pass
constexpr int kChunk = kBlock * kItems;       // 2048 keys per workgroup
constexpr int kRadixBits = 9;
constexpr int kBins = 1 << kRadixBits;        // 512

struct LssCam {       // 33 floats per camera-frame, see ocrf_hip.h
  float inv_post[9], combine[9], post_trans[3], trans[3], bda[9];
};

struct HtCam {        // 24 floats per camera-frame
  float l2i[12], aug[12];
};

// float -> int64 like x86 cvttss2si / CUDA: NaN and out-of-range give a value that fails the
// in-grid test (torch .long() on the reference's devices), never cell 0.
__device__ __forceinline__ long long trunc_ll(float x) {
  if (!(fabsf(x) < 9.0e18f)) return (long long)0x8000000000000000ull;
  return (long long)x;
}

// ---------------------------------------------------------------------------------------------
// LSS: voxel key per frustum point.  key = b*Z*Y*X + cz*Y*X + cy*X + cx, or n_vox_total if the
// point falls outside the grid.
// ---------------------------------------------------------------------------------------------
struct LssGrid {
  float lx, ly, lz, ix, iy, iz;
  int gx, gy, gz;
};

__device__ __forceinline__ unsigned lss_key_of(int p, int N, int DHW, const float* __restrict__ frustum,
                                               const LssCam* __restrict__ cams, const LssGrid& q, unsigned n_vox_total) {
  const float lx = q.lx, ly = q.ly, lz = q.lz, ix = q.ix, iy = q.iy, iz = q.iz;
  const int gx = q.gx, gy = q.gy, gz = q.gz;
  const int cam = p / DHW;                 // b*N + n
  const int cell_i = p - cam * DHW;        // (d*H + h)*W + w
  const LssCam& c = cams[cam];
  // frustum - post_trans, inv(post_rots) (view_transformer.py:130-133)
  const float fx = frustum[3 * cell_i] - c.post_trans[0];
  const float fy = frustum[3 * cell_i + 1] - c.post_trans[1];
  const float fz = frustum[3 * cell_i + 2] - c.post_trans[2];
  float x = c.inv_post[0] * fx + c.inv_post[1] * fy + c.inv_post[2] * fz;
  float y = c.inv_post[3] * fx + c.inv_post[4] * fy + c.inv_post[5] * fz;
  float z = c.inv_post[6] * fx + c.inv_post[7] * fy + c.inv_post[8] * fz;
  x = x * z;                                // un-project by depth (:138-139)
  y = y * z;
  float x2 = c.combine[0] * x + c.combine[1] * y + c.combine[2] * z;
  float y2 = c.combine[3] * x + c.combine[4] * y + c.combine[5] * z;
  float z2 = c.combine[6] * x + c.combine[7] * y + c.combine[8] * z;
  x2 = x2 + c.trans[0];
  y2 = y2 + c.trans[1];
  z2 = z2 + c.trans[2];
  const float ex = c.bda[0] * x2 + c.bda[1] * y2 + c.bda[2] * z2;
  const float ey = c.bda[3] * x2 + c.bda[4] * y2 + c.bda[5] * z2;
  const float ez = c.bda[6] * x2 + c.bda[7] * y2 + c.bda[8] * z2;
  // ((coor - lower) / interval).long(): truncation toward zero BEFORE the range test (:220-230)
  const long long cx = trunc_ll((ex - lx) / ix);
  const long long cy = trunc_ll((ey - ly) / iy);
  const long long cz = trunc_ll((ez - lz) / iz);
  unsigned key = n_vox_total;
  if (cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz) {
    const int b = cam / N;
    key = (unsigned)(((long long)b * gz + cz) * gy * gx + cy * gx + cx);
  }
  return key;
}

// The keys of one sort chunk (kChunk consecutive points, kItems per thread) AND the chunk's histogram of the first radix
// digit in one launch: the first pass of the sort used to read the 4 MB of keys back for it (one launch and ~ 6 us of
// the preparation's chain less).  table[d * n_wg + workgroup] as radix_hist_kernel leaves it.
__global__ __launch_bounds__(kBlock) void lss_keys_hist_kernel(
    int n_pts, int N, int DHW, const float* __restrict__ frustum, const LssCam* __restrict__ cams, LssGrid q,
    unsigned n_vox_total, unsigned* __restrict__ keys, int n_wg, int* __restrict__ table) {
  __shared__ int s_hist[kBins];
  for (int i = threadIdx.x; i < kBins; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * kChunk;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int p = base + it * kBlock + threadIdx.x;
    if (p < n_pts) {
      const unsigned key = lss_key_of(p, N, DHW, frustum, cams, q, n_vox_total);
      keys[p] = key;
      atomicAdd(&s_hist[key & (kBins - 1)], 1);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kBins; d += kBlock) table[(long)d * n_wg + blockIdx.x] = s_hist[d];
}

// ---------------------------------------------------------------------------------------------
// LSD radix sort, one 9-bit digit per pass: histogram -> exclusive scan of the [digit][workgroup]
// table -> stable scatter.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const unsigned* __restrict__ keys, int n, int shift,
                                                            int n_wg, int* __restrict__ table) {
  __shared__ int s_hist[kBins];
  for (int i = threadIdx.x; i < kBins; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * kChunk;
  // the thread's keys in ONE round trip (clamped addresses; `if (i < n) load` per key was a branch and a wait per key)
  unsigned key[kItems];
#pragma unroll
  for (int it = 0; it < kItems; ++it) key[it] = keys[min(base + it * kBlock + (int)threadIdx.x, n - 1)];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int i = base + it * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&s_hist[(key[it] >> shift) & (kBins - 1)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kBins; d += kBlock) table[(long)d * n_wg + blockIdx.x] = s_hist[d];
}

// exclusive prefix of v over the 256 threads of the workgroup (each thread of a scan owns 8 consecutive
// values, so a wave touches 2 KB of contiguous memory per load)
template <typename T>
__device__ __forceinline__ T block_exclusive(T v, T* s_wave, T* block_total) {
  // exclusive prefix of v over the 256 threads of the workgroup
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  T inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const T c = s_wave[w];
    if (w < wave) base += c;
    tot += c;
  }
  if (block_total) *block_total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// Single-launch prefix sums: decoupled look-back over TICKET-ordered tiles.  state[0] is the ticket
// counter, state[1 + t] the 64-bit word of tile t: bits 63:62 = 0 nothing yet | 1 the tile's own sum |
// 2 its inclusive prefix, bits 61:0 the value — one word, written and read with agent-scope atomics,
// so a reader never sees a value without its status.  A workgroup takes the next ticket before it
// does anything else: the tiles it may wait for hold smaller tickets, i.e. are already running —
// every wave reaches its exit whatever the scheduling order.  `state` must be zero before the launch.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kLbValue = (1ull << 62) - 1;

__device__ __forceinline__ void lb_publish(unsigned long long* state, int tile, unsigned status, unsigned long long v) {
  __hip_atomic_store(state + 1 + tile, ((unsigned long long)status << 62) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ticket of this workgroup (uniform); contains a barrier
__device__ __forceinline__ int lb_ticket(unsigned long long* state, int* s_tile) {
  if (threadIdx.x == 0) *s_tile = (int)atomicAdd(reinterpret_cast<unsigned*>(state), 1u);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*s_tile);      // known-uniform: what depends on it stays in SGPRs
}

template <int NW = kBlock / 64>
struct LbShared {
  unsigned long long inc[NW], none[NW], sum[NW], run;
};

// exclusive prefix of this tile given its own sum `tot` (uniform).  All 64 NW threads call; contains barriers.
// The look-back reads 64 NW predecessors per round (thread i reads tile hi - i): an agent-scope load is a
// round trip to memory, and while every tile still holds only its own sum a tile t needs ~t / (2 kBlock) rounds.
template <int NW>
__device__ __forceinline__ unsigned long long lb_tile_prefix(unsigned long long* state, int tile, unsigned long long tot,
                                                             LbShared<NW>* sh) {
  constexpr int kThreads = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    lb_publish(state, tile, tile == 0 ? 2u : 1u, tot);
    sh->run = 0;
  }
  int hi = tile - 1;
  int spins = 0;
  while (hi >= 0) {                                   // uniform: hi, and the decisions below, are the same everywhere
    const int j = hi - tid;
    const unsigned long long w = j >= 0 ? __hip_atomic_load(state + 1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : (2ull << 62);                 // before tile 0: inclusive prefix 0
    const unsigned status = (unsigned)(w >> 62);
    const unsigned long long inc = __ballot(status == 2), none = __ballot(status == 0);
    if (lane == 0) { sh->inc[wave] = inc; sh->none[wave] = none; }
    __syncthreads();
    // nearest predecessor with a full prefix, as a thread index; everything nearer must at least hold its sum
    int first = kThreads;
    bool wait = false;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      if (first < kThreads) break;
      const unsigned long long i = sh->inc[k], n = sh->none[k];
      const int f = i ? __ffsll((long long)i) - 1 : 64;
      const unsigned long long need = f >= 63 ? ~0ull : ((2ull << f) - 1ull);
      wait = wait || (n & need) != 0;
      if (f < 64) first = k * 64 + f;
    }
    __syncthreads();                                  // sh->inc / none are rewritten by the next round
    if (wait) {
      // Every tile this one waits for holds a smaller ticket and is running, so the wait ends within microseconds —
      // unless the state words are being overwritten, e.g. by a second call sharing this scratch on another stream.
      // A wave must not spin on a shared GPU for ever: after ~a second the kernel traps (the launch fails loudly).
      // A wave must not spin on a shared GPU for ever, and a trap would take the whole process down: after ~a second
      // the tile gives up, raises the error bit of the state block (bit 63 of word 0; the ticket counter is the low
      // half) and carries on with what it has — the call's last kernel turns the bit into counts = -1, which the
      // poolings read as "no points" and the host wrappers as an OcrfHipError.
      if (++spins > (1 << 20)) {
        if (tid == 0) atomicOr(state, 1ull << 63);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      continue;                                       // something nearer than `first` is unpublished: read again
    }
    unsigned long long v = tid <= first ? (w & kLbValue) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) sh->sum[wave] = v;
    __syncthreads();
    if (tid == 0) {
      unsigned long long r = sh->run;
#pragma unroll
      for (int k = 0; k < NW; ++k) r += sh->sum[k];
      sh->run = r;
    }
    if (first < kThreads) break;
    hi -= kThreads;
  }
  __syncthreads();
  const unsigned long long prefix = sh->run;
  if (tid == 0 && tile > 0) lb_publish(state, tile, 2u, prefix + tot);
  return prefix;
}

inline size_t lb_state_bytes(long n_tiles) { return ((size_t)n_tiles + 1) * 8; }

// in-place exclusive scan of n non-negative values (sums < 2^62), one launch of ceil(n / kChunk) workgroups
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_lookback_kernel(T* __restrict__ data, long n, T* __restrict__ total,
                                                               unsigned long long* __restrict__ state) {
  __shared__ T s_wave[kBlock / 64];
  __shared__ int s_tile;
  __shared__ LbShared<> s_lb;
  const int tile = lb_ticket(state, &s_tile);
  const long base = (long)tile * kChunk + (long)threadIdx.x * kItems;
  T v[kItems];
  T sum = 0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    v[i] = base + i < n ? data[base + i] : (T)0;
    sum += v[i];
  }
  T tot;
  const T local = block_exclusive<T>(sum, s_wave, &tot);
  const unsigned long long prefix = lb_tile_prefix(state, tile, (unsigned long long)tot, &s_lb);
  T run = (T)prefix + local;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
  if (total && tile == (int)gridDim.x - 1 && threadIdx.x == 0) *total = (T)prefix + tot;
}

template <typename T>
inline void scan_exclusive_lookback(T* data, long n, T* total, unsigned long long* zeroed_state, hipStream_t stream) {
  const int nb = (int)((n + kChunk - 1) / kChunk);
  ocrf::launch(OCRF_K_SCAN, scan_lookback_kernel<T>, dim3(nb), dim3(kBlock), 0, stream, data, n, total, zeroed_state);
}

// Stable scatter of one pass.  A wave owns 512 consecutive keys of the workgroup's chunk and walks
// them 64 at a time in order; the rank of a key among equal digits of its step is a ballot match,
// running per-digit positions live in LDS (LDS operations of one wave execute in order).
// EMIT (the LAST pass of the LSS preparation): the sorted (voxel, point) pairs leave as the three rank vectors at once —
// ranks_bev = key, ranks_depth = the point's flat index (view_transformer.py:232-236), ranks_feat = (b*N + n)*H*W + h*W + w
// — instead of as keys / values that one more launch turned into them.
// EMIT 2 (the last pass of a render-plan build's sort): besides the sorted keys, the plan's per-view list arrays
// (ocrf::RadixPlanEmit) — the gather launch that read the pairs back to write them is gone.
struct ScatterExtra {
  int* ranks_feat;
  int DHW, HW;
  ocrf::RadixPlanEmit plan;
};
template <bool IMPLICIT_VALS, int EMIT = 0>
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(
    const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift, int n_wg,
    const int* __restrict__ table, unsigned* __restrict__ keys_out, int* __restrict__ vals_out, ScatterExtra x) {
  __shared__ int s_pos[kBlock / 64][kBins];
  const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
  for (int i = tid; i < (kBlock / 64) * kBins; i += kBlock) (&s_pos[0][0])[i] = 0;
  __syncthreads();
  const int wbase = blockIdx.x * kChunk + wave * (kChunk / (kBlock / 64));
  unsigned key[kItems];
  int val[kItems];
  bool ok[kItems];
  // keys (and values) of the thread in ONE round trip, at clamped addresses
#pragma unroll
  for (int s = 0; s < kItems; ++s) {
    const int i = wbase + s * 64 + lane;
    ok[s] = i < n;
    key[s] = keys_in[min(i, n - 1)];
    val[s] = IMPLICIT_VALS ? i : vals_in[min(i, n - 1)];
  }
  // EMIT 2: what the value points at, requested now (a gather per key inside the ordered walk below would be a round
  // trip per step)
  int e_id[EMIT == 2 ? kItems : 1];
  float2 e_pix[EMIT == 2 ? kItems : 1];
  if constexpr (EMIT == 2) {
#pragma unroll
    for (int s = 0; s < kItems; ++s) {
      const int e = val[s];
      e_id[s] = x.plan.rec_id[e];
      const float4 q1 = x.plan.e_q1[e];
      e_pix[s] = make_float2(q1.z, q1.w);
    }
  }
#pragma unroll
  for (int s = 0; s < kItems; ++s) {
    if (!ok[s]) key[s] = 0u;
    if (ok[s]) atomicAdd(&s_pos[wave][(key[s] >> shift) & (kBins - 1)], 1);
  }
  __syncthreads();
  // per digit: global base of this workgroup, then the waves in order
  for (int d = tid; d < kBins; d += kBlock) {
    int run = table[(long)d * n_wg + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      const int c = s_pos[w][d];
      s_pos[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kItems; ++s) {
    const int digit = (int)((key[s] >> shift) & (kBins - 1));
    unsigned long long same = __ballot(ok[s]);
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const unsigned long long m = __ballot((digit >> b) & 1);
      same &= ((digit >> b) & 1) ? m : ~m;
    }
    if (ok[s]) {
      const int rank = __popcll(same & ((1ull << lane) - 1ull));
      const int pos = s_pos[wave][digit] + rank;
      keys_out[pos] = key[s];
      if constexpr (EMIT != 2) vals_out[pos] = val[s];
      if constexpr (EMIT == 1) x.ranks_feat[pos] = (val[s] / x.DHW) * x.HW + val[s] % x.HW;
      if constexpr (EMIT == 2) {
        x.plan.s_e[pos] = (unsigned)val[s];
        x.plan.s_id[pos] = (unsigned)e_id[s];
        x.plan.s_key[pos] = (key[s] & x.plan.depth_mask) + x.plan.key_base;
        x.plan.s_pix[pos] = e_pix[s];
      }
      if (rank == 0) s_pos[wave][digit] = pos + __popcll(same);      // leader advances the digit
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// lower bounds, non-empty flags, their prefix sum and the interval vectors in ONE launch (one voxel per thread,
// look-back over the tiles): starts[k] / lengths[k] of the k-th non-empty voxel, counts = {points, intervals}.
__global__ __launch_bounds__(kBlock) void lss_intervals_kernel(const unsigned* __restrict__ sorted, int n,
                                                               unsigned n_vox_total, int* __restrict__ starts,
                                                               int* __restrict__ lengths, int* __restrict__ counts,
                                                               unsigned long long* __restrict__ state,
                                                               const unsigned long long* __restrict__ all_states,
                                                               int st_words, int n_states) {
  __shared__ int s_lo[kBlock + 1];
  __shared__ int s_wave[kBlock / 64];
  __shared__ int s_tile;
  __shared__ LbShared<> s_lb;
  const int tile = lb_ticket(state, &s_tile);
  const unsigned v = (unsigned)tile * kBlock + threadIdx.x;
  auto lower_bound = [&](unsigned key) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  s_lo[threadIdx.x] = lower_bound(min(v, n_vox_total));
  if (threadIdx.x == kBlock - 1) s_lo[kBlock] = lower_bound(min(v + 1, n_vox_total));
  __syncthreads();
  const int lo = s_lo[threadIdx.x];
  const int len = v < n_vox_total ? s_lo[threadIdx.x + 1] - lo : 0;
  int tot;
  const int local = block_exclusive<int>(len > 0 ? 1 : 0, s_wave, &tot);
  const unsigned long long prefix = lb_tile_prefix(state, tile, (unsigned long long)tot, &s_lb);
  if (len > 0) {
    const int k = (int)prefix + local;
    starts[k] = lo;
    lengths[k] = len;
  }
  if (tile == (int)gridDim.x - 1 && threadIdx.x == 0) {
    // (the tile with the last ticket has every other tile's prefix behind it: no look-back of this call is still waiting)
    bool failed = false;                           // a look-back scan of this call gave up: say so in counts
    for (int k = 0; k < n_states; ++k) failed = failed || (all_states[(size_t)k * st_words] >> 63);
    counts[0] = failed ? -1 : lower_bound(n_vox_total);          // keys of dropped points are >= n_vox_total
    counts[1] = failed ? -1 : (int)prefix + tot;
  }
}

// ---------------------------------------------------------------------------------------------
// HT: one thread per (camera-frame, pillar); candidates in (camera, height) order.
// ---------------------------------------------------------------------------------------------
struct HtParams {
  int B, N, Z, Nq, Wf, Hf, D;
  float sx, ox, sy, oy, sz, oz;      // pc_range scale / offset (view_transformer_ocrf.py:690-692)
  float w_in, h_in, d0, dspan;       // image size, depth_range[0], depth_range[1]-depth_range[0]
};

// Lanes run over consecutive pillars of ONE camera-frame (blockIdx.y), so the reads of the pillar
// template are coalesced and the cheap rejections (behind the camera, out of depth range) are
// wave-coherent: a wave of pillars the camera does not see retires after a few instructions.
// The tests are evaluated cheapest first; the mask is their conjunction, the order is free.
__device__ __forceinline__ bool ht_project(const HtParams& q, const HtCam& c, float x, float y, float zz,
                                           float* u_out, float* v_out, float* d_out) {
  const float eps = 1e-5f;
  // lidar2img rows (the x, y products first, k ascending as in the reference), perspective divide,
  // image augmentation, normalisation (view_transformer_ocrf.py:700-735)
  const float cz = c.l2i[8] * x + c.l2i[9] * y + c.l2i[10] * zz + c.l2i[11] * 1.0f;
  if (!(cz > eps)) return false;
  const float d = (cz - q.d0) / q.dspan;
  if (!((d > 0.0f) && (d < 1.0f))) return false;
  const float cx = c.l2i[0] * x + c.l2i[1] * y + c.l2i[2] * zz + c.l2i[3] * 1.0f;
  const float cy = c.l2i[4] * x + c.l2i[5] * y + c.l2i[6] * zz + c.l2i[7] * 1.0f;
  const float den = fmaxf(cz, eps);
  const float u0 = cx / den, v0 = cy / den;
  float u = c.aug[0] * u0 + c.aug[1] * v0 + c.aug[2] * cz + c.aug[3] * 1.0f;
  float v = c.aug[4] * u0 + c.aug[5] * v0 + c.aug[6] * cz + c.aug[7] * 1.0f;
  u = u / q.w_in;
  v = v / q.h_in;
  *u_out = u; *v_out = v; *d_out = d;
  return (u > 0.0f) && (u < 1.0f) && (v > 0.0f) && (v < 1.0f);
}

// valid_bits[cam][pillar]: bit z = sample (cam, z, pillar) passes the mask
__global__ __launch_bounds__(kBlock) void ht_valid_kernel(HtParams q, const float* __restrict__ ref /*(Z,Nq,3) normalised*/,
                                                          const HtCam* __restrict__ cams, unsigned* __restrict__ valid_bits,
                                                          unsigned long long* __restrict__ scan_state, int scan_state_words) {
  const int pil = blockIdx.x * kBlock + threadIdx.x;
  const int cam = blockIdx.y;                  // b*N + n
  if (blockIdx.x == 0 && cam == 0)             // the look-back state of the scan two launches later
    for (int i = threadIdx.x; i < scan_state_words; i += kBlock) scan_state[i] = 0ull;
  if (pil >= q.Nq) return;
  const HtCam& c = cams[cam];
  unsigned valid = 0;
  for (int z = 0; z < q.Z; ++z) {
    const float* r = ref + ((long)z * q.Nq + pil) * 3;
    float u, v, d;
    if (ht_project(q, c, r[0] * q.sx + q.ox, r[1] * q.sy + q.oy, r[2] * q.sz + q.oz, &u, &v, &d)) valid |= 1u << z;
  }
  valid_bits[(long)cam * q.Nq + pil] = valid;
}

// Pillar projections for the colour / alpha sampling and retain_valid_pixels (view_transformer_ocrf.py:
// 1057-1066): pixel coordinates (u_norm * W_in, v_norm * H_in) and the validity mask of EVERY
// (camera, height, pillar) sample, plus the metric voxel centres — the same arithmetic as ht_project,
// without the early exits (the reference materialises the coordinates of masked samples too).
__global__ __launch_bounds__(kBlock) void ht_project_kernel(HtParams q, const float* __restrict__ ref, const HtCam* __restrict__ cams,
                                                            float2* __restrict__ pix, unsigned char* __restrict__ mask,
                                                            float* __restrict__ voxel) {
  const int pil = blockIdx.x * kBlock + threadIdx.x;
  const int cam = blockIdx.y;                  // b*N + n
  if (pil >= q.Nq) return;
  const HtCam& c = cams[cam];
  const float eps = 1e-5f;
  for (int z = 0; z < q.Z; ++z) {
    const float* r = ref + ((long)z * q.Nq + pil) * 3;
    const float x = r[0] * q.sx + q.ox, y = r[1] * q.sy + q.oy, zz = r[2] * q.sz + q.oz;
    const float cx = c.l2i[0] * x + c.l2i[1] * y + c.l2i[2] * zz + c.l2i[3] * 1.0f;
    const float cy = c.l2i[4] * x + c.l2i[5] * y + c.l2i[6] * zz + c.l2i[7] * 1.0f;
    const float cz = c.l2i[8] * x + c.l2i[9] * y + c.l2i[10] * zz + c.l2i[11] * 1.0f;
    const float den = fmaxf(cz, eps);
    const float u0 = cx / den, v0 = cy / den;
    float u = c.aug[0] * u0 + c.aug[1] * v0 + c.aug[2] * cz + c.aug[3] * 1.0f;
    float v = c.aug[4] * u0 + c.aug[5] * v0 + c.aug[6] * cz + c.aug[7] * 1.0f;
    u = u / q.w_in;
    v = v / q.h_in;
    const float d = (cz - q.d0) / q.dspan;
    const bool ok = (cz > eps) && (u > 0.0f) && (u < 1.0f) && (v > 0.0f) && (v < 1.0f) && (d > 0.0f) && (d < 1.0f);
    const long o = ((long)cam * q.Z + z) * q.Nq + pil;
    pix[o] = make_float2(u * q.w_in, v * q.h_in);
    mask[o] = ok ? 1 : 0;
    if (voxel && cam % q.N == 0) {
      float* vo = voxel + (((long)(cam / q.N) * q.Z + z) * q.Nq + pil) * 3;
      vo[0] = x; vo[1] = y; vo[2] = zz;
    }
  }
}

// per (b, pillar): number of kept samples over its cameras, packed with the non-empty flag so that
// ONE scan gives both prefix sums
__global__ __launch_bounds__(kBlock) void ht_pillar_totals_kernel(HtParams q, const unsigned* __restrict__ valid_bits,
                                                                  long long* __restrict__ cnt_flag) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= q.B * q.Nq) return;
  const int b = t / q.Nq, pil = t - b * q.Nq;
  int total = 0;
  for (int n = 0; n < q.N; ++n) total += __popc(valid_bits[((long)b * q.N + n) * q.Nq + pil]);
  cnt_flag[t] = ((long long)total << 31) | (total > 0 ? 1ll : 0ll);      // both sums stay below 2^31: one 62-bit value
}

// emit in (camera, height) order inside a pillar = the reference's flattening order under a stable sort
__global__ __launch_bounds__(kBlock) void ht_emit_kernel(HtParams q, const float* __restrict__ ref, const HtCam* __restrict__ cams,
                                                         const unsigned* __restrict__ valid_bits,
                                                         const long long* __restrict__ cnt_flag /*scanned*/,
                                                         int* __restrict__ ranks_bev, int* __restrict__ ranks_depth,
                                                         int* __restrict__ ranks_feat, int* __restrict__ starts,
                                                         int* __restrict__ lengths) {
  const int pil = blockIdx.x * kBlock + threadIdx.x;
  const int cam = blockIdx.y;
  if (pil >= q.Nq) return;
  const int b = cam / q.N, n = cam - b * q.N;
  const int t = b * q.Nq + pil;
  const unsigned valid = valid_bits[(long)cam * q.Nq + pil];
  if (n != 0 && valid == 0) return;
  const long long cf = cnt_flag[t];
  int out = (int)(cf >> 31);
  if (n == 0) {
    int total = 0;
    for (int m = 0; m < q.N; ++m) total += __popc(valid_bits[((long)b * q.N + m) * q.Nq + pil]);
    if (total > 0) {
      const int k = (int)(cf & 0x7FFFFFFFll);
      starts[k] = out;
      lengths[k] = total;
    }
    if (valid == 0) return;
  }
  for (int m = 0; m < n; ++m) out += __popc(valid_bits[((long)b * q.N + m) * q.Nq + pil]);
  const HtCam& c = cams[cam];
  for (int z = 0; z < q.Z; ++z) {
    if (!((valid >> z) & 1u)) continue;
    const float* r = ref + ((long)z * q.Nq + pil) * 3;
    float u, v, d;
    ht_project(q, c, r[0] * q.sx + q.ox, r[1] * q.sy + q.oy, r[2] * q.sz + q.oz, &u, &v, &d);
    // (coor * (W,H,D)).round().long(), clamped to the map (:806-813); rintf = half-to-even
    long long iw = (long long)rintf(u * (float)q.Wf);
    long long ih = (long long)rintf(v * (float)q.Hf);
    long long id = (long long)rintf(d * (float)q.D);
    iw = min(max(iw, 0ll), (long long)q.Wf - 1);
    ih = min(max(ih, 0ll), (long long)q.Hf - 1);
    id = min(max(id, 0ll), (long long)q.D - 1);
    const long long hw = (long long)q.Wf * q.Hf;
    long long rd = (long long)cam * (q.D * hw) + id * hw + ih * q.Wf + iw;
    long long rf = (long long)cam * hw + ih * q.Wf + iw;
    rd = min(max(rd, 0ll), (long long)q.B * q.N * q.D * hw - 1);
    rf = min(max(rf, 0ll), (long long)q.B * q.N * hw - 1);
    ranks_bev[out] = t;                     // b*Nq + pillar
    ranks_depth[out] = (int)rd;
    ranks_feat[out] = (int)rf;
    ++out;
  }
}

__global__ void ht_counts_kernel(const long long* __restrict__ total, int* __restrict__ counts,
                                 const unsigned long long* __restrict__ state) {
  const bool failed = (state[0] >> 63) != 0;             // the look-back scan gave up (lb_tile_prefix)
  counts[0] = failed ? -1 : (int)(*total >> 31);
  counts[1] = failed ? -1 : (int)(*total & 0x7FFFFFFFll);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline int radix_passes(unsigned n_vox_total) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)n_vox_total) ++bits;
  return (bits + kRadixBits - 1) / kRadixBits;
}

struct LssWs {
  size_t keys_a, keys_b, vals_a, vals_b, table, state, state_bytes, bytes;
  long state_tiles_table, state_tiles_iv;
};

inline void lss_layout(long n_pts, unsigned n_vox_total, LssWs* w) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  const long n_wg = (n_pts + kChunk - 1) / kChunk;
  w->keys_a = take((size_t)n_pts * 4);
  w->keys_b = take((size_t)n_pts * 4);
  w->vals_a = take((size_t)n_pts * 4);
  w->vals_b = take((size_t)n_pts * 4);
  w->table = take((size_t)kBins * n_wg * 4);
  // look-back states: one per radix pass (<= 4) + the interval kernel, zero-filled by ONE launch per call
  w->state_tiles_table = ((long)kBins * n_wg + kChunk - 1) / kChunk;
  w->state_tiles_iv = ((long)n_vox_total + 1 + kBlock - 1) / kBlock;
  w->state = take(4 * lb_state_bytes(w->state_tiles_table) + lb_state_bytes(w->state_tiles_iv));
  w->state_bytes = off - w->state;
  w->bytes = off;
}

struct SortWs {
  size_t keys_b, vals_a, vals_b, table, state, state_bytes, bytes;
  long state_tiles;
};

inline void sort_layout(long n, SortWs* w) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  const long n_wg = (n + kChunk - 1) / kChunk;
  w->keys_b = take((size_t)n * 4);
  w->vals_a = take((size_t)n * 4);
  w->vals_b = take((size_t)n * 4);
  w->table = take((size_t)kBins * n_wg * 4);
  w->state_tiles = ((long)kBins * n_wg + kChunk - 1) / kChunk;
  w->state = take(4 * lb_state_bytes(w->state_tiles));
  w->state_bytes = off - w->state;
  w->bytes = off;
}

}  // namespace

namespace ocrf {

// The radix sort of the LSS preparation as a service for the other translation units (raster_plan.hip sorts a
// view's Gaussians by depth bits once per plan): stable, ids = original positions.
size_t radix_sort_ids_bytes(int n) {
  if (n <= 0) return 0;
  SortWs w;
  sort_layout(n, &w);
  return w.bytes;
}

hipError_t radix_sort_ids(unsigned* keys, int n, int key_bits, void* scratch, size_t scratch_bytes,
                          const unsigned** sorted_keys, const int** sorted_ids, hipStream_t stream,
                          const RadixPlanEmit* plan_emit) {
  if (n <= 0 || key_bits <= 0 || key_bits > 32 || !keys || !scratch) return hipErrorInvalidValue;
  SortWs w;
  sort_layout(n, &w);
  if (scratch_bytes < w.bytes) return hipErrorInvalidValue;
  char* base = static_cast<char*>(scratch);
  unsigned* kbuf[2] = {keys, reinterpret_cast<unsigned*>(base + w.keys_b)};
  int* vbuf[2] = {reinterpret_cast<int*>(base + w.vals_a), reinterpret_cast<int*>(base + w.vals_b)};
  int* table = reinterpret_cast<int*>(base + w.table);
  auto* state = reinterpret_cast<unsigned long long*>(base + w.state);
  const size_t st_words = lb_state_bytes(w.state_tiles) / 8;
  hipError_t e = zero_async(state, w.state_bytes, stream);
  if (e != hipSuccess) return e;
  const int n_wg = (n + kChunk - 1) / kChunk;
  const int passes = (key_bits + kRadixBits - 1) / kRadixBits;      // <= 4
  int cur = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * kRadixBits;
    launch(OCRF_K_RADIX_HIST, radix_hist_kernel, dim3(n_wg), dim3(kBlock), 0, stream,
           static_cast<const unsigned*>(kbuf[cur]), n, shift, n_wg, table);
    scan_exclusive_lookback(table, (long)kBins * n_wg, (int*)nullptr, state + pass * st_words, stream);
    ScatterExtra x{};
    const bool emit = plan_emit != nullptr && pass == passes - 1 && pass > 0;
    if (emit) x.plan = *plan_emit;
    if (pass == 0)
      launch(OCRF_K_RADIX_SCATTER, radix_scatter_kernel<true>, dim3(n_wg), dim3(kBlock), 0, stream,
             static_cast<const unsigned*>(kbuf[cur]), static_cast<const int*>(nullptr), n, shift, n_wg,
             static_cast<const int*>(table), kbuf[cur ^ 1], vbuf[cur ^ 1], x);
    else if (emit)        // the consumer's arrays instead of the values (the sorted keys are still written)
      launch(OCRF_K_RADIX_SCATTER, radix_scatter_kernel<false, 2>, dim3(n_wg), dim3(kBlock), 0, stream,
             static_cast<const unsigned*>(kbuf[cur]), static_cast<const int*>(vbuf[cur]), n, shift, n_wg,
             static_cast<const int*>(table), kbuf[cur ^ 1], vbuf[cur ^ 1], x);
    else
      launch(OCRF_K_RADIX_SCATTER, radix_scatter_kernel<false>, dim3(n_wg), dim3(kBlock), 0, stream,
             static_cast<const unsigned*>(kbuf[cur]), static_cast<const int*>(vbuf[cur]), n, shift, n_wg,
             static_cast<const int*>(table), kbuf[cur ^ 1], vbuf[cur ^ 1], x);
    cur ^= 1;
  }
  *sorted_keys = kbuf[cur];
  *sorted_ids = (plan_emit != nullptr && passes > 1) ? nullptr : vbuf[cur];      // (not written with an emit)
  return hipGetLastError();
}

// where a radix_sort_ids(.., n, key_bits, scratch, ..) call leaves the state blocks of its look-back scans (one per
// pass): bit 63 of a block's first word says that scan gave up (the caller's last kernel checks them on the device)
void radix_sort_states(void* scratch, int n, int key_bits, const unsigned long long** states, int* n_states,
                       long* stride_words) {
  SortWs w;
  sort_layout(n, &w);
  *states = reinterpret_cast<const unsigned long long*>(static_cast<char*>(scratch) + w.state);
  *n_states = (key_bits + kRadixBits - 1) / kRadixBits;
  *stride_words = (long)(lb_state_bytes(w.state_tiles) / 8);
}

// in-place exclusive prefix sum of n non-negative ints (one look-back launch + the zero-fill of its state)
size_t exclusive_scan_bytes(long n) { return n > 0 ? align_up(lb_state_bytes((n + kChunk - 1) / kChunk), 256) : 0; }

hipError_t exclusive_scan_ints(int* data, long n, int* total, void* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (n <= 0 || !data || !scratch || scratch_bytes < exclusive_scan_bytes(n)) return hipErrorInvalidValue;
  hipError_t e = zero_async(scratch, exclusive_scan_bytes(n), stream);
  if (e != hipSuccess) return e;
  scan_exclusive_lookback<int>(data, n, total, static_cast<unsigned long long*>(scratch), stream);
  return hipGetLastError();
}

}  // namespace ocrf

extern "C" {

size_t ocrf_lss_prepare_workspace_bytes(int B, int N, int D, int H, int W, int gx, int gy, int gz) {
  if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || gx <= 0 || gy <= 0 || gz <= 0) return 0;
  LssWs w;
  lss_layout((long)B * N * D * H * W, (unsigned)((long)B * gz * gy * gx), &w);
  return w.bytes;
}

int ocrf_lss_prepare(int B, int N, int D, int H, int W, const float* frustum, const float* cams,
                     const float* grid_lower, const float* grid_interval, int gx, int gy, int gz,
                     int* ranks_bev, int* ranks_depth, int* ranks_feat, int* interval_starts,
                     int* interval_lengths, int* counts, void* workspace, size_t workspace_bytes,
                     ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || gx <= 0 || gy <= 0 || gz <= 0 || !frustum || !cams ||
      !grid_lower || !grid_interval || !ranks_bev || !ranks_depth || !ranks_feat || !interval_starts ||
      !interval_lengths || !counts)
    return (int)hipErrorInvalidValue;
  const long n_pts_l = (long)B * N * D * H * W;
  const long n_vox_l = (long)B * gz * gy * gx;
  if (n_pts_l >= (1l << 31) - kChunk || n_vox_l >= (1l << 31) - 1) return (int)hipErrorInvalidValue;
  const int n_pts = (int)n_pts_l;
  const unsigned n_vox_total = (unsigned)n_vox_l;
  LssWs w;
  lss_layout(n_pts, n_vox_total, &w);
  if (!workspace || workspace_bytes < w.bytes) return (int)hipErrorInvalidValue;
  char* base = static_cast<char*>(workspace);
  unsigned* keys[2] = {reinterpret_cast<unsigned*>(base + w.keys_a), reinterpret_cast<unsigned*>(base + w.keys_b)};
  int* vals[2] = {reinterpret_cast<int*>(base + w.vals_a), reinterpret_cast<int*>(base + w.vals_b)};
  int* table = reinterpret_cast<int*>(base + w.table);
  const int n_wg = (n_pts + kChunk - 1) / kChunk;
  // the grid constants are read on the host side of the ABI (6 floats); they are host memory
  const float lx = grid_lower[0], ly = grid_lower[1], lz = grid_lower[2];
  const float ix = grid_interval[0], iy = grid_interval[1], iz = grid_interval[2];

  auto* state = reinterpret_cast<unsigned long long*>(base + w.state);
  const size_t st_words = lb_state_bytes(w.state_tiles_table) / 8;
  hipError_t ze = ocrf::zero_async(state, w.state_bytes, stream);
  if (ze != hipSuccess) return (int)ze;
  const int passes = radix_passes(n_vox_total);
  if (passes > 4) return (int)hipErrorInvalidValue;
  // keys + the first pass's histogram in one launch; the last pass scatters straight into the rank vectors
  const LssGrid grid{lx, ly, lz, ix, iy, iz, gx, gy, gz};
  ocrf::launch(OCRF_K_LSS_KEYS, lss_keys_hist_kernel, dim3(n_wg), dim3(kBlock), 0, stream, n_pts, N, D * H * W, frustum,
               reinterpret_cast<const LssCam*>(cams), grid, n_vox_total, keys[0], n_wg, table);
  int cur = 0;
  unsigned* const rb_keys = reinterpret_cast<unsigned*>(ranks_bev);
  const int DHW = D * H * W, HW = H * W;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * kRadixBits;
    const bool last = pass == passes - 1;
    if (pass > 0)
      ocrf::launch(OCRF_K_RADIX_HIST, radix_hist_kernel, dim3(n_wg), dim3(kBlock), 0, stream,
                   static_cast<const unsigned*>(keys[cur]), n_pts, shift, n_wg, table);
    scan_exclusive_lookback(table, (long)kBins * n_wg, (int*)nullptr, state + pass * st_words, stream);
    ScatterExtra sx{};
    sx.ranks_feat = ranks_feat; sx.DHW = DHW; sx.HW = HW;
#define OCRF_LSS_SCATTER(IMPL, EMIT, KOUT, VOUT)                                                                        \
  ocrf::launch(OCRF_K_RADIX_SCATTER, radix_scatter_kernel<IMPL, EMIT>, dim3(n_wg), dim3(kBlock), 0, stream,             \
               static_cast<const unsigned*>(keys[cur]), static_cast<const int*>(IMPL ? nullptr : vals[cur]), n_pts,     \
               shift, n_wg, static_cast<const int*>(table), KOUT, VOUT, sx)
    if (pass == 0 && last) OCRF_LSS_SCATTER(true, 1, rb_keys, ranks_depth);
    else if (pass == 0) OCRF_LSS_SCATTER(true, 0, keys[cur ^ 1], vals[cur ^ 1]);
    else if (last) OCRF_LSS_SCATTER(false, 1, rb_keys, ranks_depth);
    else OCRF_LSS_SCATTER(false, 0, keys[cur ^ 1], vals[cur ^ 1]);
#undef OCRF_LSS_SCATTER
    cur ^= 1;
  }
  const int vgrid = (int)((n_vox_total + 1 + kBlock - 1) / kBlock);
  // the interval vectors from the sorted keys (= ranks_bev); its last tile also turns a look-back scan that gave up —
  // any of the sort's or its own — into counts = -1
  ocrf::launch(OCRF_K_LSS_BOUNDS, lss_intervals_kernel, dim3(vgrid), dim3(kBlock), 0, stream,
               static_cast<const unsigned*>(rb_keys), n_pts, n_vox_total, interval_starts, interval_lengths, counts,
               state + 4 * st_words, static_cast<const unsigned long long*>(state), (int)st_words, 5);
  return (int)hipGetLastError();
}

int ocrf_ht_project(int B, int N, int Z, int n_pillars, const float* ref_points, const float* cams,
                    const float* pc_range, float w_in, float h_in, float depth0, float depth1, float* pix,
                    unsigned char* mask, float* voxel, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B <= 0 || N <= 0 || Z <= 0 || B * N > 65535 || n_pillars <= 0 || !ref_points || !cams || !pc_range || !pix || !mask)
    return (int)hipErrorInvalidValue;
  HtParams q;
  q.B = B; q.N = N; q.Z = Z; q.Nq = n_pillars; q.Wf = 0; q.Hf = 0; q.D = 0;
  q.sx = (float)((double)pc_range[3] - (double)pc_range[0]); q.ox = pc_range[0];
  q.sy = (float)((double)pc_range[4] - (double)pc_range[1]); q.oy = pc_range[1];
  q.sz = (float)((double)pc_range[5] - (double)pc_range[2]); q.oz = pc_range[2];
  q.w_in = w_in; q.h_in = h_in; q.d0 = depth0; q.dspan = (float)((double)depth1 - (double)depth0);
  ocrf::launch(OCRF_K_HT_PROJECT, ht_project_kernel, dim3((n_pillars + kBlock - 1) / kBlock, B * N), dim3(kBlock), 0, stream,
               q, ref_points, reinterpret_cast<const HtCam*>(cams), reinterpret_cast<float2*>(pix), mask, voxel);
  return (int)hipGetLastError();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Per-forward calibration algebra on the device: the B*N tiny matrices of get_lidar_coor
// (view_transformer.py:128-146), get_projection (view_transformer_ocrf.py:675-685) and the render camera
// (view_transformer_ocrf.py:1135-1152 with data_utils.py:703-733) from the calibration tensors where they
// already are.  One thread per camera-frame; 3x3 inverses by cofactors in double precision, every product in
// double, ONE rounding to float at the end (the reference's host path rounds after every float32 LAPACK /
// matmul step; a CUDA run of the reference rounds differently again: the values agree to ~1 ulp).
//   lss (B*N,33): inv(post_rots) | rots inv(K) | post_trans | trans | bda
//   ht  (B*N,24): lidar2img = [K inv(rots) inv(bda) | -K inv(rots) trans] | img_aug = [post_rots | post_trans]
//   cam (B*N,36): world_view^T | full_proj^T | tanfovx | tanfovy | focal_x | focal_y, the reference's quirks
//                 kept: intrinsics as given with the network-input viewport, c2w's R and t fed where a
//                 world->view rotation / translation are expected (getWorld2View2 inverts twice: identity).
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void inv3(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double r = 1.0 / det;
  o[0] = A * r; o[1] = (c * h - b * i) * r; o[2] = (b * f - c * e) * r;
  o[3] = B * r; o[4] = (a * i - c * g) * r; o[5] = (c * d - a * f) * r;
  o[6] = C * r; o[7] = (b * g - a * h) * r; o[8] = (a * e - b * d) * r;
}
__device__ __forceinline__ void mul3(const double* x, const double* y, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[3 * r + c] = x[3 * r] * y[c] + x[3 * r + 1] * y[3 + c] + x[3 * r + 2] * y[6 + c];
}

__global__ __launch_bounds__(64) void geom_blocks_kernel(
    int B, int N, const float* __restrict__ rots, const float* __restrict__ trans, const float* __restrict__ intrins,
    const float* __restrict__ post_rots, const float* __restrict__ post_trans, const float* __restrict__ bda,
    const float* __restrict__ c2w, int H_in, int W_in, float znear, float zfar, float* __restrict__ lss,
    float* __restrict__ ht, float* __restrict__ cam) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N;
  double R[9], K[9], Pr[9], Bd[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    R[k] = rots[9 * i + k]; K[k] = intrins[9 * i + k]; Pr[k] = post_rots[9 * i + k]; Bd[k] = bda[9 * b + k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = trans[3 * i + k];
  double iPr[9], iK[9], iR[9], iB[9], comb[9], kr[9], l2r[9];
  inv3(Pr, iPr); inv3(K, iK); inv3(R, iR); inv3(Bd, iB);
  mul3(R, iK, comb);
  mul3(K, iR, kr);
  mul3(kr, iB, l2r);
  if (lss) {
    float* o = lss + 33 * i;
#pragma unroll
    for (int k = 0; k < 9; ++k) { o[k] = (float)iPr[k]; o[9 + k] = (float)comb[k]; o[24 + k] = bda[9 * b + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[18 + k] = post_trans[3 * i + k]; o[21 + k] = trans[3 * i + k]; }
  }
  if (ht) {
    float* o = ht + 24 * i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { o[4 * r + c] = (float)l2r[3 * r + c]; o[12 + 4 * r + c] = post_rots[9 * i + 3 * r + c]; }
      o[4 * r + 3] = (float)(-(kr[3 * r] * t[0] + kr[3 * r + 1] * t[1] + kr[3 * r + 2] * t[2]));
      o[12 + 4 * r + 3] = post_trans[3 * i + r];
    }
  }
  if (cam && c2w) {
    const float* M = c2w + 16 * i;
    float* o = cam + 36 * i;
    // world_view_transform = [[R^T, t], [0, 1]]^T with R = c2w[:3,:3], t = c2w[:3,3] (row-major, transposed)
    float wv[16];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) wv[4 * r + c] = M[4 * r + c];
      wv[4 * r + 3] = 0.f;
      wv[12 + r] = M[4 * r + 3];
    }
    wv[15] = 1.f;
    // getProjectionMatrix (data_utils.py:716-733) in double on the float32 intrinsics, stored transposed
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], zn = znear, zf = zfar, w = W_in, h = H_in;
    const double nfx = zn / fx, nfy = zn / fy;
    const double left = -(w - cx) * nfx, right = cx * nfx, bottom = (cy - h) * nfy, top = cy * nfy;
    float P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = 0.f;
    P[0] = (float)(2.0 * zn / (right - left));           // P[0][0]
    P[5] = (float)(2.0 * zn / (top - bottom));           // P[1][1]
    P[8] = (float)((right + left) / (right - left));     // P^T[2][0] = P[0][2]
    P[9] = (float)((top + bottom) / (top - bottom));     // P^T[2][1] = P[1][2]
    P[11] = 1.f;                                          // P^T[2][3] = P[3][2]
    P[10] = (float)(zf / (zf - zn));                      // P[2][2]
    P[14] = (float)(-(zf * zn) / (zf - zn));              // P^T[3][2] = P[2][3]
    // full_proj_transform = world_view^T-stored @ projection^T-stored, float32 like the reference's bmm
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fmaf(wv[4 * r + k], P[4 * k + c], acc);
        o[16 + 4 * r + c] = acc;
      }
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = wv[k];
    // FoV from the intrinsics as given (:1143-1146): fov = 2 atan(size / (2 f)) in float32, tan(fov / 2) in double
    const float fovx = 2.f * atanf((float)W_in / (2.f * (float)fx));
    const float fovy = 2.f * atanf((float)H_in / (2.f * (float)fy));
    const float tfx = (float)tan(0.5 * (double)fovx), tfy = (float)tan(0.5 * (double)fovy);
    o[32] = tfx; o[33] = tfy;
    o[34] = (float)W_in / (2.0f * tfx);
    o[35] = (float)H_in / (2.0f * tfy);
  }
}
}  // namespace

extern "C" {

int ocrf_geometry_blocks(int B, int N, const float* rots, const float* trans, const float* intrins,
                         const float* post_rots, const float* post_trans, const float* bda, const float* c2w,
                         int H_in, int W_in, float znear, float zfar, float* lss_block, float* ht_block,
                         float* camera_rows, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B <= 0 || N <= 0 || !rots || !trans || !intrins || !post_rots || !post_trans || !bda || H_in <= 0 || W_in <= 0 ||
      (camera_rows && !c2w))
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(geom_blocks_kernel, dim3((unsigned)((B * N + 63) / 64)), dim3(64), 0, stream, B, N, rots, trans,
                     intrins, post_rots, post_trans, bda, c2w, H_in, W_in, znear, zfar, lss_block, ht_block, camera_rows);
  return (int)hipGetLastError();
}

size_t ocrf_ht_prepare_workspace_bytes(int B, int n_pillars) {
  if (B <= 0 || n_pillars <= 0) return 0;
  const size_t n = (size_t)B * n_pillars;
  return align_up(n * 8, 256) + 256 + align_up(((n + kChunk - 1) / kChunk + 1) * 8, 256) + align_up(n * 64 * 4, 256);   // valid bits: up to 64 cameras
}

int ocrf_ht_prepare(int B, int N, int Z, int n_pillars, int Wf, int Hf, int D, const float* ref_points,
                    const float* cams, const float* pc_range, float w_in, float h_in, float depth0,
                    float depth1, int* ranks_bev, int* ranks_depth, int* ranks_feat, int* interval_starts,
                    int* interval_lengths, int* counts, void* workspace, size_t workspace_bytes,
                    ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B <= 0 || N <= 0 || N > 64 || Z <= 0 || Z > 32 || B * N > 65535 || n_pillars <= 0 || Wf <= 0 || Hf <= 0 || D <= 0 || !ref_points || !cams ||
      !pc_range || !ranks_bev || !ranks_depth || !ranks_feat || !interval_starts || !interval_lengths || !counts)
    return (int)hipErrorInvalidValue;
  if ((long)B * N * Z * n_pillars >= (1l << 31) - 1 || (long)B * N * D * Wf * Hf >= (1l << 31) - 1)
    return (int)hipErrorInvalidValue;
  const size_t need = ocrf_ht_prepare_workspace_bytes(B, n_pillars);
  if (!workspace || workspace_bytes < need) return (int)hipErrorInvalidValue;
  long long* cnt_flag = static_cast<long long*>(workspace);
  long long* total = reinterpret_cast<long long*>(static_cast<char*>(workspace) + align_up((size_t)B * n_pillars * 8, 256));
  long long* scratch = total + 32;
  unsigned* valid_bits = reinterpret_cast<unsigned*>(
      reinterpret_cast<char*>(scratch) + align_up((((size_t)B * n_pillars + kChunk - 1) / kChunk + 1) * 8, 256));
  HtParams q;
  q.B = B; q.N = N; q.Z = Z; q.Nq = n_pillars; q.Wf = Wf; q.Hf = Hf; q.D = D;
  // scale / offset as float32 scalars of double differences: what `tensor * python_float` does
  q.sx = (float)((double)pc_range[3] - (double)pc_range[0]); q.ox = pc_range[0];
  q.sy = (float)((double)pc_range[4] - (double)pc_range[1]); q.oy = pc_range[1];
  q.sz = (float)((double)pc_range[5] - (double)pc_range[2]); q.oz = pc_range[2];
  q.w_in = w_in; q.h_in = h_in; q.d0 = depth0; q.dspan = (float)((double)depth1 - (double)depth0);
  const dim3 cgrid((n_pillars + kBlock - 1) / kBlock, B * N);
  const int pgrid = (B * n_pillars + kBlock - 1) / kBlock;
  const HtCam* hc = reinterpret_cast<const HtCam*>(cams);
  // scratch holds the look-back state of the scan: (tiles + 1) words, zero-filled by the first kernel
  auto* state = reinterpret_cast<unsigned long long*>(scratch);
  const int state_words = (int)(((long)B * n_pillars + kChunk - 1) / kChunk + 1);
  ocrf::launch(OCRF_K_HT_COUNT, ht_valid_kernel, cgrid, dim3(kBlock), 0, stream, q, ref_points, hc, valid_bits, state,
               state_words);
  hipLaunchKernelGGL(ht_pillar_totals_kernel, dim3(pgrid), dim3(kBlock), 0, stream, q,
                     static_cast<const unsigned*>(valid_bits), cnt_flag);
  scan_exclusive_lookback<long long>(cnt_flag, (long)B * n_pillars, total, state, stream);
  ocrf::launch(OCRF_K_HT_EMIT, ht_emit_kernel, cgrid, dim3(kBlock), 0, stream, q, ref_points, hc,
               static_cast<const unsigned*>(valid_bits), static_cast<const long long*>(cnt_flag), ranks_bev, ranks_depth,
               ranks_feat, interval_starts, interval_lengths);
  hipLaunchKernelGGL(ht_counts_kernel, dim3(1), dim3(1), 0, stream, static_cast<const long long*>(total), counts,
                     static_cast<const unsigned long long*>(state));
  return (int)hipGetLastError();
}

}  // extern "C"

// bev_pool_v2 forward as per-tile MFMA panels (MI355X, gfx950): out[64 voxels x C] = W[64 x R] . F[R x C].
//
// The reference pools point by point: out[v] += depth[p] * feat[row(p)] (bev_pool_cuda.cu:39-47).  A tile of 64
// neighbouring voxels names each of its feature rows several times (a camera ray crosses several voxels of the tile:
// 7x for the LSS ranks, 11x for the OcRF height sampling at the 200x200 grid), so the tile's points factor into
//   W[v][r] = sum of depth[p] over the tile's points p with (voxel slot v, row slot r),   F[r] = feat row r,
// with R = the tile's UNIQUE rows: every feature row is read once per tile instead of once per point, and the
// multiply-adds run on the matrix cores — `v_mfma_f32_16x16x4_f32`: f32 in, f32 accumulate, an exact fmaf chain
// over k, so results are bitwise reproducible and within rounding of the reference's order (tests: 1e-4).
// (This is BASELINE.json's "MFMA only for the dense depth x feature outer product".)
//
// The rank-only part is a plan (ocrfdet_amd/bevpool.MfmaPoolPlan): per tile its unique rows in panels of 64, per
// panel its non-zero cells (v, r, points), the points' depth ranks in cell order, and the unit list: a tile's panels
// in groups of at most G, heaviest units first; a tile of several units reduces through write-through slabs and a
// ticket (cdna_hip_programming.md 'In-launch split-K reduction', sc1 form), the last arriver adds them in slice order.
// Per unit and panel:  feat rows -> LDS (coalesced 16-byte loads, each row once) | W zeroed, cells summed in point
// order by one thread each (plain stores: no float atomics) | 16 k-steps x C/16 MFMAs per wave (wave w owns voxels
// 16w..16w+15) | the 64 x C tile leaves through LDS in the caller's layout, 256-byte runs per channel.
#include <hip/hip_runtime.h>

#include "bev_pool_tile_out.h"
#include "launch.h"
#include "ocrf_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTV = 64;          // voxels per tile: an 8 x 8 block of one (b, z) plane — a camera ray crosses several voxels
constexpr int kTS = 8;           //   of a square block, so its feature row is shared (7x LSS / 11x HT at cfg2; 2-3x for a 64 x 1 strip)
#ifndef OCRF_MFMA_KP
#define OCRF_MFMA_KP 48
#endif
constexpr int kKP = OCRF_MFMA_KP;   // rows per panel (48: 28 KB of LDS per workgroup, five workgroups per CU = one round for 1 250 tiles)
constexpr int kMaxUnitPanels = 8; // panels per unit (plan: group <= 8)
constexpr int kLdw = kKP + 2;    // W pitch (floats), = 2 mod 16: lanes (m, k) of an A read fall on distinct banks
static_assert(kKP % 4 == 0 && (kKP + 2) % 16 == 2, "panel rows: a multiple of 16");
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct MfmaArgs {
  int C, Y, X, tx, tpp, Z, layout;      // tx = tiles per row, tpp = tiles per plane
  const int4* units;             // {tile, first panel, end panel, slice | n_slices << 16}
  const int* unit_slab;          // first slab of the unit's tile (multi-unit tiles)
  const int* panel_rows;         // [n_panels][kKP]
  const int* panel_nrows;
  const int* panel_cell_off;     // [n_panels + 1]
  const int4* cells;             // {v | r << 8 | points << 16, depth ranks of up to 3 points} or, for more points
                                 // (points field 0xFFFF), {code, first index into rd_sorted, points, -}
  const int* rd_sorted;
  const float* depth;
  const float4* feat4;
  float* out;
  float4* slabs;
  int* arrive;                   // [n_tiles], zero between calls
};

// A cell with more than three points (a near-camera voxel collects many depth bins of one pixel): rank -> depth is two
// dependent loads per point, and one point after the other that chain was the longest phase of the heaviest units
// (57 k of their 90 k cycles).  Eight points in flight, summed in point order as before.
__device__ __forceinline__ float cell_sum_many(const int* __restrict__ rd_sorted, const float* __restrict__ depth, int first,
                                               int n) {
  constexpr int kPB = 8;
  float s = 0.f;
  for (int p0 = 0; p0 < n; p0 += kPB) {
    int idx[kPB];
    float d[kPB];
#pragma unroll
    for (int k = 0; k < kPB; ++k) idx[k] = rd_sorted[first + min(p0 + k, n - 1)];
#pragma unroll
    for (int k = 0; k < kPB; ++k) d[k] = depth[idx[k]];
#pragma unroll
    for (int k = 0; k < kPB; ++k)
      if (p0 + k < n) s = (p0 + k == 0) ? d[k] : s + d[k];
  }
  return s;
}

template <int NB, bool STAMP = false>
__global__ __launch_bounds__(kBlock, NB <= 5 ? 5 : 4) void bev_pool_mfma_kernel(MfmaArgs a, unsigned long long* __restrict__ stamps) {
  OCRF_POOL_PRIO();
  constexpr int C = 16 * NB, c4 = C / 4;
  constexpr int kLdf = C + ((16 - C % 32) + 32) % 32;        // F pitch (floats) = 16 mod 32: conflict-free B reads
  constexpr int ldq = c4 | 1;                                 // output tile pitch in float4
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                            // [kTV][kLdw]
  float* Fs = smem + kTV * kLdw;                               // [kKP][kLdf]
  float4* tile = reinterpret_cast<float4*>(smem);              // [kTV][ldq], aliases Ws / Fs after the last panel
  int* s_flag = reinterpret_cast<int*>(smem + kTV * kLdw + kKP * kLdf);
  int* s_rows = s_flag + 4;                                    // [kMaxUnitPanels][kKP] row ids of the unit's panels

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int4 unit = a.units[blockIdx.x];
  const int tileid = unit.x, n_slices = unit.w >> 16, slice = unit.w & 0xFFFF;
  const int plane = tileid / a.tpp, kt = tileid % a.tpp;
  const int y0 = (kt / a.tx) * kTS, x0 = (kt % a.tx) * kTS;

  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t_prev = 0, t_acc[6] = {0, 0, 0, 0, 0, 0};      // diagnostic build only
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_prev;
      t_prev = t;
    }
  };

  // Row ids of every panel of the unit -> LDS once (one coalesced read): the feature-row loads below are then ONE
  // memory round trip per panel, and the next panel's rows are fetched into registers under this panel's MFMAs.
  const int n_pan = unit.z - unit.y;
  for (int i = tid; i < n_pan * kKP; i += kBlock) s_rows[i] = a.panel_rows[unit.y * kKP + i];
  constexpr int kFR = (kKP * c4 + kBlock - 1) / kBlock;         // float4 of a panel per thread
  float4 freg[kFR];
  int nr_next = n_pan > 0 ? a.panel_nrows[unit.y] : 0;
  __syncthreads();
  auto fetch_rows = [&](int pl, int nr) {                        // panel pl of the unit: global -> registers
    const int k4 = (nr + 3) & ~3;
#pragma unroll
    for (int j = 0; j < kFR; ++j) {
      const int i = tid + j * kBlock;
      const int r = i / c4, q = i - r * c4;
      freg[j] = make_float4(0.f, 0.f, 0.f, 0.f);      // rows that pad the last k-step are zero (0 * stale LDS could be NaN)
      if (i < k4 * c4 && r < nr) freg[j] = a.feat4[(long)s_rows[pl * kKP + r] * c4 + q];
    }
  };
  if (n_pan > 0) fetch_rows(0, nr_next);
  stamp(-1);
  for (int pl = 0; pl < n_pan; ++pl) {
    const int pn = unit.y + pl;
    const int nr = nr_next;
    const int k4 = (nr + 3) & ~3;
    const int c0 = a.panel_cell_off[pn], c1 = a.panel_cell_off[pn + 1];
    float4* Fs4 = reinterpret_cast<float4*>(Fs);
#pragma unroll
    for (int j = 0; j < kFR; ++j) {
      const int i = tid + j * kBlock;
      const int r = i / c4, q = i - r * c4;
      if (i < k4 * c4) Fs4[r * (kLdf / 4) + q] = freg[j];
    }
    for (int i = tid; i < kTV * (k4 / 2); i += kBlock) {       // W[.][0, k4) = 0, two floats per store
      const int v = i / (k4 / 2), k = i - v * (k4 / 2);
      *reinterpret_cast<float2*>(Ws + v * kLdw + 2 * k) = make_float2(0.f, 0.f);
    }
    // the cells' depth weights: record (inline depth ranks of up to three points) -> depth, summed in point order
    constexpr int kCR = 2;
    float csum[kCR];
    unsigned ccode[kCR];
#pragma unroll
    for (int j = 0; j < kCR; ++j) {
      const int c = c0 + tid + j * kBlock;
      ccode[j] = 0xFFFFFFFFu;
      csum[j] = 0.f;
      if (c < c1) {
        const int4 rec = a.cells[c];
        const int np = (int)((unsigned)rec.x >> 16);
        ccode[j] = (unsigned)rec.x & 0xFFFFu;
        if (np <= 3) {
          float s2 = a.depth[rec.y];
          if (np > 1) s2 += a.depth[rec.z];
          if (np > 2) s2 += a.depth[rec.w];
          csum[j] = s2;
        } else {
          csum[j] = cell_sum_many(a.rd_sorted, a.depth, rec.y, rec.z);
        }
      }
    }
    __syncthreads();
    stamp(0);
#pragma unroll
    for (int j = 0; j < kCR; ++j)
      if (ccode[j] != 0xFFFFFFFFu) Ws[(ccode[j] & 0xFFu) * kLdw + (ccode[j] >> 8)] = csum[j];
    for (int c = c0 + tid + kCR * kBlock; c < c1; c += kBlock) {      // panels denser than 2 cells per thread
      const int4 rec = a.cells[c];
      const int np = (int)((unsigned)rec.x >> 16);
      float s2;
      if (np <= 3) {
        s2 = a.depth[rec.y];
        if (np > 1) s2 += a.depth[rec.z];
        if (np > 2) s2 += a.depth[rec.w];
      } else {
        s2 = cell_sum_many(a.rd_sorted, a.depth, rec.y, rec.z);
      }
      Ws[((unsigned)rec.x & 0xFFu) * kLdw + (((unsigned)rec.x >> 8) & 0xFFu)] = s2;
    }
    // the next panel's rows leave for registers now: they land under this panel's MFMAs
    if (pl + 1 < n_pan) {
      nr_next = a.panel_nrows[pn + 1];
      fetch_rows(pl + 1, nr_next);
    }
    __syncthreads();
    stamp(1);
    const float* wrow = Ws + (16 * wave + (lane & 15)) * kLdw + (lane >> 4);
    const float* frow = Fs + (lane >> 4) * kLdf + (lane & 15);
    for (int kk = 0; kk < k4; kk += 4) {
      const float aw = wrow[kk];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, frow[kk * kLdf + 16 * nb], acc[nb], 0, 0, 0);
    }
    __syncthreads();                    // the next panel (or the epilogue) overwrites Ws / Fs
    stamp(2);
  }

  // C/D map: col = lane & 15 (channel), row = 4 (lane >> 4) + reg (voxel of the wave's 16)
  float* tf = reinterpret_cast<float*>(tile);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) tf[(16 * wave + 4 * (lane >> 4) + r) * (4 * ldq) + 16 * nb + (lane & 15)] = acc[nb][r];
  __syncthreads();

  stamp(3);
  auto leave = [&]() {
    if constexpr (STAMP) {
      if (tid == 0) {
        for (int k = 0; k < 6; ++k) stamps[(long)blockIdx.x * 8 + k] = t_acc[k];
        stamps[(long)blockIdx.x * 8 + 6] = (unsigned long long)(unit.z - unit.y);
        stamps[(long)blockIdx.x * 8 + 7] = (unsigned long long)n_slices;
      }
    }
  };
  // slab + ticket for tiles of several units, then the tile in the caller's layout: shared with bev_pool_panel.hip
  pool_out::Dest d;
  d.C = C; d.Y = a.Y; d.X = a.X; d.Z = a.Z; d.layout = a.layout; d.out = a.out; d.slabs = a.slabs; d.arrive = a.arrive;
  pool_out::leave<c4>(tile, s_flag, d, tileid, plane, y0, x0, n_slices, slice, n_slices > 1 ? a.unit_slab[blockIdx.x] : 0);
  stamp(5);
  leave();
}

unsigned long long* g_mfma_stamps = nullptr;

template <int NB>
size_t mfma_lds_bytes() {
  constexpr int C = 16 * NB;
  constexpr int kLdf = C + ((16 - C % 32) + 32) % 32;
  return (size_t)(kTV * kLdw + kKP * kLdf) * sizeof(float) + 16 + (size_t)kMaxUnitPanels * kKP * sizeof(int);
}

}  // namespace

extern "C" {

// Diagnostic: device buffer of n_units x 8 u64 -> the next MFMA poolings (C = 80) run the stamped build: cycles of
// {F rows + W zero, cells, MFMA, tile to LDS, slab + ticket, write-out}, panels, slices per unit.
int ocrf_diag_pool_mfma_stamps(unsigned long long* buf) { g_mfma_stamps = buf; return 0; }

int ocrf_bev_pool_mfma_panel_rows(void) { return kKP; }
int ocrf_bev_pool_mfma_tile_side(void) { return kTS; }

size_t ocrf_bev_pool_mfma_slab_bytes(int c, int n_slab_slices) {
  return (size_t)(n_slab_slices > 0 ? n_slab_slices : 1) * kTV * c * sizeof(float);
}

int ocrf_bev_pool_mfma_max_unit_panels(void) { return kMaxUnitPanels; }

int ocrf_bev_pool_v2_nchw_mfma(int c, int n_units, const int* units, const int* unit_slab, const int* panel_rows,
                               const int* panel_nrows, const int* panel_cell_off, const int* cells,
                               const int* rd_sorted, const float* depth, const float* feat, float* out,
                               int B, int Z, int Y, int X, int layout, int* arrive, void* slabs, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (c != 80 && c != 64 && c != 96 && c != 128) return (int)hipErrorInvalidValue;
  if (n_units <= 0 || !units || !unit_slab || !panel_nrows || !panel_cell_off || !cells || !rd_sorted || !depth || !feat || !out || !arrive ||
      !slabs || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || layout < 0 || layout > 2)
    return (int)hipErrorInvalidValue;
  MfmaArgs a;
  a.C = c; a.Y = Y; a.X = X; a.tx = (X + kTS - 1) / kTS; a.tpp = a.tx * ((Y + kTS - 1) / kTS); a.Z = Z; a.layout = layout;
  a.units = reinterpret_cast<const int4*>(units);
  a.unit_slab = unit_slab;
  a.panel_rows = panel_rows; a.panel_nrows = panel_nrows; a.panel_cell_off = panel_cell_off;
  a.cells = reinterpret_cast<const int4*>(cells); a.rd_sorted = rd_sorted;
  a.depth = depth; a.feat4 = reinterpret_cast<const float4*>(feat);
  a.out = out; a.slabs = static_cast<float4*>(slabs); a.arrive = arrive;
  const dim3 grid((unsigned)n_units), block(kBlock);
  unsigned long long* none = nullptr;
  if (g_mfma_stamps && c == 80) {      // diagnostic build, never used by the product path
    hipLaunchKernelGGL((bev_pool_mfma_kernel<5, true>), grid, block, mfma_lds_bytes<5>(), stream, a, g_mfma_stamps);
    return (int)hipGetLastError();
  }
  switch (c) {
    case 64: ocrf::launch(OCRF_K_BEV_POOL_MFMA, bev_pool_mfma_kernel<4>, grid, block, mfma_lds_bytes<4>(), stream, a, none); break;
    case 80: ocrf::launch(OCRF_K_BEV_POOL_MFMA, bev_pool_mfma_kernel<5>, grid, block, mfma_lds_bytes<5>(), stream, a, none); break;
    case 96: ocrf::launch(OCRF_K_BEV_POOL_MFMA, bev_pool_mfma_kernel<6>, grid, block, mfma_lds_bytes<6>(), stream, a, none); break;
    default: ocrf::launch(OCRF_K_BEV_POOL_MFMA, bev_pool_mfma_kernel<8>, grid, block, mfma_lds_bytes<8>(), stream, a, none); break;
  }
  return (int)hipGetLastError();
}

}  // extern "C"

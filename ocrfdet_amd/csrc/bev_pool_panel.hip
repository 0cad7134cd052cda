// bev_pool_v2 forward over the panel plan (bevpool.MfmaPoolPlan), cell by cell out of LDS — MI355X, gfx950.
//
// The panel plan factors a tile of 8 x 8 voxels into  out[v] = sum_r W[v][r] F[r]  with R = the tile's UNIQUE feature
// rows (7x fewer than its points for the LSS ranks, 11x for the height sampling) and W the summed depth weights of the
// points of cell (v, r) (bev_pool_cuda.cu:39-47 regrouped).  bev_pool_mfma.hip multiplies the dense W[64 x 48] panels
// on the matrix cores; but W is 5-9 % dense at the reference shapes, the 60 MFMAs of a panel are issue-bound at
// ~2 000 cycles per wave, and zeroing / scattering W and the cells' rank -> depth chains were the rest of a unit's
// time.  Here the same plan runs as a LATENCY kernel:
//   * the cell weights are summed by a pre-pass of one thread per cell (bev_pool_cell_weights_kernel, both poolings
//     of a step in ONE launch): the pooling kernel itself has no dependent gather chain left;
//   * per panel: feature rows -> LDS (each row once, coalesced 16-byte loads), the panel's cell weights and row slots
//     -> LDS (coalesced), all of it requested one panel ahead into registers;
//   * cells are stored voxel-major; LANE = VOXEL SLOT, wave = a quarter of the channels: a lane walks the cells of
//     its voxel — one {weight, row slot} entry and C/16 float4 of the row per trip, the next entry requested under
//     the current rows — acc = fma(F[r], w, acc) in registers, rows ascending, panel after panel: exactly the k
//     order of the MFMA form, so the two kernels agree bit for bit on finite inputs.  A trip is ~25 instructions for
//     up to 64 cells and a panel takes as many trips as its longest voxel has cells (cfg2: 10 of 168 cells for the
//     LSS ranks, 6 of 289 for the height sampling) — the lane-group-per-cell forms tried first spent ~20 VALU
//     instructions of index work per cell and were bound by exactly that;
//   * the rows sit in LDS at an odd float4 pitch (lanes read different rows: 16 distinct bank offsets);
//   * the 64 x C tile leaves through LDS in the caller's layout (bev_pool_tile_out.h; tiles of several units reduce
//     through write-through slabs and a ticket in slice order).
// No float atomics anywhere: results are bitwise reproducible.
#include <hip/hip_runtime.h>

#include "bev_pool_tile_out.h"
#include "launch.h"
#include "ocrf_hip.h"

namespace {

using pool_out::kTV;
constexpr int kBlock = 256;
#ifndef OCRF_MFMA_KP
#define OCRF_MFMA_KP 48
#endif
constexpr int kKP = OCRF_MFMA_KP;       // rows per panel: the plan's (ocrf_bev_pool_mfma_panel_rows())
constexpr int kMaxUnitPanels = 8;       // ocrf_bev_pool_mfma_max_unit_panels()
constexpr int kCellCap = 1024;          // cells of a panel held in LDS at once (denser panels: one window after the other)
constexpr int kCR = 2;                  // cells per thread requested one panel ahead
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PanelArgs {
  int tx, tpp;                   // tiles per row / per plane
  const int4* units;             // {tile, first panel, end panel, slice | n_slices << 16}
  const int* unit_slab;
  const int* panel_rows;         // [n_panels][kKP]
  const int* panel_nrows;
  const int* panel_cell_off;     // [n_panels + 1]
  const int* panel_voff;         // [n_panels][64]: first cell of voxel slot v, relative to the panel's first cell
  const unsigned short* cell_code;   // [n_cells] row slot | voxel slot << 8 (cells of a panel: voxel-major, rows ascending)
  const float* cw;               // [n_cells] summed depth weight of the cell
  const float4* feat4;
  int feat_records;              // bytes the feature tensor holds: the bound of its buffer resource (0x7fffffff: not stated)
  pool_out::Dest d;
};

__device__ __forceinline__ float4 fma4(float4 f, float w, float4 a) {
  a.x = fmaf(f.x, w, a.x);
  a.y = fmaf(f.y, w, a.y);
  a.z = fmaf(f.z, w, a.z);
  a.w = fmaf(f.w, w, a.w);
  return a;
}

// One thread per cell: w = depth of its points, summed in point order (the order the MFMA form uses).  Two cell
// lists per launch: the LSS and the height-sampling plan of a step read the same depth tensor.
__global__ __launch_bounds__(kBlock) void bev_pool_cell_weights_kernel(
    int n0, int blocks0, const int4* __restrict__ cells0, const int* __restrict__ rd0, float* __restrict__ cw0,
    int n1, const int4* __restrict__ cells1, const int* __restrict__ rd1, float* __restrict__ cw1,
    const float* __restrict__ depth) {
  const bool second = (int)blockIdx.x >= blocks0;
  const int n = second ? n1 : n0;
  const int4* cells = second ? cells1 : cells0;
  const int* rd = second ? rd1 : rd0;
  float* cw = second ? cw1 : cw0;
  const int c = ((int)blockIdx.x - (second ? blocks0 : 0)) * kBlock + threadIdx.x;
  const int4 rec = cells[min(c, n - 1)];
  const int np = (int)((unsigned)rec.x >> 16);
  float s;
  if (np <= 3) {
    // up to three inline depth ranks: all three requested at once (slots beyond the count repeat the first)
    const float d0 = depth[rec.y];
    const float d1 = depth[np > 1 ? rec.z : rec.y];
    const float d2 = depth[np > 2 ? rec.w : rec.y];
    s = d0;
    if (np > 1) s += d1;
    if (np > 2) s += d2;
  } else {
    constexpr int kPB = 8;               // eight points in flight, summed in point order
    const int first = rec.y, cnt = rec.z;
    s = 0.f;
    for (int p0 = 0; p0 < cnt; p0 += kPB) {
      int idx[kPB];
      float d[kPB];
#pragma unroll
      for (int k = 0; k < kPB; ++k) idx[k] = rd[first + min(p0 + k, cnt - 1)];
#pragma unroll
      for (int k = 0; k < kPB; ++k) d[k] = depth[idx[k]];
#pragma unroll
      for (int k = 0; k < kPB; ++k)
        if (p0 + k < cnt) s = (p0 + k == 0) ? d[k] : s + d[k];
    }
  }
  if (c < n) cw[c] = s;
}

// Waves per SIMD asked of the compiler: beside the render stream's persistent blend (95 VGPRs, 3-4 waves per SIMD) a SIMD
// has 130-220 free VGPRs, so 96 or fewer decides whether one or two pooling waves fit.  C = 80 builds in 96 without
// scratch; the wider rows (more accumulators per lane) get 128.
#ifdef OCRF_PANEL_WAVES
#define OCRF_PANEL_BOUNDS(C4) __launch_bounds__(kBlock, OCRF_PANEL_WAVES)      // A/B build
#else
#define OCRF_PANEL_BOUNDS(C4) __launch_bounds__(kBlock, (C4) == 20 ? 5 : ((C4) <= 24 ? 4 : 3))
#endif
template <int C4, bool STAMP = false>
__global__ OCRF_PANEL_BOUNDS(C4) void bev_pool_panel_kernel(PanelArgs a, unsigned long long* __restrict__ stamps) {
  OCRF_POOL_PRIO();
  unsigned long long t_prev = 0, t_acc[5] = {0, 0, 0, 0, 0};        // diagnostic build only
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_prev;
      t_prev = t;
    }
  };
  stamp(-1);
  static_assert(C4 % 4 == 0, "a wave owns a quarter of the channels");
  constexpr int ldq = C4 | 1;                                  // pitch (float4) of the rows in LDS and of the finished tile:
                                                               //   odd, so lanes reading different rows spread over the banks
  constexpr int K = C4 / 4;                                    // float4 of channels per lane
  constexpr int kFR = (kKP * C4 + kBlock - 1) / kBlock;        // float4 of a panel's rows per thread
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  float4* Fs = smem4;                                          // [kKP][ldq]
  float4* tile = smem4;                                        // [kTV][ldq] after the last panel
  uint2* s_ent = reinterpret_cast<uint2*>(smem4 + kTV * ldq);  // [kCellCap] {weight bits, row slot}
  int* s_voff = reinterpret_cast<int*>(s_ent + kCellCap);      // [65] (+ pad)
  int* s_flag = s_voff + 68;
  int* s_nr = s_flag + 4;                                      // [kMaxUnitPanels]
  int* s_coff = s_nr + kMaxUnitPanels;                         // [kMaxUnitPanels + 1] (+ pad)
  int* s_rows = s_coff + kMaxUnitPanels + 4;                   // [kMaxUnitPanels][kKP]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int4 unit = a.units[blockIdx.x];
  const int tileid = unit.x, n_slices = unit.w >> 16, slice = unit.w & 0xFFFF;
  const int plane = tileid / a.tpp, kt = tileid % a.tpp;
  const int y0 = (kt / a.tx) * pool_out::kTS, x0 = (kt % a.tx) * pool_out::kTS;
  const int n_pan = min(unit.z - unit.y, kMaxUnitPanels);

  // everything the panels' requests depend on, in ONE round trip: row ids, row counts, cell ranges of the unit's panels
  for (int i = tid; i < n_pan * kKP; i += kBlock) s_rows[i] = a.panel_rows[unit.y * kKP + i];
  if (tid < n_pan) s_nr[tid] = a.panel_nrows[unit.y + tid];
  if (tid <= n_pan) s_coff[tid] = a.panel_cell_off[unit.y + tid];
  __syncthreads();
  stamp(0);

  float4 acc[K];                                               // voxel slot `lane`, channels [4 K wave, 4 K (wave + 1))
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 freg[kFR];
#pragma unroll
  for (int j = 0; j < kFR; ++j) freg[j] = make_float4(0.f, 0.f, 0.f, 0.f);     // (defined on every path: else hipcc keeps the array in scratch)
  float cwreg[kCR];
  unsigned ccreg[kCR];
  int voffreg = 0;
  // panel pl of the unit: global -> registers.  Every load is issued, at a clamped address: a predicated load would
  // become a branch with its own wait.
  // (buffer loads: a 32-bit byte offset per lane instead of a 64-bit address pair per load — the host checks that the
  // tensors stay below 2 GB)
  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.feat4), 0, a.feat_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t cw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.cw), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t cc_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.cell_code), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t vo_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.panel_voff), 0, 0x7fffffff, 0x00020000);
  auto fetch = [&](int pl) __attribute__((always_inline)) {
    const int nr = s_nr[pl], c0 = s_coff[pl], nc = s_coff[pl + 1] - c0;
#pragma unroll
    for (int j = 0; j < kFR; ++j) {
      const int i = tid + j * kBlock;
      const int r = max(min(i / C4, nr - 1), 0), q = i % C4;
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (s_rows[pl * kKP + r] * C4 + q) * 16, 0, 0);
      freg[j] = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
    }
#pragma unroll
    for (int j = 0; j < kCR; ++j) {
      const int ci = max(c0 + min(tid + j * kBlock, nc - 1), 0);
      cwreg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cw_rsrc, ci * 4, 0, 0));
      ccreg[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(cc_rsrc, ci * 2, 0, 0);
    }
    voffreg = (int)__builtin_amdgcn_raw_buffer_load_b32(vo_rsrc, ((unit.y + pl) * kTV + lane) * 4, 0, 0);
  };
  if (n_pan > 0) fetch(0);

  // Lane = voxel slot: it walks ITS cells [b, e) of the window (window base cb) — the next entry is requested under
  // the current entry's rows — acc = fma(F[r], w, acc), rows ascending, panel after panel: the k order of the MFMA form.
  // A trip costs ~25 instructions for up to 64 cells; a panel takes as many trips as its longest voxel has cells.
  auto sums = [&](int b, int e) __attribute__((always_inline)) {
    const float4* fcol = Fs + wave * K;
    int c = b;
    uint2 en = s_ent[max(min(c, e - 1), 0)];
    while (c < e) {
      const uint2 nx = s_ent[min(c + 1, e - 1)];
      const float4* frow = fcol + (int)(en.y & 0xFFu) * ldq;
      float4 f[K];
#pragma unroll
      for (int k = 0; k < K; ++k) f[k] = frow[k];
      const float w = __uint_as_float(en.x);
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = fma4(f[k], w, acc[k]);
      en = nx;
      ++c;
    }
  };

  for (int pl = 0; pl < n_pan; ++pl) {
    const int c0 = s_coff[pl], ncell = s_coff[pl + 1] - c0;
    const int n0 = min(ncell, kCellCap);       // the first window of cells comes out of the registers fetch() filled
#pragma unroll
    for (int j = 0; j < kFR; ++j) {
      const int i = tid + j * kBlock;
      if (i < kKP * C4) Fs[(i / C4) * ldq + i % C4] = freg[j];
    }
#pragma unroll
    for (int j = 0; j < kCR; ++j) {
      const int ci = tid + j * kBlock;
      float wv = cwreg[j];
      unsigned cv = ccreg[j];
      asm volatile("" : "+v"(wv), "+v"(cv));          // the loads stay where fetch() issued them
      if (ci < n0) s_ent[ci] = make_uint2(__float_as_uint(wv), cv);
    }
    for (int ci = tid + kCR * kBlock; ci < n0; ci += kBlock)      // denser panels: the rest straight from memory
      s_ent[ci] = make_uint2(__float_as_uint(a.cw[c0 + ci]), (unsigned)a.cell_code[c0 + ci]);
    if (wave == 0) s_voff[lane] = voffreg;
    if (tid == kTV) s_voff[kTV] = ncell;
    if (pl + 1 < n_pan) fetch(pl + 1);         // lands under this panel's sums
    __syncthreads();
    stamp(1);
    const int b = s_voff[lane], e = s_voff[lane + 1];
    for (int cb = 0;;) {
      sums(max(b, cb) - cb, min(e, cb + kCellCap) - cb);
      cb += kCellCap;
      if (cb >= ncell) break;
      // a panel denser than the window (synthetic rank vectors): the next kCellCap cells; a lane's run cut by a window
      // border goes on where it stopped (same order)
      __syncthreads();
      for (int ci = tid; ci < min(kCellCap, ncell - cb); ci += kBlock)
        s_ent[ci] = make_uint2(__float_as_uint(a.cw[c0 + cb + ci]), (unsigned)a.cell_code[c0 + cb + ci]);
      __syncthreads();
    }
    __syncthreads();                           // the next panel (or the finished tile) overwrites Fs / the cell window
    stamp(2);
  }

#pragma unroll
  for (int k = 0; k < K; ++k) tile[lane * ldq + wave * K + k] = acc[k];
  __syncthreads();
  stamp(3);
  pool_out::leave<C4>(tile, s_flag, a.d, tileid, plane, y0, x0, n_slices, slice, n_slices > 1 ? a.unit_slab[blockIdx.x] : 0);
  stamp(4);
  if constexpr (STAMP) {
    if (tid == 0) {
      for (int k = 0; k < 5; ++k) stamps[(long)blockIdx.x * 8 + k] = t_acc[k];
      stamps[(long)blockIdx.x * 8 + 5] = (unsigned long long)n_pan;
      stamps[(long)blockIdx.x * 8 + 6] = (unsigned long long)(n_pan > 0 ? s_coff[n_pan] - s_coff[0] : 0);
      stamps[(long)blockIdx.x * 8 + 7] = (unsigned long long)n_slices;
    }
  }
}

unsigned long long* g_panel_stamps = nullptr;

template <int C4>
size_t panel_lds_bytes() {
  constexpr int ldq = C4 | 1;
  return (size_t)(kTV * ldq) * 16 + (size_t)kCellCap * 8 + (68 + 4 + kMaxUnitPanels + kMaxUnitPanels + 4 + kMaxUnitPanels * kKP) * 4;
}

}  // namespace

extern "C" {

// Diagnostic: device buffer of n_units x 8 u64 -> the next panel poolings (C = 80) run the stamped build: cycles of
// {unit header, wait for the panel's rows + cells, sums, tile to LDS, slab + write-out}, panels, cells, slices per unit.
int ocrf_diag_pool_panel_stamps(unsigned long long* buf) { g_panel_stamps = buf; return 0; }

int ocrf_bev_pool_cell_weights(int n_cells0, const int* cells0, const int* rd_sorted0, float* cw0, int n_cells1,
                               const int* cells1, const int* rd_sorted1, float* cw1, const float* depth,
                               ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_cells0 < 0 || n_cells1 < 0 || !depth) return (int)hipErrorInvalidValue;
  if (n_cells0 > 0 && (!cells0 || !rd_sorted0 || !cw0)) return (int)hipErrorInvalidValue;
  if (n_cells1 > 0 && (!cells1 || !rd_sorted1 || !cw1)) return (int)hipErrorInvalidValue;
  const int b0 = (n_cells0 + kBlock - 1) / kBlock, b1 = (n_cells1 + kBlock - 1) / kBlock;
  if (b0 + b1 == 0) return 0;
  ocrf::launch(OCRF_K_BEV_POOL_CELL_WEIGHTS, bev_pool_cell_weights_kernel, dim3((unsigned)(b0 + b1)), dim3(kBlock), 0, stream,
               n_cells0, b0, reinterpret_cast<const int4*>(cells0), rd_sorted0, cw0, n_cells1,
               reinterpret_cast<const int4*>(cells1), rd_sorted1, cw1, depth);
  return (int)hipGetLastError();
}

int ocrf_bev_pool_v2_nchw_panel(int c, int n_units, const int* units, const int* unit_slab, const int* panel_rows,
                                const int* panel_nrows, const int* panel_cell_off, const int* panel_voff,
                                const unsigned short* cell_code, const float* cw, const float* feat, float* out, int B, int Z,
                                int Y, int X, int layout, int* arrive, void* slabs, size_t feat_bytes, ocrf_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (c != 80 && c != 64 && c != 96 && c != 128) return (int)hipErrorInvalidValue;
  // 32-bit byte offsets into feat and the output (bev_pool_cuda.cu:39-47 indexes with int): refused at the C boundary
  if (feat_bytes >= (1ull << 31) || (size_t)B * Z * Y * X * c * 4 >= (1ull << 31)) return (int)hipErrorInvalidValue;
  if (n_units <= 0 || !units || !unit_slab || !panel_rows || !panel_nrows || !panel_cell_off || !panel_voff || !cell_code || !cw ||
      !feat || !out || !arrive || !slabs || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || layout < 0 || layout > 2)
    return (int)hipErrorInvalidValue;
  PanelArgs a;
  a.tx = (X + pool_out::kTS - 1) / pool_out::kTS;
  a.tpp = a.tx * ((Y + pool_out::kTS - 1) / pool_out::kTS);
  a.units = reinterpret_cast<const int4*>(units);
  a.unit_slab = unit_slab;
  a.panel_rows = panel_rows; a.panel_nrows = panel_nrows; a.panel_cell_off = panel_cell_off; a.panel_voff = panel_voff;
  a.cell_code = cell_code; a.cw = cw;
  a.feat4 = reinterpret_cast<const float4*>(feat);
  a.feat_records = feat_bytes ? (int)feat_bytes : 0x7fffffff;      // (0: not stated — the reference's ABI has no sizes either)
  a.d.C = c; a.d.Y = Y; a.d.X = X; a.d.Z = Z; a.d.layout = layout;
  a.d.out = out; a.d.slabs = static_cast<float4*>(slabs); a.d.arrive = arrive;
  const dim3 grid((unsigned)n_units), block(kBlock);
  unsigned long long* none = nullptr;
  if (g_panel_stamps && c == 80) {      // diagnostic build, never used by the product path
    hipLaunchKernelGGL((bev_pool_panel_kernel<20, true>), grid, block, panel_lds_bytes<20>(), stream, a, g_panel_stamps);
    return (int)hipGetLastError();
  }
  switch (c) {
    case 64: ocrf::launch(OCRF_K_BEV_POOL_PANEL, bev_pool_panel_kernel<16>, grid, block, panel_lds_bytes<16>(), stream, a, none); break;
    case 80: ocrf::launch(OCRF_K_BEV_POOL_PANEL, bev_pool_panel_kernel<20>, grid, block, panel_lds_bytes<20>(), stream, a, none); break;
    case 96: ocrf::launch(OCRF_K_BEV_POOL_PANEL, bev_pool_panel_kernel<24>, grid, block, panel_lds_bytes<24>(), stream, a, none); break;
    default: ocrf::launch(OCRF_K_BEV_POOL_PANEL, bev_pool_panel_kernel<32>, grid, block, panel_lds_bytes<32>(), stream, a, none); break;
  }
  return (int)hipGetLastError();
}

}  // extern "C"

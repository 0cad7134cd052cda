// Library identification and the per-kernel event timer of libocrf_hip.so.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "launch.h"
#include "ocrf_hip.h"

namespace {

struct Timer {
  int kernel_id = 0;
  std::vector<hipEvent_t> start, stop;
  int used = 0;
};

std::mutex g_mu;
std::vector<Timer*> g_armed;              // armed timers (a few at most: one per kernel id of interest)
volatile int g_n_armed = 0;               // fast-path test without the lock

}  // namespace

namespace {

__global__ void zero_words_kernel(unsigned* __restrict__ dst, size_t n_words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) dst[i] = 0u;
}

__global__ void diag_stamp_kernel(unsigned long long* __restrict__ slot) { *slot = wall_clock64(); }

// where does a workgroup run?  out[b] = XCC_ID << 16 | HW_ID[15:0] (gfx9: CU_ID 11:8, SH_ID 12, SE_ID 15:13); each
// workgroup spins a little so that the launch spreads over every CU its stream may use
__global__ void diag_where_kernel(unsigned* __restrict__ out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xFu) << 16) | (hw & 0xFFFFu);
}

}  // namespace

namespace ocrf {

hipError_t zero_async(void* dst, size_t bytes, hipStream_t stream) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     static_cast<unsigned*>(dst), n);
  return hipGetLastError();
}

bool timer_next(int kernel_id, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_n_armed) return false;   // fast path: no lock when no timer is armed
  std::lock_guard<std::mutex> lk(g_mu);
  for (Timer* t : g_armed) {
    if (t->kernel_id != kernel_id || t->used >= (int)t->start.size()) continue;
    *start = t->start[t->used];
    *stop = t->stop[t->used];
    ++t->used;
    return true;
  }
  return false;
}

}  // namespace ocrf

extern "C" {

const char* ocrf_version(void) { return "ocrf_hip 0.1 gfx950"; }

int ocrf_diag_stamp(unsigned long long* slot, void* stream) {
  hipLaunchKernelGGL(diag_stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), slot);
  return (int)hipGetLastError();
}

int ocrf_diag_where(int n_blocks, unsigned* out, int spin_ticks, void* stream) {
  if (n_blocks <= 0 || !out) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(diag_where_kernel, dim3(n_blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out, spin_ticks);
  return (int)hipGetLastError();
}

const char* ocrf_kernel_name(int kernel_id) {
  switch (kernel_id) {
    case OCRF_K_BEV_POOL_FWD: return "bev_pool_tile_kernel<false>";
    case OCRF_K_BEV_POOL_INTERVAL: return "bev_pool_interval_kernel";
    case OCRF_K_BEV_POOL_MFMA: return "bev_pool_mfma_kernel<*>";
    case OCRF_K_BEV_POOL_PANEL: return "bev_pool_panel_kernel<*>";
    case OCRF_K_BEV_POOL_CELL_WEIGHTS: return "bev_pool_cell_weights_kernel";
    case OCRF_K_BEV_POOL_GRAD: return "bev_pool_grad_vec_kernel";
    case OCRF_K_RASTER_PREPROCESS: return "raster_preprocess_kernel";
    case OCRF_K_RASTER_BLEND: return "raster_blend_kernel<false, false, *>";
    case OCRF_K_RASTER_BLEND_BWD: return "raster_blend_kernel<false, true, true>";
    case OCRF_K_RASTER_PRE_BWD: return "raster_preprocess_backward_kernel";
    case OCRF_K_RASTER_GATHER: return "raster_scatter_kernel";
    case OCRF_K_RASTER_PLAN_UPDATE: return "raster_plan_update_kernel";
    case OCRF_K_RASTER_BLEND_SORTED: return "raster_blend_sorted_kernel<*>";
    case OCRF_K_RASTER_SCAN: return "raster_bucket_scan_kernel";
    case OCRF_K_HOA_STATS: return "hoa_channel_stats_kernel";
    case OCRF_K_HOA_MASK_GATE: return "hoa_mask_gate_kernel";
    case OCRF_K_HOA_HEIGHT_MAX: return "hoa_height_max_kernel";
    case OCRF_K_HOA_HEIGHT_GATE: return "hoa_height_gate_kernel";
    case OCRF_K_HOA_UNET_BLOCK: return "hoa_unet_block_kernel";
    case OCRF_K_HOA_OUT_CONV: return "hoa_gated_conv1x1_kernel";
    case OCRF_K_HOA1_ATTN: return "hoa1_attention_upsample_kernel";
    case OCRF_K_HOA1_UP: return "(unused)";
    case OCRF_K_HOA1_Q: return "(unused)";
    case OCRF_K_HOA1_KV: return "hoa1_kv_kernel";
    case OCRF_K_HOA_DW3X3: return "hoa_dw3x3_kernel";
    case OCRF_K_HOA_DW3X3_WGRAD: return "hoa_dw3x3_wgrad_kernel";
    case OCRF_K_LSS_KEYS: return "lss_keys_hist_kernel";
    case OCRF_K_RADIX_HIST: return "radix_hist_kernel";
    case OCRF_K_SCAN: return "scan_apply_kernel<*>";
    case OCRF_K_RADIX_SCATTER: return "radix_scatter_kernel<*>";
    case OCRF_K_LSS_BOUNDS: return "lss_intervals_kernel";
    case OCRF_K_LSS_EMIT: return "(unused)";
    case OCRF_K_HT_COUNT: return "ht_valid_kernel";
    case OCRF_K_HT_EMIT: return "ht_emit_kernel";
    case OCRF_K_HT_PROJECT: return "ht_project_kernel";
    case OCRF_K_NECK_PREFILTER: return "neck_prefilter_kernel";
    case OCRF_K_NECK_SAMPLE: return "neck_pillar_sample_mean_kernel<*>";
    case OCRF_K_NECK_RETAIN: return "neck_retain_scatter_kernel";
    case OCRF_K_NECK_HEADS: return "neck_gauss_heads_kernel";
    case OCRF_K_NECK_NERF_ALPHA: return "neck_nerf_alpha_kernel";
    case OCRF_K_NECK_NERF_RENDER: return "neck_nerf_render_kernel";
    case OCRF_K_NECK_FUSION: return "neck_dual_fusion_kernel<*>";
    case OCRF_K_NECK_PLANE_PASS: return "neck_plane_pass_kernel<*>";
    case OCRF_K_NECK_CHANNEL_MLP: return "neck_channel_mlp_kernel";
    case OCRF_K_NECK_SCALED_STATS: return "neck_scaled_channel_stats_kernel";
    case OCRF_K_NECK_CBAM_TAIL: return "neck_cbam_tail_kernel";
    default: return "";
  }
}

int ocrf_timer_create(int capacity, void** timer_out) {
  if (capacity <= 0 || !timer_out) return (int)hipErrorInvalidValue;
  Timer* t = new Timer();
  t->start.resize(capacity);
  t->stop.resize(capacity);
  for (int i = 0; i < capacity; ++i) {
    hipError_t e = hipEventCreate(&t->start[i]);
    if (e == hipSuccess) e = hipEventCreate(&t->stop[i]);
    if (e != hipSuccess) { delete t; return (int)e; }
  }
  *timer_out = t;
  return 0;
}

int ocrf_timer_arm(void* timer, int kernel_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  Timer* t = static_cast<Timer*>(timer);
  if (!t) {
    g_armed.clear();
  } else {
    t->kernel_id = kernel_id;
    t->used = 0;
    bool present = false;
    for (Timer* q : g_armed) present = present || (q == t);
    if (!present) g_armed.push_back(t);
  }
  g_n_armed = (int)g_armed.size();
  return 0;
}

int ocrf_timer_read(void* timer, float* ms_out, int capacity, int* count_out) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t || !count_out) return (int)hipErrorInvalidValue;
  int n = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    n = t->used;
  }
  if (n > capacity) n = capacity;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventSynchronize(t->stop[i]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_out[i], t->start[i], t->stop[i]);
    if (e != hipSuccess) return (int)e;
  }
  *count_out = n;
  return 0;
}

// A HIP stream restricted to a set of compute units (hipExtStreamCreateWithCUMask) and / or with a priority:
// the hot path's two chains (VALU-bound renders, latency-bound pools + HOA) each get their own part of the chip
// instead of interleaving workgroup by workgroup.  cu_mask may be null (all CUs).
int ocrf_stream_create(const uint32_t* cu_mask, int n_words, int priority, void** stream_out) {
  if (!stream_out || n_words < 0) return (int)hipErrorInvalidValue;
  hipStream_t s = nullptr;
  hipError_t e;
  if (cu_mask && n_words > 0) {
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
  } else {
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
  }
  if (e != hipSuccess) return (int)e;
  *stream_out = s;
  return 0;
}

// A 32-bit value written to device memory in stream order WITHOUT a kernel launch (hipStreamWriteValue32): the hot
// path's "main chain busy" hint for the persistent blend (ocrf_rasterize_planned, yield_if).
int ocrf_stream_write_value32(int* ptr, int value, ocrf_stream_t stream) {
  if (!ptr) return (int)hipErrorInvalidValue;
  return (int)hipStreamWriteValue32(static_cast<hipStream_t>(stream), ptr, (uint32_t)value, 0);
}

// Node census of a captured hipGraph (host call): kernels, memsets, memcpys, everything else.  Replaying a graph that
// holds MEMSET nodes after an intervening hipMemcpyAsync faults on ROCm 7.2 / gfx950 (launch.h: zero_async), so the
// owners of captured graphs refuse them.
int ocrf_graph_node_census(void* graph, int* n_kernel, int* n_memset, int* n_memcpy, int* n_other) {
  if (!graph || !n_kernel || !n_memset || !n_memcpy || !n_other) return (int)hipErrorInvalidValue;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(static_cast<hipGraph_t>(graph), nullptr, &n);
  if (e != hipSuccess) return (int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n) {
    e = hipGraphGetNodes(static_cast<hipGraph_t>(graph), nodes.data(), &n);
    if (e != hipSuccess) return (int)e;
  }
  *n_kernel = *n_memset = *n_memcpy = *n_other = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e != hipSuccess) return (int)e;
    if (t == hipGraphNodeTypeKernel) ++*n_kernel;
    else if (t == hipGraphNodeTypeMemset) ++*n_memset;
    else if (t == hipGraphNodeTypeMemcpy) ++*n_memcpy;
    else ++*n_other;
  }
  return 0;
}

int ocrf_stream_destroy(void* stream) {
  return stream ? (int)hipStreamDestroy(static_cast<hipStream_t>(stream)) : 0;
}

int ocrf_timer_destroy(void* timer) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t) return 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_armed.size(); ++i)
      if (g_armed[i] == t) { g_armed.erase(g_armed.begin() + i); break; }
    g_n_armed = (int)g_armed.size();
  }
  for (auto e : t->start) (void)hipEventDestroy(e);
  for (auto e : t->stop) (void)hipEventDestroy(e);
  delete t;
  return 0;
}

}  // extern "C"

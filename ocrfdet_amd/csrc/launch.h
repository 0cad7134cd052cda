// Internal launch helper shared by the .hip files: a plain hipLaunchKernelGGL unless a kernel
// timer (ocrf_timer_*, include/ocrf_hip.h) is armed for this kernel id, in which case the launch
// is bracketed by a hipEvent pair on the launch stream (hipExtLaunchKernelGGL), so bench.py can
// read the exact device duration of one kernel inside its timed region.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "ocrf_hip.h"

// A latency-bound kernel of the step's main chain may put its waves in front of the render stream's persistent blend in
// the SIMDs' issue arbitration (priority, then AGE: the blend's waves are always the oldest and always ready).
// -DOCRF_MAIN_PRIO=n builds it in (A/B, tools/build_variant.sh); measured in round 4: see DESIGN 5.
#ifdef OCRF_MAIN_PRIO
#define OCRF_MAIN_CHAIN_PRIO() __builtin_amdgcn_s_setprio(OCRF_MAIN_PRIO)
#else
#define OCRF_MAIN_CHAIN_PRIO() ((void)0)
#endif

#ifdef OCRF_POOL_PRIO_LEVEL
#define OCRF_POOL_PRIO() __builtin_amdgcn_s_setprio(OCRF_POOL_PRIO_LEVEL)
#else
#define OCRF_POOL_PRIO() ((void)0)
#endif

namespace ocrf {

bool timer_next(int kernel_id, hipEvent_t* start, hipEvent_t* stop);

template <typename F, typename... Args>
inline void launch(int kernel_id, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                   Args... args) {
  hipEvent_t a = nullptr, b = nullptr;
  if (timer_next(kernel_id, &a, &b)) {
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, stream, a, b, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, stream, args...);
  }
}

// Zero-fill of `bytes` (a multiple of 4, dst 4-byte aligned) as an ordinary kernel instead of
// hipMemsetAsync: in a captured hipGraph the memset becomes a memset NODE, and replaying a graph that
// holds memset nodes after ANY intervening hipMemcpyAsync on the stream ended in GPU memory faults on
// ROCm 7.2 / gfx950 (found by bisecting a captured neck step: kernel-only graphs survive the same
// sequence).  A kernel is also cheaper than the memset path for these sizes (32-320 KB).
hipError_t zero_async(void* dst, size_t bytes, hipStream_t stream);

// raster_plan.hip: diagnostic knobs reached through ocrf_tune_set (keys 10-19)
void raster_plan_tune(int key, int value);
// hoa.hip: keys 20-29
void hoa_tune(int key, int value);
// hoa_v2b.hip: where a deferred ocrf_hoa_v2b_forward (out == NULL) left decoder1's activations (B,4,H,W) and per-tile
// maxima (B,tiles,4) in its workspace, and the output conv / decoder1 gate weights inside the packed weight vector
void v2b_deferred_pointers(const void* workspace, const float* weights, int B, int H, int W, const float** d1,
                           const float** pm, int* tiles, const float** out_w, const float** out_b, const float** g_w1,
                           const float** g_w2);

}  // namespace ocrf

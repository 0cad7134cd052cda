// Microbenchmark for "HOA-2 as one launch": what does a stage boundary cost on gfx950 —
//   (a) a dependent launch on one stream (what the six-launch HOA-2 pays five times), against
//   (b) a grid-wide barrier inside one persistent kernel (what a fused HOA-2 would pay five times),
// both with the payload of a U-Net block's boundary: every workgroup publishes 16 per-tile maxima before the boundary
// and reads all tiles' maxima after it (gate_from_tiles_dev in csrc/hoa.hip).  Grids are those of the blocks at cfg2
// (B = 2): 32, 98 and 338 workgroups of 256 threads.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ float stage_payload(float* tiles, int n_wg, int stage, float seed) {
  // publish 16 maxima of this workgroup, as a block's epilogue does
  if (threadIdx.x < 16) tiles[((stage & 1) * 16 + threadIdx.x) * 512 + blockIdx.x] = seed + threadIdx.x;
  return seed;
}

__device__ float read_payload(const float* tiles, int n_wg, int stage) {
  // rebuild 16 channel maxima from every workgroup's entry, as a consumer's prologue does
  __shared__ float s_m[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wave; c < 16; c += 4) {
    float m = -1e30f;
    for (int t = lane; t < n_wg; t += 64)
      m = fmaxf(m, tiles[((stage & 1) * 16 + c) * 512 + t]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    if (lane == 0) s_m[c] = m;
  }
  __syncthreads();
  float r = s_m[threadIdx.x & 15];
  __syncthreads();
  return r;
}

// (a) one stage per launch
__global__ __launch_bounds__(256) void stage_kernel(float* tiles, int stage, float* sink, int payload) {
  float v = 0.f;
  if (stage > 0 && payload) v = read_payload(tiles, gridDim.x, stage - 1);
  if (payload) v = stage_payload(tiles, gridDim.x, stage, v * 1e-9f + 1.f);
  if (v == 123.456f) sink[0] = v;
}

// (b) all stages in one launch, a sense-free counting barrier between them (every workgroup resident)
template <int SLEEP>
__global__ __launch_bounds__(256) void fused_kernel(float* tiles, int stages, unsigned* bar, float* sink, int payload) {
  float v = 0.f;
  for (int s = 0; s < stages; ++s) {
    if (s > 0 && payload) v = read_payload(tiles, gridDim.x, s - 1);
    if (payload) v = stage_payload(tiles, gridDim.x, s, v * 1e-9f + 1.f);
    if (s + 1 < stages) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(bar, 1u);
        const unsigned want = (unsigned)(s + 1) * gridDim.x;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(SLEEP);
        __threadfence();
      }
      __syncthreads();
    }
  }
  if (v == 123.456f) sink[0] = v;
}

int main() {
  float *tiles, *sink;
  unsigned* bar;
  CHECK(hipMalloc(&tiles, 2 * 16 * 512 * sizeof(float)));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&bar, 64));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int stages = 6, reps = 200;
  for (int payload = 0; payload < 2; ++payload)
  for (int n_wg : {32, 98, 338, 512}) {
    float ms_a = 0.f, ms_b = 0.f, ms_c = 0.f;
    for (int warm = 0; warm < 2; ++warm) {
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r)
        for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(stage_kernel, dim3(n_wg), dim3(256), 0, st, tiles, s, sink, payload);
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_a, e0, e1));
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) {
        CHECK(hipMemsetAsync(bar, 0, 4, st));
        hipLaunchKernelGGL(fused_kernel<8>, dim3(n_wg), dim3(256), 0, st, tiles, stages, bar, sink, payload);
      }
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_b, e0, e1));
      // the fused kernel with ONE stage: its fixed cost (memset + launch), to take out of (b)
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) {
        CHECK(hipMemsetAsync(bar, 0, 4, st));
        hipLaunchKernelGGL(fused_kernel<8>, dim3(n_wg), dim3(256), 0, st, tiles, 1, bar, sink, payload);
      }
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_c, e0, e1));
    }
    const float per_a = ms_a * 1e3f / reps, per_b = ms_b * 1e3f / reps, per_c = ms_c * 1e3f / reps;
    printf("payload %d workgroups %3d: six launches %6.2f us (%.2f us per stage) | one launch, five grid barriers %6.2f us "
           "(one-stage launch %5.2f us -> %.2f us per barrier + stage)\n",
           payload, n_wg, per_a, per_a / stages, per_b, per_c, (per_b - per_c) / (stages - 1));
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}

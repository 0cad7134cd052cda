// Microbenchmark: issue cost of the VALU instructions the blend loop is made of (gfx950), in SIMD cycles per wave64
// instruction, at 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(unsigned long long* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = 0.5f, a5 = 0.25f, a6 = 1.5f, a7 = 2.5f;
  float b0 = 1.0001f, b1 = 0.9999f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1)); }
    if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                                "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q)); }
    if (OP == 2) { asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q)); }
    if (OP == 3) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    if (OP == 4) { asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                                "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc"); }
    if (OP == 5) { asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n"
                                "v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1)); }
    if (OP == 6) { asm volatile("v_cmp_lt_f32 s[40:41], %0, %4\n v_cndmask_b32 %0, %0, %5, s[40:41]\n v_cmp_lt_f32 s[42:43], %1, %4\n v_cndmask_b32 %1, %1, %5, s[42:43]\n"
                                "v_cmp_lt_f32 s[44:45], %2, %4\n v_cndmask_b32 %2, %2, %5, s[44:45]\n v_cmp_lt_f32 s[46:47], %3, %4\n v_cndmask_b32 %3, %3, %5, s[46:47]\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"); }
    if (OP == 7) { asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q)); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y == 12345.678f) out[0] = 0;
}

template <int OP>
double run(int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;      // 256-thread blocks: 4 waves, one per SIMD
  unsigned long long* d;
  hipMalloc(&d, blocks * 4 * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  hipFree(d);
  double s = 0;
  for (auto v : h) s += (double)v;
  return s / h.size() / (iters * 8.0);      // cycles of one wave's lifetime per instruction
}

int main() {
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_cmp+v_cndmask (vcc)", "v_min_f32 / v_mul_f32", "v_cmp+v_cndmask (sgpr pair)", "v_pk_add_f32"};
  printf("%-30s %10s %10s %10s   (wave-lifetime cycles per instruction; SIMD cycles per instruction = that / waves)\n", "instruction", "1 wave", "2 waves", "4 waves");
#define ROW(OP) { double a = run<OP>(1, 4096), b = run<OP>(2, 4096), c = run<OP>(4, 4096); \
    printf("%-30s %10.2f %10.2f %10.2f   -> SIMD: %.2f %.2f %.2f\n", names[OP], a, b, c, a, b / 2, c / 4); }
  ROW(0) ROW(1) ROW(2) ROW(7) ROW(3) ROW(4) ROW(6) ROW(5)
  return 0;
}

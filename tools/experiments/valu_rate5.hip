// VALU issue-rate probe 5 (gfx950): packed fp32 (v_pk_*_f32 on VGPR pairs), v_log_f32 and the fp64 instructions of the per-lane scalar kernels,
// relative to v_fma_f32 (16 independent chains per thread).   hipcc --offload-arch=gfx950 -O3 valu_rate5.hip -o valu_rate5
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP32(name, str)                                                                                           \
  struct name { using T = float; static __device__ __forceinline__ void op(T& a, T b, T c) { asm volatile(str : "+v"(a) : "v"(b), "v"(c)); } \
                static const char* nm() { return #name; } };
#define OP64(name, str)                                                                                           \
  struct name { using T = double; static __device__ __forceinline__ void op(T& a, T b, T c) { asm volatile(str : "+v"(a) : "v"(b), "v"(c)); } \
                static const char* nm() { return #name; } };
OP32(fma_ref, "v_fma_f32 %0, %0, %1, %2")
OP32(log_, "v_log_f32 %0, %0")
OP32(rcp, "v_rcp_f32 %0, %0")
OP64(pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(pk_add, "v_pk_add_f32 %0, %0, %1")
OP64(pk_mul, "v_pk_mul_f32 %0, %0, %1")
OP64(fma64, "v_fma_f64 %0, %0, %1, %2")
OP64(add64, "v_add_f64 %0, %0, %1")
OP64(mul64, "v_mul_f64 %0, %0, %1")
OP64(rcp64, "v_rcp_f64 %0, %0")
OP64(rsq64, "v_rsq_f64 %0, %0")
OP64(sqrt64, "v_sqrt_f64 %0, %0")
OP64(div_scale64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
OP64(div_fmas64, "v_div_fmas_f64 %0, %0, %1, %2")
OP64(div_fixup64, "v_div_fixup_f64 %0, %0, %1, %2")
template <class O>
__global__ __launch_bounds__(256) void k(typename O::T* out, int iters, float seed) {
  typename O::T a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  typename O::T b = seed * 1.0000001f, c = seed * 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) O::op(a[i], b, c);
  }
  typename O::T s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// one ACTIVE lane per wave (the per-lane scalar kernels): does the instruction still cost a full wave64 issue?
template <class O>
__global__ __launch_bounds__(64) void k1(typename O::T* out, int iters, float seed) {
  if (threadIdx.x != 0) return;
  typename O::T a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i;
  typename O::T b = seed * 1.0000001f, c = seed * 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) O::op(a[i], b, c);
  }
  typename O::T s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x] = s;
}
template <class O> void run(double ref_ms[3]) {
  typename O::T* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int j = 0;
  printf("%-14s", O::nm());
  for (int wg_per_cu : {2, 8}) {
    int blocks = 256 * wg_per_cu, iters = 20000;
    k<O><<<blocks, 256>>>(d, 100, 1.f);
    hipEventRecord(e0); k<O><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ref_ms[j] == 0) ref_ms[j] = ms;
    printf("  waves/SIMD %d: %8.3f ms = %.2f x v_fma_f32", wg_per_cu, ms, ms / ref_ms[j]);
    ++j;
  }
  {
    k1<O><<<1, 64>>>(d, 100, 1.f);
    hipEventRecord(e0); k1<O><<<1, 64>>>(d, 20000, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ref_ms[2] == 0) ref_ms[2] = ms;
    printf("  one lane of one wave: %8.3f ms = %.2f x (%.1f ns per instruction)", ms, ms / ref_ms[2], ms * 1e6 / (20000.0 * 16));
  }
  printf("\n");
  hipFree(d);
}
int main() {
  double ref[3] = {0, 0, 0};
  run<fma_ref>(ref); run<log_>(ref); run<rcp>(ref); run<pk_fma>(ref); run<pk_add>(ref); run<pk_mul>(ref);
  run<fma64>(ref); run<add64>(ref); run<mul64>(ref); run<rcp64>(ref); run<rsq64>(ref); run<sqrt64>(ref);
  run<div_scale64>(ref); run<div_fmas64>(ref); run<div_fixup64>(ref);
  return 0;
}

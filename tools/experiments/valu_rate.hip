// VALU issue-rate probe for gfx950: wave64 instructions per cycle per SIMD for v_fma_f32, v_pk_fma_f32, v_mul_f32+v_add_f32, v_rcp_f32,
// v_cvt, v_cndmask, at 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
  const float m = 1.0000001f, c = 1e-7f;
  const v2f pm = {m, m}, pc = {c, c};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_rcpf(a[i]);
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { int r; asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(a[i])); a[i] = __int_as_float(r); }
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] > 0.5f ? a[(i + 1) & 15] : m;
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { int r; asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(a[i])), "v"(77), "v"(3)); a[i] = __int_as_float(r); }
    }
  }
  float s = 0;
  if (MODE == 1) { for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y; } else { for (int i = 0; i < 16; ++i) s += a[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per_iter_instr, int flops_per_instr) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu : {1, 2, 4, 8}) {
    int blocks = 256 * wg_per_cu, iters = 20000;
    k<MODE><<<blocks, 256>>>(d, 100, 1.f);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * per_iter_instr;            // wave64 instructions
    double per_simd_per_s = wave_instr / 1024 / (ms * 1e-3);
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f G wave-instr/s/SIMD -> %.2f cycles/instr at 2.4 GHz, %.1f TFLOP/s\n", name, wg_per_cu, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 16, 2);
  run<1>("v_pk_fma_f32", 8, 4);
  run<2>("v_rcp_f32", 16, 1);
  run<3>("v_cvt_flr", 16, 1);
  run<4>("cmp+cndmask", 32, 1);
  run<5>("v_mad_u32_u24", 16, 1);
  return 0;
}
